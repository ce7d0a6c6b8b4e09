// gemm_f32.hip — the fp32-accurate MFMA GEMMs of libtoad_hip.so (gfx950): the Linear layers of TOAD's MIL path and the
// convolutions-as-GEMMs of the feature extractor. One translation unit (kernels and launchers must share a TU without
// -fgpu-rdc), organised as:
//   gemm_nt_f32.inc    generic exact-fp32 128x128 NT kernel (the fallback for shapes the persistent kernels do not take); the per-XCD tile
//                      plan and the shared epilogue of the persistent kernels
//   gemm_tn.inc        wgrad: the generic exact-fp32 TN kernel, the split-M plan, slab reduction, transpose
//   gemm_h2.inc        the MIL GEMMs (forward / dgrad / wgrad) and the extractor's wide convolutions: persistent 256x256 kernels on the
//                      fp16 pipe with two-piece operands (x*s = h + m, three MFMA terms, power-of-two scales from per-256-row abs-max arrays)
//   gemm_pt.inc        plane-tiled prepared bags: the splitter and the weight-gradient kernel that reads them by LDS-DMA + transposing LDS reads
//   gemm_narrow.inc    the narrow-N weight-plane format, convolution geometry and gather modes of the extractor
//   gemm_stream.inc    narrow-N kernels of the extractor: A streamed through registers; 3x3 convolutions with the activation halo in LDS
//   this file          shared constants, launch selection, the extern "C" entry points declared in include/toad_hip.h
// (Round 1's exact-fp32 / split-bf16 persistent kernels and the LDS-staged narrow kernels were A/B arms; they left the tree in round 4 -
//  tools/ab/README.md names the commit that still holds them.)
//
// Two product shapes cover every GEMM on the path:
//   gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)        both operands reduction-contiguous
//             forward  Y = act(X W^T + b)             models/model_toad.py:59,62,21,25
//             dgrad    dX = (dY (W^T)^T + add)*mask   W^T read transposed in place by the plane splitter
//   gemm_tn : C[I,J] = sum_m A[m,I] . B[m,J]         both operands reduction-strided
//             wgrad    dW = dY^T X, split over m, deterministic slab reduction
// Why not plain bf16: parity with the reference's PyTorch-CPU path is 1e-4 on fp32 outputs and bf16 operands miss it
// (1e-3..6e-3); gfx950 has no TF32/xf32. DESIGN.md 4 gives the arithmetic and the measurements.
#include "common.h"

#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace toad {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NT_LD = 36;                                  // padded LDS row (floats), conflict-free b128
constexpr int NT_TILE = BM * NT_LD;                        // floats per operand tile
constexpr int NT_SMEM = 2 * 2 * NT_TILE * (int)sizeof(float);   // 73,728 B -> 2 blocks / CU
constexpr int TN_LD = 128;
constexpr int TN_TILE = BK * TN_LD;
constexpr int TN_SMEM = 2 * 2 * TN_TILE * (int)sizeof(float);   // 65,536 B

// EpiScalars (common.h): epilogue scalars shared by every NT kernel (plain scalars only: pointers stay direct kernel arguments)

// XCD-aware block -> tile map: all column tiles of one row tile run on the same XCD (same L2),
// back to back, so the A panel is fetched from HBM once and re-read from L2.
__device__ __forceinline__ bool map_tile(int tiles_m, int tiles_n, int &tm, int &tn) {
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, q = b / kNumXCD;
    tm = (q / tiles_n) * kNumXCD + xcd;
    tn = q % tiles_n;
    return tm < tiles_m;
}

#include "gemm_nt_f32.inc"
#include "gemm_tn.inc"
#include "gemm_h2.inc"
#include "gemm_pt.inc"
#include "gemm_narrow.inc"
#include "gemm_stream.inc"
#include "stem_halo.inc"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// One-time initialisation of this translation unit: kernel attributes (dynamic LDS sizes). The library has ONE arithmetic per
// product shape and reads no environment variable anywhere; std::call_once makes the first call from any thread complete the
// attribute calls before any launch (PyTorch runs backward on its own thread; the ingest workers are threads too).
// Residual GEMMs with a reduction of at most this many elements run on the narrow (streamed) tiles whatever their width: measured
// (tools/gemm_shape_bench.py, same box) M=262144 K=64 N=256 +residual 170 -> 133 us; K=128 equal, K=256 slower (A is re-split per
// 128-column tile)
constexpr int kNarrowResKmax = 64;
static std::once_flag g_cfg_once;
static void cfg() {
    std::call_once(g_cfg_once, [] {
#define TOAD_ATTR(K, BYTES) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)
#define TOAD_H2_ATTR(P, A_, M_) TOAD_ATTR((gemm_nt_h2_big_kernel<P, A_, M_, 0>), H2_SMEM)
        TOAD_H2_ATTR(false, false, 0); TOAD_H2_ATTR(false, false, 1); TOAD_H2_ATTR(false, false, 2);
        TOAD_H2_ATTR(true, false, 0); TOAD_H2_ATTR(true, false, 1); TOAD_H2_ATTR(true, false, 2);
        TOAD_H2_ATTR(false, true, 0); TOAD_H2_ATTR(false, true, 1); TOAD_H2_ATTR(false, true, 2);
#undef TOAD_H2_ATTR
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 0, 1>), H2_SMEM);
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 0, 2>), H2_SMEM_PT);
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 0, 3>), H2_SMEM_RUN);
        TOAD_ATTR((gemm_nt_h2_big_kernel<true, false, 2, 4>), H2_SMEM);          // batched pooled addend (ragged multi-slide step)
        // half-height tiles (short operands; nt_half_tiles below): the MIL step's five epilogues
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 0, 0, 128>), H2_SMEM_HALF);
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 2, 0, 128>), H2_SMEM_HALF);
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 1, 0, 128>), H2_SMEM_HALF);
        TOAD_ATTR((gemm_nt_h2_big_kernel<true, false, 2, 0, 128>), H2_SMEM_HALF);
        TOAD_ATTR((gemm_nt_h2_big_kernel<true, false, 2, 4, 128>), H2_SMEM_HALF);
        TOAD_ATTR((gemm_nt_h2_big_kernel<false, false, 0, 3, 128>), H2_SMEM_HALF_RUN);
        TOAD_ATTR(gemm_tn_h2_big_kernel<false>, TN2_SMEM);
        TOAD_ATTR(gemm_tn_h2_big_kernel<true>, TN2_SMEM);
        TOAD_ATTR(gemm_tn_h2_batch_kernel, TN2_SMEM);
        TOAD_ATTR(gemm_tn_pt_kernel, TP_SMEM);
        TOAD_ATTR(gemm_nt_f32_kernel, NT_SMEM);
        TOAD_ATTR(gemm_tn_f32_kernel, TN_SMEM);
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 2, GATHER_NONE>), (StreamCfg<2, 2>::SMEM));
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 4, GATHER_NONE>), (StreamCfg<2, 4>::SMEM));
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 2, GATHER_CONV>), (StreamCfg<2, 2>::SMEM));
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 4, GATHER_CONV>), (StreamCfg<2, 4>::SMEM));
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 2, GATHER_STEM>), (StreamCfg<2, 2>::SMEM));
        TOAD_ATTR((gemm_nt_h2_stream_kernel<2, 2, GATHER_STEM_POOL>), (StreamCfg<2, 2>::SMEM + STEM_POOL_LDS));
        TOAD_ATTR(stem_halo_pool_kernel, SH_SMEM);
        TOAD_ATTR(conv3x3_h2_halo_kernel<2>, 160 * 1024);
        TOAD_ATTR(conv3x3_h2_halo_kernel<4>, 160 * 1024);
#undef TOAD_ATTR
    });
}

// max |x| over n floats -> out[0] (bit pattern of a non-negative float, zeroed here): the tensor-wide abs-max a narrow / extractor GEMM
// scales its A operand with when no producer handed one over (the per-op entry points; inside toad_resnet50_trunc_fwd_f32 every
// producer emits it)
__global__ __launch_bounds__(256) void gmax_kernel(const float *__restrict__ x, int64_t n4, float *__restrict__ out) {
    __shared__ float s[4];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        mx = __builtin_fmaxf(mx, h2_absmax4(__builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x) + i)));
    mx = h2_wave_max(mx);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) h2_atomic_amax(out, __builtin_fmaxf(__builtin_fmaxf(s[0], s[1]), __builtin_fmaxf(s[2], s[3])));
}
int launch_gmax(const float *x, int64_t n, float *out, hipStream_t st, const char *what) {
    (void)hipMemsetAsync(out, 0, sizeof(float), st);
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gmax_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, n / 4, out);
    return check_launch(what);
}

// C[M,N] = act(A' W^T + bias + addend) with N <= 128 on the narrow kernels; A' = A[M,K] (MODE = GATHER_NONE) or the implicit
// im2col of the NHWC activation A described by cg. `ws` as for launch_nt (planes and inverse scales live behind the slab area).
// a_gmax: device scalar max |A| (of the whole activation); y_gmax (optional): receives max |C| by atomic max (zeroed by the caller).
template <int RA, int NB, int MODE>
static int launch_narrow_t(const float *A, int64_t lda, const float *a_gmax, const float *W, int64_t ldw, float *C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, const float *bias, int relu, const float *addend, const ConvGeom &cg, float *y_gmax, void *ws,
                           hipStream_t st, const char *what) {
    (void)cfg();
    char *w = reinterpret_cast<char *>(ws) + (size_t)PB_GRID * PB * PB * sizeof(float);
    constexpr int TNW = NB * 32;         // tile width (RA only names the instantiation; the streamed kernels run 64-row wave tiles)
    const int tiles_n = (int)((N + TNW - 1) / TNW);
    unsigned short *planes = reinterpret_cast<unsigned short *>(w);
    float *binv;
    // the streamed kernel walks an implicit convolution's k-stages channel-chunk outer, tap inner (gemm_stream.inc): its planes are split in that order
    const int taps = MODE == GATHER_CONV ? (int)(K / cg.C) : 1;
    const int nk = (int)(K / BK), nkp = (nk + 1) & ~1;                  // the streamed kernel walks k-stages in pairs: an odd count gets a zero stage
    binv = reinterpret_cast<float *>(w + (size_t)tiles_n * TNW * (size_t)nkp * BK * 4);
    hipLaunchKernelGGL(split_planes_narrow_h2_kernel<NB>, dim3((tiles_n * TNW + 3) / 4), dim3(256), 0, st, W, ldw, planes, binv, (int)N, (int)K, tiles_n,
                       taps, cg.C, nkp);
    if (int rc = check_launch(what)) return rc;
    if (MODE == GATHER_CONV) {       // 3x3 / 1 / 1 with whole 256-pixel row blocks: the activation halo lives in LDS (gemm_stream.inc)
        const int W = cg.W, H = cg.H;
        const bool shape = cg.kw == 3 && K == 9 * (int64_t)cg.C && cg.stride == 1 && cg.pad == 1 && cg.Ho == H && cg.Wo == W && W >= 8 && W <= 128 &&
                           (W & (W - 1)) == 0 && H % (256 / W) == 0 && cg.C % BK == 0 && M % 256 == 0;
        if (shape) {
            using HCfg = HaloCfg<NB>;
            const int hb = HCfg::halo_bytes(W);
            hipLaunchKernelGGL(conv3x3_h2_halo_kernel<NB>, dim3(2 * PB_GRID), dim3(HCfg::THREADS), HCfg::smem(W), st, A, a_gmax, planes, binv, C, ldc, (int)M, (int)N,
                               (int)K, bias, relu, addend, cg, y_gmax, (int)(M / 256), tiles_n, hb);
            return check_launch(what);
        }
    }
    // A streamed through registers (gemm_stream.inc): wave tile 64 x 64 / 64 x 128
    constexpr int SRA = 2;
    using SCfg = StreamCfg<SRA, NB>;
    const int stiles_m = (int)((M + SCfg::TM - 1) / SCfg::TM);
    if constexpr (MODE == GATHER_STEM_POOL) {        // C = the POOLED output; every workgroup a contiguous range of row-pair tiles (ext_stem_conv checked the geometry)
        hipLaunchKernelGGL((gemm_nt_h2_stream_kernel<SRA, NB, MODE>), dim3(std::min(stiles_m, SCfg::WG_PER_CU * PB_GRID)), dim3(SCfg::THREADS), SCfg::SMEM + STEM_POOL_LDS, st,
                           A, lda, a_gmax, planes, binv, C, ldc, (int)M, (int)N, (int)K, bias, relu, addend, cg, y_gmax, stiles_m, tiles_n);
        return check_launch(what);
    }
    hipLaunchKernelGGL((gemm_nt_h2_stream_kernel<SRA, NB, MODE>), dim3(SCfg::WG_PER_CU * PB_GRID), dim3(SCfg::THREADS), SCfg::SMEM, st, A, lda, a_gmax, planes, binv,
                       C, ldc, (int)M, (int)N, (int)K, bias, relu, addend, cg, y_gmax, stiles_m, tiles_n);
    return check_launch(what);
}

static bool narrow_ok(int64_t M, int64_t N, int64_t K, int64_t ldc, const float *bias, const float *addend, void *ws) {
    const bool wide_ok = addend && K <= kNarrowResKmax;       // residual GEMMs with a short reduction: epilogue-bound, see DESIGN 10
    return ws && (N <= 128 || wide_ok) && N % 4 == 0 && K % BK == 0 && ldc % 4 == 0 && M < (1ll << 31) &&
           (!bias || aligned16(bias)) && (!addend || aligned16(addend));
}

// ---- h2 (fp16 two-piece) path ---------------------------------------------------------------------------------
// shapes the persistent h2 NT kernel serves (32-bit row offsets of A, whole 32-deep stages, 16-byte output rows)
bool h2_nt_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
    return M >= 1 && K % BK == 0 && N % 4 == 0 && ldc % 4 == 0 && lda % 4 == 0 && (uint64_t)M * lda * 4 < (1ull << 32);
}
// the self-measuring operand mode keeps one bit per item of a workgroup: at most 64 items (M up to ~1 M rows at N = 512)
bool nt_run_ok(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + PB - 1) / PB) * ((N + PB - 1) / PB);
    return h2_nt_ok(M, N, K, K, N) && (tiles + kNumXCD * PB_BLOCKS_PER_XCD - 1) / (kNumXCD * PB_BLOCKS_PER_XCD) + 2 <= 64;
}
size_t h2_planes_bytes(int64_t N, int64_t K) { return (size_t)((N + PB - 1) / PB) * PB * (size_t)K * 4; }   // two fp16 planes
size_t h2_slab_bytes() { return (size_t)PB_GRID * PB * PB * sizeof(float); }
size_t h2_binv_bytes(int64_t N) { return (size_t)((N + PB - 1) / PB) * PB * sizeof(float); }

// Half-height (128 x 256) tiles, one whole-K item per workgroup and no K-split / fix-up launch (gemm_h2.inc, TM = 128): for operands short enough
// that every XCD holds at most 32 such tiles AND long enough that the 256 x 256 plan could only cut its tiles into two K-slices (its cap of ~64
// slabs per launch, nt_plan): measured on the 10k-patch step (profiles/r06c_*), a two-slice GEMM + its fix-up launch take 46 / 52 / 63 us
// (K = 512 / 768 / 1024) against 33 / 47 / 62 us for whole-K half tiles (a 128-row stage costs 1.8 us, the LOAD phase of a wave bounds it, a
// 256-row stage 2.1). Shorter operands (a 2,000-patch bag: three or more slices per tile, a 256-patch bag: sixteen one-stage slices) stay on the
// K-split, which puts more CUs on them than whole-K items would.
// The rule depends on (M, N) ONLY, never on K: the one-bit ReLU image a forward GEMM writes (whole tiles only; K-split remainder tiles go through
// the fix-up kernel, which reads the fp32 activations instead) is read by the dgrad of the same layer, a GEMM with the same M and N and another
// K - both must agree on which tiles are whole.
bool nt_half_tiles(int64_t M, int64_t N) {
    const int tiles_n = (int)((N + PB - 1) / PB);
    const int64_t tm128 = (M + 127) / 128, tm256 = (M + PB - 1) / PB;
    if (((tm128 + kNumXCD - 1) / kNumXCD) * tiles_n > PB_BLOCKS_PER_XCD) return false;
    const int64_t rem_max = ((tm256 + kNumXCD - 1) / kNumXCD) * tiles_n, rem_all = tm256 * tiles_n;        // (no full round: rem = all tiles)
    int64_t g = PB_BLOCKS_PER_XCD / rem_max, cap = 64 / rem_all;
    if (cap < 2) cap = 2;
    if (g > cap) g = cap;
    return g <= 2;
}

// split up to 6 weight operands into planes + inverse row scales with ONE launch
int launch_split_h2(const H2Operand *ops, int n, float *zero, int zero_n, hipStream_t st, const char *what, float *zero2, int zero2_n) {
    H2SplitBatch b;
    b.n = n;
    b.zero = zero;
    b.zero_n = zero ? zero_n : 0;
    b.zero2 = zero2;
    b.zero2_n = zero2 ? zero2_n : 0;
    int waves = 0;
    for (int i = 0; i < 6; ++i) {
        if (i < n) {
            const int tiles_n = (int)((ops[i].N + PB - 1) / PB);
            b.d[i] = H2SplitDesc{ops[i].src, ops[i].sn, ops[i].sk, ops[i].planes, ops[i].binv, (int)ops[i].N, (int)ops[i].K, tiles_n, waves};
            waves += (ops[i].sn == 1 && ops[i].sk != 1) ? tiles_n * PB / 4 : tiles_n * PB;       // transposed source: a workgroup (4 waves) per 16 rows; else a wave per row
        } else {
            b.d[i] = H2SplitDesc{nullptr, 0, 0, nullptr, nullptr, 0, 0, 0, INT32_MAX};
        }
    }
    b.total_waves = waves;
    hipLaunchKernelGGL(split_planes_h2_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, b);
    return check_launch(what);
}
int launch_absmax(const float *X, int64_t ld, int64_t M, int64_t K, float *amax, bool zero, hipStream_t st, const char *what) {
    const int nblk = (int)h2_nblk(M);
    if (zero) (void)hipMemsetAsync(amax, 0, (size_t)nblk * sizeof(float), st);
    hipLaunchKernelGGL(absmax_rows256_kernel, dim3(nblk <= 8 ? 16 : 4, nblk), dim3(256), 0, st, X, ld, (int)M, (int)K, amax);
    return check_launch(what);
}
size_t pt_bytes_host(int64_t rows, int64_t cols) { return pt_bytes(rows, cols); }
// X [M][K] fp32 -> plane-tiled form (gemm_pt.inc) with the per-row-tile exponents of `amax` (its toad_absmax_rows256_f32 array)
int launch_pt_split(const float *X, int64_t ld, int64_t M, int64_t K, const float *amax, unsigned short *pt, hipStream_t st, const char *what) {
    hipLaunchKernelGGL(pt_split_kernel, dim3((unsigned)pt_col_stages(K), (unsigned)pt_row_tiles(M)), dim3(256), 0, st, X, ld, (int)M, (int)K, amax, pt,
                       (int)pt_col_stages(K));
    return check_launch(what);
}
// C = epi(A . B^T) with pre-split B (planes + binv) and the abs-max array of A; y_amax (zeroed by the caller) receives the abs-max of C
int launch_nt_h2(const float *A, int64_t lda, const float *a_amax, const unsigned short *planes, const float *binv, float *C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend,
                        const float *mask_src, const unsigned long long *mask_bits, H2Pool pool, float *slabs, float *y_amax,
                        unsigned long long *bits_out, hipStream_t st, const char *what, int a_mode, int a_stride, int y_stride, float *a_amax_out,
                        int *slab_ke) {
    const int tiles_m = (int)((M + PB - 1) / PB), tiles_n = (int)((N + PB - 1) / PB);
    (void)cfg();
    // Staggered workgroup starts (gemm_h2.inc) for launches of at least two tiles per workgroup: the 32 workgroups of an XCD start spread over
    // ~2/5 of a tile period (1.95 us per k-step + ~8 us of epilogue), so their epilogue store bursts and LOAD phases no longer coincide.
    // Same-box sweep on the 100k-patch step (profiles/r04u_stagger_mil_step.txt): 2.125 -> 2.083 ms; per GEMM the best spread is about a third
    // of its period (K = 512: 10-15 us, K = 1024: 20-25 us). Round 3's four-phase version of this was neutral. Timing only: values never change.
    if (es.stagger == 0 && (int64_t)tiles_m * tiles_n >= 2 * PB_GRID) es.stagger = (int)(((K / BK) * 195 + 800) * 2 / 5);
    // a_amax == NULL with a_amax_out (fp32 A, plain forward): the kernel measures A itself, stage by stage (AMODE 3, gemm_h2.inc), fills
    // a_amax_out (zeroed by the caller) and slab_ke; the fix-up then reads the completed array
    const bool run_mode = a_mode == TOAD_X_F32 && !a_amax && a_amax_out;
    if (run_mode) {
        if (addend || mask_src || mask_bits || pool.T > 0 || a_stride != 1 || !slab_ke) { set_error("%s: the self-measuring operand mode is a plain forward with per-block scales", what); return TOAD_EINVAL; }
        if (nt_half_tiles(M, N)) {
            hipLaunchKernelGGL((gemm_nt_h2_big_kernel<false, false, 0, 3, 128>), dim3(PB_GRID), dim3(512), H2_SMEM_HALF_RUN, st, A, lda, (const float *)nullptr, planes,
                               binv, C, ldc, (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, 0, slabs, y_amax, bits_out, (int)((M + 127) / 128), tiles_n, a_stride, y_stride, a_amax_out, slab_ke);
            return check_launch(what);
        }
        hipLaunchKernelGGL((gemm_nt_h2_big_kernel<false, false, 0, 3>), dim3(PB_GRID), dim3(512), H2_SMEM_RUN, st, A, lda, (const float *)nullptr, planes,
                           binv, C, ldc, (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, 0, slabs, y_amax, bits_out, tiles_m, tiles_n, a_stride, y_stride, a_amax_out, slab_ke);
        int rcr = check_launch(what);
        if (rcr) return rcr;
        int max_rem = 0, rem_all = 0;
        for (int x = 0; x < kNumXCD; ++x) { const NtPlan pl = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)); if (pl.g > 1) { rem_all += pl.rem; if (pl.rem > max_rem) max_rem = pl.rem; } }
        if (max_rem > 0) {
            const H2Pool np{nullptr, nullptr, nullptr, 0};
            if (rem_all > 8)
                hipLaunchKernelGGL(nt_fixup_h2_kernel<4>, dim3(16, max_rem, kNumXCD), dim3(256), 0, st, (const float *)slabs, (const float *)a_amax_out, binv, C, ldc,
                                   (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, np.a_raw, np.stats, np.dM, 0, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)slab_ke);
            else
                hipLaunchKernelGGL(nt_fixup_h2_kernel<1>, dim3(64, max_rem, kNumXCD), dim3(256), 0, st, (const float *)slabs, (const float *)a_amax_out, binv, C, ldc,
                                   (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, np.a_raw, np.stats, np.dM, 0, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)slab_ke);
            rcr = check_launch(what);
        }
        return rcr;
    }
    if (a_mode != TOAD_X_F32) {   // A is fp16 [M, lda halves] or plane-tiled: plain forward only (no addend / mask / pooling variants are instantiated)
        if (addend || mask_src || mask_bits || pool.T > 0) { set_error("%s: the fp16 / plane-tiled operand kernels have no addend / mask / pooling epilogue", what); return TOAD_EINVAL; }
        if (a_mode == TOAD_X_PT)
            hipLaunchKernelGGL((gemm_nt_h2_big_kernel<false, false, 0, 2>), dim3(PB_GRID), dim3(512), H2_SMEM_PT, st, A, lda, a_amax, planes,
                               binv, C, ldc, (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, 0, slabs, y_amax, bits_out, tiles_m, tiles_n, a_stride, y_stride, (float *)nullptr, (int *)nullptr);
        else
        hipLaunchKernelGGL((gemm_nt_h2_big_kernel<false, false, 0, 1>), dim3(PB_GRID), dim3(512), H2_SMEM, st, A, lda, (const float *)nullptr, planes,
                           binv, C, ldc, (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, 0, slabs, y_amax, bits_out, tiles_m, tiles_n, a_stride, y_stride, (float *)nullptr, (int *)nullptr);
        int rc16 = check_launch(what);
        if (rc16) return rc16;
        int max_rem16 = 0, rem_all16 = 0;
        for (int x = 0; x < kNumXCD; ++x) { const NtPlan pl = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)); if (pl.g > 1) { rem_all16 += pl.rem; if (pl.rem > max_rem16) max_rem16 = pl.rem; } }
        if (max_rem16 > 0) {   // the fix-up takes the A scale from a_amax: fp16 operands carry scale 1 -> NULL selects exponent 0
            const H2Pool np{nullptr, nullptr, nullptr, 0};
            const float *fx_amax = a_mode == TOAD_X_PT ? a_amax : nullptr;
            if (rem_all16 > 8)
                hipLaunchKernelGGL(nt_fixup_h2_kernel<4>, dim3(16, max_rem16, kNumXCD), dim3(256), 0, st, (const float *)slabs, fx_amax, binv, C, ldc,
                                   (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, np.a_raw, np.stats, np.dM, 0, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)nullptr);
            else
                hipLaunchKernelGGL(nt_fixup_h2_kernel<1>, dim3(64, max_rem16, kNumXCD), dim3(256), 0, st, (const float *)slabs, fx_amax, binv, C, ldc,
                                   (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, (const float *)nullptr, np.a_raw, np.stats, np.dM, 0, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)nullptr);
            rc16 = check_launch(what);
        }
        return rc16;
    }
    if (pool.T > 0 && addend) { set_error("%s: an addend buffer and the recomputed pooling addend are mutually exclusive", what); return TOAD_EINVAL; }
    // whole tiles read the one-bit ReLU image when the caller has it (mask_bits), the fix-up kernel always reads the fp32 mask_src
    const float *msrc = mask_bits ? reinterpret_cast<const float *>(mask_bits) : mask_src;
#define TOAD_LAUNCH_H2(P, A_, M_)                                                                                                     \
    hipLaunchKernelGGL((gemm_nt_h2_big_kernel<P, A_, M_>), dim3(PB_GRID), dim3(512), H2_SMEM, st, A, lda, a_amax, planes, binv, C, ldc, (int)M, \
                       (int)N, (int)K, bias, es, addend, msrc, pool.a_raw, pool.stats, pool.dM, pool.T, slabs, y_amax, bits_out, tiles_m, tiles_n, a_stride, y_stride, (float *)nullptr, (int *)nullptr)
    const int msk = mask_bits ? 2 : (mask_src ? 1 : 0);
    if (mask_bits && !mask_src) { set_error("%s: the one-bit ReLU image needs the fp32 relu_src as well (remainder tiles)", what); return TOAD_EINVAL; }
    // short operands: half-height tiles, whole K per workgroup, no fix-up launch (the epilogues the MIL step uses; per-block abs-max arrays)
    if (!addend && (pool.T == 0 || msk == 2) && a_stride == 1 && y_stride == 1 && nt_half_tiles(M, N)) {
        const int tm128 = (int)((M + 127) / 128);
#define TOAD_LAUNCH_H2_HALF(P, M_, AM)                                                                                                   \
        hipLaunchKernelGGL((gemm_nt_h2_big_kernel<P, false, M_, AM, 128>), dim3(PB_GRID), dim3(512), H2_SMEM_HALF, st, A, lda, a_amax, planes, binv, C, ldc, \
                           (int)M, (int)N, (int)K, bias, es, (const float *)nullptr, msrc, pool.a_raw, pool.stats, pool.dM, pool.T, slabs, y_amax, bits_out,    \
                           tm128, tiles_n, a_stride, y_stride, (float *)nullptr, (int *)nullptr)
        if (pool.T >> 8) {
            if ((pool.T & 255) != 2 || !aligned16(pool.a_raw)) { set_error("%s: the batched pooled addend is instantiated for 2 tasks on the one-bit ReLU image", what); return TOAD_EINVAL; }
            TOAD_LAUNCH_H2_HALF(true, 2, 4);
        } else if (pool.T > 0) TOAD_LAUNCH_H2_HALF(true, 2, 0);
        else if (msk == 2) TOAD_LAUNCH_H2_HALF(false, 2, 0);
        else if (msk == 1) TOAD_LAUNCH_H2_HALF(false, 1, 0);       // (fp32 mask: the per-op dgrad without a bit image; same tiles, same sums as with one)
        else TOAD_LAUNCH_H2_HALF(false, 0, 0);
#undef TOAD_LAUNCH_H2_HALF
        return check_launch(what);
    }
    if (pool.T >> 8) {       // batched pooled addend: per-row records + per-slide dM (the ragged multi-slide step; gemm_h2_epilogue.inc PBATCH)
        if ((pool.T & 255) != 2 || msk != 2 || !aligned16(pool.a_raw)) { set_error("%s: the batched pooled addend is instantiated for 2 tasks on the one-bit ReLU image", what); return TOAD_EINVAL; }
        hipLaunchKernelGGL((gemm_nt_h2_big_kernel<true, false, 2, 4>), dim3(PB_GRID), dim3(512), H2_SMEM, st, A, lda, a_amax, planes, binv, C, ldc, (int)M, (int)N, (int)K, bias,
                           es, (const float *)nullptr, msrc, pool.a_raw, pool.stats, pool.dM, pool.T, slabs, y_amax, bits_out, tiles_m, tiles_n, a_stride, y_stride,
                           (float *)nullptr, (int *)nullptr);
    } else if (pool.T > 0) { if (msk == 2) TOAD_LAUNCH_H2(true, false, 2); else if (msk == 1) TOAD_LAUNCH_H2(true, false, 1); else TOAD_LAUNCH_H2(true, false, 0); }
    else if (addend) { if (msk == 2) TOAD_LAUNCH_H2(false, true, 2); else if (msk == 1) TOAD_LAUNCH_H2(false, true, 1); else TOAD_LAUNCH_H2(false, true, 0); }
    else { if (msk == 2) TOAD_LAUNCH_H2(false, false, 2); else if (msk == 1) TOAD_LAUNCH_H2(false, false, 1); else TOAD_LAUNCH_H2(false, false, 0); }
#undef TOAD_LAUNCH_H2
    int rc = check_launch(what);
    if (rc) return rc;
    int max_rem = 0;                                   // fix-up grid: only as many tile rows as some XCD has remainder tiles
    for (int x = 0; x < kNumXCD; ++x) { const NtPlan pl = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)); if (pl.g > 1 && pl.rem > max_rem) max_rem = pl.rem; }
    if (max_rem > 0) {
        int rem_all = 0;
        for (int x = 0; x < kNumXCD; ++x) { const NtPlan pl = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)); if (pl.g > 1) rem_all += pl.rem; }
        if (rem_all > 8)
            hipLaunchKernelGGL(nt_fixup_h2_kernel<4>, dim3(16, max_rem, kNumXCD), dim3(256), 0, st, (const float *)slabs, a_amax, binv, C, ldc,
                               (int)M, (int)N, (int)K, bias, es, addend, mask_src, pool.a_raw, pool.stats, pool.dM, pool.T, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)nullptr);
        else
            hipLaunchKernelGGL(nt_fixup_h2_kernel<1>, dim3(64, max_rem, kNumXCD), dim3(256), 0, st, (const float *)slabs, a_amax, binv, C, ldc,
                               (int)M, (int)N, (int)K, bias, es, addend, mask_src, pool.a_raw, pool.stats, pool.dM, pool.T, y_amax, tiles_m, tiles_n, a_stride, y_stride, (const int *)nullptr);
        rc = check_launch(what);
    }
    return rc;
}

// Self-contained NT product for the per-op entry points: B[n,k] = Bsrc[n*bsn + k*bsk]. Uses the h2 kernel when the shape allows
// (splitting B and, when a_amax == NULL, measuring A inside `ws`), else the older paths (launch_nt). y_amax, when requested, is
// always produced (by the epilogue, or by a pass over C on the fallback paths).
static int launch_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend,
                     const float *mask_src, void *ws, hipStream_t st, const char *what);
static int launch_nt_auto(const float *A, int64_t lda, const float *a_amax, const float *B, int64_t ldb, float *C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend, const float *mask_src,
                          const unsigned long long *mask_bits, H2Pool pool, float *y_amax, unsigned long long *bits_out, void *ws,
                          hipStream_t st, const char *what) {
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias && !aligned16(bias)) || (addend && !aligned16(addend)) ||
        (mask_src && !aligned16(mask_src))) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (M > INT32_MAX - BM || N > INT32_MAX - BN || K > INT32_MAX - BK) { set_error("%s: dimension too large", what); return TOAD_ESHAPE; }
    if (y_amax) (void)hipMemsetAsync(y_amax, 0, (size_t)h2_nblk(M) * sizeof(float), st);
    if (ws && ldb == K && h2_nt_ok(M, N, K, lda, ldc)) {
        char *w = reinterpret_cast<char *>(ws);
        float *slabs = reinterpret_cast<float *>(w);
        w += (size_t)PB_GRID * PB * PB * sizeof(float);
        unsigned short *planes = reinterpret_cast<unsigned short *>(w);
        w += h2_planes_bytes(N, K);
        float *binv = reinterpret_cast<float *>(w);
        w += h2_binv_bytes(N);
        float *amax_ws = reinterpret_cast<float *>(w);
        int *slab_ke = reinterpret_cast<int *>(amax_ws + h2_nblk(M) + 64);
        // A without an abs-max array: a plain forward measures it inside the GEMM (AMODE 3, like the whole-slide calls: the two routes stay
        // bitwise equal); a product with an addend / mask / pooling epilogue measures it with a pass of its own
        const bool self_measure = !a_amax && !addend && !mask_src && !mask_bits && pool.T == 0 && nt_run_ok(M, N, K);
        if (!a_amax && !self_measure) {
            if (int rc = launch_absmax(A, lda, M, K, amax_ws, true, st, what)) return rc;
            a_amax = amax_ws;
        }
        const H2Operand op{B, ldb, 1, N, K, planes, binv};
        if (self_measure) {
            if (int rc = launch_split_h2(&op, 1, amax_ws, (int)h2_nblk(M), st, what)) return rc;      // (the split launch zeroes the array the GEMM fills)
            return launch_nt_h2(A, lda, nullptr, planes, binv, C, ldc, M, N, K, bias, es, nullptr, nullptr, nullptr, pool, slabs, y_amax, bits_out, st, what,
                                TOAD_X_F32, 1, 1, amax_ws, slab_ke);
        }
        if (int rc = launch_split_h2(&op, 1, nullptr, 0, st, what)) return rc;
        return launch_nt_h2(A, lda, a_amax, planes, binv, C, ldc, M, N, K, bias, es, addend, mask_src, mask_bits, pool, slabs, y_amax, bits_out, st, what);
    }
    if (pool.T > 0) { set_error("%s: the recomputed pooling addend needs the h2 kernel (K %% 32 == 0, M*K*4 < 2^32, workspace)", what); return TOAD_ESHAPE; }
    if (bits_out) { set_error("%s: the one-bit ReLU image is only produced by the h2 kernel (check toad_linear_h2_ok)", what); return TOAD_ESHAPE; }
    int rc = launch_nt(A, lda, B, ldb, C, ldc, M, N, K, bias, es, addend, mask_src, ws, st, what);
    if (rc || !y_amax) return rc;
    hipLaunchKernelGGL(absmax_rows256_kernel, dim3(h2_nblk(M) <= 8 ? 16 : 4, (int)h2_nblk(M)), dim3(256), 0, st, C, ldc, (int)M, (int)N, y_amax);
    return check_launch(what);
}

static int launch_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend,
                     const float *mask_src, void *ws, hipStream_t st, const char *what) {
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (M > INT32_MAX - BM || N > INT32_MAX - BN || K > INT32_MAX - BK) { set_error("%s: dimension too large", what); return TOAD_ESHAPE; }
    if (K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0) { set_error("%s: reduction dim %lld must be a multiple of 4", what, (long long)K); return TOAD_ESHAPE; }
    if (!aligned16(A) || !aligned16(B) || !aligned16(C)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    (void)cfg();
    // Every shape the fp16 two-piece kernels take was routed to them by the caller (launch_nt_auto / ext_linear); what arrives here is
    // the remainder (K % 32 != 0, no workspace, > 2^32-byte operands) for the generic exact-fp32 128x128 kernel.
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const int grid = kNumXCD * ((tiles_m + kNumXCD - 1) / kNumXCD) * tiles_n;
    note_fallback_launch();
    hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3(grid), dim3(256), NT_SMEM, st, A, lda, B, ldb, C, ldc, (int)M, (int)N,
                       (int)K, bias, es, addend, mask_src, tiles_m, tiles_n);
    return check_launch(what);
}

struct WgradPlan { int nsplit, rows_per_split, tiles_i, tiles_j; };
// Split-M plan. All tiles of one split run on one XCD (split s -> XCD s % 8) so the dY / X panels of
// that split are fetched once into that XCD's L2. For the CUs to finish together every XCD must get
// the same number of blocks and that number must be a multiple of its 32 CUs:
//   tiles * splits_per_xcd % 32 == 0, with >= 2 blocks per CU when the reduction is long enough.
static WgradPlan wgrad_plan(int64_t M, int64_t N, int64_t K) {
    WgradPlan p;
    p.tiles_i = (int)((N + BM - 1) / BM);
    p.tiles_j = (int)((K + BN - 1) / BN);
    const int tiles = p.tiles_i * p.tiles_j;
    const int64_t max_splits = (M + 255) / 256;            // >= 8 reduction steps per split
    int best = 1;
    if (max_splits >= kNumXCD) {
        int spx = 1;                                        // splits per XCD
        while ((tiles * spx) % 32 != 0 && spx < 32) ++spx; // smallest balanced count
        const int unit = spx;
        while (tiles * spx < 64 && (int64_t)(spx + unit) * kNumXCD <= max_splits) spx += unit;   // >= 2 blocks / CU
        if ((int64_t)spx * kNumXCD > max_splits) spx = (int)(max_splits / kNumXCD);
        if (spx < 1) spx = 1;
        best = spx * kNumXCD;
    } else {
        best = (int)max_splits;
    }
    int64_t rps = (M + best - 1) / best;
    rps = (rps + BK - 1) / BK * BK;
    p.rows_per_split = (int)rps;
    p.nsplit = (int)((M + rps - 1) / rps);
    return p;
}

}  // namespace toad

using namespace toad;

extern "C" size_t toad_linear_ws_bytes(int64_t M, int64_t N, int64_t K) {
    (void)M;
    // one 256x256 fp32 slab per persistent block (64 MiB) + room for the weight planes (6 bytes per element reserved since ABI 1; the
    // two fp16 planes take 4)
    // K rounded up to an EVEN number of 32-deep stages: the streamed narrow kernels walk stages in pairs and their planes carry a zero
    // stage when the count is odd (at K = 32 that doubles the planes: 4 x 64 bytes per weight row > the 6 x 32 reserved before)
    const size_t tiles_n = (size_t)((N + PB - 1) / PB), kpad = (size_t)((K + 2 * BK - 1) / (2 * BK) * (2 * BK));
    // (the h2 path needs 4 bytes per weight element + inverse scales + the abs-max array of A; the 6-byte planes of the older split cover it)
    return (size_t)PB_GRID * PB * PB * sizeof(float) + tiles_n * PB * kpad * 6 + tiles_n * PB * sizeof(float) +
           (size_t)(h2_nblk(M > 0 ? M : 1) + 64) * sizeof(float) + 256 + PB_GRID * sizeof(int);     // ... + the K-slice exponents of a self-measured operand
}

extern "C" int toad_linear_h2_ok(int64_t M, int64_t N, int64_t K) { return h2_nt_ok(M, N, K, K, N) ? 1 : 0; }
static bool tn_big_ok(int64_t M, int64_t N, int64_t K);
// fp16 bags: both the first Linear (NT, A = bag) and its weight gradient (TN, B = bag) must take the fp16 two-piece kernels
extern "C" int toad_mil_x16_ok(int64_t N) { return (N >= 64 && h2_nt_ok(N, 512, 1024, 1024, 512) && tn_big_ok(N, 512, 1024)) ? 1 : 0; }   // (tiny fp16 bags are up-cast)
extern "C" size_t toad_relu_bits_bytes(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)((M + PB - 1) / PB) * (size_t)((N + PB - 1) / PB) * 8 * 2 * 64 * sizeof(unsigned long long);   // 8 KB per 256 x 256 tile
}
extern "C" size_t toad_amax_floats(int64_t rows) { return (size_t)h2_nblk(rows > 0 ? rows : 1); }

extern "C" int toad_absmax_rows256_f32(const float *X, int64_t M, int64_t K, float *amax, void *stream) {
    const char *what = "toad_absmax_rows256_f32";
    if (!X || !amax || M <= 0 || K <= 0 || K % 4 != 0) { set_error("%s: bad argument", what); return TOAD_EINVAL; }
    if (M > INT32_MAX - 256) { set_error("%s: M too large", what); return TOAD_ESHAPE; }
    if (!aligned16(X)) { set_error("%s: X must be 16-byte aligned", what); return TOAD_EALIGN; }
    return launch_absmax(X, K, M, K, amax, true, (hipStream_t)stream, what);
}

static int check_ws(void *ws, size_t ws_bytes, int64_t M, int64_t N, int64_t K, const char *what) {
    if (ws && ws_bytes < toad_linear_ws_bytes(M, N, K)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    if (ws && !aligned16(ws)) { set_error("%s: workspace must be 16-byte aligned", what); return TOAD_EALIGN; }
    return TOAD_OK;
}

extern "C" int toad_linear_act_fwd_f32(const float *X, const float *W, const float *bias, float *Y, int64_t M,
                                        int64_t K, int64_t N, int act, float drop_p, uint64_t drop_seed,
                                        const float *x_amax, float *y_amax, uint64_t *relu_bits_out, void *ws, size_t ws_bytes,
                                        void *stream) {
    const char *what = "toad_linear_act_fwd_f32";
    if (!X || !W || !Y) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    if (int rc = check_ws(ws, ws_bytes, M, N, K, what)) return rc;
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(drop_p, drop_seed)};
    if (relu_bits_out && act != TOAD_ACT_RELU) { set_error("%s: relu_bits_out needs act = RELU", what); return TOAD_EINVAL; }
    return launch_nt_auto(X, K, x_amax, W, K, Y, N, M, N, K, bias, es, nullptr, nullptr, nullptr, H2Pool{nullptr, nullptr, nullptr, 0}, y_amax,
                          reinterpret_cast<unsigned long long *>(relu_bits_out), ws, (hipStream_t)stream, what);
}

// ---- the extractor's GEMMs (models/resnet_custom.py:19-119 as NHWC GEMMs, conv.hip): Y = act(X' W^T + bias + residual) -------------
// x_gmax: device scalar max |X| of the whole activation (NULL: measured here, one extra read); y_gmax: device scalar that receives
// max |Y| by atomic max (the caller zeroed it; NULL: not wanted). One tensor-wide power-of-two scale per activation: gemm_narrow.inc.
static float *ext_scratch_scalar(void *ws, int64_t M, int64_t N, int64_t K) {          // a float inside the workspace's abs-max area
    const size_t tiles_n = (size_t)((N + PB - 1) / PB), kpad = (size_t)((K + 2 * BK - 1) / (2 * BK) * (2 * BK));        // as toad_linear_ws_bytes
    return reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + (size_t)PB_GRID * PB * PB * sizeof(float) + tiles_n * PB * kpad * 6 + tiles_n * PB * sizeof(float)) + 8;
}
int toad::ext_linear(const float *X, const float *x_gmax, const float *W, const float *bias, const float *residual, float *Y, float *y_gmax,
                     int64_t M, int64_t K, int64_t N, int act, void *ws, size_t ws_bytes, hipStream_t st, const char *what) {
    if (!X || !W || !Y) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (residual && (N % 4 != 0 || !aligned16(residual))) { set_error("%s: residual needs N %% 4 == 0 and 16-byte alignment", what); return TOAD_ESHAPE; }
    if (int rc = check_ws(ws, ws_bytes, M, N, K, what)) return rc;
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(0.f, 0)};
    const bool narrow = narrow_ok(M, N, K, N, bias, residual, ws) && (uint64_t)M * K * 4 < (1ull << 32);
    const bool big = !narrow && ws && h2_nt_ok(M, N, K, K, N);
    if ((narrow || big) && !x_gmax) {
        float *g = ext_scratch_scalar(ws, M, N, K);
        if (int rc = launch_gmax(X, M * K, g, st, what)) return rc;
        x_gmax = g;
    }
    if (narrow) {
        const ConvGeom none{0, 0, 0, 0, 0, 0, 0, 0};
        if (N <= 64) return launch_narrow_t<2, 2, GATHER_NONE>(X, K, x_gmax, W, K, Y, N, M, N, K, bias, es.relu, residual, none, y_gmax, ws, st, what);
        return launch_narrow_t<1, 4, GATHER_NONE>(X, K, x_gmax, W, K, Y, N, M, N, K, bias, es.relu, residual, none, y_gmax, ws, st, what);
    }
    if (big) {
        // wide outputs (1x1 expansions / downsamples, the 256-channel 3x3 after im2col): the persistent 256x256 kernel of the MIL path
        // with the tensor-wide scalars in place of its per-block abs-max arrays (stride 0)
        if (!aligned16(X) || !aligned16(W) || !aligned16(Y) || (bias && !aligned16(bias))) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
        char *w = reinterpret_cast<char *>(ws);
        float *slabs = reinterpret_cast<float *>(w);
        w += (size_t)PB_GRID * PB * PB * sizeof(float);
        unsigned short *planes = reinterpret_cast<unsigned short *>(w);
        w += h2_planes_bytes(N, K);
        float *binv = reinterpret_cast<float *>(w);
        const H2Operand op{W, K, 1, N, K, planes, binv};
        if (int rc = launch_split_h2(&op, 1, nullptr, 0, st, what)) return rc;
        // short reductions: a tile is a few k-steps of matrix work followed by 256-512 KB of epilogue traffic, and 256 workgroups in phase make
        // the GEMM the SUM of the two. Spread the starts over one expected tile period (10 ns ticks: ~1.95 us per k-step, ~8 us per 256 KB of
        // epilogue traffic; measured on the extractor's shapes, tools/ab/duo_ext_bench.py: -8..-10 % on the residual GEMMs, neutral from K = 1024)
        if (K <= 512 && M >= 32 * 1024) es.stagger = (int)std::min<int64_t>((K / BK) * 195 + (residual ? 1600 : 800), 3000);
        return launch_nt_h2(X, K, x_gmax, planes, binv, Y, N, M, N, K, bias, es, residual, nullptr, nullptr, H2Pool{nullptr, nullptr, nullptr, 0}, slabs,
                            y_gmax, nullptr, st, what, TOAD_X_F32, 0, 0);
    }
    if (int rc = launch_nt(X, K, W, K, Y, N, M, N, K, bias, es, residual, nullptr, ws, st, what)) return rc;
    return y_gmax ? launch_gmax(Y, M * N, y_gmax, st, what) : TOAD_OK;
}

extern "C" int toad_linear_act_res_fwd_f32(const float *X, const float *W, const float *bias, const float *residual, float *Y,
                                            int64_t M, int64_t K, int64_t N, int act, void *ws, size_t ws_bytes, void *stream) {
    return ext_linear(X, nullptr, W, bias, residual, Y, nullptr, M, K, N, act, ws, ws_bytes, (hipStream_t)stream, "toad_linear_act_res_fwd_f32");
}

int toad::ext_conv_nhwc(const float *X, const float *x_gmax, const float *Wf, const float *bias, const float *residual, float *Y, float *y_gmax, int B,
                        int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, int act, void *ws, size_t ws_bytes, hipStream_t st,
                        const char *what) {
    if (!X || !Wf || !Y || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || pad > 8 || H + 8 >= 32768 || W + 8 >= 32768) { set_error("%s: bad geometry", what); return TOAD_ESHAPE; }
    if (Cin <= 0 || Cin % BK != 0) { set_error("%s: Cin must be a multiple of %d (use toad_im2col_nhwc_f32 + toad_linear_act_res_fwd_f32 otherwise)", what, BK); return TOAD_ESHAPE; }
    if (Cout <= 0 || Cout > 512 || Cout % 4 != 0) { set_error("%s: implicit path needs Cout <= 512, a multiple of 4 (use im2col + linear for wider layers)", what); return TOAD_ESHAPE; }
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho < 1 || Wo < 1) { set_error("%s: empty output", what); return TOAD_ESHAPE; }
    const int64_t M = (int64_t)B * Ho * Wo, K = (int64_t)kh * kw * Cin;
    if ((uint64_t)B * H * W * Cin * 4 >= (1ull << 31) || M >= (1ll << 31)) { set_error("%s: activation too large for 32-bit offsets (split the batch)", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(Wf) || !aligned16(Y)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (int rc = check_ws(ws, ws_bytes, M, Cout, K, what)) return rc;
    if (!narrow_ok(M, Cout <= 128 ? Cout : 128, K, Cout, bias, residual, ws)) { set_error("%s: bias / residual must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (!x_gmax) {
        float *g = ext_scratch_scalar(ws, M, Cout, K);
        if (int rc = launch_gmax(X, (int64_t)B * H * W * Cin, g, st, what)) return rc;
        x_gmax = g;
    }
    const ConvGeom cg{H, W, Cin, Ho, Wo, kw, stride, pad};
    if (Cout <= 64)
        return launch_narrow_t<2, 2, GATHER_CONV>(X, 0, x_gmax, Wf, K, Y, Cout, M, Cout, K, bias, act == TOAD_ACT_RELU, residual, cg, y_gmax, ws, st, what);
    return launch_narrow_t<1, 4, GATHER_CONV>(X, 0, x_gmax, Wf, K, Y, Cout, M, Cout, K, bias, act == TOAD_ACT_RELU, residual, cg, y_gmax, ws, st, what);
}
extern "C" int toad_conv_nhwc_f32(const float *X, const float *Wf, const float *bias, const float *residual, float *Y, int B, int H,
                                  int W, int Cin, int kh, int kw, int stride, int pad, int Cout, int act, void *ws, size_t ws_bytes,
                                  void *stream) {
    return ext_conv_nhwc(X, nullptr, Wf, bias, residual, Y, nullptr, B, H, W, Cin, kh, kw, stride, pad, Cout, act, ws, ws_bytes, (hipStream_t)stream,
                         "toad_conv_nhwc_f32");
}

int toad::ext_stem_conv(const float *Xs, const float *x_gmax, const float *Wf, const float *bias, float *Y, float *y_gmax, int B, int Ho, int Wo, int act,
                        void *ws, size_t ws_bytes, hipStream_t st, const char *what, bool pooled) {
    if (!Xs || !Wf || !Y || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (B <= 0 || Ho <= 0 || Wo <= 0) { set_error("%s: bad geometry", what); return TOAD_ESHAPE; }
    const int Hs = Ho + 3, Ws = Wo + 3;
    const int64_t M = (int64_t)B * Ho * Wo;
    if ((uint64_t)B * Hs * Ws * 48 >= (1ull << 31) || M >= (1ll << 31)) { set_error("%s: batch too large for 32-bit offsets (split it)", what); return TOAD_ESHAPE; }
    if (!aligned16(Xs) || !aligned16(Wf) || !aligned16(Y)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (int rc = check_ws(ws, ws_bytes, M, 64, 192, what)) return rc;
    if (!narrow_ok(M, 64, 192, 64, bias, nullptr, ws)) { set_error("%s: bias must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (!x_gmax) {
        float *g = ext_scratch_scalar(ws, M, 64, 192);
        if (int rc = launch_gmax(Xs, (int64_t)B * Hs * Ws * 12, g, st, what)) return rc;
        x_gmax = g;
    }
    const ConvGeom cg{Hs, Ws, 12, Ho, Wo, 0, 0, 0};
    if (pooled) {
        // the 3x3/2 max-pool in the stem's epilogue (gemm_stream.inc): a tile = two whole image rows, ReLU outputs
        if (!stem_pool_ok(Ho, Wo, act)) { set_error("%s: the pooled stem needs Wo = 128, an even Ho and ReLU", what); return TOAD_ESHAPE; }
        return launch_narrow_t<2, 2, GATHER_STEM_POOL>(Xs, 0, x_gmax, Wf, 192, Y, 64, M, 64, 192, bias, 1, nullptr, cg, y_gmax, ws, st, what);
    }
    return launch_narrow_t<2, 2, GATHER_STEM>(Xs, 0, x_gmax, Wf, 192, Y, 64, M, 64, 192, bias, act == TOAD_ACT_RELU, nullptr, cg, y_gmax, ws, st, what);
}
bool toad::stem_pool_ok(int Ho, int Wo, int act) { return Wo == 128 && Ho > 0 && Ho % 2 == 0 && act == TOAD_ACT_RELU; }
extern "C" int toad_stem_conv_s2d_f32(const float *Xs, const float *Wf, const float *bias, float *Y, int B, int Ho, int Wo, int act,
                                      void *ws, size_t ws_bytes, void *stream) {
    return ext_stem_conv(Xs, nullptr, Wf, bias, Y, nullptr, B, Ho, Wo, act, ws, ws_bytes, (hipStream_t)stream, "toad_stem_conv_s2d_f32");
}
// The stem AND nn.MaxPool2d(3, 2, 1) as one kernel (models/resnet_custom.py:96-99): Yp [B, Ho/2, Wo/2, 64]. Shapes: Wo = 128, Ho a multiple of 32
// (256 x 256 tiles); others take toad_stem_conv_s2d_f32 + toad_maxpool3x3s2_nhwc_f32.
extern "C" int toad_stem_conv_pool_s2d_f32(const float *Xs, const float *Wf, const float *bias, float *Yp, int B, int Ho, int Wo, void *ws, size_t ws_bytes,
                                           void *stream) {
    return ext_stem_conv(Xs, nullptr, Wf, bias, Yp, nullptr, B, Ho, Wo, TOAD_ACT_RELU, ws, ws_bytes, (hipStream_t)stream, "toad_stem_conv_pool_s2d_f32", true);
}

// The stem + ReLU + 3x3/2 max-pool straight from NCHW tiles (stem_halo.inc): no space-to-depth image, no fragment loads from global memory.
bool toad::stem_nchw_pool_ok(int H, int W) { return W == 256 && H >= 4 && H % 4 == 0; }
int toad::ext_stem_nchw_pool(const float *X, const float *Wf, const float *bias, float *Yp, float *y_gmax, int B, int H, int W, void *ws, size_t ws_bytes,
                             hipStream_t st, const char *what) {
    if (!X || !Wf || !Yp || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || !stem_nchw_pool_ok(H, W)) { set_error("%s: needs W = 256 and H %% 4 == 0 (other tiles: toad_stem_s2d_nchw_f32 + toad_stem_conv_s2d_f32 + toad_maxpool3x3s2_nhwc_f32)", what); return TOAD_ESHAPE; }
    const int64_t M = (int64_t)B * (H / 2) * 128;
    if (M >= (1ll << 31)) { set_error("%s: batch too large for 32-bit offsets (split it)", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(Wf) || !aligned16(Yp) || (bias && !aligned16(bias))) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (int rc = check_ws(ws, ws_bytes, M, 64, 192, what)) return rc;
    (void)cfg();
    char *w = reinterpret_cast<char *>(ws) + (size_t)PB_GRID * PB * PB * sizeof(float);
    unsigned short *planes = reinterpret_cast<unsigned short *>(w);
    float *binv = reinterpret_cast<float *>(w + (size_t)64 * 6 * BK * 4);
    hipLaunchKernelGGL(split_planes_narrow_h2_kernel<2>, dim3(16), dim3(256), 0, st, Wf, (int64_t)192, planes, binv, 64, 192, 1, 1, 12, 6);
    if (int rc = check_launch(what)) return rc;
    const int tiles = B * (H / 4);
    hipLaunchKernelGGL(stem_halo_pool_kernel, dim3(std::min(tiles, 2 * PB_GRID)), dim3(256), SH_SMEM, st, X, planes, binv, bias, Yp, B, H, y_gmax, tiles);      // two workgroups per CU
    return check_launch(what);
}
extern "C" int toad_stem_pool_nchw_f32(const float *X, const float *Wf, const float *bias, float *Yp, int B, int H, int W, void *ws, size_t ws_bytes, void *stream) {
    return ext_stem_nchw_pool(X, Wf, bias, Yp, nullptr, B, H, W, ws, ws_bytes, (hipStream_t)stream, "toad_stem_pool_nchw_f32");
}

extern "C" int toad_linear_dgrad_f32(const float *dY, const float *WT, const float *addend, const float *relu_src,
                                      float mask_scale, float *dX, int64_t M, int64_t N, int64_t K,
                                      const float *pool_a_raw, const float *pool_stats, const float *pool_dM, int pool_T,
                                      const float *dy_amax, float *dx_amax, const uint64_t *relu_bits, void *ws, size_t ws_bytes,
                                      void *stream) {
    const char *what = "toad_linear_dgrad_f32";
    if (!dY || !WT || !dX) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (pool_T < 0 || pool_T > 2 || (pool_T > 0 && (!pool_a_raw || !pool_stats || !pool_dM))) { set_error("%s: bad pooling-addend arguments", what); return TOAD_EINVAL; }
    if (pool_T > 0 && (!aligned16(pool_dM) || (pool_T == 2 && ((uintptr_t)pool_a_raw & 7)))) { set_error("%s: pooling-addend pointers misaligned", what); return TOAD_EALIGN; }
    if (int rc = check_ws(ws, ws_bytes, M, K, N, what)) return rc;
    // dX[M,K] = dY[M,N] . WT[K,N]^T : an NT product with reduction dim N
    EpiScalars es{0, mask_scale, make_drop(0.f, 0)};
    if (relu_bits && (!relu_src || !h2_nt_ok(M, K, N, N, K) || !ws)) { set_error("%s: relu_bits needs relu_src, a workspace and the h2 kernel", what); return TOAD_EINVAL; }
    return launch_nt_auto(dY, N, dy_amax, WT, N, dX, K, M, K, N, nullptr, es, addend, relu_src, reinterpret_cast<const unsigned long long *>(relu_bits),
                          H2Pool{pool_a_raw, pool_stats, pool_dM, pool_T}, dx_amax, nullptr, ws, (hipStream_t)stream, what);
}

// (round 6: any number of rows - until then bags below 64 patches took the exact-fp32 128 x 128 kernel; the staging clamps its row reads and zeroes
//  rows beyond the operand, so a one-row bag is one 32-row stage like the last stage of any other)
static bool tn_big_ok(int64_t M, int64_t N, int64_t K) { return M >= 1 && N >= 4 && K >= 4; }

extern "C" size_t toad_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const WgradPlan p = wgrad_plan(M, N, K);
    const TnPlan q = tn_plan(M, N, K);
    const int ns = p.nsplit > q.nsplit ? p.nsplit : q.nsplit;
    // slabs (+ bias slabs), then: 2 scale floats, 2 abs-max arrays (operands measured here when the caller has none)
    return (size_t)ns * (size_t)(N * K + N) * sizeof(float) + (size_t)(2 * h2_nblk(M) + 64) * sizeof(float);
}

// dW = beta*dW + dY^T X (+ db). dy_amax / x_amax: abs-max arrays of the operands (NULL -> measured here).
int toad::launch_wgrad(const float *dY, const float *dy_amax, const float *X, const float *x_amax, float *dW, float *db, int64_t M,
                        int64_t N, int64_t K, float beta, void *ws, hipStream_t st, const char *what, int x_mode, WgradDeferred *defer) {
    // x_mode: TOAD_X_F32 = fp32 [M][K]; TOAD_X_F16 = fp16 [M][K]; TOAD_X_PT = plane-tiled (gemm_pt.inc; x_amax = its per-row-tile abs-max)
    if (x_mode != TOAD_X_F32 && !tn_big_ok(M, N, K)) { set_error("%s: an fp16 / plane-tiled input operand needs the h2 wgrad kernels", what); return TOAD_ESHAPE; }
    if (x_mode == TOAD_X_PT && !x_amax) { set_error("%s: a plane-tiled operand comes with its abs-max array", what); return TOAD_EINVAL; }
    float *slab = (float *)ws;
    int nsplit;
    int rc;
    bool h2 = false;
    float *scales = nullptr;
    if (tn_big_ok(M, N, K)) {
        const TnPlan q = tn_plan(M, N, K);
        nsplit = q.nsplit;
        float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
        {
            h2 = true;
            scales = slab + (size_t)nsplit * (size_t)(N * K + N);
            float *amax_ws = scales + 16;
            if (!dy_amax) { if ((rc = launch_absmax(dY, N, M, N, amax_ws, true, st, what))) return rc; dy_amax = amax_ws; }
            if (x_mode == TOAD_X_PT) {
                hipLaunchKernelGGL(gemm_tn_pt_kernel, dim3(PB_GRID), dim3(512), TP_SMEM, st, dY, N, dy_amax, reinterpret_cast<const unsigned short *>(X),
                                   x_amax, (int)pt_col_stages(K), slab, cs, scales, (int)M, (int)N, (int)K, q.rows_per_split, q.ti, q.tj, q.nsplit);
            } else if (x_mode == TOAD_X_F16) {
                hipLaunchKernelGGL(gemm_tn_h2_big_kernel<true>, dim3(PB_GRID), dim3(512), TN2_SMEM, st, dY, N, dy_amax, X, K, (const float *)nullptr, slab, cs,
                                   scales, (int)M, (int)N, (int)K, q.rows_per_split, q.ti, q.tj, q.nsplit);
            } else {
                if (!x_amax) { float *a2 = amax_ws + h2_nblk(M) + 16; if ((rc = launch_absmax(X, K, M, K, a2, true, st, what))) return rc; x_amax = a2; }
                hipLaunchKernelGGL(gemm_tn_h2_big_kernel<false>, dim3(PB_GRID), dim3(512), TN2_SMEM, st, dY, N, dy_amax, X, K, x_amax, slab, cs, scales,
                                   (int)M, (int)N, (int)K, q.rows_per_split, q.ti, q.tj, q.nsplit);
            }
        }
        rc = check_launch(what);
    } else {
        const WgradPlan p = wgrad_plan(M, N, K);
        nsplit = p.nsplit;
        float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
        const int tiles = p.tiles_i * p.tiles_j;
        const int grid = kNumXCD * ((p.nsplit + kNumXCD - 1) / kNumXCD) * tiles;
        note_fallback_launch();
        hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(grid), dim3(256), TN_SMEM, st, dY, N, X, K, slab, cs, (int)M, (int)N,
                           (int)K, p.rows_per_split, p.tiles_i, p.tiles_j, p.nsplit);
        rc = check_launch(what);
    }
    if (rc) return rc;
    float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
    const int64_t n = N * K, n2 = db ? N : 0;
    int rgrid = (int)(((n + n2) / 4 + 255) / 256);
    if (rgrid > 4096) rgrid = 4096;
    if (h2 && defer) {                 // the caller reduces this and its other weight gradients in one launch (launch_wgrad_reduce)
        *defer = WgradDeferred{slab, dW, n, cs, db, n2, nsplit, beta, scales};
        return TOAD_OK;
    }
    if (defer) defer->slab = nullptr;  // reduced here (not the h2 path)
    if (h2)
        hipLaunchKernelGGL(slab_reduce_h2_kernel, dim3(rgrid), dim3(256), 0, st, slab, dW, n, cs, db, n2, nsplit, beta, scales);
    else
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(rgrid), dim3(256), 0, st, slab, dW, n, cs, db, n2, nsplit, beta);
    return check_launch(what);
}

// ---- up to three weight gradients of one backward pass in ONE launch (gemm_tn_h2_batch_kernel, gemm_h2.inc) ----------------------------------
// Plan: as many row splits as keep every item on its own workgroup (PB_GRID / tiles of all products), each at least four 32-row stages deep for
// bags that have the rows (one stage for tiny ones), all of the same depth.
struct TnBatchPlan { int nsplit, rows_per_split, tiles_all; };
static TnBatchPlan tn_batch_plan(int64_t M, const WgradJob *jobs, int n) {
    TnBatchPlan p{0, 0, 0};
    for (int i = 0; i < n; ++i) p.tiles_all += (int)(((jobs[i].N + PB - 1) / PB) * ((jobs[i].K + PB - 1) / PB));
    if (p.tiles_all <= 0 || p.tiles_all > PB_GRID) return p;
    int64_t s = PB_GRID / p.tiles_all;
    const int64_t max_splits = M >= 8192 ? (M + 127) / 128 : (M + 31) / 32;
    if (s > max_splits) s = max_splits;
    if (s < 1) s = 1;
    int64_t rps = (M + s - 1) / s;
    rps = (rps + BK - 1) / BK * BK;
    p.rows_per_split = (int)rps;
    p.nsplit = (int)((M + rps - 1) / rps);
    return p;
}
bool toad::wgrad_batch_ok(int64_t M, const WgradJob *jobs, int n, size_t ws_bytes_each) {
    if (n < 2 || n > 3 || M < 64 || M > kTnBatchMaxRows) return false;
    const TnBatchPlan p = tn_batch_plan(M, jobs, n);
    if (p.nsplit < 1) return false;
    for (int i = 0; i < n; ++i) {
        if (!jobs[i].dY || !jobs[i].X || !jobs[i].dy_amax || !jobs[i].x_amax || !jobs[i].dW || !jobs[i].ws || !tn_big_ok(M, jobs[i].N, jobs[i].K)) return false;
        const size_t need = (size_t)p.nsplit * (size_t)(jobs[i].N * jobs[i].K + jobs[i].N) * sizeof(float) + (size_t)(2 * h2_nblk(M) + 64) * sizeof(float);
        if (need > ws_bytes_each) return false;
    }
    return true;
}
int toad::launch_wgrad_batch(const WgradJob *jobs, int n, int64_t M, float beta, hipStream_t st, const char *what, WgradDeferred *defer) {
    const TnBatchPlan p = tn_batch_plan(M, jobs, n);
    (void)cfg();
    const float *A[3], *aam[3], *B[3], *bam[3]; float *slab[3], *cs[3], *sc[3]; int I[3], J[3];
    for (int i = 0; i < 3; ++i) {
        const WgradJob &j = jobs[i < n ? i : 0];
        A[i] = j.dY; aam[i] = j.dy_amax; B[i] = j.X; bam[i] = j.x_amax; I[i] = (int)j.N; J[i] = (int)j.K;
        slab[i] = reinterpret_cast<float *>(j.ws);
        cs[i] = j.db ? slab[i] + (size_t)p.nsplit * j.N * j.K : nullptr;
        sc[i] = slab[i] + (size_t)p.nsplit * (size_t)(j.N * j.K + j.N);
        if (i < n) defer[i] = WgradDeferred{slab[i], j.dW, j.N * j.K, cs[i], j.db, j.db ? j.N : 0, p.nsplit, beta, sc[i]};
    }
    hipLaunchKernelGGL(gemm_tn_h2_batch_kernel, dim3(PB_GRID), dim3(512), TN2_SMEM, st, A[0], aam[0], B[0], bam[0], slab[0], cs[0], sc[0], I[0], J[0],
                       A[1], aam[1], B[1], bam[1], slab[1], cs[1], sc[1], I[1], J[1], A[2], aam[2], B[2], bam[2], slab[2], cs[2], sc[2], I[2], J[2], n, (int)M,
                       p.rows_per_split, p.nsplit);
    return check_launch(what);
}

int toad::launch_wgrad_reduce(const WgradDeferred *d, int count, hipStream_t st, const char *what) {
    SlabReduceBatch b{};
    int64_t e0 = 0;
    for (int i = 0; i < count && b.count < 3; ++i) {
        if (!d[i].slab) continue;
        b.d[b.count] = SlabReduceDesc{d[i].slab, d[i].out, d[i].n, d[i].slab2, d[i].out2, d[i].n2, d[i].nsplit, d[i].beta, d[i].scales, e0};
        e0 += (d[i].n + d[i].n2) / 4;
        ++b.count;
    }
    if (b.count == 0) return TOAD_OK;
    b.total4 = e0;
    int grid = (int)((e0 + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(slab_reduce_h2_batch_kernel, dim3(grid), dim3(256), 0, st, b);
    return check_launch(what);
}

extern "C" int toad_linear_wgrad_f32(const float *dY, const float *X, float *dW, float *db, int64_t M, int64_t N,
                                      int64_t K, float beta, const float *dy_amax, const float *x_amax, void *ws, size_t ws_bytes,
                                      void *stream) {
    const char *what = "toad_linear_wgrad_f32";
    if (!dY || !X || !dW || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (N % 4 != 0 || K % 4 != 0) { set_error("%s: N and K must be multiples of 4", what); return TOAD_ESHAPE; }
    if (M > INT32_MAX - 4096) { set_error("%s: M too large", what); return TOAD_ESHAPE; }
    if (!aligned16(dY) || !aligned16(X) || !aligned16(dW) || !aligned16(ws)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_linear_wgrad_ws_bytes(M, N, K)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    return launch_wgrad(dY, dy_amax, X, x_amax, dW, db, M, N, K, beta, ws, (hipStream_t)stream, what);
}

extern "C" int toad_linear_wgrad_xp_f32(const float *dY, const void *Xp, const float *x_amax, float *dW, float *db, int64_t M, int64_t N,
                                         int64_t K, float beta, const float *dy_amax, void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_linear_wgrad_xp_f32";
    if (!dY || !Xp || !x_amax || !dW || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (N % 4 != 0 || K % 8 != 0) { set_error("%s: N must be a multiple of 4, K of 8", what); return TOAD_ESHAPE; }
    if (M > INT32_MAX - 4096) { set_error("%s: M too large", what); return TOAD_ESHAPE; }
    if (!aligned16(dY) || !aligned16(Xp) || !aligned16(dW) || !aligned16(ws)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_linear_wgrad_ws_bytes(M, N, K)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    return launch_wgrad(dY, dy_amax, reinterpret_cast<const float *>(Xp), x_amax, dW, db, M, N, K, beta, ws, (hipStream_t)stream, what, TOAD_X_PT);
}

extern "C" int toad_transpose_f32(const float *in, float *out, int64_t rows, int64_t cols, void *stream) {
    if (!in || !out || rows <= 0 || cols <= 0) { set_error("toad_transpose_f32: bad argument"); return TOAD_EINVAL; }
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)rows, (int)cols);
    return check_launch("toad_transpose_f32");
}

// mask[e] = 0 or 1/(1-p) for flat element e (the multiplier the kernels apply): test/debug helper
__global__ __launch_bounds__(256) void dropout_mask_kernel(float *out, int64_t n, DropArgs d) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        out[e] = d.thresh ? drop_keep((uint64_t)e, d) : 1.f;
}
extern "C" int toad_dropout_mask_f32(float *out, int64_t n, float drop_p, uint64_t drop_seed, void *stream) {
    if (!out || n <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) { set_error("toad_dropout_mask_f32: bad argument"); return TOAD_EINVAL; }
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, make_drop(drop_p, drop_seed));
    return check_launch("toad_dropout_mask_f32");
}

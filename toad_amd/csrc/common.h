// common.h — shared helpers for the gfx950 kernels of libtoad_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "toad_hip.h"

namespace toad {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;      // gfx950 wavefront
constexpr int kNumXCD = 8;     // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)

void set_error(const char *fmt, ...);
int check_launch(const char *what);
void note_fallback_launch();      // an exact-fp32 fallback GEMM kernel was launched (toad_fallback_launches, capi.hip)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- 16-lane ("DPP row") all-reduce: every lane of each 16-lane row ends with the row's sum.
// quad_perm[1,0,3,2] -> quad_perm[2,3,0,1] -> row_half_mirror -> row_mirror. All four are
// single-instruction DPP modifiers on gfx9, no LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_allreduce_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// ---- stateless dropout (nn.Dropout(0.25) of models/model_toad.py:27-29,61,64 in train mode) ----------
// keep(element) = hash(seed, flat element index) >= p * 2^32 ; kept values are scaled by 1/(1-p).
// The mask is never stored: backward recomputes it from (seed, index).
struct DropArgs { uint64_t seed; uint32_t thresh; float scale; };   // thresh == 0 -> dropout off
__host__ __device__ inline DropArgs make_drop(float p, uint64_t seed) {
    DropArgs d;
    d.seed = seed;
    d.thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
    d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    return d;
}
__device__ __forceinline__ uint32_t drop_hash(uint64_t idx, uint64_t seed) {
    uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
    const uint32_t y = (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32);
    x *= 0x9E3779B1u; x ^= x >> 15; x += y * 0x85EBCA77u;
    x *= 0xC2B2AE3Du; x ^= x >> 13; x *= 0x27D4EB2Fu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_keep(uint64_t idx, const DropArgs &d) {   // 0 or 1/(1-p)
    return drop_hash(idx, d.seed) >= d.thresh ? d.scale : 0.f;
}

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// ---- abs-max arrays (csrc/gemm_h2.inc): amax[b] = max |x| over rows [256 b, 256 b + 256) of a tensor, stored as the bit
// pattern of a non-negative float so that an unsigned atomic max orders it. Producers (GEMM epilogues, pooling backward)
// fill it; the fp16 two-piece GEMMs derive their power-of-two operand scales from it.
constexpr int H2_ROWBLK = 256;
__device__ __forceinline__ float h2_wave_max(float v) {          // wave-wide max of non-negative floats
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    return v;
}
// the same without lane-index registers (ds_bpermute needs one VGPR per shuffle distance: inside a GEMM main loop that runs at the register
// limit they spill): DPP inside the 16-lane rows, then the four row results through scalar registers. Every lane returns the wave maximum.
__device__ __forceinline__ float h2_wave_max_dpp(float v) {
    v = __builtin_fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = __builtin_fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = __builtin_fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = __builtin_fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}
__device__ __forceinline__ void h2_atomic_amax(float *slot, float v) {     // v >= 0
    atomicMax(reinterpret_cast<unsigned *>(slot), __builtin_bit_cast(unsigned, v));
}
__device__ __forceinline__ float h2_absmax4(f32x4 v) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                           __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
}
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// streaming store: the line is not kept dirty in L2 for the next kernel to flush (GEMM outputs are consumed by the NEXT launch)
__device__ __forceinline__ void st4s(float *p, f32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
}

// how the bag (the A operand of the first Linear, the B operand of its weight gradient) lies in memory
enum { TOAD_X_F32 = 0,      // fp32 [N][1024]
       TOAD_X_F16 = 1,      // fp16 [N][1024] (feature stores kept in half precision)
       TOAD_X_PT = 2 };     // plane-tiled two-piece form written by toad_bag_prepare_f32 (csrc/gemm_pt.inc)

// ---- host-side launchers of the fp16 two-piece GEMMs (defined in gemm_f32.hip next to their kernels; used by step.hip) ----
struct H2Operand { const float *src; int64_t sn, sk; int64_t N, K; unsigned short *planes; float *binv; };   // B[n,k] = src[n*sn + k*sk]
struct H2Pool { const float *a_raw, *stats, *dM; int T; };                                                   // recomputed pooling addend (T = 0: none)
struct EpiScalars { int relu; float mask_scale; DropArgs drop; int stagger = 0; };                           // epilogue scalars of every NT kernel; stagger: gemm_h2.inc
bool h2_nt_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc);
bool nt_half_tiles(int64_t M, int64_t N);               // fp32-A products of this shape run on half-height tiles (no K-split: every tile writes / reads the ReLU bit image)
bool nt_run_ok(int64_t M, int64_t N, int64_t K);       // the first GEMM may measure its fp32 A operand itself (gemm_h2.inc AMODE 3)
size_t h2_planes_bytes(int64_t N, int64_t K);
size_t h2_binv_bytes(int64_t N);
size_t h2_slab_bytes();
int launch_split_h2(const H2Operand *ops, int n, float *zero, int zero_n, hipStream_t st, const char *what, float *zero2 = nullptr,
                    int zero2_n = 0);                                                                            // also zeroes zero[0..zero_n) and zero2[0..zero2_n)
int launch_absmax(const float *X, int64_t ld, int64_t M, int64_t K, float *amax, bool zero, hipStream_t st, const char *what);
int launch_nt_h2(const float *A, int64_t lda, const float *a_amax, const unsigned short *planes, const float *binv, float *C,
                 int64_t ldc, int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend,
                 const float *mask_src, const unsigned long long *mask_bits, H2Pool pool, float *slabs, float *y_amax,
                 unsigned long long *bits_out, hipStream_t st, const char *what, int a_mode = TOAD_X_F32, int a_stride = 1, int y_stride = 1,
                 float *a_amax_out = nullptr, int *slab_ke = nullptr);   // a_amax == NULL + a_amax_out (zeroed) + slab_ke [256]: A is measured inside the GEMM
size_t pt_bytes_host(int64_t rows, int64_t cols);                                                          // bytes of a plane-tiled tensor
int launch_pt_split(const float *X, int64_t ld, int64_t M, int64_t K, const float *amax, unsigned short *pt, hipStream_t st, const char *what);
int launch_pool_bwd(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *A_raw, const float *stats,
                    const float *M, const float *dM, const float *dA_ext, float *dPa, float *dPb, int64_t ldd, float *dH, float *dWc, float *dbc,
                    float beta, float *dp_amax, bool zero_amax, void *ws, size_t ws_bytes, int64_t N, int L, int D, int T, float drop_p,
                    uint64_t seed_a, uint64_t seed_b, hipStream_t st);
// the extractor's GEMMs with tensor-wide abs-max scalars threaded from producer to consumer (gemm_f32.hip; used by conv.hip)
int launch_gmax(const float *x, int64_t n, float *out, hipStream_t st, const char *what);                   // out[0] = max |x| (zeroed here)
int ext_linear(const float *X, const float *x_gmax, const float *W, const float *bias, const float *residual, float *Y, float *y_gmax,
               int64_t M, int64_t K, int64_t N, int act, void *ws, size_t ws_bytes, hipStream_t st, const char *what);
int ext_conv_nhwc(const float *X, const float *x_gmax, const float *Wf, const float *bias, const float *residual, float *Y, float *y_gmax, int B,
                  int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, int act, void *ws, size_t ws_bytes, hipStream_t st,
                  const char *what);
int ext_stem_conv(const float *Xs, const float *x_gmax, const float *Wf, const float *bias, float *Y, float *y_gmax, int B, int Ho, int Wo, int act,
                  void *ws, size_t ws_bytes, hipStream_t st, const char *what, bool pooled = false);
bool stem_nchw_pool_ok(int H, int W);                 // the stem + pool may run straight from the NCHW tiles (stem_halo.inc)
int ext_stem_nchw_pool(const float *X, const float *Wf, const float *bias, float *Yp, float *y_gmax, int B, int H, int W, void *ws, size_t ws_bytes,
                       hipStream_t st, const char *what);
bool stem_pool_ok(int Ho, int Wo, int act);           // the stem may take the 3x3/2 max-pool into its epilogue (gemm_stream.inc)
// batched pooling launches of the ragged multi-slide step (gated_pool.hip): blockIdx.y = slide, row ranges from the DEVICE array seg_dev [B+1]
size_t pool_batch_ws_bytes(int B, int L, int D, int T);
int launch_pool_fwd_batch(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *bc, float *A_raw, float *M,
                          int m_stride, float *stats, int s_stride, void *ws, const int64_t *seg_dev, int B, int64_t max_n, int L, int D, int T,
                          float drop_p, uint64_t seed_a, uint64_t seed_b, hipStream_t st);
int launch_pool_bwd_batch(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *A_raw, const float *stats,
                          int s_stride, const float *M, const float *dM, int m_stride, float *dPa, float *dPb, int64_t ldd, float *dH, float *dWc,
                          float *dbc, float beta, void *ws, const int64_t *seg_dev, int B, int64_t max_n, int L, int D, int T, float drop_p,
                          uint64_t seed_a, uint64_t seed_b, hipStream_t st, float *dp_amax = nullptr, float *row_bound = nullptr, int64_t n_rows = 0);
// batched heads + weighted CE + heads backward of the ragged multi-slide step (heads.hip): one workgroup per slide, then the head-weight gradients
struct HeadsBatch {          // per-slide records, all with the same byte stride `rec` (slide b of array p: (char *)p + b * rec)
    const float *M; float *Mcat, *logits, *yprob; int64_t *yhat; float *slog, *sprob; int64_t *shat; float *dM, *dl, *ds; size_t rec;
};
int launch_heads_batch(const HeadsBatch &hb, const float *sex, const float *Wcls, const float *bcls, const float *Wsite, const float *bsite,
                       const int64_t *label, const int64_t *site, float w_cls, float w_site, float *loss_out, float *dWcls, float *dbcls,
                       float *dWsite, float *dbsite, float beta, int B, int L, int C, hipStream_t st);
struct WgradDeferred;
int launch_wgrad(const float *dY, const float *dy_amax, const float *X, const float *x_amax, float *dW, float *db, int64_t M,
                 int64_t N, int64_t K, float beta, void *ws, hipStream_t st, const char *what, int x_mode = TOAD_X_F32,
                 struct WgradDeferred *defer = nullptr);
// a weight gradient whose slab reduction was deferred (launch_wgrad with `defer`): up to three are reduced by ONE launch_wgrad_reduce
struct WgradDeferred { const float *slab; float *out; int64_t n; const float *slab2; float *out2; int64_t n2; int nsplit; float beta; const float *scales; };
int launch_wgrad_reduce(const WgradDeferred *d, int count, hipStream_t st, const char *what);
// two or three weight gradients over the SAME M rows in one launch (gemm_tn_h2_batch_kernel): fp32 operands with their abs-max arrays, one slab
// area (toad_linear_wgrad_ws_bytes) each; always deferred - the caller reduces them with launch_wgrad_reduce
constexpr int64_t kTnBatchMaxRows = 262144;
struct WgradJob { const float *dY, *dy_amax, *X, *x_amax; float *dW, *db; int64_t N, K; void *ws; };
bool wgrad_batch_ok(int64_t M, const WgradJob *jobs, int n, size_t ws_bytes_each);
int launch_wgrad_batch(const WgradJob *jobs, int n, int64_t M, float beta, hipStream_t st, const char *what, WgradDeferred *defer);

}  // namespace toad

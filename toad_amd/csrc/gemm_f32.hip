// gemm_f32.hip — exact-fp32 MFMA GEMMs for the Linear layers of TOAD's MIL path (gfx950).
//
// Why fp32 MFMA: parity with the reference's PyTorch-CPU path is 1e-4 on fp32 outputs and
// plain bf16 operands miss it (1e-3..6e-3). gfx950 has no TF32/xf32, but
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at the fp32 vector peak (157.3 TF).
//
// Two kernels cover every product on the path:
//   gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)        both operands reduction-contiguous
//             forward  Y = act(X W^T + b)             models/model_toad.py:59,62,21,25
//             dgrad    dX = (dY (W^T)^T + add)*mask   with WT = W^T materialised once (2 MB)
//   gemm_tn : C[I,J] = sum_m A[m,I] . B[m,J]         both operands reduction-strided
//             wgrad    dW = dY^T X, split over m, deterministic slab reduction
//
// Tiling (both): 128x128 block tile, 32-deep reduction step, 256 threads = 4 waves in 2x2,
// each wave a 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 acc VGPRs). LDS is double
// buffered through registers (global -> VGPR -> LDS) so the next tile's HBM latency hides
// under 64 MFMAs (4096 issue cycles) per wave; one barrier per step.
// The k-order inside a step is permuted (lane-half hi supplies k = 8q+4hi+s) so the NT
// operand fragments are single ds_read_b128's; LDS rows are padded to 36 floats which makes
// those reads conflict-free for the gfx950 ds_read_b128 lane groups.
#include "common.h"

namespace toad {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NT_LD = 36;                                  // padded LDS row (floats), conflict-free b128
constexpr int NT_TILE = BM * NT_LD;                        // floats per operand tile
constexpr int NT_SMEM = 2 * 2 * NT_TILE * (int)sizeof(float);   // 73,728 B -> 2 blocks / CU
constexpr int TN_LD = 128;
constexpr int TN_TILE = BK * TN_LD;
constexpr int TN_SMEM = 2 * 2 * TN_TILE * (int)sizeof(float);   // 65,536 B

// XCD-aware block -> tile map: all column tiles of one row tile run on the same XCD (same L2),
// back to back, so the A panel is fetched from HBM once and re-read from L2.
__device__ __forceinline__ bool map_tile(int tiles_m, int tiles_n, int &tm, int &tn) {
    const int b = blockIdx.x;
    const int xcd = b % kNumXCD, q = b / kNumXCD;
    tm = (q / tiles_n) * kNumXCD + xcd;
    tn = q % tiles_n;
    return tm < tiles_m;
}

// ------------------------------------------------------------------------------------------
// NT: C[M,N] = epi(A[M,K] B[N,K]^T);   epi: +bias[col], +addend[row,col], relu, mask(mask_src>0)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
    float *C, int64_t ldc, int M, int N, int K,
    const float *__restrict__ bias, int relu, const float *addend, const float *__restrict__ mask_src,
    int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int tm, tn;
    if (!map_tile(tiles_m, tiles_n, tm, tn)) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x;
    const int c4 = tid & 7, r0 = tid >> 3;     // staging: float4 column, first row
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;

    const float *ap[4], *bp[4];
    bool aok[4], bok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ra = m0 + r0 + 32 * j, rb = n0 + r0 + 32 * j;
        aok[j] = ra < M;
        bok[j] = rb < N;
        ap[j] = A + (int64_t)(aok[j] ? ra : 0) * lda + c4 * 4;
        bp[j] = B + (int64_t)(bok[j] ? rb : 0) * ldb + c4 * 4;
    }
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#ifdef TOAD_ABLATE_NO_GLOAD      // tools/ubench only: measure the loop without its global loads
        if (k0 > 0) return;
#endif
        const bool kok = (k0 + c4 * 4) < K;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = (aok[j] && kok) ? ld4(ap[j] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef TOAD_ABLATE_NO_GLOAD_B
            if (k0 > 0) continue;
#endif
            rb[j] = (bok[j] && kok) ? ld4(bp[j] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * 2 * NT_TILE, *Bs = As + NT_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st4(As + (r0 + 32 * j) * NT_LD + c4 * 4, ra[j]);
            st4(Bs + (r0 + 32 * j) * NT_LD + c4 * 4, rb[j]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Fragment pipeline: the four 8-deep k-groups of a tile are read one group ahead of the MFMAs
    // that consume them, and the tile barrier sits BEFORE the last group's MFMAs, so the first
    // fragments of the next tile are fetched under 16 MFMAs (1024 issue cycles) as well. A wave
    // therefore never waits on LDS latency with an empty matrix pipe; it parks only for barrier skew.
    struct Frag { f32x4 a0, a1, b0, b1; };
    const int frag_off_a = (wm * 64 + li) * NT_LD + hi * 4;
    const int frag_off_b = NT_TILE + (wn * 64 + li) * NT_LD + hi * 4;
    auto fread = [&](int buf, int q) {
        const float *base = smem + buf * 2 * NT_TILE;
        Frag f;
#ifdef TOAD_ABLATE_NO_FREAD
        if (q >= 0) { asm volatile("" : "=v"(f.a0), "=v"(f.a1), "=v"(f.b0), "=v"(f.b1)); return f; }
#endif
        f.a0 = ld4(base + frag_off_a + q * 8);
        f.a1 = ld4(base + frag_off_a + 32 * NT_LD + q * 8);
        f.b0 = ld4(base + frag_off_b + q * 8);
        f.b1 = ld4(base + frag_off_b + 32 * NT_LD + q * 8);
        return f;
    };
    auto mma16 = [&](const Frag &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b1[s], acc[1][1], 0, 0, 0);
        }
    };

    const int nk = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    Frag f0 = fread(0, 0);
    for (int t = 0; t < nk; ++t) {
        const bool more = (t + 1) < nk;
        const int buf = t & 1;
        if (more) gload((t + 1) * BK);
        // sched_barrier(0) pins "issue the next group's ds_reads, THEN this group's 16 MFMAs":
        // left alone, hipcc sinks the reads to just before their first use to save 16 VGPRs.
        Frag f1 = fread(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f0);
        __builtin_amdgcn_sched_barrier(0);
        Frag f2 = fread(buf, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f1);
        __builtin_amdgcn_sched_barrier(0);
        Frag f3 = fread(buf, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f2);
        __builtin_amdgcn_sched_barrier(0);
#ifndef TOAD_ABLATE_NO_LSTORE
        if (more) lstore(buf ^ 1);
#endif
#ifndef TOAD_ABLATE_NO_BARRIER
        __syncthreads();
#endif
        if (more) f0 = fread(buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f3);
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: acc reg r of lane (li,hi) is row (r&3)+8*(r>>2)+4*hi, column li of the 32x32 tile.
    // Rows are clamped (not branched) for the addend/mask loads so all 16 loads of a sub-tile are
    // issued back to back; only the stores are predicated.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn * 64 + b * 32 + li;
        const bool cok = col < N;
        const int colc = cok ? col : N - 1;
        const float bv = bias ? bias[colc] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rbase = m0 + wm * 64 + a * 32 + 4 * hi;
            float add[16], msk[16];
            if (addend) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
                    add[r] = addend[(int64_t)row * ldc + colc];
                }
            }
            if (mask_src) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
                    msk[r] = mask_src[(int64_t)row * ldc + colc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[a][b][r] + bv;
                if (addend) v += add[r];
                if (relu) v = v > 0.f ? v : 0.f;
                if (mask_src) v = msk[r] > 0.f ? v : 0.f;
                if (cok && row < M) C[(int64_t)row * ldc + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN (wgrad): slab[s][I,J] = sum_{m in split s} A[m,I] B[m,J];  colsum slab[s][I] = sum_m A[m,I]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_tn_f32_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
    float *slab, float *colsum_slab, int Mred, int I, int J, int rows_per_split,
    int tiles_i, int tiles_j, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles = tiles_i * tiles_j;
    const int bid = blockIdx.x;
    const int xcd = bid % kNumXCD, q = bid / kNumXCD;
    const int split = (q / tiles) * kNumXCD + xcd;   // all tiles of one split share an XCD's L2
    if (split >= nsplit) return;
    const int tile = q % tiles;
    const int ti = tile / tiles_j, tj = tile % tiles_j;
    const int i0 = ti * BM, j0 = tj * BN;
    const int mbeg = split * rows_per_split;
    const int mend = min(Mred, mbeg + rows_per_split);

    const int tid = threadIdx.x;
    const int c4 = tid & 31, r0 = tid >> 5;     // staging: float4 column (of 32), first row (of 8)
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const bool aok = (i0 + c4 * 4) < I, bok = (j0 + c4 * 4) < J;
    const float *ap = A + i0 + c4 * 4, *bp = B + j0 + c4 * 4;

    f32x4 ra[4], rb[4];
    auto gload = [&](int mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mt + r0 + 8 * j;
            const bool ok = m < mend;
            ra[j] = (ok && aok) ? ld4(ap + (int64_t)m * lda) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[j] = (ok && bok) ? ld4(bp + (int64_t)m * ldb) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * 2 * TN_TILE, *Bs = As + TN_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st4(As + (r0 + 8 * j) * TN_LD + c4 * 4, ra[j]);
            st4(Bs + (r0 + 8 * j) * TN_LD + c4 * 4, rb[j]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum = 0.f;
    const bool do_colsum = (colsum_slab != nullptr) && (tj == 0) && (tid < BM);

    // operand fragments: lane (li,hi) reads floats (2li, 2li+1) of row 2s+hi -> sub-tiles 0/1.
    // Same fragment pipeline as the NT kernel: 4 groups of 4 k-steps, read one group ahead,
    // barrier before the last group's MFMAs.
    struct Frag { f32x2 a[4], b[4]; };
    const int frag_off_a = hi * TN_LD + wm * 64 + 2 * li;
    const int frag_off_b = TN_TILE + hi * TN_LD + wn * 64 + 2 * li;
    auto fread = [&](int buf, int q) {
        const float *base = smem + buf * 2 * TN_TILE;
        Frag f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f.a[s] = *reinterpret_cast<const f32x2 *>(base + frag_off_a + 2 * (4 * q + s) * TN_LD);
            f.b[s] = *reinterpret_cast<const f32x2 *>(base + frag_off_b + 2 * (4 * q + s) * TN_LD);
        }
        return f;
    };
    auto mma16 = [&](const Frag &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][0], f.b[s][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][0], f.b[s][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][1], f.b[s][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][1], f.b[s][1], acc[1][1], 0, 0, 0);
        }
    };
    auto colsum = [&](int buf) {
        if (do_colsum) {
            const float *As = smem + buf * 2 * TN_TILE;
#pragma unroll
            for (int r = 0; r < BK; ++r) bsum += As[r * TN_LD + tid];
        }
    };

    const int nk = (mend - mbeg + BK - 1) / BK;
    if (nk > 0) {
        gload(mbeg);
        lstore(0);
    }
    __syncthreads();
    Frag f0;
    if (nk > 0) f0 = fread(0, 0);
    for (int t = 0; t < nk; ++t) {
        const bool more = (t + 1) < nk;
        const int buf = t & 1;
        if (more) gload(mbeg + (t + 1) * BK);
        Frag f1 = fread(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f0);
        __builtin_amdgcn_sched_barrier(0);
        Frag f2 = fread(buf, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f1);
        __builtin_amdgcn_sched_barrier(0);
        Frag f3 = fread(buf, 3);
        colsum(buf);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f2);
        __builtin_amdgcn_sched_barrier(0);
        if (more) lstore(buf ^ 1);
        __syncthreads();
        if (more) f0 = fread(buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma16(f3);
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: sub-tile (a,b) element (ri, li) is output (i0+wm*64+2*ri+a, j0+wn*64+2*li+b)
    float *out = slab + (int64_t)split * I * J;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int row = i0 + wm * 64 + 2 * ri + a;
            const int col = j0 + wn * 64 + 2 * li;
            if (row < I && col < J) {   // J % 2 == 0 (J % 4 == 0 is required)
                f32x2 v = {acc[a][0][r], acc[a][1][r]};
                *reinterpret_cast<f32x2 *>(out + (int64_t)row * J + col) = v;
            }
        }
    }
    if (do_colsum && (i0 + tid) < I) colsum_slab[(int64_t)split * I + i0 + tid] = bsum;
}

// out[e] = beta*out[e] + sum_s slab[s][e]   (fixed order -> run-to-run deterministic).
// One launch reduces the weight slabs (n floats each) and, behind them, the bias slabs (n2 each).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slab, float *out, int64_t n,
                                                           const float *__restrict__ slab2, float *out2, int64_t n2,
                                                           int nsplit, float beta) {
    const int64_t n4 = n >> 2, m4 = n2 >> 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4 + m4; e += (int64_t)gridDim.x * blockDim.x) {
        const bool second = e >= n4;
        const float *src = second ? slab2 + (e - n4) * 4 : slab + e * 4;
        float *dst = second ? out2 + (e - n4) * 4 : out + e * 4;
        const int64_t stride = second ? n2 : n;
        f32x4 s = ld4(src);
        for (int k = 1; k < nsplit; ++k) s += ld4(src + (int64_t)k * stride);
        if (beta != 0.f) s += beta * ld4(dst);
        st4(dst, s);
    }
}

// 32x32 LDS transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(int64_t)(by + j) * cols + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < cols && by + tx < rows) out[(int64_t)(bx + j) * rows + by + tx] = tile[tx][j];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int launch_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float *bias, int relu, const float *addend,
                     const float *mask_src, hipStream_t st, const char *what) {
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (M > INT32_MAX - BM || N > INT32_MAX - BN || K > INT32_MAX - BK) { set_error("%s: dimension too large", what); return TOAD_ESHAPE; }
    if (K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0) { set_error("%s: reduction dim %lld must be a multiple of 4", what, (long long)K); return TOAD_ESHAPE; }
    if (!aligned16(A) || !aligned16(B) || !aligned16(C)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NT_SMEM);
        attr_set = true;
    }
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const int grid = kNumXCD * ((tiles_m + kNumXCD - 1) / kNumXCD) * tiles_n;
    hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3(grid), dim3(256), NT_SMEM, st, A, lda, B, ldb, C, ldc, (int)M, (int)N,
                       (int)K, bias, relu, addend, mask_src, tiles_m, tiles_n);
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

extern "C" int toad_linear_act_fwd_f32(const float *X, const float *W, const float *bias, float *Y, int64_t M,
                                        int64_t K, int64_t N, int act, void *stream) {
    if (!X || !W || !Y) { set_error("toad_linear_act_fwd_f32: null pointer"); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("toad_linear_act_fwd_f32: bad act %d", act); return TOAD_EINVAL; }
    return launch_nt(X, K, W, K, Y, N, M, N, K, bias, act == TOAD_ACT_RELU, nullptr, nullptr, (hipStream_t)stream,
                     "toad_linear_act_fwd_f32");
}

extern "C" int toad_linear_dgrad_f32(const float *dY, const float *WT, const float *addend, const float *relu_src,
                                      float *dX, int64_t M, int64_t N, int64_t K, void *stream) {
    if (!dY || !WT || !dX) { set_error("toad_linear_dgrad_f32: null pointer"); return TOAD_EINVAL; }
    // dX[M,K] = dY[M,N] . WT[K,N]^T : an NT product with reduction dim N
    return launch_nt(dY, N, WT, N, dX, K, M, K, N, nullptr, 0, addend, relu_src, (hipStream_t)stream,
                     "toad_linear_dgrad_f32");
}

extern "C" size_t toad_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const WgradPlan p = wgrad_plan(M, N, K);
    return (size_t)p.nsplit * (size_t)(N * K + N) * sizeof(float);
}

extern "C" int toad_linear_wgrad_f32(const float *dY, const float *X, float *dW, float *db, int64_t M, int64_t N,
                                      int64_t K, float beta, void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_linear_wgrad_f32";
    if (!dY || !X || !dW || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (N % 4 != 0 || K % 4 != 0) { set_error("%s: N and K must be multiples of 4", what); return TOAD_ESHAPE; }
    if (M > INT32_MAX - 4096) { set_error("%s: M too large", what); return TOAD_ESHAPE; }
    if (!aligned16(dY) || !aligned16(X) || !aligned16(dW) || !aligned16(ws)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_linear_wgrad_ws_bytes(M, N, K)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM);
        attr_set = true;
    }
    const WgradPlan p = wgrad_plan(M, N, K);
    float *slab = (float *)ws;
    float *cs = db ? slab + (size_t)p.nsplit * N * K : nullptr;
    const int tiles = p.tiles_i * p.tiles_j;
    const int grid = kNumXCD * ((p.nsplit + kNumXCD - 1) / kNumXCD) * tiles;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(grid), dim3(256), TN_SMEM, st, dY, N, X, K, slab, cs, (int)M, (int)N,
                       (int)K, p.rows_per_split, p.tiles_i, p.tiles_j, p.nsplit);
    int rc = check_launch(what);
    if (rc) return rc;
    const int64_t n = N * K, n2 = db ? N : 0;
    int rgrid = (int)(((n + n2) / 4 + 255) / 256);
    if (rgrid > 4096) rgrid = 4096;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rgrid), dim3(256), 0, st, slab, dW, n, cs, db, n2, p.nsplit, beta);
    return check_launch(what);
}

extern "C" int toad_transpose_f32(const float *in, float *out, int64_t rows, int64_t cols, void *stream) {
    if (!in || !out || rows <= 0 || cols <= 0) { set_error("toad_transpose_f32: bad argument"); return TOAD_EINVAL; }
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)rows, (int)cols);
    return check_launch("toad_transpose_f32");
}

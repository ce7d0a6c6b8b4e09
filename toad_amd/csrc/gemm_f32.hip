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

#include <stdlib.h>
#include <type_traits>

namespace toad {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NT_LD = 36;                                  // padded LDS row (floats), conflict-free b128
constexpr int NT_TILE = BM * NT_LD;                        // floats per operand tile
constexpr int NT_SMEM = 2 * 2 * NT_TILE * (int)sizeof(float);   // 73,728 B -> 2 blocks / CU
constexpr int TN_LD = 128;
constexpr int TN_TILE = BK * TN_LD;
constexpr int TN_SMEM = 2 * 2 * TN_TILE * (int)sizeof(float);   // 65,536 B

// epilogue scalars shared by every NT kernel (plain scalars only: pointers stay direct kernel arguments)
struct EpiScalars { int relu; float mask_scale; DropArgs drop; };

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
    const float *__restrict__ bias, EpiScalars es, const float *addend, const float *__restrict__ mask_src,
    int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int relu = es.relu;
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
                if (es.drop.thresh) v *= drop_keep((uint64_t)row * (uint64_t)N + (uint64_t)col, es.drop);
                if (mask_src) v = msk[r] > 0.f ? v * es.mask_scale : 0.f;
                if (cok && row < M) C[(int64_t)row * ldc + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// NT, persistent 256x256 (the fast path; K % 32 == 0)
//
// Why this shape. Ablations (tools/ubench/gemm_ablate.cpp) showed the 128x128 kernels pay ~10 % for
// operand staging even when the data is never waited for: a 128x128x32 step moves 32 KB per 1.05
// MFLOP = 8 B/clk/CU at full MFMA rate, against a ~10-11 B/clk/CU vector-memory path. A 256x256
// tile halves the bytes per FLOP (3.9 B/clk/CU) and needs 25 % fewer LDS fragment reads per MFMA.
//
//  * one persistent block of 8 waves per CU (grid = 256), waves 4(m) x 2(n), each 64x128 = 2x4
//    MFMA 32x32 accumulators (128 acc VGPRs);
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane): no staging VGPRs, no
//    ds_write pass. LDS-DMA writes lane-linear, so the tile image is unpadded [256 pos][32 floats]
//    and bank conflicts are removed by an XOR swizzle applied to the SOURCE address and to the
//    fragment reads: 16-B chunk c of position p is stored at chunk c ^ ((p >> 1) & 7);
//  * the B image is column-interleaved: position 32b+li of a wave's 128 columns holds column
//    4li+b, so the four 32-wide MFMA sub-tiles of a lane are 4 CONSECUTIVE output columns and
//    the epilogue moves 16 B per lane (bias/addend/mask loads and the store);
//  * the k-loop runs across work items (the next item's first stage is in flight while the
//    epilogue of the current one drains); fragments are read one k-group ahead and the stage
//    barrier sits before the last group's MFMAs;
//  * tile quantisation: every XCD plans its own tiles (row tiles x, x+8, ...; all column tiles of a
//    row tile together). Full rounds are dealt to its 32 blocks; the REMAINDER tiles are split
//    along K into g = 32/R pieces whose raw accumulators go to fp32 slabs, summed in fixed order
//    (deterministic) by nt_fixup_kernel, which also applies the epilogue. Small bags (fewer tiles
//    than CUs) are parallelised by the same mechanism.
// ------------------------------------------------------------------------------------------
constexpr int PB = 256;                                         // tile edge
constexpr int PB_TILE = PB * BK;                                // floats per operand stage
constexpr int PB_SMEM = 2 * 2 * PB_TILE * (int)sizeof(float);   // 131,072 B: one block per CU
constexpr int PB_BLOCKS_PER_XCD = 32;
constexpr int PB_GRID = kNumXCD * PB_BLOCKS_PER_XCD;
constexpr int PB_GMAX = 16;                                     // max K-split of a remainder tile

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct NtPlan { int mt_x, tiles, rounds, rem, g; };
__host__ __device__ inline NtPlan nt_plan(int xcd, int tiles_m, int tiles_n, int nk) {
    NtPlan p;
    p.mt_x = (tiles_m - xcd + kNumXCD - 1) / kNumXCD;
    p.tiles = p.mt_x * tiles_n;
    p.rounds = p.tiles / PB_BLOCKS_PER_XCD;
    p.rem = p.tiles - p.rounds * PB_BLOCKS_PER_XCD;
    int g = p.rem > 0 ? PB_BLOCKS_PER_XCD / p.rem : 0;
    if (g > nk) g = nk;
    if (g > PB_GMAX) g = PB_GMAX;
    p.g = g;
    return p;
}

// shared by the GEMM (direct tiles) and the fix-up kernel (slab sums): v = one float4 of 4 consecutive columns.
// NOTE: bias/addend/mask are passed to the kernels as DIRECT pointer arguments. Inside a by-value struct
// hipcc does not infer the global address space, emits FLAT loads, and a pending FLAT access makes it
// put `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every LDS access that follows (found the hard way).
__device__ __forceinline__ f32x4 apply_epilogue(f32x4 v, const EpiScalars &e, const float *add_p, const float *msk_p, f32x4 bv,
                                                uint64_t flat_idx) {
    v += bv;
    if (add_p) v += ld4(add_p);
    if (e.relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f; }
    if (e.drop.thresh) {               // forward dropout after the activation
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= drop_keep(flat_idx + k, e.drop);
    }
    if (msk_p) {                       // backward: relu (and dropout) mask of the saved activation, times 1/(1-p)
        const f32x4 m = ld4(msk_p);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = m[k] > 0.f ? v[k] * e.mask_scale : 0.f;
    }
    return v;
}

// same, with the addend / mask vectors already loaded
__device__ __forceinline__ f32x4 apply_epilogue_v(f32x4 v, const EpiScalars &e, f32x4 av, bool has_add, f32x4 mv, bool has_msk, f32x4 bv,
                                                  uint64_t flat_idx) {
    v += bv;
    if (has_add) v += av;
    if (e.relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f; }
    if (e.drop.thresh) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= drop_keep(flat_idx + k, e.drop);
    }
    if (has_msk) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mv[k] > 0.f ? v[k] * e.mask_scale : 0.f;
    }
    return v;
}

__global__ __launch_bounds__(512, 2) void gemm_nt_f32_big_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
    float *C, int64_t ldc, int M, int N, int K, const float *__restrict__ bias, EpiScalars es, const float *addend,
    const float *__restrict__ mask_src, float *__restrict__ slabs, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const int nk = K / BK;

    // ---- work list of this block
    const int xcd = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    const NtPlan pl = nt_plan(xcd, tiles_m, tiles_n, nk);
    const bool has_part = j < pl.rem * pl.g;
    const int n_items = pl.rounds + (has_part ? 1 : 0);
    if (n_items == 0) return;
    // item i < rounds: full tile q = j + i*32; item == rounds: k-slice `part` of remainder tile
    const int part_tile = pl.g ? pl.rounds * PB_BLOCKS_PER_XCD + j / pl.g : 0;
    const int part = pl.g ? j % pl.g : 0;
    const int part_k0 = pl.g ? (part * nk) / pl.g : 0, part_k1 = pl.g ? ((part + 1) * nk) / pl.g : 0;
    auto item_tile = [&](int i) { return i < pl.rounds ? j + i * PB_BLOCKS_PER_XCD : part_tile; };
    auto item_k0 = [&](int i) { return i < pl.rounds ? 0 : part_k0; };
    auto item_k1 = [&](int i) { return i < pl.rounds ? nk : part_k1; };
    int total = pl.rounds * nk + (has_part ? part_k1 - part_k0 : 0);

    // ---- staging. Wave w copies positions [32w, 32w+32) of the A image and of the B image, 8 per
    //      instruction; lane l -> position 8jj + (l >> 3), physical chunk l & 7.
    const char *Ab = reinterpret_cast<const char *>(A), *Bb = reinterpret_cast<const char *>(B);
    unsigned aoff[4], boff[4];
    auto set_tile = [&](int q, int &m0, int &n0) {
        m0 = ((q / tiles_n) * kNumXCD + xcd) * PB;
        n0 = (q % tiles_n) * PB;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int p = wave * 32 + 8 * jj + (lane >> 3);
            const int ch = (lane & 7) ^ ((p >> 1) & 7);
            const int pp = p & 127;
            const int bcol = (p & 128) + 4 * (pp & 31) + (pp >> 5);         // column-interleaved B image
            aoff[jj] = (unsigned)min(m0 + p, M - 1) * (unsigned)(lda * 4) + ch * 16;   // clamped rows are never stored
            boff[jj] = (unsigned)min(n0 + bcol, N - 1) * (unsigned)(ldb * 4) + ch * 16;
        }
    };
    // LDS-DMA is issued through inline asm: with the builtin, hipcc protects a possible alias between the
    // DMA's LDS write and the following ds_reads with `s_waitcnt vmcnt(0)` right after the issue, which
    // serialises load latency with the MFMAs. The asm form is invisible to that analysis; completion is
    // enforced by hand with ONE `s_waitcnt vmcnt(0)` in front of the stage barrier (dma_wait). M0 (the LDS
    // destination base) is compiler-reserved: saved, written and restored inside the same statement.
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    auto dma1 = [&](const char *sbase, unsigned voff, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
    };
    auto dma = [&](int buf, int kt) {
        const unsigned dst = lds_base + (unsigned)(buf * 2 * PB_TILE + wave * 32 * BK) * 4u;
        const char *ak = Ab + kt * (BK * 4), *bk = Bb + kt * (BK * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            dma1(ak, aoff[jj], dst + jj * 8 * BK * 4);
            dma1(bk, boff[jj], dst + (PB_TILE + jj * 8 * BK) * 4);
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- fragments: lane (li,hi), k-group q reads logical chunk 2q+hi of its positions.
    // Register budget (128 accumulators): B fragments are ROLLING - the 4 registers of sub-tile b are
    // re-read for the next k-group right after the 8 MFMAs that consume them have issued (an MFMA
    // reads its operands at issue) - and only the 8 A registers are double-buffered.
    const int sw = (li >> 1) & 7;
    const int apos = (wm * 64 + li) * BK, bpos = PB_TILE + (wn * 128 + li) * BK;
    auto chunk = [&](int q) { return ((2 * q + hi) ^ sw) * 4; };
    auto read_a = [&](int buf, int q, f32x4 (&fa)[2]) {
        const float *base = smem + buf * 2 * PB_TILE + apos + chunk(q);
        fa[0] = ld4(base);
        fa[1] = ld4(base + 32 * BK);
    };
    auto read_b = [&](int buf, int q, int b) { return ld4(smem + buf * 2 * PB_TILE + bpos + b * 32 * BK + chunk(q)); };
    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    f32x4 fa[2], fb[4];
    // one k-group: 4 x (8 MFMAs on sub-tile column b, then refill fb[b] from (nbuf, nq)); A for the next
    // group is fetched up front into na and swapped in at the end. `refill` = false on the very last group.
    auto group = [&](int nbuf, int nq, bool refill) {
        f32x4 na[2];
        if (refill) read_a(nbuf, nq, na);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s], fb[b][s], acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][s], fb[b][s], acc[1][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (refill) fb[b] = read_b(nbuf, nq, b);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (refill) { fa[0] = na[0]; fa[1] = na[1]; }
    };

    // ---- epilogue. Lane (li,hi) holds, for sub-tile a and acc reg r, the 4 consecutive columns
    //      n0 + wn*128 + 4li .. +3 of row m0 + wm*64 + a*32 + (r&3) + 8(r>>2) + 4hi.
    // Addressing: row and column split into a wave-uniform part (SGPR base pointer per row) and ONE
    // 32-bit per-lane element offset voff = 4*hi*ldc + 4*li that is the same for every row; otherwise
    // LICM hoists 32 per-lane 64-bit row offsets out of the step loop (64 VGPRs -> spills).
    auto epilogue = [&](int m0, int n0, bool partial) {
        // opaque copies: everything derived below is then NOT loop-invariant for LICM, which would
        // otherwise precompute ~60 per-lane values before the step loop and spill them
        int li4 = 4 * li, hi4 = 4 * hi;
        asm volatile("" : "+v"(li4), "+v"(hi4));
        const int voff = hi4 * (int)ldc + li4;
        const int svoff = hi4 * PB + li4;
        const int ucol = n0 + wn * 128;
        const bool cok = (ucol + li4) < N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && cok) bv = ld4(bias + ucol + li4);
        // consume the load on EVERY path: a load destination that is still "pending" on some path at the
        // back-edge makes hipcc guard the next step's ds_reads (same registers, WAW) with vmcnt(0), and
        // that in-order wait also covers the freshly issued LDS-DMA of the next stage.
        asm volatile("" : "+v"(bv));
        float *slab = slabs + (int64_t)blockIdx.x * PB * PB + wn * 128;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int urow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2);      // + 4*hi is in voff
                f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                if (partial) {
                    st4(slab + urow * PB + svoff, v);
                } else if (cok && (m0 + urow + hi4) < M) {
                    const int64_t uoff = (int64_t)(m0 + urow) * ldc + ucol;            // wave-uniform
                    st4s(C + uoff + voff, apply_epilogue(v, es, addend ? addend + uoff + voff : nullptr,
                                                        mask_src ? mask_src + uoff + voff : nullptr, bv,
                                                        (uint64_t)(m0 + urow + hi4) * (uint64_t)N + (uint64_t)(ucol + li4)));
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most 4 rows of loads in flight: bounded registers
            }
        }
    };

    int it = 0, kt = item_k0(0), kend = item_k1(0);     // step being computed
    int nit = 0, nkt = kt, nkend = kend;                // step being staged
    int cur_m0, cur_n0, nxt_m0, nxt_n0;
    set_tile(item_tile(0), cur_m0, cur_n0);
    nxt_m0 = cur_m0; nxt_n0 = cur_n0;
    dma(0, kt);
    zero_acc();
    dma_wait();
    __syncthreads();
    read_a(0, 0, fa);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = read_b(0, 0, b);
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        const bool more = (step + 1) < total;
        if (more) {                        // stage the next step (possibly the next item's first k-step)
            if (++nkt == nkend) { ++nit; nkt = item_k0(nit); nkend = item_k1(nit); set_tile(item_tile(nit), nxt_m0, nxt_n0); }
            dma(buf ^ 1, nkt);
        }
        group(buf, 1, true);
        group(buf, 2, true);
        group(buf, 3, true);
        dma_wait();                        // this wave's share of the next stage has landed (also drains older stores) ...
        __syncthreads();                   // ... everyone's has, and this stage is fully read
        group(buf ^ 1, 0, more);           // last k-group from registers; refills come from the NEXT stage
        if (++kt == kend) {                // item finished: stores are fire-and-forget, the next item's first
            epilogue(cur_m0, cur_n0, it >= pl.rounds);     // stage is already in flight
            zero_acc();
            ++it;
            kt = item_k0(it); kend = item_k1(it);
            cur_m0 = nxt_m0; cur_n0 = nxt_n0;
        }
    }
}

// Remainder tiles: C tile = epilogue(sum of the g K-slice slabs), fixed order. grid = (64, 31, 8):
// x = 4-row strip of the tile (256 threads = 4 rows x 64 float4 columns: one float4 per thread, so a
// small-bag fix-up still spreads over 64 blocks per tile), y = remainder tile index of the XCD, z = XCD.
__global__ __launch_bounds__(256) void nt_fixup_kernel(const float *__restrict__ slabs, float *C, int64_t ldc, int M, int N,
                                                        int K, const float *__restrict__ bias, EpiScalars es, const float *addend,
                                                        const float *__restrict__ mask_src, int tiles_m, int tiles_n) {
    const int xcd = blockIdx.z, tr = blockIdx.y;
    const NtPlan pl = nt_plan(xcd, tiles_m, tiles_n, K / BK);
    if (tr >= pl.rem || pl.g == 0) return;
    const int q = pl.rounds * PB_BLOCKS_PER_XCD + tr;
    const int m0 = ((q / tiles_n) * kNumXCD + xcd) * PB, n0 = (q % tiles_n) * PB;
    const int c4 = threadIdx.x & 63, lrow = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int col = n0 + c4 * 4;
    if (col >= N || m0 + lrow >= M) return;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ld4(bias + col);
    // slab of K-slice p was written by block (xcd + 8*(tr*g + p))
    const float *sp = slabs + ((int64_t)(xcd + kNumXCD * (tr * pl.g)) * PB + lrow) * PB + c4 * 4;
    f32x4 v = ld4(sp);
    for (int p = 1; p < pl.g; ++p) v += ld4(sp + (int64_t)p * kNumXCD * PB * PB);
    const int64_t off = (int64_t)(m0 + lrow) * ldc + col;
    st4s(C + off, apply_epilogue(v, es, addend ? addend + off : nullptr, mask_src ? mask_src + off : nullptr, bv,
                                (uint64_t)(m0 + lrow) * (uint64_t)N + (uint64_t)col));
}

// ------------------------------------------------------------------------------------------
// NT, persistent 256x256, SPLIT-bf16 arithmetic (fp32-equivalent results on the bf16 matrix pipe)
//
// x = h + m + l with three bf16 (24 significand bits); a product keeps the six terms
// hh + hm + mh + mm + hl + lh (dropped: <= 2^-24 relative), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16. Measured (tools/ubench/bf16x3_probe.hip): max error 4.85e-6 vs 4.68e-6 for
// the exact-fp32 MFMA chain at K = 1024 - indistinguishable - at 1/2.67 of the matrix-pipe time.
//  * A (activations, fp32 in HBM): staged by LDS-DMA exactly like gemm_nt_f32_big_kernel and split in
//    registers when its fragments are read: 16 floats per lane per 16-deep step, ~5.5 VALU ops each,
//    i.e. ~88 VALU against 48 MFMAs - hidden in the MFMA shadow.
//  * B (weights): pre-split ONCE per call by split_planes_kernel into three bf16 planes already in the
//    LDS stage order (column-interleaved positions, 16-B chunks XOR-swizzled with (pos >> 2) & 3), so its
//    LDS-DMA is a linear copy and its fragments are single conflict-free ds_read_b128's (8 bf16).
//  * stage = 32 k: A 32 KB + 3 x 16 KB planes = 80 KB; two stages = the whole 160 KB LDS of the CU.
//    A stage is consumed as 8 units (2 k16 steps x 4 column sub-tiles) of 12 MFMAs; B planes are read one
//    unit ahead, A one k16 step ahead, the stage barrier sits before the last unit.
// Work decomposition, epilogue, K-split tail and fix-up are those of gemm_nt_f32_big_kernel.
// ------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SP_A_BYTES = PB * BK * 4;                         // 32,768
constexpr int SP_PLANE_BYTES = PB * BK * 2;                     // 16,384
constexpr int SP_STAGE_BYTES = SP_A_BYTES + 3 * SP_PLANE_BYTES; // 81,920
constexpr int SP_SMEM = 2 * SP_STAGE_BYTES;                     // 163,840 = all of LDS

// bits of x reduced to bf16, low 16 bits cleared. Default: truncation (1 VALU op). Measured against fp64 on
// GEMM outputs and on all 14 gradients (tools/gemm_accuracy.py, tools/grad_errors.py): truncation and
// round-to-nearest (-DTOAD_SPLIT_RN, 3 ops) are equally accurate - both at the exact-fp32 kernel's error level -
// because three 8-bit pieces hold all 24 significand bits either way; RN costs ~10 % of the kernel's speed.
__device__ __forceinline__ unsigned bf16_rn(unsigned u) {
#ifdef TOAD_SPLIT_RN
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
#else
    return u & 0xFFFF0000u;
#endif
}
// Bp[tile][kstage][plane][pos 0..255][phys chunk 0..3][8 bf16]  <-  B[n, k] = src[n * sn + k * sk]
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ src, int64_t sn, int64_t sk,
                                                            unsigned short *__restrict__ Bp, int N, int K, int tiles_n) {
    const int nk = K / BK;
    const int64_t total = (int64_t)tiles_n * nk * PB * 4;                 // one thread per (tile, stage, pos, logical chunk)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int chunk = (int)(t & 3), pos = (int)((t >> 2) & (PB - 1));
        const int64_t ts = t >> 10;                                       // tile * nk + stage
        const int stage = (int)(ts % nk), tile = (int)(ts / nk);
        const int pp = pos & 127;
        const int col = tile * PB + (pos & 128) + 4 * (pp & 31) + (pp >> 5);
        const int phys = chunk ^ ((pos >> 2) & 3);
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = stage * BK + chunk * 8 + e;
            const float x = col < N ? src[(int64_t)col * sn + (int64_t)k * sk] : 0.f;
            const unsigned hb = bf16_rn(__builtin_bit_cast(unsigned, x));
            const float r1 = x - __builtin_bit_cast(float, hb);
            const unsigned mb = bf16_rn(__builtin_bit_cast(unsigned, r1));
            const float r2 = r1 - __builtin_bit_cast(float, mb);
            h[e] = (unsigned short)(hb >> 16); m[e] = (unsigned short)(mb >> 16);
            l[e] = (unsigned short)(bf16_rn(__builtin_bit_cast(unsigned, r2)) >> 16);
        }
        unsigned short *dst = Bp + ts * (3 * PB * BK) + pos * BK + phys * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dst[e] = h[e]; dst[PB * BK + e] = m[e]; dst[2 * PB * BK + e] = l[e]; }
    }
}

// split 8 fp32 into three bf16x8 planes h, m, l with x = h + m + l (+ < 2^-24 |x|): every residual x - h,
// (x - h) - m is exact in fp32.
__device__ __forceinline__ void split8(const f32x4 &x0, const f32x4 &x1, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        hh[e] = bf16_rn(__builtin_bit_cast(unsigned, x));
        const float r1 = x - __builtin_bit_cast(float, hh[e]);
        mm[e] = bf16_rn(__builtin_bit_cast(unsigned, r1));
        const float r2 = r1 - __builtin_bit_cast(float, mm[e]);
        ll[e] = bf16_rn(__builtin_bit_cast(unsigned, r2));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hp, mp, lp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {      // dword e = bf16 of element 2e (low half) | bf16 of element 2e+1 (high half)
        hp[e] = __builtin_amdgcn_perm(hh[2 * e + 1], hh[2 * e], 0x07060302u);
        mp[e] = __builtin_amdgcn_perm(mm[2 * e + 1], mm[2 * e], 0x07060302u);
        lp[e] = __builtin_amdgcn_perm(ll[2 * e + 1], ll[2 * e], 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp); m = __builtin_bit_cast(bf16x8, mp); l = __builtin_bit_cast(bf16x8, lp);
}

__global__ __launch_bounds__(512, 2) void gemm_nt_split_big_kernel(
    const float *__restrict__ A, int64_t lda, const unsigned short *__restrict__ Bp,
    float *C, int64_t ldc, int M, int N, int K, const float *__restrict__ bias, EpiScalars es, const float *addend,
    const float *__restrict__ mask_src, float *__restrict__ slabs, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const int nk = K / BK;

    // ---- work list (identical to gemm_nt_f32_big_kernel)
    const int xcd = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    const NtPlan pl = nt_plan(xcd, tiles_m, tiles_n, nk);
    const bool has_part = j < pl.rem * pl.g;
    const int n_items = pl.rounds + (has_part ? 1 : 0);
    if (n_items == 0) return;
    const int part_tile = pl.g ? pl.rounds * PB_BLOCKS_PER_XCD + j / pl.g : 0;
    const int part = pl.g ? j % pl.g : 0;
    const int part_k0 = pl.g ? (part * nk) / pl.g : 0, part_k1 = pl.g ? ((part + 1) * nk) / pl.g : 0;
    auto item_tile = [&](int i) { return i < pl.rounds ? j + i * PB_BLOCKS_PER_XCD : part_tile; };
    auto item_k0 = [&](int i) { return i < pl.rounds ? 0 : part_k0; };
    auto item_k1 = [&](int i) { return i < pl.rounds ? nk : part_k1; };
    const int total = pl.rounds * nk + (has_part ? part_k1 - part_k0 : 0);

    // ---- staging
    const char *Ab = reinterpret_cast<const char *>(A), *Bpb = reinterpret_cast<const char *>(Bp);
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    unsigned aoff[4];
    const unsigned lane16 = lane * 16u;
    auto dma1 = [&](const char *sbase, unsigned voff, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
    };
    auto set_tile = [&](int q, int &m0, int &n0, int &tn) {
        m0 = ((q / tiles_n) * kNumXCD + xcd) * PB;
        tn = q % tiles_n;
        n0 = tn * PB;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int p = wave * 32 + 8 * jj + (lane >> 3);
            const int ch = (lane & 7) ^ ((p >> 1) & 7);
            aoff[jj] = (unsigned)min(m0 + p, M - 1) * (unsigned)(lda * 4) + ch * 16;
        }
    };
    auto dma = [&](int buf, int kt, int tn) {
        const unsigned dst = lds_base + (unsigned)buf * SP_STAGE_BYTES;
        const char *ak = Ab + kt * (BK * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma1(ak, aoff[jj], dst + (wave * 32 + jj * 8) * (BK * 4));
        const char *bk = Bpb + (int64_t)(tn * nk + kt) * (3 * SP_PLANE_BYTES) + wave * 6144;
#pragma unroll
        for (int i = 0; i < 6; ++i) dma1(bk + i * 1024, lane16, dst + SP_A_BYTES + wave * 6144 + i * 1024);
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- fragment addressing
    const int swA = (li >> 1) & 7, swB = (li >> 2) & 3;
    const int aposf = (wm * 64 + li) * BK;                                   // floats, + a*32*BK
    const int bposb = SP_A_BYTES + (wn * 128 + li) * 64;                     // bytes, + plane*16384 + b*32*64
    auto read_a = [&](int buf, int s, f32x4 (&f)[2][2]) {                    // raw fp32 of k16 step s: [a][lo/hi 4 floats]
        const float *base = smem + buf * (SP_STAGE_BYTES / 4) + aposf;
        const int c0 = ((4 * s + 2 * hi) ^ swA) * 4, c1 = ((4 * s + 2 * hi + 1) ^ swA) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a) { f[a][0] = ld4(base + a * 32 * BK + c0); f[a][1] = ld4(base + a * 32 * BK + c1); }
    };
    auto read_b = [&](int buf, int s, int b, bf16x8 (&q)[3]) {
        const char *base = reinterpret_cast<const char *>(smem) + buf * SP_STAGE_BYTES + bposb + b * 32 * 64 +
                           ((2 * s + hi) ^ swB) * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) q[p] = *reinterpret_cast<const bf16x8 *>(base + p * SP_PLANE_BYTES);
    };
    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    bf16x8 ap[2][3], an[2][3];           // split A fragments of the current / next k16 step: [sub-tile a][h, m, l]
    f32x4 fa[2][2];                      // raw A fragments of the next k16 step
    bf16x8 bq[3], bn[3];                 // B planes of the current / next unit
    auto convert_a = [&](bf16x8 (&dst)[2][3]) {
#pragma unroll
        for (int a = 0; a < 2; ++a) split8(fa[a][0], fa[a][1], dst[a][0], dst[a][1], dst[a][2]);
    };
    auto mma12 = [&](int b) {            // six terms, small ones first; the two row sub-tiles alternate
#define TOAD_T(PA, PB_) \
        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][PA], bq[PB_], acc[0][b], 0, 0, 0); \
        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1][PA], bq[PB_], acc[1][b], 0, 0, 0);
        TOAD_T(2, 0) TOAD_T(0, 2) TOAD_T(1, 1) TOAD_T(1, 0) TOAD_T(0, 1) TOAD_T(0, 0)
#undef TOAD_T
    };
    // one unit = 12 MFMAs on column sub-tile b with the current planes; the next unit's B planes are fetched first.
    // `conv`: also split the raw A fragments of the next k16 step into `an` INSIDE this unit's scheduling
    // region, so the ~180 VALU ops of the conversion issue in the shadow of the 12 MFMAs.
    auto unit = [&](int b, int nbuf, int ns, int nb, bool fetch, bool conv) {
        if (fetch) read_b(nbuf, ns, nb, bn);
        __builtin_amdgcn_sched_barrier(0);
        mma12(b);
        if (conv) convert_a(an);
        __builtin_amdgcn_sched_barrier(0);
        if (fetch) { bq[0] = bn[0]; bq[1] = bn[1]; bq[2] = bn[2]; }
        if (conv) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) ap[a][p] = an[a][p];
        }
    };

    auto epilogue = [&](int m0, int n0, bool partial) {
        int li4 = 4 * li, hi4 = 4 * hi;
        asm volatile("" : "+v"(li4), "+v"(hi4));
        const int voff = hi4 * (int)ldc + li4;
        const int svoff = hi4 * PB + li4;
        const int ucol = n0 + wn * 128;
        const bool cok = (ucol + li4) < N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && cok) bv = ld4(bias + ucol + li4);
        asm volatile("" : "+v"(bv));
        float *slab = slabs + (int64_t)blockIdx.x * PB * PB + wn * 128;
        if (!partial && addend && !mask_src) {
            // residual epilogue (convolution + skip connection): the per-row `if (in range) { load; add; store }` below makes
            // every row a dependent HBM round trip; here 8 rows of the residual are loaded back to back from CLAMPED
            // addresses (no branch in between), then finished and stored. (With a mask as well - the dgrads - 8 rows
            // of two operands do not fit the register budget: those keep the loop below.)
            const int colc = cok ? ucol + li4 : 0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    f32x4 av[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int r = r0 + g;
                        const int rowc = min(m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + hi4, M - 1);
                        av[g] = ld4(addend + (int64_t)rowc * ldc + colc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int r = r0 + g;
                        const int urow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2);
                        if (cok && (m0 + urow + hi4) < M) {
                            f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                            v += bv; v += av[g];
                            if (es.relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f; }
                            if (es.drop.thresh) {
                                const uint64_t fi = (uint64_t)(m0 + urow + hi4) * (uint64_t)N + (uint64_t)(ucol + li4);
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] *= drop_keep(fi + k, es.drop);
                            }
                            st4s(C + (int64_t)(m0 + urow) * ldc + ucol + voff, v);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        if (!partial && mask_src) {
            // dgrad epilogue (mask, optionally addend): 4 rows of both operands in flight, clamped addresses, no branch between loads
            const bool has_add = addend != nullptr;
            const int colc = cok ? ucol + li4 : 0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) {
                    f32x4 av[4], mv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int r = r0 + g;
                        const int rowc = min(m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + hi4, M - 1);
                        const int64_t off = (int64_t)rowc * ldc + colc;
                        mv[g] = ld4(mask_src + off);
                        av[g] = has_add ? ld4(addend + off) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int r = r0 + g;
                        const int urow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2);
                        if (cok && (m0 + urow + hi4) < M) {
                            f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                            st4s(C + (int64_t)(m0 + urow) * ldc + ucol + voff,
                                 apply_epilogue_v(v, es, av[g], has_add, mv[g], true, bv,
                                                  (uint64_t)(m0 + urow + hi4) * (uint64_t)N + (uint64_t)(ucol + li4)));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int urow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2);
                f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                if (partial) {
                    st4(slab + urow * PB + svoff, v);
                } else if (cok && (m0 + urow + hi4) < M) {
                    const int64_t uoff = (int64_t)(m0 + urow) * ldc + ucol;
                    st4s(C + uoff + voff, apply_epilogue(v, es, addend ? addend + uoff + voff : nullptr,
                                                         mask_src ? mask_src + uoff + voff : nullptr, bv,
                                                         (uint64_t)(m0 + urow + hi4) * (uint64_t)N + (uint64_t)(ucol + li4)));
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int it = 0, kt = item_k0(0), kend = item_k1(0);
    int nit = 0, nkt = kt, nkend = kend;
    int cur_m0, cur_n0, cur_tn, nxt_m0, nxt_n0, nxt_tn;
    set_tile(item_tile(0), cur_m0, cur_n0, cur_tn);
    nxt_m0 = cur_m0; nxt_n0 = cur_n0; nxt_tn = cur_tn;
    dma(0, kt, cur_tn);
    zero_acc();
    dma_wait();
    __syncthreads();
    read_a(0, 0, fa);
    read_b(0, 0, 0, bq);
    convert_a(ap);
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        const bool more = (step + 1) < total;
        if (more) {
            if (++nkt == nkend) { ++nit; nkt = item_k0(nit); nkend = item_k1(nit); set_tile(item_tile(nit), nxt_m0, nxt_n0, nxt_tn); }
            dma(buf ^ 1, nkt, nxt_tn);
        }
        // k16 step 0: units 0..3 (raw A of step 1 is fetched up front and split during unit 3)
        read_a(buf, 1, fa);
        unit(0, buf, 0, 1, true, false);
        unit(1, buf, 0, 2, true, false);
        unit(2, buf, 0, 3, true, false);
        unit(3, buf, 1, 0, true, true);
        // k16 step 1: units 4..7
        unit(0, buf, 1, 1, true, false);
        unit(1, buf, 1, 2, true, false);
        unit(2, buf, 1, 3, true, false);
        dma_wait();
        __syncthreads();                    // next stage landed everywhere; this stage is fully read
        if (more) read_a(buf ^ 1, 0, fa);
        unit(3, buf ^ 1, 0, 0, more, more);
        if (++kt == kend) {
            epilogue(cur_m0, cur_n0, it >= pl.rounds);
            zero_acc();
            ++it;
            kt = item_k0(it); kend = item_k1(it);
            cur_m0 = nxt_m0; cur_n0 = nxt_n0; cur_tn = nxt_tn;
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

// ------------------------------------------------------------------------------------------
// TN, persistent 256x256 (wgrad fast path): slab[s][I,J] = sum_{m in split s} A[m,I] B[m,J]
//
// Same machinery as gemm_nt_f32_big_kernel (one 8-wave block per CU, LDS-DMA stages issued through
// inline asm, fragments read ahead, stage barrier before the last k-group, work items pipelined
// across their boundaries). Differences:
//  * the reduction runs over ROWS (patches): a stage is 32 rows x 256 columns of each operand, one
//    1-KiB LDS-DMA per row, lane-linear -> the image needs no swizzle: A fragments are ds_read_b64
//    (columns 2li,2li+1 -> sub-tiles a=0,1) and B fragments ds_read_b128 (columns 4li..4li+3 ->
//    sub-tiles b=0..3, so the slab stores are 16 B), both conflict-free as laid out;
//  * rows past the end of a split must contribute ZERO (they are reduction terms): the wave that
//    staged such a row overwrites it with zeros after its own DMA has landed, before the barrier;
//  * every item ends in a raw 256x256 slab (+ the column sums of A for the bias gradient when the
//    item owns column tile 0); slab_reduce_kernel sums the splits in fixed order.
// ------------------------------------------------------------------------------------------
struct TnPlan { int ti, tj, tiles, spx, nsplit, rows_per_split; };
static TnPlan tn_plan(int64_t M, int64_t I, int64_t J) {
    TnPlan p;
    p.ti = (int)((I + PB - 1) / PB);
    p.tj = (int)((J + PB - 1) / PB);
    p.tiles = p.ti * p.tj;
    int spx = PB_BLOCKS_PER_XCD / p.tiles;                       // one item per block where possible
    if (spx < 1) spx = 1;
    const int64_t max_splits = M >= 8192 ? (M + 127) / 128 : (M + 31) / 32;   // >= 4 stages per split (>= 1 for short bags)
    while (spx > 1 && (int64_t)spx * kNumXCD > max_splits) --spx;
    int ns = spx * kNumXCD;
    if (ns > max_splits) ns = (int)(max_splits < 1 ? 1 : max_splits);
    int64_t rps = (M + ns - 1) / ns;
    rps = (rps + BK - 1) / BK * BK;
    p.rows_per_split = (int)rps;
    p.nsplit = (int)((M + rps - 1) / rps);
    p.spx = (p.nsplit + kNumXCD - 1) / kNumXCD;
    return p;
}

__global__ __launch_bounds__(512, 2) void gemm_tn_f32_big_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ slab, float *__restrict__ colsum_slab, int Mred, int I, int J, int rows_per_split,
    int ti, int tj, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const int tiles = ti * tj;

    // ---- items of this block: XCD x owns splits x, x+8, ...; item idx -> (split_local, tile)
    const int xcd = blockIdx.x % kNumXCD, jb = blockIdx.x / kNumXCD;
    const int splits_x = (nsplit - xcd + kNumXCD - 1) / kNumXCD;
    const int items_x = splits_x * tiles;
    const int n_items = jb < items_x ? (items_x - jb + PB_BLOCKS_PER_XCD - 1) / PB_BLOCKS_PER_XCD : 0;
    if (n_items == 0) return;
    auto item_split = [&](int i) { return ((jb + i * PB_BLOCKS_PER_XCD) / tiles) * kNumXCD + xcd; };
    auto item_tile = [&](int i) { return (jb + i * PB_BLOCKS_PER_XCD) % tiles; };
    auto split_steps = [&](int sp) {
        const int mb = sp * rows_per_split, me = min(Mred, mb + rows_per_split);
        return (me - mb + BK - 1) / BK;
    };
    int total = 0;
    for (int i = 0; i < n_items; ++i) total += split_steps(item_split(i));

    // ---- staging: wave w copies rows 4w..4w+3 of both images (one 1-KiB DMA per row, lane l -> floats 4l..4l+3)
    const char *Ab = reinterpret_cast<const char *>(A), *Bb = reinterpret_cast<const char *>(B);
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    auto dma1 = [&](const char *sbase, unsigned voff, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
    };
    unsigned avoff = 0, bvoff = 0;           // per-lane column byte offsets inside a row (clamped at the tile edge)
    int st_i0 = 0, st_j0 = 0, st_mend = 0;   // tile / split end of the item being staged
    auto set_item = [&](int i, int &mrow) {
        const int sp = item_split(i), tl = item_tile(i);
        st_i0 = (tl / tj) * PB; st_j0 = (tl % tj) * PB;
        mrow = sp * rows_per_split;
        st_mend = min(Mred, mrow + rows_per_split);
        avoff = (unsigned)min(st_i0 + 4 * lane, I - 4) * 4u;
        bvoff = (unsigned)min(st_j0 + 4 * lane, J - 4) * 4u;
    };
    // stage rows [m, m+32) of the staged item into buffer `buf`; returns how many of this wave's 4 rows are real
    auto dma = [&](int buf, int m) {
        const unsigned dst = lds_base + (unsigned)(buf * 2 * PB_TILE + wave * 4 * PB) * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(m + wave * 4 + r, Mred - 1);               // clamped: valid address, zeroed later if past the split
            dma1(Ab + (int64_t)row * (lda * 4), avoff, dst + r * PB * 4);
            dma1(Bb + (int64_t)row * (ldb * 4), bvoff, dst + (PB_TILE + r * PB) * 4);
        }
    };
    auto zero_tail = [&](int buf, int m, int mend) {                   // rare: only the last stage of a ragged split
        if (m + 32 <= mend) return;
        float *img = smem + buf * 2 * PB_TILE + wave * 4 * PB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + wave * 4 + r >= mend) {
                st4(img + r * PB + 4 * lane, f32x4{0.f, 0.f, 0.f, 0.f});
                st4(img + PB_TILE + r * PB + 4 * lane, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- fragments: k2-step s (0..15) uses image rows 2s+hi. 4-deep ring, refilled right after use.
    const int aoffs = hi * PB + wi * 64 + 2 * li, boffs = PB_TILE + hi * PB + wj * 128 + 4 * li;
    f32x2 fa[4];
    f32x4 fb[4];
    auto fread = [&](int buf, int s2, int slot) {
        const float *base = smem + buf * 2 * PB_TILE + 2 * s2 * PB;
        fa[slot] = *reinterpret_cast<const f32x2 *>(base + aoffs);
        fb[slot] = ld4(base + boffs);
    };
    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    // one k-group = 4 k2-steps; after the 8 MFMAs of a step its ring slot is refilled from (nbuf, ns0 + step)
    auto group = [&](int nbuf, int ns0, bool refill) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k][a], fb[k][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (refill) fread(nbuf, ns0 + k, k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float bsum = 0.f;
    auto colsum = [&](int buf, bool on) {     // column sums of the A image (bias gradient), threads 0..255
        if (on && tid < PB) {
            const float *As = smem + buf * 2 * PB_TILE;
#pragma unroll 8
            for (int r = 0; r < BK; ++r) bsum += As[r * PB + tid];
        }
    };
    auto epilogue = [&](int sp, int i0, int j0, bool with_colsum) {
        int li4 = 4 * li, hi4 = 4 * hi;
        asm volatile("" : "+v"(li4), "+v"(hi4));                 // keep LICM from hoisting ~60 per-lane offsets
        float *out = slab + (int64_t)sp * I * J;
        const int col = j0 + wj * 128 + li4;
        const bool cok = col < J;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * 64 + 2 * ((r & 3) + 8 * (r >> 2) + hi4) + a;
                f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                if (cok && row < I) st4(out + (int64_t)row * J + col, v);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (with_colsum && tid < PB && (i0 + tid) < I) colsum_slab[(int64_t)sp * I + i0 + tid] = bsum;
        bsum = 0.f;
    };

    // ---- the step loop (flattened over items)
    int it = 0, nit = 0;
    int m_stage = 0;                          // first row of the stage being loaded
    set_item(0, m_stage);
    int cur_sp = item_split(0), cur_i0 = st_i0, cur_j0 = st_j0, cur_mend = st_mend;
    int kt = 0, kend = split_steps(cur_sp);  // step being computed
    int nkt = 0, nkend = kend;                // step being staged
    bool cur_cs = colsum_slab != nullptr && cur_j0 == 0;
    dma(0, m_stage);
    zero_acc();
    dma_wait();
    zero_tail(0, m_stage, st_mend);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) fread(0, k, k);
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        const bool more = (step + 1) < total;
        if (more) {
            if (++nkt == nkend) { ++nit; nkt = 0; set_item(nit, m_stage); nkend = split_steps(item_split(nit)); }
            else m_stage += BK;
            dma(buf ^ 1, m_stage);
        }
        group(buf, 4, true);
        group(buf, 8, true);
        colsum(buf, cur_cs);
        group(buf, 12, true);
        dma_wait();
        if (more) zero_tail(buf ^ 1, m_stage, st_mend);
        __syncthreads();
        group(buf ^ 1, 0, more);
        if (++kt == kend) {
            epilogue(cur_sp, cur_i0, cur_j0, cur_cs);
            zero_acc();
            ++it;
            if (it < n_items) {
                cur_sp = item_split(it); kt = 0; kend = split_steps(cur_sp);
                cur_i0 = st_i0; cur_j0 = st_j0; cur_mend = st_mend;
                cur_cs = colsum_slab != nullptr && cur_j0 == 0;
            }
        }
    }
    (void)cur_mend;
}

// ------------------------------------------------------------------------------------------
// TN, persistent 256x256, SPLIT-bf16 arithmetic (wgrad on the bf16 matrix pipe)
//
// Same work decomposition, staging (fp32 rows by LDS-DMA), ragged-tail zeroing, slabs and column sums as
// gemm_tn_f32_big_kernel; the product is computed like gemm_nt_split_big_kernel (x = h+m+l, six terms).
// Both operands are activations here, so both are split in registers. The reduction runs over image ROWS:
// for a 16-deep MFMA step lane (li, hi) gathers rows 16s+8hi .. +7 of its columns - 8 ds_read_b64 for
// the two A sub-tiles and 8 ds_read_b128 for the four B sub-tiles (conflict-free as in the fp32 kernel) -
// and packs each sub-tile's 8 row values into three bf16x8 planes. 48 floats are split per lane per step
// (~5.5 VALU ops each) against 48 MFMAs; the second wave of the SIMD computes while this one gathers.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8s(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hh[e] = bf16_rn(__builtin_bit_cast(unsigned, x[e]));
        const float r1 = x[e] - __builtin_bit_cast(float, hh[e]);
        mm[e] = bf16_rn(__builtin_bit_cast(unsigned, r1));
        const float r2 = r1 - __builtin_bit_cast(float, mm[e]);
        ll[e] = bf16_rn(__builtin_bit_cast(unsigned, r2));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hp, mp, lp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hp[e] = __builtin_amdgcn_perm(hh[2 * e + 1], hh[2 * e], 0x07060302u);
        mp[e] = __builtin_amdgcn_perm(mm[2 * e + 1], mm[2 * e], 0x07060302u);
        lp[e] = __builtin_amdgcn_perm(ll[2 * e + 1], ll[2 * e], 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp); m = __builtin_bit_cast(bf16x8, mp); l = __builtin_bit_cast(bf16x8, lp);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_split_big_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ slab, float *__restrict__ colsum_slab, int Mred, int I, int J, int rows_per_split,
    int ti, int tj, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const int tiles = ti * tj;

    const int xcd = blockIdx.x % kNumXCD, jb = blockIdx.x / kNumXCD;
    const int splits_x = (nsplit - xcd + kNumXCD - 1) / kNumXCD;
    const int items_x = splits_x * tiles;
    const int n_items = jb < items_x ? (items_x - jb + PB_BLOCKS_PER_XCD - 1) / PB_BLOCKS_PER_XCD : 0;
    if (n_items == 0) return;
    auto item_split = [&](int i) { return ((jb + i * PB_BLOCKS_PER_XCD) / tiles) * kNumXCD + xcd; };
    auto item_tile = [&](int i) { return (jb + i * PB_BLOCKS_PER_XCD) % tiles; };
    auto split_steps = [&](int sp) {
        const int mb = sp * rows_per_split, me = min(Mred, mb + rows_per_split);
        return (me - mb + BK - 1) / BK;
    };
    int total = 0;
    for (int i = 0; i < n_items; ++i) total += split_steps(item_split(i));

    const char *Ab = reinterpret_cast<const char *>(A), *Bb = reinterpret_cast<const char *>(B);
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    auto dma1 = [&](const char *sbase, unsigned voff, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
    };
    unsigned avoff = 0, bvoff = 0;
    int st_i0 = 0, st_j0 = 0, st_mend = 0;
    auto set_item = [&](int i, int &mrow) {
        const int sp = item_split(i), tl = item_tile(i);
        st_i0 = (tl / tj) * PB; st_j0 = (tl % tj) * PB;
        mrow = sp * rows_per_split;
        st_mend = min(Mred, mrow + rows_per_split);
        avoff = (unsigned)min(st_i0 + 4 * lane, I - 4) * 4u;
        bvoff = (unsigned)min(st_j0 + 4 * lane, J - 4) * 4u;
    };
    auto dma = [&](int buf, int m) {
        const unsigned dst = lds_base + (unsigned)(buf * 2 * PB_TILE + wave * 4 * PB) * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(m + wave * 4 + r, Mred - 1);
            dma1(Ab + (int64_t)row * (lda * 4), avoff, dst + r * PB * 4);
            dma1(Bb + (int64_t)row * (ldb * 4), bvoff, dst + (PB_TILE + r * PB) * 4);
        }
    };
    auto zero_tail = [&](int buf, int m, int mend) {
        if (m + 32 <= mend) return;
        float *img = smem + buf * 2 * PB_TILE + wave * 4 * PB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + wave * 4 + r >= mend) {
                st4(img + r * PB + 4 * lane, f32x4{0.f, 0.f, 0.f, 0.f});
                st4(img + PB_TILE + r * PB + 4 * lane, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- fragments of one 16-deep step: rows 16s + 8hi + j (j = 0..7) of this lane's columns
    const int aoffs = 8 * hi * PB + wi * 64 + 2 * li, boffs = PB_TILE + 8 * hi * PB + wj * 128 + 4 * li;
    f32x2 ra[8];
    f32x4 rb[8];
    auto read_ra = [&](int buf, int s) {
        const float *base = smem + buf * 2 * PB_TILE + 16 * s * PB + aoffs;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ra[jj] = *reinterpret_cast<const f32x2 *>(base + jj * PB);
    };
    auto read_rb = [&](int buf, int s) {
        const float *base = smem + buf * 2 * PB_TILE + 16 * s * PB + boffs;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) rb[jj] = ld4(base + jj * PB);
    };
    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    bf16x8 ap[2][3], bq[3];
    auto conv_a = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float x[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) x[jj] = ra[jj][a];
            split8s(x, ap[a][0], ap[a][1], ap[a][2]);
        }
    };
    auto conv_b = [&](int b) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) x[jj] = rb[jj][b];
        split8s(x, bq[0], bq[1], bq[2]);
    };
    auto mma12 = [&](int b) {
#define TOAD_T(PA, PB_) \
        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][PA], bq[PB_], acc[0][b], 0, 0, 0); \
        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1][PA], bq[PB_], acc[1][b], 0, 0, 0);
        TOAD_T(2, 0) TOAD_T(0, 2) TOAD_T(1, 1) TOAD_T(1, 0) TOAD_T(0, 1) TOAD_T(0, 0)
#undef TOAD_T
    };
    float bsum = 0.f;
    auto colsum = [&](int buf, bool on) {
        if (on && tid < PB) {
            const float *As = smem + buf * 2 * PB_TILE;
#pragma unroll 8
            for (int r = 0; r < BK; ++r) bsum += As[r * PB + tid];
        }
    };
    auto epilogue = [&](int sp, int i0, int j0, bool with_colsum) {
        int li4 = 4 * li, hi4 = 4 * hi;
        asm volatile("" : "+v"(li4), "+v"(hi4));
        float *out = slab + (int64_t)sp * I * J;
        const int col = j0 + wj * 128 + li4;
        const bool cok = col < J;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * 64 + 2 * ((r & 3) + 8 * (r >> 2) + hi4) + a;
                f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                if (cok && row < I) st4(out + (int64_t)row * J + col, v);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (with_colsum && tid < PB && (i0 + tid) < I) colsum_slab[(int64_t)sp * I + i0 + tid] = bsum;
        bsum = 0.f;
    };

    int it = 0, nit = 0;
    int m_stage = 0;
    set_item(0, m_stage);
    int cur_sp = item_split(0), cur_i0 = st_i0, cur_j0 = st_j0;
    int kt = 0, kend = split_steps(cur_sp);
    int nkt = 0, nkend = kend;
    bool cur_cs = colsum_slab != nullptr && cur_j0 == 0;
    dma(0, m_stage);
    zero_acc();
    dma_wait();
    zero_tail(0, m_stage, st_mend);
    __syncthreads();
    read_ra(0, 0);
    read_rb(0, 0);
    // Gather pipeline without a second raw register set: the A gather of the next 16-deep step is re-issued as
    // soon as A has been split, the B gather right after the LAST B sub-tile has been split (its 12 MFMAs are
    // still to come), and for the second step of a stage the stage barrier sits at that same point.
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        const bool more = (step + 1) < total;
        if (more) {
            if (++nkt == nkend) { ++nit; nkt = 0; set_item(nit, m_stage); nkend = split_steps(item_split(nit)); }
            else m_stage += BK;
            dma(buf ^ 1, m_stage);
        }
        // ---- 16-deep step 0
        conv_a();
        __builtin_amdgcn_sched_barrier(0);
        read_ra(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 3; ++b) { conv_b(b); mma12(b); }
        conv_b(3);
        __builtin_amdgcn_sched_barrier(0);
        read_rb(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma12(3);
        colsum(buf, cur_cs);
        // ---- 16-deep step 1
        conv_a();
#pragma unroll
        for (int b = 0; b < 3; ++b) { conv_b(b); mma12(b); }
        conv_b(3);
        __builtin_amdgcn_sched_barrier(0);
        dma_wait();
        if (more) zero_tail(buf ^ 1, m_stage, st_mend);
        __syncthreads();                    // next stage landed everywhere; this stage is fully read (ra/rb consumed)
        if (more) { read_ra(buf ^ 1, 0); read_rb(buf ^ 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mma12(3);
        if (++kt == kend) {
            epilogue(cur_sp, cur_i0, cur_j0, cur_cs);
            zero_acc();
            ++it;
            if (it < n_items) {
                cur_sp = item_split(it); kt = 0; kend = split_steps(cur_sp);
                cur_i0 = st_i0; cur_j0 = st_j0;
                cur_cs = colsum_slab != nullptr && cur_j0 == 0;
            }
        }
    }
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
// Narrow-N split-bf16 NT kernel, optionally with an implicit convolution gather on the A operand.
//
// The persistent 256x256 kernel above wastes 75 % / 50 % of its MFMAs when the weight operand has only 64 / 128 rows
// (the 1x1 and 3x3 convolutions of ResNet layer1 / layer2, the 7x7 stem). Here all 8 waves sit along M and every wave
// owns the full tile width: tile = (8 * RA * 32) x (NB * 32) with <RA, NB> = <2, 2> (512 x 64) or <1, 4> (256 x 128);
// no A fragment is converted twice (the 4x2 wave grid of the big kernel converts every A row in both wave columns).
// Same arithmetic (x = h + m + l, six terms, small first), same LDS image (XOR-swizzled 128-B fp32 rows for A,
// column-interleaved pre-swizzled bf16 planes for B), same unit pipeline (B planes one unit ahead, A one k16 step
// ahead, stage barrier before the last unit), cross-tile pipelining; no K-split (M / TM >> 256 tiles).
//
// CONV: A is never materialised. Row m = output pixel (b, oy, ox) of an NHWC activation X[B, H, W, C] (C % 32 == 0) and
// K = kh * kw * C in (ky, kx, c) order, so k-stage kt lies inside ONE tap (ky, kx) at channel offset c0: each lane's
// LDS-DMA source is X + ((b*H + oy*s - p + ky)*W + ox*s - p + kx)*C*4 + c0*4 + chunk*16, or - outside the image - a
// 16-B zero written to the lane's LDS slot instead of the DMA. This deletes the im2col kernels and their 9x HBM
// write + read traffic: neighbouring taps / tiles hit the same activation lines in the XCD's L2 (tiles are
// dealt to XCDs in contiguous ranges for that reason).
// ------------------------------------------------------------------------------------------
struct ConvGeom { int H, W, C, Ho, Wo, kw, stride, pad; };    // plain ints only (pointers in by-value structs become FLAT)
enum { GATHER_NONE = 0, GATHER_CONV = 1, GATHER_STEM = 2 };     // how the A operand of the narrow kernel is addressed

template <int NB>
__global__ __launch_bounds__(256) void split_planes_narrow_kernel(const float *__restrict__ src, int64_t sn, unsigned short *__restrict__ Bp,
                                                                   int N, int K, int tiles_n) {
    constexpr int TN = NB * 32;
    const int nk = K / BK;
    const int64_t total = (int64_t)tiles_n * nk * TN * 4;                 // one thread per (tile, stage, pos, logical chunk)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int chunk = (int)(t & 3), pos = (int)((t >> 2) % TN);
        const int64_t ts = (t >> 2) / TN;                                 // tile * nk + stage
        const int stage = (int)(ts % nk), tile = (int)(ts / nk);
        const int col = tile * TN + NB * (pos & 31) + (pos >> 5);         // pos = 32 b + li  <->  column NB * li + b
        const int phys = chunk ^ ((pos >> 2) & 3);
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = stage * BK + chunk * 8 + e;
            const float x = col < N ? src[(int64_t)col * sn + k] : 0.f;
            const unsigned hb = bf16_rn(__builtin_bit_cast(unsigned, x));
            const float r1 = x - __builtin_bit_cast(float, hb);
            const unsigned mb = bf16_rn(__builtin_bit_cast(unsigned, r1));
            const float r2 = r1 - __builtin_bit_cast(float, mb);
            h[e] = (unsigned short)(hb >> 16); m[e] = (unsigned short)(mb >> 16);
            l[e] = (unsigned short)(bf16_rn(__builtin_bit_cast(unsigned, r2)) >> 16);
        }
        unsigned short *dst = Bp + ts * (3 * TN * BK) + pos * BK + phys * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dst[e] = h[e]; dst[TN * BK + e] = m[e]; dst[2 * TN * BK + e] = l[e]; }
    }
}

template <int RA, int NB> struct NarrowCfg {
    static constexpr int TM = 8 * RA * 32, TN = NB * 32;
    static constexpr int A_BYTES = TM * BK * 4, PL_BYTES = TN * BK * 2;
    static constexpr int STAGE = A_BYTES + 3 * PL_BYTES, SMEM = 2 * STAGE;
    static constexpr int B_INSTR = 3 * PL_BYTES / 1024;                  // 1 KB (one wave-wide 16-B DMA) each
    static constexpr int A_INSTR = RA * 4;                                // per wave: 8 rows x 128 B each
};

template <int RA, int NB, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_nt_split_narrow_kernel(
    const float *__restrict__ A, int64_t lda, const unsigned short *__restrict__ Bp, float *C, int64_t ldc, int M, int N,
    int K, const float *__restrict__ bias, int relu, const float *addend, ConvGeom cg, int tiles_m, int tiles_n) {
    using Cfg = NarrowCfg<RA, NB>;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int nk = K / BK;

    // ---- work list: XCD x owns a contiguous range of tiles (row-major over (tile_m, tile_n)); slot s of the XCD's
    // 32 blocks takes tiles s, s + 32, ... of that range, so concurrently running tiles are neighbours in M
    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int tiles = tiles_m * tiles_n;
    const int t_lo = (int)(((int64_t)tiles * xcd) / kNumXCD), t_hi = (int)(((int64_t)tiles * (xcd + 1)) / kNumXCD);
    const int n_items = (t_hi - t_lo - slot + PB_BLOCKS_PER_XCD - 1) / PB_BLOCKS_PER_XCD;
    if (t_hi - t_lo <= slot) return;
    auto item_tile = [&](int i) { return t_lo + slot + i * PB_BLOCKS_PER_XCD; };
    const int total = n_items * nk;

    // ---- staging
    const char *Bpb = reinterpret_cast<const char *>(Bp);
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lane16 = lane * 16u;
    int aoff[Cfg::A_INSTR];              // per-lane byte offset of its row's 16-B chunk (CONV: of pixel (oy*s-p, ox*s-p), may be < 0)
    int ayx[MODE == GATHER_CONV ? Cfg::A_INSTR : 1];    // CONV: (oy*s - p + 8) << 16 | (ox*s - p + 8)
    auto dma1 = [&](const char *sbase, unsigned voff, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
    };
    auto set_tile = [&](int q, int &m0, int &n0, int &tn) {
        m0 = (q / tiles_n) * TM;
        tn = q % tiles_n;
        n0 = tn * TN;
#pragma unroll
        for (int jj = 0; jj < Cfg::A_INSTR; ++jj) {
            const int p = wave * (RA * 32) + 8 * jj + (lane >> 3);
            const int ch = (lane & 7) ^ ((p >> 1) & 7);
            const int m = min(m0 + p, M - 1);
            if (MODE == GATHER_CONV) {
                const int ox = m % cg.Wo, t = m / cg.Wo;
                const int oy = t % cg.Ho, b = t / cg.Ho;
                const int iy0 = oy * cg.stride - cg.pad, ix0 = ox * cg.stride - cg.pad;
                aoff[jj] = (((b * cg.H + iy0) * cg.W + ix0) * cg.C) * 4 + ch * 16;
                ayx[jj] = ((iy0 + 8) << 16) | (ix0 + 8);
            } else if (MODE == GATHER_STEM) {      // space-to-depth image Xs[b, Y, X, 12]: window of pixel (oy, ox) starts at (oy, ox)
                const int ox = m % cg.Wo, t = m / cg.Wo;
                const int oy = t % cg.Ho, b = t / cg.Ho;
                aoff[jj] = ((b * cg.H + oy) * cg.W + ox) * 48;
            } else {
                aoff[jj] = (int)((unsigned)m * (unsigned)(lda * 4)) + ch * 16;
            }
        }
    };
    // stage cursor of the stage being LOADED: k-stage index and, for CONV, its tap (ky, kx) and channel offset c0
    struct Cur { int kt, ky, kx, c0; };
    auto cur_reset = [&](Cur &c) { c.kt = 0; c.ky = 0; c.kx = 0; c.c0 = 0; };
    auto cur_next = [&](Cur &c) {
        ++c.kt;
        if (MODE == GATHER_CONV) { c.c0 += BK; if (c.c0 == cg.C) { c.c0 = 0; if (++c.kx == cg.kw) { c.kx = 0; ++c.ky; } } }
    };
    auto dma = [&](int buf, const Cur &c, int tn) {
        const unsigned dst = lds_base + (unsigned)buf * Cfg::STAGE;
        const char *Ab = reinterpret_cast<const char *>(A);
        if (MODE == GATHER_STEM) {
            // K = 4 window rows x 48 contiguous floats (4 s2d pixels x 12 channels): 16-B chunk g of the K row lives in
            // window row g / 12 at chunk g % 12; always inside the pre-padded image
#pragma unroll
            for (int jj = 0; jj < Cfg::A_INSTR; ++jj) {
                const int p = wave * (RA * 32) + 8 * jj + (lane >> 3);
                const int g = c.kt * 8 + ((lane & 7) ^ ((p >> 1) & 7));
                const int qy = (g * 43) >> 9;                         // g / 12 for g < 48
                dma1(Ab, (unsigned)(aoff[jj] + (qy * (3 * cg.W - 12) + g) * 16), dst + (wave * (RA * 32) + jj * 8) * (BK * 4));
            }
        } else if (MODE == GATHER_CONV) {
            const int soff = ((c.ky * cg.W + c.kx) * cg.C + c.c0) * 4;
#pragma unroll
            for (int jj = 0; jj < Cfg::A_INSTR; ++jj) {
                const unsigned ldst = dst + (wave * (RA * 32) + jj * 8) * (BK * 4);
                const int iy = (ayx[jj] >> 16) - 8 + c.ky, ix = (ayx[jj] & 0xFFFF) - 8 + c.kx;
                if ((unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W) {
                    dma1(Ab, (unsigned)(aoff[jj] + soff), ldst);
                } else {
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(smem) + (ldst - lds_base) + lane16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
            const char *ak = Ab + c.kt * (BK * 4);
#pragma unroll
            for (int jj = 0; jj < Cfg::A_INSTR; ++jj) dma1(ak, (unsigned)aoff[jj], dst + (wave * (RA * 32) + jj * 8) * (BK * 4));
        }
        const char *bk = Bpb + (int64_t)(tn * nk + c.kt) * (3 * Cfg::PL_BYTES);
#pragma unroll
        for (int i = 0; i < (Cfg::B_INSTR + 7) / 8; ++i) {
            const int q = wave + 8 * i;
            if (q < Cfg::B_INSTR) dma1(bk + q * 1024, lane16, dst + Cfg::A_BYTES + q * 1024);
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- fragment addressing
    const int swA = (li >> 1) & 7, swB = (li >> 2) & 3;
    const int aposf = (wave * (RA * 32) + li) * BK;                          // floats, + a*32*BK
    const int bposb = Cfg::A_BYTES + li * 64;                                // bytes, + plane*PL_BYTES + b*32*64
    auto read_a = [&](int buf, int s, f32x4 (&f)[RA][2]) {
        const float *base = smem + buf * (Cfg::STAGE / 4) + aposf;
        const int c0 = ((4 * s + 2 * hi) ^ swA) * 4, c1 = ((4 * s + 2 * hi + 1) ^ swA) * 4;
#pragma unroll
        for (int a = 0; a < RA; ++a) { f[a][0] = ld4(base + a * 32 * BK + c0); f[a][1] = ld4(base + a * 32 * BK + c1); }
    };
    auto read_b = [&](int buf, int s, int b, bf16x8 (&q)[3]) {
        const char *base = reinterpret_cast<const char *>(smem) + buf * Cfg::STAGE + bposb + b * 32 * 64 + ((2 * s + hi) ^ swB) * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) q[p] = *reinterpret_cast<const bf16x8 *>(base + p * Cfg::PL_BYTES);
    };
    f32x16 acc[RA][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    bf16x8 ap[RA][3], an[RA][3];
    f32x4 fa[RA][2];
    bf16x8 bq[3], bn[3];
    auto convert_a = [&](bf16x8 (&dst)[RA][3]) {
#pragma unroll
        for (int a = 0; a < RA; ++a) split8(fa[a][0], fa[a][1], dst[a][0], dst[a][1], dst[a][2]);
    };
    auto mma = [&](int b) {              // six terms, small ones first; row sub-tiles alternate
#define TOAD_T(PA, PB_) \
        _Pragma("unroll") for (int a = 0; a < RA; ++a) \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[a][PA], bq[PB_], acc[a][b], 0, 0, 0);
        TOAD_T(2, 0) TOAD_T(0, 2) TOAD_T(1, 1) TOAD_T(1, 0) TOAD_T(0, 1) TOAD_T(0, 0)
#undef TOAD_T
    };
    auto unit = [&](int b, int nbuf, int ns, int nb, bool fetch, bool conv) {
        if (fetch) read_b(nbuf, ns, nb, bn);
        __builtin_amdgcn_sched_barrier(0);
        mma(b);
        if (conv) convert_a(an);
        __builtin_amdgcn_sched_barrier(0);
        if (fetch) { bq[0] = bn[0]; bq[1] = bn[1]; bq[2] = bn[2]; }
        if (conv) {
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) ap[a][p] = an[a][p];
        }
    };

    typedef float fvec __attribute__((ext_vector_type(NB)));
    constexpr int EG = 16;               // residual rows loaded back to back (clamped addresses, no branch in between)
    auto epilogue = [&](int m0, int n0) {
        int lic = NB * li, hi4 = 4 * hi;
        asm volatile("" : "+v"(lic), "+v"(hi4));
        const int col = n0 + lic;
        const bool cok = col < N;
        fvec bv;
#pragma unroll
        for (int e = 0; e < NB; ++e) bv[e] = 0.f;
        if (bias && cok) bv = *reinterpret_cast<const fvec *>(bias + col);
        asm volatile("" : "+v"(bv));
        const bool has_add = addend != nullptr;
        const int colc = cok ? col : 0;
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            fvec av[EG];
#pragma unroll
            for (int g = 0; g < EG; ++g) {
                const int rowc = min(m0 + wave * (RA * 32) + a * 32 + (g & 3) + 8 * (g >> 2) + hi4, M - 1);
#pragma unroll
                for (int e = 0; e < NB; ++e) av[g][e] = 0.f;
                if (has_add) av[g] = *reinterpret_cast<const fvec *>(addend + (int64_t)rowc * ldc + colc);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * (RA * 32) + a * 32 + (r & 3) + 8 * (r >> 2) + hi4;
                if (cok && row < M) {
                    fvec v;
#pragma unroll
                    for (int e = 0; e < NB; ++e) v[e] = acc[a][e][r];
                    v += bv;
                    v += av[r];
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < NB; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    __builtin_nontemporal_store(v, reinterpret_cast<fvec *>(C + (int64_t)row * ldc + col));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int kt = 0, nit = 0;
    Cur nc;
    cur_reset(nc);
    int cur_m0, cur_n0, cur_tn, nxt_m0, nxt_n0, nxt_tn;
    set_tile(item_tile(0), cur_m0, cur_n0, cur_tn);
    nxt_m0 = cur_m0; nxt_n0 = cur_n0; nxt_tn = cur_tn;
    dma(0, nc, cur_tn);
    zero_acc();
    dma_wait();
    __syncthreads();
    read_a(0, 0, fa);
    read_b(0, 0, 0, bq);
    convert_a(ap);
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        const bool more = (step + 1) < total;
        if (more) {
            cur_next(nc);
            if (nc.kt == nk) { ++nit; cur_reset(nc); set_tile(item_tile(nit), nxt_m0, nxt_n0, nxt_tn); }
            dma(buf ^ 1, nc, nxt_tn);
        }
        // k16 step 0: units 0..NB-1 (raw A of step 1 is fetched up front and split during the last unit)
        read_a(buf, 1, fa);
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) unit(b, buf, 0, b + 1, true, false);
        unit(NB - 1, buf, 1, 0, true, true);
        // k16 step 1
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) unit(b, buf, 1, b + 1, true, false);
        dma_wait();
        __syncthreads();                    // next stage landed everywhere; this stage is fully read
        if (more) read_a(buf ^ 1, 0, fa);
        unit(NB - 1, buf ^ 1, 0, 0, more, more);
        if (++kt == nk) {
            epilogue(cur_m0, cur_n0);
            zero_acc();
            kt = 0;
            cur_m0 = nxt_m0; cur_n0 = nxt_n0; cur_tn = nxt_tn;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int narrow_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("TOAD_GEMM_NARROW"); v = e ? atoi(e) : 1; }   // A/B knob; default on
    return v;
}

// C[M,N] = act(A' W^T + bias + addend) with N <= 128 on the narrow kernels; A' = A[M,K] (geom == nullptr) or the implicit
// im2col of the NHWC activation A described by *geom. `ws` as for launch_nt (the bf16 planes live behind the slab area).
template <int RA, int NB, int MODE>
static int launch_narrow_t(const float *A, int64_t lda, const float *W, int64_t ldw, float *C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, const float *bias, int relu, const float *addend, const ConvGeom &cg, void *ws,
                           hipStream_t st, const char *what) {
    using Cfg = NarrowCfg<RA, NB>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_split_narrow_kernel<RA, NB, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_set = true;
    }
    const int tiles_m = (int)((M + Cfg::TM - 1) / Cfg::TM), tiles_n = (int)((N + Cfg::TN - 1) / Cfg::TN);
    unsigned short *planes = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(ws) + (size_t)PB_GRID * PB * PB * sizeof(float));
    const int64_t pthreads = (int64_t)tiles_n * (K / BK) * Cfg::TN * 4;
    int pgrid = (int)((pthreads + 255) / 256);
    if (pgrid > 4096) pgrid = 4096;
    hipLaunchKernelGGL(split_planes_narrow_kernel<NB>, dim3(pgrid), dim3(256), 0, st, W, ldw, planes, (int)N, (int)K, tiles_n);
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL((gemm_nt_split_narrow_kernel<RA, NB, MODE>), dim3(PB_GRID), dim3(512), Cfg::SMEM, st, A, lda, planes, C, ldc,
                       (int)M, (int)N, (int)K, bias, relu, addend, cg, tiles_m, tiles_n);
    return check_launch(what);
}

static int narrow_res_kmax() {
    static int v = -1;
    // measured (tools/gemm_shape_bench.py, same box): M=262144 K=64 N=256 +residual 170 -> 133 us on the 256x128 narrow tiles
    // (16 residual rows in flight per wave); K=128 equal, K=256 slower (A is re-split per 128-column tile) -> default 64
    if (v < 0) { const char *e = getenv("TOAD_NARROW_RES_KMAX"); v = e ? atoi(e) : 64; }
    return v;
}
static bool narrow_ok(int64_t M, int64_t N, int64_t K, int64_t ldc, const float *bias, const float *addend, void *ws) {
    const bool wide_ok = addend && K <= narrow_res_kmax();       // residual GEMMs with a short reduction: epilogue-bound, see DESIGN 10
    return narrow_enabled() && ws && (N <= 128 || wide_ok) && N % 4 == 0 && K % BK == 0 && ldc % 4 == 0 && M < (1ll << 31) &&
           (!bias || aligned16(bias)) && (!addend || aligned16(addend));
}

static int launch_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend,
                     const float *mask_src, void *ws, hipStream_t st, const char *what) {
    if (M <= 0 || N <= 0 || K <= 0) { set_error("%s: non-positive dimension", what); return TOAD_EINVAL; }
    if (M > INT32_MAX - BM || N > INT32_MAX - BN || K > INT32_MAX - BK) { set_error("%s: dimension too large", what); return TOAD_ESHAPE; }
    if (K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0) { set_error("%s: reduction dim %lld must be a multiple of 4", what, (long long)K); return TOAD_ESHAPE; }
    if (!aligned16(A) || !aligned16(B) || !aligned16(C)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NT_SMEM);
        attr_set = true;
    }
    static int use_big = -1;
    static float *slab_ws = nullptr;
    if (use_big < 0) {
        const char *e = getenv("TOAD_GEMM_BIG");           // A/B knob; default on
        use_big = e ? atoi(e) : 1;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_f32_big_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM);
    }
    static int use_split = -1;
    if (use_split < 0) {
        const char *e = getenv("TOAD_GEMM_SPLIT");         // A/B knob
        use_split = e ? atoi(e) : 1;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_split_big_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM);
    }
    if (use_big && use_split && !es.drop.thresh && !mask_src && ldb == K && (uint64_t)M * lda * 4 < (1ull << 32) &&
        narrow_ok(M, N, K, ldc, bias, addend, ws)) {
        const ConvGeom none{0, 0, 0, 0, 0, 0, 0, 0};
        if (N <= 64) return launch_narrow_t<2, 2, GATHER_NONE>(A, lda, B, ldb, C, ldc, M, N, K, bias, es.relu, addend, none, ws, st, what);
        return launch_narrow_t<1, 4, GATHER_NONE>(A, lda, B, ldb, C, ldc, M, N, K, bias, es.relu, addend, none, ws, st, what);
    }
    if (use_big && use_split && ws && K % BK == 0 && N % 4 == 0 && ldc % 4 == 0 && (uint64_t)M * lda * 4 < (1ull << 32)) {
        // B is given as B[n, k] = Bsrc[n * bsn + k * bsk]; split it into pre-swizzled bf16 planes behind the slabs
        const int tiles_m = (int)((M + PB - 1) / PB), tiles_n = (int)((N + PB - 1) / PB);
        unsigned short *planes = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(ws) + (size_t)PB_GRID * PB * PB * sizeof(float));
        const int64_t pthreads = (int64_t)tiles_n * (K / BK) * PB * 4;
        int pgrid = (int)((pthreads + 255) / 256);
        if (pgrid > 4096) pgrid = 4096;
        hipLaunchKernelGGL(split_planes_kernel, dim3(pgrid), dim3(256), 0, st, B, ldb, (int64_t)1, planes, (int)N, (int)K, tiles_n);
        int rc = check_launch(what);
        if (rc) return rc;
        hipLaunchKernelGGL(gemm_nt_split_big_kernel, dim3(PB_GRID), dim3(512), SP_SMEM, st, A, lda, planes, C, ldc, (int)M,
                           (int)N, (int)K, bias, es, addend, mask_src, (float *)ws, tiles_m, tiles_n);
        rc = check_launch(what);
        if (rc) return rc;
        int max_rem = 0;                                   // fix-up grid: only as many tile rows as some XCD has remainder tiles
        for (int x = 0; x < kNumXCD; ++x) { const int r = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)).rem; if (r > max_rem) max_rem = r; }
        if (max_rem > 0) {
            hipLaunchKernelGGL(nt_fixup_kernel, dim3(64, max_rem, kNumXCD), dim3(256), 0, st, (const float *)ws, C,
                               ldc, (int)M, (int)N, (int)K, bias, es, addend, mask_src, tiles_m, tiles_n);
            rc = check_launch(what);
        }
        return rc;
    }
    if (use_big && ws && K % BK == 0 && N % 4 == 0 && ldc % 4 == 0 && (uint64_t)M * lda * 4 < (1ull << 32) &&
        (uint64_t)N * ldb * 4 < (1ull << 32)) {
        const int tiles_m = (int)((M + PB - 1) / PB), tiles_n = (int)((N + PB - 1) / PB);
        hipLaunchKernelGGL(gemm_nt_f32_big_kernel, dim3(PB_GRID), dim3(512), PB_SMEM, st, A, lda, B, ldb, C, ldc, (int)M,
                           (int)N, (int)K, bias, es, addend, mask_src, (float *)ws, tiles_m, tiles_n);
        int rc = check_launch(what);
        if (rc) return rc;
        int max_rem = 0;                                   // fix-up grid: only as many tile rows as some XCD has remainder tiles
        for (int x = 0; x < kNumXCD; ++x) { const int r = nt_plan(x, tiles_m, tiles_n, (int)(K / BK)).rem; if (r > max_rem) max_rem = r; }
        if (max_rem > 0) {
            hipLaunchKernelGGL(nt_fixup_kernel, dim3(64, max_rem, kNumXCD), dim3(256), 0, st, (const float *)ws, C,
                               ldc, (int)M, (int)N, (int)K, bias, es, addend, mask_src, tiles_m, tiles_n);
            rc = check_launch(what);
        }
        return rc;
    }
    (void)slab_ws;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const int grid = kNumXCD * ((tiles_m + kNumXCD - 1) / kNumXCD) * tiles_n;
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
    // one 256x256 fp32 slab per persistent block (64 MiB) + the three bf16 planes of the weight operand
    const size_t tiles_n = (size_t)((N + PB - 1) / PB), kpad = (size_t)((K + BK - 1) / BK * BK);
    return (size_t)PB_GRID * PB * PB * sizeof(float) + tiles_n * PB * kpad * 6 + 256;
}

static int check_ws(void *ws, size_t ws_bytes, int64_t M, int64_t N, int64_t K, const char *what) {
    if (ws && ws_bytes < toad_linear_ws_bytes(M, N, K)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    if (ws && !aligned16(ws)) { set_error("%s: workspace must be 16-byte aligned", what); return TOAD_EALIGN; }
    return TOAD_OK;
}

extern "C" int toad_linear_act_fwd_f32(const float *X, const float *W, const float *bias, float *Y, int64_t M,
                                        int64_t K, int64_t N, int act, float drop_p, uint64_t drop_seed, void *ws,
                                        size_t ws_bytes, void *stream) {
    const char *what = "toad_linear_act_fwd_f32";
    if (!X || !W || !Y) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    if (int rc = check_ws(ws, ws_bytes, M, N, K, what)) return rc;
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(drop_p, drop_seed)};
    return launch_nt(X, K, W, K, Y, N, M, N, K, bias, es, nullptr, nullptr, ws, (hipStream_t)stream, what);
}

extern "C" int toad_linear_act_res_fwd_f32(const float *X, const float *W, const float *bias, const float *residual, float *Y,
                                            int64_t M, int64_t K, int64_t N, int act, void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_linear_act_res_fwd_f32";
    if (!X || !W || !Y) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (residual && (N % 4 != 0 || !aligned16(residual))) { set_error("%s: residual needs N %% 4 == 0 and 16-byte alignment", what); return TOAD_ESHAPE; }
    if (int rc = check_ws(ws, ws_bytes, M, N, K, what)) return rc;
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(0.f, 0)};
    return launch_nt(X, K, W, K, Y, N, M, N, K, bias, es, residual, nullptr, ws, (hipStream_t)stream, what);
}

extern "C" int toad_conv_nhwc_f32(const float *X, const float *Wf, const float *bias, const float *residual, float *Y, int B, int H,
                                  int W, int Cin, int kh, int kw, int stride, int pad, int Cout, int act, void *ws, size_t ws_bytes,
                                  void *stream) {
    const char *what = "toad_conv_nhwc_f32";
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
    const ConvGeom cg{H, W, Cin, Ho, Wo, kw, stride, pad};
    if (Cout <= 64)
        return launch_narrow_t<2, 2, GATHER_CONV>(X, 0, Wf, K, Y, Cout, M, Cout, K, bias, act == TOAD_ACT_RELU, residual, cg, ws, (hipStream_t)stream, what);
    return launch_narrow_t<1, 4, GATHER_CONV>(X, 0, Wf, K, Y, Cout, M, Cout, K, bias, act == TOAD_ACT_RELU, residual, cg, ws, (hipStream_t)stream, what);
}

extern "C" int toad_stem_conv_s2d_f32(const float *Xs, const float *Wf, const float *bias, float *Y, int B, int Ho, int Wo, int act,
                                      void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_stem_conv_s2d_f32";
    if (!Xs || !Wf || !Y || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (act != TOAD_ACT_NONE && act != TOAD_ACT_RELU) { set_error("%s: bad act %d", what, act); return TOAD_EINVAL; }
    if (B <= 0 || Ho <= 0 || Wo <= 0) { set_error("%s: bad geometry", what); return TOAD_ESHAPE; }
    const int Hs = Ho + 3, Ws = Wo + 3;
    const int64_t M = (int64_t)B * Ho * Wo;
    if ((uint64_t)B * Hs * Ws * 48 >= (1ull << 31) || M >= (1ll << 31)) { set_error("%s: batch too large for 32-bit offsets (split it)", what); return TOAD_ESHAPE; }
    if (!aligned16(Xs) || !aligned16(Wf) || !aligned16(Y)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (int rc = check_ws(ws, ws_bytes, M, 64, 192, what)) return rc;
    if (!narrow_ok(M, 64, 192, 64, bias, nullptr, ws)) { set_error("%s: bias must be 16-byte aligned", what); return TOAD_EALIGN; }
    const ConvGeom cg{Hs, Ws, 12, Ho, Wo, 0, 0, 0};
    return launch_narrow_t<2, 2, GATHER_STEM>(Xs, 0, Wf, 192, Y, 64, M, 64, 192, bias, act == TOAD_ACT_RELU, nullptr, cg, ws, (hipStream_t)stream, what);
}

extern "C" int toad_linear_dgrad_f32(const float *dY, const float *WT, const float *addend, const float *relu_src,
                                      float mask_scale, float *dX, int64_t M, int64_t N, int64_t K, void *ws,
                                      size_t ws_bytes, void *stream) {
    const char *what = "toad_linear_dgrad_f32";
    if (!dY || !WT || !dX) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (int rc = check_ws(ws, ws_bytes, M, K, N, what)) return rc;
    // dX[M,K] = dY[M,N] . WT[K,N]^T : an NT product with reduction dim N
    EpiScalars es{0, mask_scale, make_drop(0.f, 0)};
    return launch_nt(dY, N, WT, N, dX, K, M, K, N, nullptr, es, addend, relu_src, ws, (hipStream_t)stream, what);
}

static bool tn_big_ok(int64_t M, int64_t N, int64_t K) {
    static int use_big = -1;
    if (use_big < 0) {
        const char *e = getenv("TOAD_GEMM_BIG");           // A/B knob; default on
        use_big = e ? atoi(e) : 1;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_f32_big_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM);
    }
    return use_big && M >= 64 && N >= 4 && K >= 4;
}

extern "C" size_t toad_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const WgradPlan p = wgrad_plan(M, N, K);
    const TnPlan q = tn_plan(M, N, K);
    const int ns = p.nsplit > q.nsplit ? p.nsplit : q.nsplit;
    return (size_t)ns * (size_t)(N * K + N) * sizeof(float);
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
    float *slab = (float *)ws;
    int nsplit;
    int rc;
    if (tn_big_ok(M, N, K)) {
        const TnPlan q = tn_plan(M, N, K);
        nsplit = q.nsplit;
        float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
        static int tn_split = -1;
        if (tn_split < 0) {
            const char *e = getenv("TOAD_GEMM_SPLIT");     // A/B knob
            tn_split = e ? atoi(e) : 1;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_split_big_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM);
        }
        if (tn_split)
            hipLaunchKernelGGL(gemm_tn_split_big_kernel, dim3(PB_GRID), dim3(512), PB_SMEM, st, dY, N, X, K, slab, cs, (int)M,
                               (int)N, (int)K, q.rows_per_split, q.ti, q.tj, q.nsplit);
        else
            hipLaunchKernelGGL(gemm_tn_f32_big_kernel, dim3(PB_GRID), dim3(512), PB_SMEM, st, dY, N, X, K, slab, cs, (int)M,
                               (int)N, (int)K, q.rows_per_split, q.ti, q.tj, q.nsplit);
        rc = check_launch(what);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM);
            attr_set = true;
        }
        const WgradPlan p = wgrad_plan(M, N, K);
        nsplit = p.nsplit;
        float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
        const int tiles = p.tiles_i * p.tiles_j;
        const int grid = kNumXCD * ((p.nsplit + kNumXCD - 1) / kNumXCD) * tiles;
        hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(grid), dim3(256), TN_SMEM, st, dY, N, X, K, slab, cs, (int)M, (int)N,
                           (int)K, p.rows_per_split, p.tiles_i, p.tiles_j, p.nsplit);
        rc = check_launch(what);
    }
    if (rc) return rc;
    float *cs = db ? slab + (size_t)nsplit * N * K : nullptr;
    const int64_t n = N * K, n2 = db ? N : 0;
    int rgrid = (int)(((n + n2) / 4 + 255) / 256);
    if (rgrid > 4096) rgrid = 4096;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rgrid), dim3(256), 0, st, slab, dW, n, cs, db, n2, nsplit, beta);
    return check_launch(what);
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

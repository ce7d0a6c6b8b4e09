// gated_pool.hip — fused gated-attention scores + online-softmax pooling (forward and backward).
//
// Replaces, in ONE pass over the bag (models/model_toad.py:37-40 and :92,:97-98):
//     a = tanh(Pa); b = sigmoid(Pb); A_raw = (a*b) Wc^T + bc; A = softmax_N(A_raw^T); M = A @ H
// HBM-bound: 4*(2D+L+T) bytes per patch are read exactly once (5,128 B at D=384, L=512, T=2),
// ~7.4 kFLOP per patch -> 1.4 FLOP/B.  No operand is re-read, nothing is staged twice.
//
// Work decomposition (gfx950, wave64):
//  * 32 lanes own one patch row (LPR), so a wave streams 2 rows per step; every 32-lane group reads
//    contiguous 512-B pieces of its row with 16-B/lane NON-TEMPORAL loads straight into registers (the rows are
//    read exactly once, so there is nothing to stage through LDS: a register stream reaches 0.92-0.94 of what a
//    pure read stream gets on this part). A wave step keeps 3+3 (Pa,Pb) + 4 (H) dwordx4 loads in flight per lane.
//  * the D-long dot products with Wc are reduced with four DPP steps inside each 16-lane DPP row
//    (quad_perm, quad_perm, row_half_mirror, row_mirror) and one cross-row exchange: no LDS, no ds_bpermute.
//  * online softmax: each 32-lane group carries (m_t, l_t, acc_t[L/32 columns]) in registers;
//    the accumulator is rescaled only when a step raises the running max (wave-uniform branch).
//  * groups -> waves -> block are merged once at the end through LDS, each block writes one
//    (m, l, acc) partial; a small second kernel merges the partials, normalises and also
//    emits (max, sum) per task for the backward pass.
// Row steps are dealt block-cyclically so all blocks sweep HBM together; two 4-wave workgroups per CU (grid 512) keep the
// memory pipe busy while one of them computes.
#include "common.h"

#include <math.h>
#include <stdlib.h>

namespace toad {

constexpr float kLog2e = 1.4426950408889634f;
#ifndef TOAD_POOL_WAVES
#define TOAD_POOL_WAVES 4
#endif
constexpr int NW = TOAD_POOL_WAVES;        // waves per block
constexpr int POOL_THREADS = 64 * NW;
constexpr int LPR = 32;                    // lanes per patch row (a wave streams 64/LPR rows per step)
constexpr int RPW = 64 / LPR;              // rows per wave step
constexpr int ROWS_PER_BLOCK_STEP = NW * RPW;

// all-reduce over the LPR lanes that share a row (DPP inside 16-lane rows, one cross-row exchange)
__device__ __forceinline__ float row_allreduce_sum(float v) {
    v = row16_allreduce_sum(v);
    if (LPR == 32) v += __shfl_xor(v, 16);
    return v;
}
// sum over the RPW row-groups of a wave: lanes c, c+LPR, ... hold the same columns
__device__ __forceinline__ float groups_sum(float v) {
    if (LPR == 16) v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// streamed-once operands (P, H rows) use non-temporal loads: measured in the training step
// (bench.py, N=100k) the fused forward drops 105.6 -> 80.7 us and the backward 182 -> 167 us
// (the plain-load arm was deleted with the other A/B switches in round 4; docs/HISTORY.md has the measurement).
__device__ __forceinline__ f32x4 ld4s(const float *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
}

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }

// a = tanh(x), b = sigmoid(y) from two v_exp_f32 and one v_rcp_f32.
//   u = e^{2x} (x clamped to +-20: tanh saturates to 1-4e-18, u stays finite), v = e^{-y}
//   a = (u-1)/(u+1), b = 1/(1+v).  Absolute error ~1e-7 (fp32 eps) everywhere.
// (round 3: ONE reciprocal for both - r = 1 / ((u+1)(1+v)), a = (u-1)(1+v) r, b = (u+1) r - the backward kernel is co-limited by its
//  transcendental rate, and this is a quarter of them. y is clamped at -40 so that (u+1)(1+v) <= e^80 stays finite: sigmoid(-40) = 4e-18.)
__device__ __forceinline__ void gate_ab(float x, float y, float &a, float &b) {
    x = __builtin_fminf(__builtin_fmaxf(x, -20.f), 20.f);
    y = __builtin_fmaxf(y, -40.f);
    const float u = __builtin_amdgcn_exp2f(x * (2.f * kLog2e));
    const float v1 = 1.f + __builtin_amdgcn_exp2f(y * (-kLog2e));
    const float r = __builtin_amdgcn_rcpf((u + 1.f) * v1);
    a = ((u - 1.f) * v1) * r;
    b = (u + 1.f) * r;
}
// g = a*b with a single reciprocal: (u-1) / ((u+1)(1+v))
__device__ __forceinline__ float gate_g(float x, float y) {
    x = __builtin_fminf(__builtin_fmaxf(x, -20.f), 20.f);
    const float u = __builtin_amdgcn_exp2f(x * (2.f * kLog2e));
    const float v = __builtin_amdgcn_exp2f(y * (-kLog2e));
    return (u - 1.f) * __builtin_amdgcn_rcpf((u + 1.f) * (1.f + v));
}

// Partial record written by each block: [T][L] acc then [T][2] (m,l), padded to a multiple of 4 floats so that every record
// (and the 16-byte accesses into it) stays 16-byte aligned for any T
__host__ __device__ inline int64_t pool_partial_floats(int L, int T) { return ((int64_t)T * L + 2 * T + 3) & ~(int64_t)3; }

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// GEN = false: the shape IS the template (T tasks, D = 128 DQ, L = 128 LQ): no masking anywhere - the tuned instantiations of the
// shapes TOAD builds (models/model_toad.py:56,66) and of Attn_Net_Gated's defaults (:19).
// GEN = true: ONE covering instantiation (T = 4, D <= 512, L <= 1024) serves every other shape Attn_Net_Gated's constructor can be
// given within those limits (any D, L multiple of 4 / 8, any n_tasks <= 4): the run-time shape (Dr, Lr, Tr) sets the strides, and
// columns / tasks beyond it carry zeros (zero attention weights, no loads, no stores).
template <int T, int DQ /* = D/(4*LPR) float4 per lane */, int LQ /* = L/(4*LPR) float4 per lane */, bool POOL, bool GEN = false>
__global__ __launch_bounds__(POOL_THREADS) void gated_pool_fwd_kernel(
    const float *__restrict__ Pa, const float *__restrict__ Pb, int64_t ldp, const float *__restrict__ H,
    const float *__restrict__ Wc, const float *__restrict__ bc, float *__restrict__ A_raw,
    float *__restrict__ partials, int N, DropArgs drop_a, DropArgs drop_b, int Dr, int Lr, int Tr, const int64_t *__restrict__ seg) {
    constexpr int D = DQ * 4 * LPR, L = LQ * 4 * LPR;
    const int D_ = GEN ? Dr : D, L_ = GEN ? Lr : L, T_ = GEN ? Tr : T;      // run-time shape (== the template's unless GEN)
    // Batched launch (seg != NULL; the ragged multi-slide step): blockIdx.y = slide, whose rows [seg[y], seg[y+1]) of the shared
    // activations are pooled on their own; every (slide, block) writes its own partial record, workgroups beyond a short slide's
    // rows write an empty one (max = -inf), which the merge ignores.
    if (seg) {
        const int64_t r0 = seg[blockIdx.y];
        N = (int)(seg[blockIdx.y + 1] - r0);
        Pa += r0 * ldp; Pb += r0 * ldp; A_raw += r0 * T_;
        if (POOL) H += r0 * L_;
        // slide y draws its tanh- / sigmoid-branch masks with seeds + 2 y G: the base seeds are seed + 3 G and seed + 4 G (step.hip drop_seeds), so the
        // tanh seeds stay on odd and the sigmoid seeds on even multiples of G and no slide's mask repeats another slide's other branch
        drop_a.seed += (uint64_t)blockIdx.y * (2ull * 0x9E3779B97F4A7C15ull); drop_b.seed += (uint64_t)blockIdx.y * (2ull * 0x9E3779B97F4A7C15ull);
    }
    const int pblk = blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, c = lane % LPR;     // LPR-lane group = one row; c = float4 slot
    const bool dropping = drop_a.thresh != 0;       // train-mode Dropout(0.25) after tanh and after sigmoid
    auto dcol_ok = [&](int j) { return !GEN || (c + LPR * j) * 4 < D_; };
    auto lcol_ok = [&](int j) { return !GEN || (c + LPR * j) * 4 < L_; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Wc columns owned by this lane: float4 index c + LPR j
    f32x4 wc[T][DQ];
    float bcv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        bcv[t] = (t < T_) ? bc[t] : 0.f;
#pragma unroll
        for (int j = 0; j < DQ; ++j) wc[t][j] = (t < T_ && dcol_ok(j)) ? ld4(Wc + t * D_ + (c + LPR * j) * 4) : zero4;
    }

    float m[T], l[T];
    f32x4 acc[T][POOL ? LQ : 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int j = 0; j < (POOL ? LQ : 1); ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (N + ROWS_PER_BLOCK_STEP - 1) / ROWS_PER_BLOCK_STEP;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row = tile * ROWS_PER_BLOCK_STEP + wave * RPW + grp;
        const bool valid = row < N;
        const int64_t rr = valid ? row : 0;
        const float *pa = Pa + rr * ldp + c * 4;
        const float *pb = Pb + rr * ldp + c * 4;
        f32x4 xa[DQ], xb[DQ];
#pragma unroll
        for (int j = 0; j < DQ; ++j) {
            xa[j] = dcol_ok(j) ? ld4s(pa + 4 * LPR * j) : zero4;
            xb[j] = dcol_ok(j) ? ld4s(pb + 4 * LPR * j) : zero4;
        }
        f32x4 hv[POOL ? LQ : 1];
        if (POOL) {
            const float *hp = H + rr * L_ + c * 4;
#pragma unroll
            for (int j = 0; j < LQ; ++j) hv[j] = lcol_ok(j) ? ld4s(hp + 4 * LPR * j) : zero4;
        }

        float s[T];
#pragma unroll
        for (int t = 0; t < T; ++t) s[t] = 0.f;
#pragma unroll
        for (int j = 0; j < DQ; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g;
                if (dropping) {
                    float a, b;
                    gate_ab(xa[j][e], xb[j][e], a, b);
                    const uint64_t idx = (uint64_t)rr * D_ + (uint64_t)((c + LPR * j) * 4 + e);
                    g = (a * drop_keep(idx, drop_a)) * (b * drop_keep(idx, drop_b));
                } else {
                    g = gate_g(xa[j][e], xb[j][e]);
                }
#pragma unroll
                for (int t = 0; t < T; ++t) s[t] = fmaf(g, wc[t][j][e], s[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) s[t] = row_allreduce_sum(s[t]) + bcv[t];

        if (valid && c == 0) {
            if (T == 2 && !GEN) {
                *reinterpret_cast<f32x2 *>(A_raw + (int64_t)row * 2) = f32x2{s[0], s[T - 1]};
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (t < T_) A_raw[(int64_t)row * T_ + t] = s[t];
            }
        }

        if (POOL) {
            float mn[T];
            bool grow = false;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                mn[t] = valid ? __builtin_fmaxf(m[t], s[t]) : m[t];
                grow |= mn[t] > m[t];
            }
            if (__any(grow)) {   // wave-uniform: rescale the running sums to the new max
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float f = (mn[t] > m[t]) ? fast_exp(m[t] - mn[t]) : 1.f;   // exp(-inf)=0 on first row
                    l[t] *= f;
#pragma unroll
                    for (int j = 0; j < LQ; ++j) acc[t][j] *= f;
                    m[t] = mn[t];
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float w = valid ? fast_exp(s[t] - m[t]) : 0.f;
                l[t] += w;
#pragma unroll
                for (int j = 0; j < LQ; ++j) acc[t][j] += w * hv[j];
            }
        }
    }

    if (!POOL) return;

    // ---- merge the 16 (group, wave) partials of this block -------------------------------
    __shared__ float sm_m[NW * RPW][T];
    // exact shapes: one image per wave, summed by all threads; GEN (T*L up to 4096 floats): ONE image the waves add into in turn
    // (64 KB of static LDS is the limit, and this code runs once per workgroup)
    __shared__ __attribute__((aligned(16))) float sm_acc[GEN ? 1 : NW][T][L];
    __shared__ float sm_l[NW][T];
    __shared__ float sm_mb[T];
    if (c == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) sm_m[wave * RPW + grp][t] = m[t];
    }
    __syncthreads();
    f32x4 mg[T][LQ];                       // this lane's columns, rescaled to the block maximum and summed over the wave's row groups
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float mb = sm_m[0][t];
#pragma unroll
        for (int k = 1; k < NW * RPW; ++k) mb = __builtin_fmaxf(mb, sm_m[k][t]);
        const float f = (m[t] == -INFINITY) ? 0.f : fast_exp(m[t] - mb);
        float lt = groups_sum(l[t] * f);      // every lane of a row group carries the same l
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            f32x4 v = acc[t][j] * f;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = groups_sum(v[e]);
            mg[t][j] = v;
            if (!GEN && grp == 0) st4(&sm_acc[wave][t][(c + LPR * j) * 4], v);
        }
        if (lane == 0) sm_l[wave][t] = lt;
        if (tid == 0) sm_mb[t] = mb;   // block max (same value in every thread)
    }
    if (GEN) {
        for (int wv = 0; wv < NW; ++wv) {      // fixed order: deterministic
            if (wave == wv && grp == 0) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int j = 0; j < LQ; ++j) {
                        float *dst = &sm_acc[0][t][(c + LPR * j) * 4];
                        st4(dst, wv == 0 ? mg[t][j] : ld4(dst) + mg[t][j]);
                    }
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
    }
    float *out = partials + (int64_t)pblk * pool_partial_floats(L_, T_);
    for (int e = tid; e < T_ * L_ / 4; e += POOL_THREADS) {
        const int t = e / (L_ / 4), q = e % (L_ / 4);
        f32x4 v = ld4(&sm_acc[0][t][q * 4]);
        if (!GEN) {
#pragma unroll
            for (int w = 1; w < NW; ++w) v += ld4(&sm_acc[w][t][q * 4]);
        }
        st4(out + t * L_ + q * 4, v);
    }
    if (tid < T_) {
        float lsum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) lsum += sm_l[w][tid];
        out[T_ * L_ + 2 * tid] = sm_mb[tid];
        out[T_ * L_ + 2 * tid + 1] = lsum;
    }
}

// Merge G block partials: M[t,:] = sum_b exp(m_b - m) acc_b / l ; stats[t] = (m, l).
// grid = T * ceil(L/32) blocks; block = 256 threads = 8 float4 column groups (32 columns) x 32 partial slices. Every block recomputes
// the (tiny) global max / sum of its task; exp2(-inf) = 0 makes empty partials vanish without a branch. The launch is latency: a chain
// of dependent steps (max -> weights -> weighted accumulation -> store). Here ALL global loads of a thread - its (max, sum) pairs and
// the 16 partial rows of its columns - are issued up front in one round trip (the rows do not depend on the maximum, only their
// weights do; those go through LDS), and the reductions are wave shuffles plus one LDS exchange between the four waves instead of
// eight-step LDS trees and a serial 128-term tail. The pooling forward this kernel finishes streams its 512.8 MB in 75 us, so every
// microsecond here is 1 % of the op.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void gated_pool_combine_kernel(const float *__restrict__ partials, int G, int L,
                                                                  int T, float *__restrict__ M,
                                                                  float *__restrict__ stats, int m_stride, int s_stride) {
    // blockIdx.y = slide of a batched launch (0 otherwise): its G partial records, its M / stats record
    partials += (int64_t)blockIdx.y * G * pool_partial_floats(L, T);
    M += (int64_t)blockIdx.y * m_stride; stats += (int64_t)blockIdx.y * s_stride;
    __shared__ float red[2][4];
    __shared__ float wgt[512];                       // exp(m_b - m) of the first 512 partials (the launch grid of the forward is capped there)
    __shared__ __attribute__((aligned(16))) float sacc[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rec = pool_partial_floats(L, T);
    const int blocks_per_t = (L + 31) / 32;
    const int t = blockIdx.x / blocks_per_t, col0 = (blockIdx.x % blocks_per_t) * 32;
    const float *ml = partials + T * L + 2 * t;
    const int q = tid & 7, slice = tid >> 3;
    const bool cok = col0 + q * 4 < L;
    const float *pcol = partials + t * L + col0 + q * 4;

    // round trip 1: the (max, sum) pairs this thread owns AND - they do not depend on them - the first 16 partial rows of its columns
    float pm[2], pl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = tid + 256 * i;
        pm[i] = b < G ? ml[b * rec] : -INFINITY;
        pl[i] = b < G ? ml[b * rec + 1] : 0.f;
    }
    f32x4 pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int b = slice + 32 * i;
        pv[i] = (cok && b < G) ? ld4(pcol + b * rec) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float mx = __builtin_fmaxf(pm[0], pm[1]);
    for (int b = tid + 512; b < G; b += 256) mx = __builtin_fmaxf(mx, ml[b * rec]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[0][wave] = mx;
    __syncthreads();
    mx = __builtin_fmaxf(__builtin_fmaxf(red[0][0], red[0][1]), __builtin_fmaxf(red[0][2], red[0][3]));
    const float w0 = fast_exp(pm[0] - mx), w1 = fast_exp(pm[1] - mx);
    wgt[tid] = w0; wgt[tid + 256] = w1;
    float ls = pl[0] * w0 + pl[1] * w1;
    for (int b = tid + 512; b < G; b += 256) ls += ml[b * rec + 1] * fast_exp(ml[b * rec] - mx);
    ls = wave_sum(ls);
    if (lane == 0) red[1][wave] = ls;
    __syncthreads();
    // weighted accumulation in a fixed order: partials slice, slice + 32, ... (all already in registers)
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) a += wgt[slice + 32 * i] * pv[i];
    if (cok) {
        for (int b = slice + 512; b < G; b += 32) a += fast_exp(ml[b * rec] - mx) * ld4(pcol + b * rec);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {            // sum the 8 slices of this wave that share q (lane bits 3..5), fixed order
        a[e] += __shfl_xor(a[e], 8);
        a[e] += __shfl_xor(a[e], 16);
        a[e] += __shfl_xor(a[e], 32);
    }
    if (lane < 8) st4(&sacc[wave][q * 4], a);
    ls = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    __syncthreads();
    if (tid < 32 && col0 + tid < L) M[t * L + col0 + tid] = ((sacc[0][tid] + sacc[1][tid]) + (sacc[2][tid] + sacc[3][tid])) / ls;
    if (blockIdx.x % blocks_per_t == 0 && tid == 0) {
        stats[2 * t] = mx;
        stats[2 * t + 1] = ls;
    }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// Per-block partial: [T][D] dWc then [T] dbc, padded to a multiple of 4 floats (16-byte aligned records)
__host__ __device__ inline int64_t bwd_partial_floats(int D, int T) { return ((int64_t)T * D + T + 3) & ~(int64_t)3; }

template <int T, int DQ, int LQ, bool GEN = false>      // GEN: see gated_pool_fwd_kernel
__global__ __launch_bounds__(POOL_THREADS) void gated_pool_bwd_kernel(
    const float *__restrict__ Pa, const float *__restrict__ Pb, int64_t ldp, const float *__restrict__ H,
    const float *__restrict__ Wc, const float *__restrict__ A_raw, const float *__restrict__ stats,
    const float *__restrict__ Mp, const float *__restrict__ dM, const float *__restrict__ dA_ext,
    float *__restrict__ dPa, float *__restrict__ dPb, int64_t ldd, float *__restrict__ dH,
    float *__restrict__ partials, float *__restrict__ dp_amax, int N, DropArgs drop_a, DropArgs drop_b, int Dr, int Lr, int Tr,
    const int64_t *__restrict__ seg, int m_stride, int s_stride) {
    constexpr int D = DQ * 4 * LPR, L = LQ * 4 * LPR;
    const int D_ = GEN ? Dr : D, L_ = GEN ? Lr : L, T_ = GEN ? Tr : T;
    if (seg) {          // batched launch: blockIdx.y = slide (see gated_pool_fwd_kernel); its softmax statistics, M and dM records
        const int64_t r0 = seg[blockIdx.y];
        N = (int)(seg[blockIdx.y + 1] - r0);
        Pa += r0 * ldp; Pb += r0 * ldp; H += r0 * L_; A_raw += r0 * T_; dPa += r0 * ldd; dPb += r0 * ldd;
        if (dH) dH += r0 * L_;
        if (dA_ext) dA_ext += r0 * T_;
        if (dp_amax) dp_amax += r0;                  // batched: ONE bound per row of the concatenation (slides do not line up with the 256-row blocks)
        stats += (int64_t)blockIdx.y * s_stride; Mp += (int64_t)blockIdx.y * m_stride; dM += (int64_t)blockIdx.y * m_stride;
        // slide y draws its tanh- / sigmoid-branch masks with seeds + 2 y G: the base seeds are seed + 3 G and seed + 4 G (step.hip drop_seeds), so the
        // tanh seeds stay on odd and the sigmoid seeds on even multiples of G and no slide's mask repeats another slide's other branch
        drop_a.seed += (uint64_t)blockIdx.y * (2ull * 0x9E3779B97F4A7C15ull); drop_b.seed += (uint64_t)blockIdx.y * (2ull * 0x9E3779B97F4A7C15ull);
    }
    const int pblk = blockIdx.y * gridDim.x + blockIdx.x;
    const bool dropping = drop_a.thresh != 0;
    __shared__ __attribute__((aligned(16))) float s_dm[T][L];
    __shared__ __attribute__((aligned(16))) float s_wc[T][D];
    __shared__ __attribute__((aligned(16))) float s_red[NW][T][D];
    __shared__ float s_c[T];
    __shared__ float s_db[NW][T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, c = lane % LPR;
    auto dcol_ok = [&](int j) { return !GEN || (c + LPR * j) * 4 < D_; };
    auto lcol_ok = [&](int j) { return !GEN || (c + LPR * j) * 4 < L_; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (GEN) {          // columns / tasks beyond the run-time shape read as zeros below
        for (int e = tid; e < T * L; e += POOL_THREADS) s_dm[e / L][e % L] = 0.f;
        for (int e = tid; e < T * D; e += POOL_THREADS) s_wc[e / D][e % D] = 0.f;
        __syncthreads();
    }
    for (int e = tid; e < T_ * L_; e += POOL_THREADS) s_dm[e / L_][e % L_] = dM[e];
    for (int e = tid; e < T_ * D_; e += POOL_THREADS) s_wc[e / D_][e % D_] = Wc[e];
    // c_t = dM[t] . M[t]  (one wave per task, fixed order)
    if (wave < T_) {
        float p = 0.f;
        for (int e = lane; e < L_; e += 64) p = fmaf(dM[wave * L_ + e], Mp[wave * L_ + e], p);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
        if (lane == 0) s_c[wave] = p;
    }
    __syncthreads();

    float mt[T], il[T], ct[T], wcm[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        mt[t] = (t < T_) ? stats[2 * t] : 0.f;
        il[t] = (t < T_) ? 1.f / stats[2 * t + 1] : 0.f;
        ct[t] = (t < T_) ? s_c[t] : 0.f;
        float mw = 0.f;                                     // max |Wc[t,:]| for the abs-max bound of dP below
        for (int e = lane; e < D; e += 64) mw = __builtin_fmaxf(mw, __builtin_fabsf(s_wc[t][e]));
        wcm[t] = h2_wave_max(mw);
    }
    f32x4 dwc[T][DQ];
    float dbc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        dbc[t] = 0.f;
#pragma unroll
        for (int j = 0; j < DQ; ++j) dwc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (N + ROWS_PER_BLOCK_STEP - 1) / ROWS_PER_BLOCK_STEP;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row = tile * ROWS_PER_BLOCK_STEP + wave * RPW + grp;
        const bool valid = row < N;
        const int64_t rr = valid ? row : 0;
        // issue every load of the step up front
        const float *hp = H + rr * L_ + c * 4;
        f32x4 hv[LQ];
#pragma unroll
        for (int j = 0; j < LQ; ++j) hv[j] = lcol_ok(j) ? ld4s(hp + 4 * LPR * j) : zero4;
        const float *pa = Pa + rr * ldp + c * 4;
        const float *pb = Pb + rr * ldp + c * 4;
        f32x4 xa[DQ], xb[DQ];
#pragma unroll
        for (int j = 0; j < DQ; ++j) {
            xa[j] = dcol_ok(j) ? ld4s(pa + 4 * LPR * j) : zero4;
            xb[j] = dcol_ok(j) ? ld4s(pb + 4 * LPR * j) : zero4;
        }
        float p[T], ds[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float s = (t < T_) ? A_raw[rr * T_ + t] : 0.f;
            p[t] = (valid && t < T_) ? fast_exp(s - mt[t]) * il[t] : 0.f;
            ds[t] = (dA_ext && valid && t < T_) ? dA_ext[rr * T_ + t] : 0.f;
        }

        // --- H side: dot_t = dM[t].H[row], dH[row] = sum_t p_t dM[t] (dH == NULL: the dgrad of the attention Linear recomputes it)
        float dot[T];
#pragma unroll
        for (int t = 0; t < T; ++t) dot[t] = 0.f;
        float *dhp = dH + rr * L_ + c * 4;
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 d = ld4(&s_dm[t][(c + LPR * j) * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot[t] = fmaf(d[e], hv[j][e], dot[t]);
                o += p[t] * d;
            }
            if (valid && dH && lcol_ok(j)) st4(dhp + 4 * LPR * j, o);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            dot[t] = row_allreduce_sum(dot[t]);
            ds[t] += p[t] * (dot[t] - ct[t]);
            if (c == 0) dbc[t] += ds[t];
        }

        // --- P side
        float *dpa = dPa + rr * ldd + c * 4;
        float *dpb = dPb + rr * ldd + c * 4;
#pragma unroll
        for (int j = 0; j < DQ; ++j) {
            f32x4 oa, ob;
            f32x4 w[T];
#pragma unroll
            for (int t = 0; t < T; ++t) w[t] = ld4(&s_wc[t][(c + LPR * j) * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a, b;
                gate_ab(xa[j][e], xb[j][e], a, b);
                float ka = 1.f, kb = 1.f;           // dropout multipliers (0 or 1/(1-p)), recomputed from the seed
                if (dropping) {
                    const uint64_t idx = (uint64_t)rr * D_ + (uint64_t)((c + LPR * j) * 4 + e);
                    ka = drop_keep(idx, drop_a);
                    kb = drop_keep(idx, drop_b);
                }
                const float ad = a * ka, bd = b * kb;
                const float g = ad * bd;
                float dg = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    dg = fmaf(ds[t], w[t][e], dg);
                    dwc[t][j][e] = fmaf(ds[t], g, dwc[t][j][e]);
                }
                oa[e] = dg * bd * ka * (1.f - a * a);
                ob[e] = dg * ad * kb * b * (1.f - b);
            }
            if (valid && dcol_ok(j)) {
                st4(dpa + 4 * LPR * j, oa);
                st4(dpb + 4 * LPR * j, ob);
            }
        }
        if (dp_amax) {
            // (dp_amax here = the fine table in the workspace.) The abs-max array only has to BOUND |dP| from above (it picks a power-of-two scale with 2 bits of headroom to spare
            // per 4x of slack), so instead of reducing 24 stored values per lane it uses, per row,
            //   |dPa|, |dPb| <= |dg| ka kb,  |dg| <= sum_t |dS_t| max_e |Wc[t,e]|     (|a|, b, 1-a^2 <= 1; ka, kb <= 1/(1-p))
            // which every lane of the row already holds: one cross-row exchange and one atomic per wave step. A step's rows lie
            // inside one 256-row block (256 % ROWS_PER_BLOCK_STEP == 0).
            float bound = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) bound = fmaf(__builtin_fabsf(ds[t]), wcm[t], bound);
            if (dropping) bound *= drop_a.scale * drop_b.scale;
            bound = valid ? bound : 0.f;
            if (seg) {                                   // batched: per-row table, folded 256 rows per slot by bwd_partial_reduce_kernel
                if (c == 0 && valid) dp_amax[row] = bound;
            } else {
                if (RPW == 2) bound = __builtin_fmaxf(bound, __shfl_xor(bound, 32));
                else bound = h2_wave_max(bound);
                // one plain store per wave step into a fine-grained table; bwd_partial_reduce_kernel folds 128 entries into each
                // 256-row slot. (Device-scope atomics bypass the per-XCD L2s: 50,000 of them cost this kernel 30-40 us at N = 100k.)
                if (lane == 0) dp_amax[tile * NW + wave] = bound;
            }
        }
    }

    // ---- block reduction of dWc / dbc partials (fixed order) ----------------------------
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float b = groups_sum(dbc[t]);   // non-zero only on c == 0 lanes
        if (lane == 0) s_db[wave][t] = b;
#pragma unroll
        for (int j = 0; j < DQ; ++j) {
            f32x4 v = dwc[t][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = groups_sum(v[e]);
            if (grp == 0) st4(&s_red[wave][t][(c + LPR * j) * 4], v);
        }
    }
    __syncthreads();
    float *out = partials + (int64_t)pblk * bwd_partial_floats(D_, T_);
    for (int e = tid; e < T_ * D_ / 4; e += POOL_THREADS) {
        const int t = e / (D_ / 4), q = e % (D_ / 4);
        f32x4 v = ld4(&s_red[0][t][q * 4]);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += ld4(&s_red[w][t][q * 4]);
        st4(out + t * D_ + q * 4, v);
    }
    if (tid < T_) {
        float bsum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) bsum += s_db[w][tid];
        out[T_ * D_ + tid] = bsum;
    }
}

// out[e] = beta*out[e] + sum_b partials[b][e]; e < n (two destinations: dWc [T*D] then dbc [T]).
// block = 4 outputs x 64 partial slices, fixed summation order.
__global__ __launch_bounds__(256) void bwd_partial_reduce_kernel(const float *__restrict__ partials, int G, int64_t rec,
                                                                  int n_w, int n_b, float *dWc, float *dbc,
                                                                  float beta, int nred, const float *__restrict__ fine,
                                                                  int n_fine, float *__restrict__ dp_amax, int per_row) {
    __shared__ float red[64][4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nred) {
        // extra workgroups: abs-max slot of 256-row block rb = max over its 32 eight-row steps x NW waves of the fine table
        // (per_row: the batched launch's table holds one bound per row)
        const int PER = per_row ? H2_ROWBLK : (H2_ROWBLK / ROWS_PER_BLOCK_STEP) * NW;
        const int rb = blockIdx.x - nred;
        float v = 0.f;
        for (int e = tid; e < PER; e += 256) { const int i = rb * PER + e; if (i < n_fine) v = __builtin_fmaxf(v, fine[i]); }
        v = h2_wave_max(v);
        if ((tid & 63) == 0) red[tid >> 6][0] = v;
        __syncthreads();
        if (tid == 0) dp_amax[rb] = __builtin_fmaxf(__builtin_fmaxf(red[0][0], red[1][0]), __builtin_fmaxf(red[2][0], red[3][0]));
        return;
    }
    const int o = tid & 3, slice = tid >> 2;
    const int e = blockIdx.x * 4 + o;
    float v = 0.f;
    if (e < n_w + n_b)
        for (int b = slice; b < G; b += 64) v += partials[b * rec + e];
    red[slice][o] = v;
    __syncthreads();
    if (tid < 4 && e < n_w + n_b) {
        v = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) v += red[k][tid];
        float *dst = e < n_w ? dWc + e : dbc + (e - n_w);
        *dst = (beta != 0.f ? beta * *dst : 0.f) + v;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Workgroups of the two pool kernels (each persistent over row steps, 4 waves). Two per CU: with one, a CU's four waves issue a step's
// ten 16-byte loads per lane, wait, compute, store - the memory pipe idles during the compute; a second workgroup fills it
// (rocprofv3, 100k patches: forward 84 -> 75 us = 6.85 TB/s, backward 172 -> 140 us; three per CU is slower again, profiles/r02ba_*).
static int pool_grid(int64_t N, bool bwd = false) {
    const int64_t ntiles = (N + ROWS_PER_BLOCK_STEP - 1) / ROWS_PER_BLOCK_STEP;
    (void)bwd;
    const int64_t cap = 512;                 // (round 6: 256 / 384 / 1024 / 2048 workgroups for bags below 32k patches all measured slower, profiles/r06y_pool_grid_sweep.txt)
    return (int)(ntiles < cap ? ntiles : cap);
}

// Shapes: the tuned instantiations (no masking) for what TOAD builds - TOAD_fc_mtl_concat "big" / "small" (L 512, D 384 / 256, 2 tasks,
// models/model_toad.py:56,66) and Attn_Net_Gated's constructor defaults (L 1024, D 256, 1 task, :19) - and ONE covering instantiation
// (GEN: up to 4 tasks, D <= 512, L <= 1024) for every other (L, D, n_tasks) the constructor accepts inside those limits.
constexpr int GEN_T = 4, GEN_DQ = 4, GEN_LQ = 8;
static bool exact_shape(int L, int D, int T) { return (T == 1 || T == 2) && (D == 256 || D == 384) && (L == 512 || L == 1024); }
static bool shape_ok(int L, int D, int T) {
    return exact_shape(L, D, T) || (T >= 1 && T <= GEN_T && D >= 4 && D <= GEN_DQ * 4 * LPR && D % 4 == 0 && L >= 8 && L <= GEN_LQ * 4 * LPR && L % 8 == 0);
}

template <bool POOL>
static void launch_fwd(int L, int D, int T, int grid, hipStream_t st, const float *Pa, const float *Pb, int64_t ldp,
                       const float *H, const float *Wc, const float *bc, float *A_raw, float *partials, int N,
                       DropArgs da, DropArgs db, const int64_t *seg = nullptr, int nseg = 1) {
#define TOAD_FWD_CASE(TT, DD, LL)                                                                             \
    if (T == TT && D == DD && L == LL) {                                                                      \
        hipLaunchKernelGGL((gated_pool_fwd_kernel<TT, DD / (4 * LPR), LL / (4 * LPR), POOL>), dim3(grid, nseg), dim3(POOL_THREADS), 0, st, \
                           Pa, Pb, ldp, H, Wc, bc, A_raw, partials, N, da, db, D, L, T, seg);                 \
        return;                                                                                               \
    }
    TOAD_FWD_CASE(2, 384, 512)
    TOAD_FWD_CASE(2, 256, 512)
    TOAD_FWD_CASE(1, 384, 512)
    TOAD_FWD_CASE(1, 256, 512)
    TOAD_FWD_CASE(2, 384, 1024)
    TOAD_FWD_CASE(2, 256, 1024)
    TOAD_FWD_CASE(1, 384, 1024)
    TOAD_FWD_CASE(1, 256, 1024)
#undef TOAD_FWD_CASE
    hipLaunchKernelGGL((gated_pool_fwd_kernel<GEN_T, GEN_DQ, GEN_LQ, POOL, true>), dim3(grid, nseg), dim3(POOL_THREADS), 0, st, Pa, Pb, ldp, H, Wc, bc,
                       A_raw, partials, N, da, db, D, L, T, seg);
}

static void launch_bwd(int L, int D, int T, int grid, hipStream_t st, const float *Pa, const float *Pb, int64_t ldp,
                       const float *H, const float *Wc, const float *A_raw, const float *stats, const float *M,
                       const float *dM, const float *dA_ext, float *dPa, float *dPb, int64_t ldd, float *dH,
                       float *partials, float *dp_amax, int N, DropArgs da, DropArgs db, const int64_t *seg = nullptr, int nseg = 1,
                       int m_stride = 0, int s_stride = 0) {
#define TOAD_BWD_CASE(TT, DD, LL)                                                                             \
    if (T == TT && D == DD && L == LL) {                                                                      \
        hipLaunchKernelGGL((gated_pool_bwd_kernel<TT, DD / (4 * LPR), LL / (4 * LPR)>), dim3(grid, nseg), dim3(POOL_THREADS), 0, st, Pa, \
                           Pb, ldp, H, Wc, A_raw, stats, M, dM, dA_ext, dPa, dPb, ldd, dH, partials, dp_amax, N, da, db, D, L, T, seg, m_stride, s_stride); \
        return;                                                                                               \
    }
    TOAD_BWD_CASE(2, 384, 512)
    TOAD_BWD_CASE(2, 256, 512)
    TOAD_BWD_CASE(1, 384, 512)
    TOAD_BWD_CASE(1, 256, 512)
    TOAD_BWD_CASE(2, 384, 1024)
    TOAD_BWD_CASE(2, 256, 1024)
    TOAD_BWD_CASE(1, 384, 1024)
    TOAD_BWD_CASE(1, 256, 1024)
#undef TOAD_BWD_CASE
    hipLaunchKernelGGL((gated_pool_bwd_kernel<GEN_T, GEN_DQ, GEN_LQ, true>), dim3(grid, nseg), dim3(POOL_THREADS), 0, st, Pa, Pb, ldp, H, Wc, A_raw, stats,
                       M, dM, dA_ext, dPa, dPb, ldd, dH, partials, dp_amax, N, da, db, D, L, T, seg, m_stride, s_stride);
}

}  // namespace toad

using namespace toad;

extern "C" size_t toad_gated_pool_ws_bytes(int64_t N, int L, int D, int T) {
    if (N <= 0 || !shape_ok(L, D, T)) return 0;
    return (size_t)pool_grid(N) * (size_t)pool_partial_floats(L, T) * sizeof(float);
}

extern "C" int toad_gated_pool_fwd_f32(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc,
                                        const float *bc, float *A_raw, float *M, float *stats, void *ws,
                                        size_t ws_bytes, int64_t N, int L, int D, int T, float drop_p,
                                        uint64_t seed_a, uint64_t seed_b, void *stream) {
    const char *what = "toad_gated_pool_fwd_f32";
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    const DropArgs da = make_drop(drop_p, seed_a), db = make_drop(drop_p, seed_b);
    if (!Pa || !Pb || !Wc || !bc || !A_raw) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (N <= 0 || N > INT32_MAX - 64) { set_error("%s: bad N=%lld", what, (long long)N); return TOAD_EINVAL; }
    if (!shape_ok(L, D, T)) { set_error("%s: unsupported shape L=%d D=%d T=%d (1 <= T <= 4, D <= 512 multiple of 4, L <= 1024 multiple of 8)", what, L, D, T); return TOAD_ESHAPE; }
    if (ldp < D || ldp % 4 != 0) { set_error("%s: bad ldp=%lld", what, (long long)ldp); return TOAD_ESHAPE; }
    if (!aligned16(Pa) || !aligned16(Pb) || !aligned16(Wc) || (H && !aligned16(H)) || (T == 2 && exact_shape(L, D, T) && ((uintptr_t)A_raw & 7))) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    hipStream_t st = (hipStream_t)stream;
    const int grid = pool_grid(N);
    if (!H) {
        if (M || stats) { set_error("%s: M/stats requested without H", what); return TOAD_EINVAL; }
        launch_fwd<false>(L, D, T, grid, st, Pa, Pb, ldp, nullptr, Wc, bc, A_raw, nullptr, (int)N, da, db);
        return check_launch(what);
    }
    if (!M || !stats || !ws) { set_error("%s: null output/workspace", what); return TOAD_EINVAL; }
    if (!aligned16(ws)) { set_error("%s: workspace must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_gated_pool_ws_bytes(N, L, D, T)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    launch_fwd<true>(L, D, T, grid, st, Pa, Pb, ldp, H, Wc, bc, A_raw, (float *)ws, (int)N, da, db);
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(gated_pool_combine_kernel, dim3(T * ((L + 31) / 32)), dim3(256), 0, st, (const float *)ws, grid, L, T, M, stats, 0, 0);
    return check_launch(what);
}

// ---- batched (ragged multi-slide) launches: ONE launch pools every slide of a batch on its row range of the shared activations ----
// grid = (gx, B): gx workgroups per slide, sized for the longest slide and capped so that B * gx <= 4096 partial records.
static int batch_gx(int64_t max_n, int B) {
    int64_t gx = (max_n + ROWS_PER_BLOCK_STEP - 1) / ROWS_PER_BLOCK_STEP;
    const int64_t cap = 4096 / B > 0 ? 4096 / B : 1;
    if (gx > cap) gx = cap;
    if (gx > 512) gx = 512;
    return (int)(gx < 1 ? 1 : gx);
}
size_t toad::pool_batch_ws_bytes(int B, int L, int D, int T) {       // forward partials, then backward partials (either fits 4096 records)
    const size_t f = (size_t)4096 * (size_t)pool_partial_floats(L, T) * sizeof(float), b = (size_t)4096 * (size_t)bwd_partial_floats(D, T) * sizeof(float);
    (void)B;
    return (f > b ? f : b) + 256;
}
int toad::launch_pool_fwd_batch(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *bc, float *A_raw, float *M,
                                int m_stride, float *stats, int s_stride, void *ws, const int64_t *seg_dev, int B, int64_t max_n, int L, int D, int T,
                                float drop_p, uint64_t seed_a, uint64_t seed_b, hipStream_t st) {
    const char *what = "toad_gated_pool_fwd_f32 (batched)";
    if (!shape_ok(L, D, T) || B < 1 || B > 4096) { set_error("%s: unsupported shape", what); return TOAD_ESHAPE; }
    const DropArgs da = make_drop(drop_p, seed_a), db = make_drop(drop_p, seed_b);
    const int gx = batch_gx(max_n, B);
    launch_fwd<true>(L, D, T, gx, st, Pa, Pb, ldp, H, Wc, bc, A_raw, (float *)ws, 0, da, db, seg_dev, B);
    if (int rc = check_launch(what)) return rc;
    hipLaunchKernelGGL(gated_pool_combine_kernel, dim3(T * ((L + 31) / 32), B), dim3(256), 0, st, (const float *)ws, gx, L, T, M, stats, m_stride, s_stride);
    return check_launch(what);
}
int toad::launch_pool_bwd_batch(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *A_raw, const float *stats,
                                int s_stride, const float *M, const float *dM, int m_stride, float *dPa, float *dPb, int64_t ldd, float *dH, float *dWc,
                                float *dbc, float beta, void *ws, const int64_t *seg_dev, int B, int64_t max_n, int L, int D, int T, float drop_p,
                                uint64_t seed_a, uint64_t seed_b, hipStream_t st, float *dp_amax, float *row_bound, int64_t n_rows) {
    const char *what = "toad_gated_pool_bwd_f32 (batched)";
    if (!shape_ok(L, D, T) || B < 1 || B > 4096) { set_error("%s: unsupported shape", what); return TOAD_ESHAPE; }
    const DropArgs da = make_drop(drop_p, seed_a), db = make_drop(drop_p, seed_b);
    const int gx = batch_gx(max_n, B);
    // dp_amax (optional) + row_bound (n_rows floats of scratch): the abs-max ARRAY of dP over the concatenation, from a per-row bound table
    launch_bwd(L, D, T, gx, st, Pa, Pb, ldp, H, Wc, A_raw, stats, M, dM, nullptr, dPa, dPb, ldd, dH, (float *)ws, dp_amax ? row_bound : nullptr, 0, da, db, seg_dev, B,
               m_stride, s_stride);
    if (int rc = check_launch(what)) return rc;
    const int n = T * D + T, nred = (n + 3) / 4;
    const int nblk = dp_amax ? (int)((n_rows + H2_ROWBLK - 1) / H2_ROWBLK) : 0;
    hipLaunchKernelGGL(bwd_partial_reduce_kernel, dim3(nred + nblk), dim3(256), 0, st, (const float *)ws, gx * B, bwd_partial_floats(D, T), T * D, T, dWc, dbc, beta, nred,
                       (const float *)row_bound, (int)n_rows, dp_amax, 1);
    return check_launch(what);
}

static size_t bwd_partials_bytes(int64_t N, int D, int T) { return ((size_t)pool_grid(N, true) * (size_t)bwd_partial_floats(D, T) * sizeof(float) + 255) & ~(size_t)255; }
static int64_t bwd_fine_floats(int64_t N) { return ((N + ROWS_PER_BLOCK_STEP - 1) / ROWS_PER_BLOCK_STEP) * NW; }
extern "C" size_t toad_gated_pool_bwd_ws_bytes(int64_t N, int L, int D, int T) {
    if (N <= 0 || !shape_ok(L, D, T)) return 0;
    // per-workgroup dWc / dbc partials, then the fine abs-max table (one float per wave per 8-row step)
    return bwd_partials_bytes(N, D, T) + (size_t)bwd_fine_floats(N) * sizeof(float);
}

int toad::launch_pool_bwd(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc, const float *A_raw,
                          const float *stats, const float *M, const float *dM, const float *dA_ext, float *dPa, float *dPb, int64_t ldd,
                          float *dH, float *dWc, float *dbc, float beta, float *dp_amax, bool zero_amax, void *ws, size_t ws_bytes, int64_t N,
                          int L, int D, int T, float drop_p, uint64_t seed_a, uint64_t seed_b, hipStream_t st) {
    const char *what = "toad_gated_pool_bwd_f32";
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    const DropArgs da = make_drop(drop_p, seed_a), db = make_drop(drop_p, seed_b);
    if (!Pa || !Pb || !H || !Wc || !A_raw || !stats || !M || !dM || !dPa || !dPb || !dWc || !dbc || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (N <= 0 || N > INT32_MAX - 64) { set_error("%s: bad N", what); return TOAD_EINVAL; }
    if (!shape_ok(L, D, T)) { set_error("%s: unsupported shape L=%d D=%d T=%d", what, L, D, T); return TOAD_ESHAPE; }
    if (ldp < D || ldp % 4 != 0 || ldd < D || ldd % 4 != 0) { set_error("%s: bad row stride", what); return TOAD_ESHAPE; }
    if (!aligned16(Pa) || !aligned16(Pb) || !aligned16(H) || !aligned16(dPa) || !aligned16(dPb) || (dH && !aligned16(dH)) || !aligned16(ws)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_gated_pool_bwd_ws_bytes(N, L, D, T)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    const int grid = pool_grid(N, true);
    static_assert(H2_ROWBLK % ROWS_PER_BLOCK_STEP == 0, "a block step must not straddle two abs-max blocks");
    (void)zero_amax;                                   // every slot is overwritten (no atomics): nothing to zero
    float *fine = dp_amax ? reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + bwd_partials_bytes(N, D, T)) : nullptr;
    launch_bwd(L, D, T, grid, st, Pa, Pb, ldp, H, Wc, A_raw, stats, M, dM, dA_ext, dPa, dPb, ldd, dH, (float *)ws, fine, (int)N, da, db);
    int rc = check_launch(what);
    if (rc) return rc;
    const int n = T * D + T, nred = (n + 3) / 4;
    const int nblk = dp_amax ? (int)((N + H2_ROWBLK - 1) / H2_ROWBLK) : 0;
    hipLaunchKernelGGL(bwd_partial_reduce_kernel, dim3(nred + nblk), dim3(256), 0, st, (const float *)ws, grid,
                       bwd_partial_floats(D, T), T * D, T, dWc, dbc, beta, nred, (const float *)fine, (int)bwd_fine_floats(N), dp_amax, 0);
    return check_launch(what);
}

extern "C" int toad_gated_pool_bwd_f32(const float *Pa, const float *Pb, int64_t ldp, const float *H, const float *Wc,
                                        const float *A_raw, const float *stats, const float *M, const float *dM,
                                        const float *dA_ext, float *dPa, float *dPb, int64_t ldd, float *dH,
                                        float *dWc, float *dbc, float beta, float *dp_amax, void *ws, size_t ws_bytes,
                                        int64_t N, int L, int D, int T, float drop_p, uint64_t seed_a, uint64_t seed_b,
                                        void *stream) {
    return launch_pool_bwd(Pa, Pb, ldp, H, Wc, A_raw, stats, M, dM, dA_ext, dPa, dPb, ldd, dH, dWc, dbc, beta, dp_amax, true, ws, ws_bytes,
                           N, L, D, T, drop_p, seed_a, seed_b, (hipStream_t)stream);
}

// heads.hip — the two classifier heads of TOAD_fc_mtl_concat and the caller's weighted CE.
//
// models/model_toad.py:99-107 (concat sex, Linear(513,C), Linear(513,2), topk, softmax) and
// utils/core_utils_mtl_concat.py:213-215 (0.75*CE + 0.25*CE).  O(10 kFLOP) per slide: these
// kernels exist to keep the per-slide tail at three launches, not for throughput.
#include "common.h"

#include <math.h>

namespace toad {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// softmax + first-argmax of n logits by one wave
__device__ void wave_softmax_argmax(const float *lg, int n, float *prob, int64_t *hat, int lane) {
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, lg[i]);
    mx = wave_max(mx);
    int best = INT32_MAX;
    for (int i = lane; i < n; i += 64)
        if (lg[i] == mx && i < best) best = i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(lg[i] - mx);
    s = wave_sum(s);
    for (int i = lane; i < n; i += 64) prob[i] = expf(lg[i] - mx) / s;
    if (lane == 0) *hat = best == INT32_MAX ? 0 : best;
}

__global__ __launch_bounds__(1024) void heads_fwd_kernel(const float *__restrict__ M, const float *__restrict__ sex,
                                                         const float *__restrict__ Wcls, const float *__restrict__ bcls,
                                                         const float *__restrict__ Wsite, const float *__restrict__ bsite,
                                                         float *Mcat, float *logits, float *Y_prob, int64_t *Y_hat,
                                                         float *site_logits, float *site_prob, int64_t *site_hat, int L,
                                                         int C) {
    extern __shared__ float s_m[];   // [2][L+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = L + 1;
    const float sx = sex[0];
    for (int e = tid; e < 2 * LP; e += 1024) {
        const int t = e / LP, k = e % LP;
        const float v = k < L ? M[t * L + k] : sx;
        s_m[e] = v;
        Mcat[e] = v;
    }
    __syncthreads();
    // rows 0..C-1: classifier on Mcat[0]; rows C, C+1: site classifier on Mcat[1]
    for (int r = wave; r < C + 2; r += 16) {
        const float *w = r < C ? Wcls + (int64_t)r * LP : Wsite + (int64_t)(r - C) * LP;
        const float *x = r < C ? s_m : s_m + LP;
        float p = 0.f;
        for (int k = lane; k < LP; k += 64) p = fmaf(w[k], x[k], p);
        p = wave_sum(p);
        if (lane == 0) {
            if (r < C) logits[r] = p + bcls[r];
            else site_logits[r - C] = p + bsite[r - C];
        }
    }
    __syncthreads();
    if (wave == 0) wave_softmax_argmax(logits, C, Y_prob, Y_hat, lane);
    if (wave == 1) wave_softmax_argmax(site_logits, 2, site_prob, site_hat, lane);
}

// grid.y = C + 2 weight rows (classifier rows then the two site rows) + 1 extra row for dM;
// grid.x covers the L+1 columns. Every element is independent.
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float *__restrict__ Mcat, const float *__restrict__ dlogits,
                                                         const float *__restrict__ dsite, const float *__restrict__ Wcls,
                                                         const float *__restrict__ Wsite, const float *__restrict__ dMcat_ext,
                                                         float *dWcls, float *dbcls, float *dWsite, float *dbsite,
                                                         float *dM, float *dsex, float beta, int L, int C) {
    const int LP = L + 1;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (k >= LP) return;
    if (r < C) {
        const int64_t o = (int64_t)r * LP + k;
        dWcls[o] = (beta != 0.f ? beta * dWcls[o] : 0.f) + dlogits[r] * Mcat[k];
        if (k == 0) dbcls[r] = (beta != 0.f ? beta * dbcls[r] : 0.f) + dlogits[r];
    } else if (r < C + 2) {
        const int c = r - C, o = c * LP + k;
        dWsite[o] = (beta != 0.f ? beta * dWsite[o] : 0.f) + dsite[c] * Mcat[LP + k];
        if (k == 0) dbsite[c] = (beta != 0.f ? beta * dbsite[c] : 0.f) + dsite[c];
    } else if (k < L) {
        float d0 = 0.f;
        for (int c = 0; c < C; ++c) d0 = fmaf(dlogits[c], Wcls[(int64_t)c * LP + k], d0);
        const float d1 = fmaf(dsite[1], Wsite[LP + k], dsite[0] * Wsite[k]);
        dM[k] = d0 + (dMcat_ext ? dMcat_ext[k] : 0.f);
        dM[L + k] = d1 + (dMcat_ext ? dMcat_ext[LP + k] : 0.f);
    } else if (dsex) {
        // k == L: the `sex` column that models/model_toad.py:99 concatenates to BOTH pooled rows -> its gradient sums both heads
        float d = 0.f;
        for (int c = 0; c < C; ++c) d = fmaf(dlogits[c], Wcls[(int64_t)c * LP + L], d);
        d = fmaf(dsite[1], Wsite[LP + L], fmaf(dsite[0], Wsite[L], d));
        if (dMcat_ext) d += dMcat_ext[L] + dMcat_ext[LP + L];
        dsex[0] = d;
    }
}

// one wave: loss and d loss / d logits for w_cls*CE(logits,label) + w_site*CE(site_logits,site)
__device__ float wave_ce(const float *lg, int n, int64_t y, float wgt, float *dlg, int lane) {
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, lg[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(lg[i] - mx);
    s = wave_sum(s);
    // a label outside [0, n) makes torch's CrossEntropyLoss raise; a kernel cannot, so it poisons the loss and the gradient
    // (NaN propagates to every parameter gradient and is impossible to miss) instead of reading out of bounds
    const bool bad = y < 0 || y >= n;
    for (int i = lane; i < n; i += 64) dlg[i] = bad ? __builtin_nanf("") : wgt * (expf(lg[i] - mx) / s - (i == (int)y ? 1.f : 0.f));
    return bad ? __builtin_nanf("") : (logf(s) + mx) - lg[y];
}
__global__ __launch_bounds__(64) void mtl_ce_kernel(const float *logits, const float *site_logits, const int64_t *label,
                                                     const int64_t *site, float w_cls, float w_site, float *loss_out,
                                                     float *dlogits, float *dsite, int C) {
    const int lane = threadIdx.x;
    const float lc = wave_ce(logits, C, label[0], w_cls, dlogits, lane);
    const float ls = wave_ce(site_logits, 2, site[0], w_site, dsite, lane);
    if (lane == 0) {
        loss_out[0] = w_cls * lc + w_site * ls;
        loss_out[1] = lc;
        loss_out[2] = ls;
    }
}

// ------------------------------------------------------------------------------------------
// heads forward + weighted CE + heads backward in ONE single-workgroup launch (the per-slide tail of the fused step:
// three dependent launches of O(10 kFLOP) cost ~25 us of pure latency on small bags). Same arithmetic, in the same order,
// as heads_fwd_kernel -> mtl_ce_kernel -> heads_bwd_kernel, so the results are bitwise those of the three-kernel chain.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void heads_ce_fused_kernel(
    const float *__restrict__ M, const float *__restrict__ sex, const float *__restrict__ Wcls, const float *__restrict__ bcls,
    const float *__restrict__ Wsite, const float *__restrict__ bsite, const int64_t *__restrict__ label,
    const int64_t *__restrict__ site, float w_cls, float w_site, float *Mcat, float *logits, float *Y_prob, int64_t *Y_hat,
    float *site_logits, float *site_prob, int64_t *site_hat, float *loss_out, float *dlogits, float *dsite, float *dWcls,
    float *dbcls, float *dWsite, float *dbsite, float *dM, float beta, int L, int C, int cache_w, int64_t rec) {
    // rec != 0: batched launch (the ragged multi-slide step): blockIdx.x = slide b; every per-slide pointer moves by b * rec BYTES (sex /
    // label / site by b elements, loss_out by 3 b), dlogits / dsite are exported per slide, and the head-weight gradients - a sum over
    // the batch - are formed afterwards by heads_wgrad_batch_kernel (dWcls == NULL here)
    if (rec) {
        const int64_t b = blockIdx.x, o = b * rec;
#define TOAD_MV(p) p = reinterpret_cast<decltype(p)>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)o)
        TOAD_MV(M); TOAD_MV(Mcat); TOAD_MV(logits); TOAD_MV(Y_prob); TOAD_MV(Y_hat); TOAD_MV(site_logits); TOAD_MV(site_prob); TOAD_MV(site_hat);
        TOAD_MV(dM); TOAD_MV(dlogits); TOAD_MV(dsite);
#undef TOAD_MV
        sex += b; label += b; site += b; loss_out += 3 * b;
    }
    extern __shared__ __attribute__((aligned(16))) float s_all[];   // [2][L+1] Mcat | [C] logits | [2] site logits | [C] dlogits | [2] dsite | [C+2] biases | pad to 16 B | cache_w: [C+2][L+1] weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = L + 1;
    float *s_m = s_all, *s_lg = s_all + 2 * LP, *s_sl = s_lg + C, *s_dl = s_sl + 2, *s_ds = s_dl + C, *s_b = s_ds + 2;
    // cache_w: the head weights are parked in LDS, so the backward's dM = dlogits . Wcls + dsite . Wsite reads LDS instead of walking C
    // dependent global loads per thread (that loop was ~10 of this kernel's ~21 us, and the kernel runs once per slide: 64 times in a
    // 64-slide batch of small bags)
    float *s_w = s_all + ((2 * LP + 3 * C + 6 + 3) & ~3);      // 16-byte aligned (vector stores of the weight copy)
    // Round 6: ONE global round trip in front of the first barrier. The kernel is a latency chain (one workgroup, O(10 kFLOP)): the row
    // dot products used to walk their weight row with a load -> fma dependency per 64-column step (9 round trips per row, two rows for
    // the first waves: ~12 of 15.5 us at 18 classes), then fetched the bias, then - in the loss phase - the labels. Now every thread issues
    // all of its loads back to back - the pooled features, the head weights as ONE flat copy into LDS (eight independent loads in flight per
    // thread and batch), the biases, the labels - and the dot products read LDS in the same k order (same sums, bit for bit).
    const float sx = sex[0];
    const int64_t lab = label[0], sit = site[0];
    const int nW = C * LP, nAll = (C + 2) * LP;
    const bool al16 = ((reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(Wcls)) & 15) == 0;
    if (cache_w && al16 && L % 4 == 0 && 2 * L <= 4096 && C + 2 <= 1024 && nW <= 16 * 1024 && 2 * LP <= 2048) {
        // every load of the thread is unconditional (indices clamped to valid elements) and issued before the first use; the two big pieces -
        // the pooled features and the classifier rows - move as 16-byte vectors (few instructions: see the note on the backward below)
        const int nW4 = nW >> 2, m4 = (2 * L) >> 2;
        const f32x4 mv = ld4(M + 4 * min(tid, m4 - 1));
        f32x4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = ld4(Wcls + 4 * min(tid + 1024 * u, nW4 - 1));
        const float wt = Wcls[min(4 * nW4 + tid, nW - 1)];                 // the (nW mod 4) elements behind the last whole vector
        float sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) sv[u] = Wsite[min(tid + 1024 * u, 2 * LP - 1)];
        const int bi = min(tid, C + 1);
        const float bval = *(bi < C ? bcls + bi : bsite + (bi - C));
        if (tid < m4) {
            const int e = 4 * tid, t = e >= L ? 1 : 0, o = e + t;          // Mcat[t][k] sits at t * (L + 1) + k
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_m[o + q] = mv[q]; Mcat[o + q] = mv[q]; }
        }
        if (tid < 2) { s_m[tid * LP + L] = sx; Mcat[tid * LP + L] = sx; }
        if (tid < C + 2) s_b[tid] = bval;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 1024 * u;
            if (i < nW4) st4(s_w + 4 * i, wv[u]);
        }
        if (4 * nW4 + tid < nW) s_w[4 * nW4 + tid] = wt;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 1024 * u;
            if (i < 2 * LP) s_w[nW + i] = sv[u];
        }
    } else {
        for (int e = tid; e < 2 * LP; e += 1024) {
            const int t = e / LP, k = e % LP;
            const float v = k < L ? M[t * L + k] : sx;
            s_m[e] = v;
            Mcat[e] = v;
        }
        for (int i = tid; i < C + 2; i += 1024) s_b[i] = i < C ? bcls[i] : bsite[i - C];
        if (cache_w) {
            for (int base = 0; base < nAll; base += 8 * 1024) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = base + u * 1024 + tid;
                    v[u] = e < nW ? Wcls[e] : (e < nAll ? Wsite[e - nW] : 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = base + u * 1024 + tid;
                    if (e < nAll) s_w[e] = v[u];
                }
            }
        }
    }
    __syncthreads();
    for (int r = wave; r < C + 2; r += 16) {
        const float *x = r < C ? s_m : s_m + LP;
        float p = 0.f;
        if (cache_w) {
            const float *w = s_w + r * LP;
            for (int k = lane; k < LP; k += 64) p = fmaf(w[k], x[k], p);
        } else {
            const float *w = r < C ? Wcls + (int64_t)r * LP : Wsite + (int64_t)(r - C) * LP;
            for (int k = lane; k < LP; k += 64) p = fmaf(w[k], x[k], p);
        }
        p = wave_sum(p);
        if (lane == 0) {
            const float v = p + s_b[r];
            if (r < C) { logits[r] = v; s_lg[r] = v; }
            else { site_logits[r - C] = v; s_sl[r - C] = v; }
        }
    }
    __syncthreads();
    if (wave == 0) wave_softmax_argmax(s_lg, C, Y_prob, Y_hat, lane);
    if (wave == 1) wave_softmax_argmax(s_sl, 2, site_prob, site_hat, lane);
    if (wave == 2) {
        const float lc = wave_ce(s_lg, C, lab, w_cls, s_dl, lane);
        const float ls = wave_ce(s_sl, 2, sit, w_site, s_ds, lane);
        if (lane == 0) { loss_out[0] = w_cls * lc + w_site * ls; loss_out[1] = lc; loss_out[2] = ls; }
    }
    __syncthreads();
    if (dlogits) for (int i = tid; i < C; i += 1024) dlogits[i] = s_dl[i];
    if (dsite && tid < 2) dsite[tid] = s_ds[tid];
    // backward: (C + 2) weight rows of L+1 columns (one wave per row, lanes along the row: no index arithmetic per element - the whole kernel
    // runs on ONE CU, 1,024 threads share 64 lanes per clock, so every instruction per thread is 16 cycles of the launch), then the two dM rows
    if (dWcls) {                                        // (batched: the weight gradients are summed over the slides by heads_wgrad_batch_kernel)
        for (int r = wave; r < C + 2; r += 16) {
            const bool cls = r < C;
            const float g = cls ? s_dl[r] : s_ds[r - C];
            const float *x = cls ? s_m : s_m + LP;
            float *dst = cls ? dWcls + (int64_t)r * LP : dWsite + (int64_t)(r - C) * LP;
            for (int k = lane; k < LP; k += 64) dst[k] = (beta != 0.f ? beta * dst[k] : 0.f) + g * x[k];
            if (lane == 0) {
                float *db = cls ? dbcls + r : dbsite + (r - C);
                *db = (beta != 0.f ? beta * *db : 0.f) + g;
            }
        }
    }
    for (int e = tid; e < 2 * L; e += 1024) {           // dM[0, k] = dlogits . Wcls[:, k], dM[1, k] = dsite . Wsite[:, k]
        const int t = e >= L ? 1 : 0, k = e - t * L;
        float d;
        if (t == 0) {
            d = 0.f;
            if (cache_w) { for (int c = 0; c < C; ++c) d = fmaf(s_dl[c], s_w[c * LP + k], d); }
            else { for (int c = 0; c < C; ++c) d = fmaf(s_dl[c], Wcls[(int64_t)c * LP + k], d); }
        } else {
            d = cache_w ? fmaf(s_ds[1], s_w[(C + 1) * LP + k], s_ds[0] * s_w[C * LP + k]) : fmaf(s_ds[1], Wsite[LP + k], s_ds[0] * Wsite[k]);
        }
        dM[e] = d;
    }
}

// Head-weight gradients of a batch: dWcls[c, :] = beta dWcls[c, :] + sum_b dlogits_b[c] Mcat_b[0, :], dWsite[s, :] likewise with Mcat_b[1, :],
// biases = sums of dlogits / dsite. One thread per output element, slides summed in index order (deterministic); the inputs are a few
// hundred KB and stay in L2.
__global__ __launch_bounds__(256) void heads_wgrad_batch_kernel(const float *__restrict__ dl, const float *__restrict__ ds, const float *__restrict__ Mcat,
                                                                 int64_t rec, float *dWcls, float *dbcls, float *dWsite, float *dbsite, float beta, int B,
                                                                 int L, int C) {
    const int LP = L + 1;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (C + 2) * LP) return;
    const int r = e / LP, k = e % LP;
    const bool cls = r < C;
    const int c = cls ? r : r - C;
    const char *g = reinterpret_cast<const char *>(cls ? dl : ds), *m = reinterpret_cast<const char *>(Mcat);
    float acc = 0.f, bacc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float gv = reinterpret_cast<const float *>(g + (int64_t)b * rec)[c];
        acc = fmaf(gv, reinterpret_cast<const float *>(m + (int64_t)b * rec)[(cls ? 0 : LP) + k], acc);
        bacc += gv;
    }
    float *dw = cls ? dWcls + (int64_t)c * LP + k : dWsite + (int64_t)c * LP + k;
    *dw = (beta != 0.f ? beta * *dw : 0.f) + acc;
    if (k == 0) { float *db = cls ? dbcls + c : dbsite + c; *db = (beta != 0.f ? beta * *db : 0.f) + bacc; }
}

// Adam over one flat buffer, same update as torch.optim.Adam (L2 weight decay folded into the gradient;
// utils/utils.py:63-70 get_optim: Adam(lr, weight_decay=reg)). One launch for all 1.19 M parameters.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
    const int64_t n4 = n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        f32x4 pp = ld4(p + 4 * e), gg = ld4(g + 4 * e), mm = ld4(m + 4 * e), vv = ld4(v + 4 * e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] + wd * pp[k];
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] -= (lr / bc1) * (mm[k] / denom);
        }
        st4(p + 4 * e, pp); st4(m + 4 * e, mm); st4(v + 4 * e, vv);
    }
}

}  // namespace toad

using namespace toad;

extern "C" int toad_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int64_t step, void *stream) {
    const char *what = "toad_adam_step_f32";
    if (!p || !g || !m || !v || n <= 0 || step < 1) { set_error("%s: bad argument", what); return TOAD_EINVAL; }
    if (n % 4 != 0 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) { set_error("%s: n must be a multiple of 4 and pointers 16-byte aligned", what); return TOAD_EALIGN; }
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    int grid = (int)((n / 4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
    return check_launch(what);
}

extern "C" int toad_heads_fwd_f32(const float *M, const float *sex, const float *Wcls, const float *bcls,
                                   const float *Wsite, const float *bsite, float *Mcat, float *logits, float *Y_prob,
                                   int64_t *Y_hat, float *site_logits, float *site_prob, int64_t *site_hat, int L, int C,
                                   void *stream) {
    const char *what = "toad_heads_fwd_f32";
    if (!M || !sex || !Wcls || !bcls || !Wsite || !bsite || !Mcat || !logits || !Y_prob || !Y_hat || !site_logits || !site_prob || !site_hat) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (L <= 0 || L > 8192 || C <= 0 || C > 1024) { set_error("%s: unsupported L=%d C=%d", what, L, C); return TOAD_ESHAPE; }
    hipLaunchKernelGGL(heads_fwd_kernel, dim3(1), dim3(1024), 2 * (L + 1) * sizeof(float), (hipStream_t)stream, M, sex, Wcls,
                       bcls, Wsite, bsite, Mcat, logits, Y_prob, Y_hat, site_logits, site_prob, site_hat, L, C);
    return check_launch(what);
}

extern "C" int toad_heads_bwd_f32(const float *Mcat, const float *dlogits, const float *dsite, const float *Wcls,
                                   const float *Wsite, const float *dMcat_ext, float *dWcls, float *dbcls, float *dWsite,
                                   float *dbsite, float *dM, float *dsex, float beta, int L, int C, void *stream) {
    const char *what = "toad_heads_bwd_f32";
    if (!Mcat || !dlogits || !dsite || !Wcls || !Wsite || !dWcls || !dbcls || !dWsite || !dbsite || !dM) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (L <= 0 || L > 8192 || C <= 0 || C > 1024 || C > L) { set_error("%s: unsupported L=%d C=%d", what, L, C); return TOAD_ESHAPE; }
    hipLaunchKernelGGL(heads_bwd_kernel, dim3((L + 1 + 255) / 256, C + 3), dim3(256), 0, (hipStream_t)stream, Mcat, dlogits, dsite,
                       Wcls, Wsite, dMcat_ext, dWcls, dbcls, dWsite, dbsite, dM, dsex, beta, L, C);
    return check_launch(what);
}

extern "C" int toad_mtl_ce_fwd_bwd_f32(const float *logits, const float *site_logits, const int64_t *label,
                                        const int64_t *site, float w_cls, float w_site, float *loss_out, float *dlogits,
                                        float *dsite, int C, void *stream) {
    const char *what = "toad_mtl_ce_fwd_bwd_f32";
    if (!logits || !site_logits || !label || !site || !loss_out || !dlogits || !dsite) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (C <= 0 || C > 1024) { set_error("%s: unsupported C=%d", what, C); return TOAD_ESHAPE; }
    hipLaunchKernelGGL(mtl_ce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logits, site_logits, label, site, w_cls,
                       w_site, loss_out, dlogits, dsite, C);
    return check_launch(what);
}

extern "C" int toad_heads_ce_fused_f32(const float *M, const float *sex, const float *Wcls, const float *bcls, const float *Wsite,
                                        const float *bsite, const int64_t *label, const int64_t *site, float w_cls, float w_site,
                                        float *Mcat, float *logits, float *Y_prob, int64_t *Y_hat, float *site_logits,
                                        float *site_prob, int64_t *site_hat, float *loss_out, float *dlogits, float *dsite,
                                        float *dWcls, float *dbcls, float *dWsite, float *dbsite, float *dM, float beta, int L,
                                        int C, void *stream) {
    const char *what = "toad_heads_ce_fused_f32";
    if (!M || !sex || !Wcls || !bcls || !Wsite || !bsite || !label || !site || !Mcat || !logits || !Y_prob || !Y_hat || !site_logits ||
        !site_prob || !site_hat || !loss_out || !dWcls || !dbcls || !dWsite || !dbsite || !dM) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (L <= 0 || L > 8192 || C <= 0 || C > 1024 || C > L) { set_error("%s: unsupported L=%d C=%d", what, L, C); return TOAD_ESHAPE; }
    size_t smem = (size_t)((2 * (L + 1) + 3 * C + 6 + 3) & ~3) * sizeof(float);
    const size_t wbytes = (size_t)(C + 2) * (L + 1) * sizeof(float);
    const int cache_w = smem + wbytes <= 60 * 1024 ? 1 : 0;          // 18 classes x 513: 41 KB (below the 64 KB a launch gets without an attribute)
    if (cache_w) smem += wbytes;
    hipLaunchKernelGGL(heads_ce_fused_kernel, dim3(1), dim3(1024), smem, (hipStream_t)stream, M, sex, Wcls, bcls, Wsite, bsite, label,
                       site, w_cls, w_site, Mcat, logits, Y_prob, Y_hat, site_logits, site_prob, site_hat, loss_out, dlogits, dsite,
                       dWcls, dbcls, dWsite, dbsite, dM, beta, L, C, cache_w, (int64_t)0);
    return check_launch(what);
}

int toad::launch_heads_batch(const HeadsBatch &hb, const float *sex, const float *Wcls, const float *bcls, const float *Wsite, const float *bsite,
                             const int64_t *label, const int64_t *site, float w_cls, float w_site, float *loss_out, float *dWcls, float *dbcls,
                             float *dWsite, float *dbsite, float beta, int B, int L, int C, hipStream_t st) {
    const char *what = "toad_heads_ce_fused_f32 (batched)";
    if (L <= 0 || L > 8192 || C <= 0 || C > 1024 || C > L || B < 1) { set_error("%s: unsupported L=%d C=%d B=%d", what, L, C, B); return TOAD_ESHAPE; }
    if ((size_t)2 * (L + 1) * sizeof(float) > hb.rec || (size_t)C * sizeof(float) > hb.rec) { set_error("%s: per-slide record too small", what); return TOAD_EWORKSPACE; }
    size_t smem = (size_t)((2 * (L + 1) + 3 * C + 6 + 3) & ~3) * sizeof(float);
    const size_t wbytes = (size_t)(C + 2) * (L + 1) * sizeof(float);
    const int cache_w = smem + wbytes <= 60 * 1024 ? 1 : 0;
    if (cache_w) smem += wbytes;
    hipLaunchKernelGGL(heads_ce_fused_kernel, dim3(B), dim3(1024), smem, st, hb.M, sex, Wcls, bcls, Wsite, bsite, label, site, w_cls, w_site, hb.Mcat,
                       hb.logits, hb.yprob, hb.yhat, hb.slog, hb.sprob, hb.shat, loss_out, hb.dl, hb.ds, (float *)nullptr, (float *)nullptr,
                       (float *)nullptr, (float *)nullptr, hb.dM, beta, L, C, cache_w, (int64_t)hb.rec);
    if (int rc = check_launch(what)) return rc;
    hipLaunchKernelGGL(heads_wgrad_batch_kernel, dim3(((C + 2) * (L + 1) + 255) / 256), dim3(256), 0, st, (const float *)hb.dl, (const float *)hb.ds,
                       (const float *)hb.Mcat, (int64_t)hb.rec, dWcls, dbcls, dWsite, dbsite, beta, B, L, C);
    return check_launch(what);
}

// Flat SGD (get_optim's SGD branch, utils/utils.py:66-67: optim.SGD(lr, momentum=0.9, weight_decay=reg)) over one buffer:
//   g' = g + wd*p;  buf = momentum*buf + g' (buf = g' on the first step);  p -= lr*buf      - torch.optim.SGD's update.
__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ mom,
                                                   int64_t n, float lr, float momentum, float wd, int first) {
    const int64_t n4 = n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        f32x4 pp = ld4(p + 4 * e), gg = ld4(g + 4 * e);
        f32x4 bb = mom ? ld4(mom + 4 * e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] + wd * pp[k];
            bb[k] = (mom && !first) ? momentum * bb[k] + gk : gk;
            pp[k] -= lr * bb[k];
        }
        st4(p + 4 * e, pp);
        if (mom) st4(mom + 4 * e, bb);
    }
}
extern "C" int toad_sgd_step_f32(float *p, const float *g, float *momentum_buf, int64_t n, float lr, float momentum,
                                  float weight_decay, int64_t step, void *stream) {
    const char *what = "toad_sgd_step_f32";
    if (!p || !g || n <= 0 || step < 1 || (momentum != 0.f && !momentum_buf)) { set_error("%s: bad argument", what); return TOAD_EINVAL; }
    if (n % 4 != 0 || !aligned16(p) || !aligned16(g) || (momentum_buf && !aligned16(momentum_buf))) { set_error("%s: n must be a multiple of 4 and pointers 16-byte aligned", what); return TOAD_EALIGN; }
    int grid = (int)((n / 4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, momentum != 0.f ? momentum_buf : nullptr, n, lr,
                       momentum, weight_decay, step == 1 ? 1 : 0);
    return check_launch(what);
}

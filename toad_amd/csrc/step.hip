// step.hip — one C-ABI call for the whole per-slide training step of TOAD's MIL path:
// forward (trunk, stacked attention GEMM, fused pool, heads), the caller's weighted CE
// (utils/core_utils_mtl_concat.py:213-215) and the full backward, sequenced in C++ over a
// caller-owned arena. Same kernels, same order and same results as the per-op entry points
// (toad_amd/functional.py); what disappears is ~30 host round trips and ~40 allocations per
// slide, which dominate for the small bags of real cohorts (a 256-patch step is launch-bound).
#include "common.h"

namespace toad {

// Sub-buffers start on 2 MiB boundaries (like separate large device allocations do): with 256-B packing the
// pool kernels, which stream P, H, dP and dH concurrently, ran 9-17 % slower (HBM channel aliasing between the
// streams); measured with rocprofv3 on the same kernels, profiles/.
static inline size_t align256(size_t x) { return (x + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1); }

struct Arena {
    char *base; size_t off, cap;
    template <typename T> T *take(size_t n) {
        T *p = reinterpret_cast<T *>(base + off);
        off = align256(off + n * sizeof(T));
        return p;
    }
};

struct StepDims { int L0, L, D, T; };
static const StepDims kDims = {1024, 512, 384, 2};     // size_arg = "big" (models/model_toad.py:56)

}  // namespace toad

using namespace toad;

extern "C" size_t toad_mil_step_ws_bytes(int64_t N, int C, int D) {
    if (N <= 0 || C <= 0 || (D != 256 && D != 384)) return 0;
    const int L0 = kDims.L0, L = kDims.L, T = kDims.T;
    size_t b = 0;
    auto add = [&](size_t n) { b += align256(n); };
    add((size_t)N * L * 4);            // H1
    add((size_t)N * L * 4);            // H
    add((size_t)N * 2 * D * 4);        // P
    add((size_t)N * T * 4);            // A_raw
    add(T * 2 * 4); add(T * L * 4); add(T * (L + 1) * 4);      // stats, M, Mcat
    add(C * 4); add(C * 4); add(8); add(8); add(8); add(8); add(8);   // logits, Y_prob, Y_hat, site_logits, site_prob, site_hat, (pad)
    add(C * 4); add(8); add(T * L * 4);                          // dlogits, dsite, dM
    add((size_t)N * 2 * D * 4);        // dP
    add((size_t)N * L * 4);            // dH -> dZ2 (in place)
    add((size_t)N * L * 4);            // dZ1
    add((size_t)L * 2 * D * 4); add((size_t)L * L * 4);          // WabT, W2T
    add(toad_gated_pool_ws_bytes(N, L, D, T));
    add(toad_gated_pool_bwd_ws_bytes(N, L, D, T));
    add(toad_linear_ws_bytes(N, L, L0));
    size_t wg = toad_linear_wgrad_ws_bytes(N, 2 * D, L);
    size_t w2 = toad_linear_wgrad_ws_bytes(N, L, L), w1 = toad_linear_wgrad_ws_bytes(N, L, L0);
    if (w2 > wg) wg = w2;
    if (w1 > wg) wg = w1;
    add(wg);
    (void)L0;
    return b + ((size_t)1 << 21);
}

// events: NULL, or 18 hipEvent_t: [0,1] bracket the fused pool forward, [2+2i, 3+2i] bracket GEMM call i
// (fwd1, fwd2, fwd_ab, wgrad_ab, dgrad_ab, wgrad_2, dgrad_2, wgrad_1) - for bench.py's roofline figures.
extern "C" int toad_mil_step_f32(const float *const *params, float *const *grads, float beta, const float *X,
                                  const float *sex, const int64_t *label, const int64_t *site, float w_cls,
                                  float w_site, int64_t N, int C, int D, float drop_p, uint64_t seed,
                                  float *loss_out, float *logits_out, float *site_logits_out, void *ws,
                                  size_t ws_bytes, void **events, void *stream) {
    const char *what = "toad_mil_step_f32";
    if (!params || !grads || !X || !sex || !label || !site || !loss_out || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (ws_bytes < toad_mil_step_ws_bytes(N, C, D) || toad_mil_step_ws_bytes(N, C, D) == 0) { set_error("%s: workspace too small or bad shape", what); return TOAD_EWORKSPACE; }
    if (!aligned16(ws)) { set_error("%s: workspace must be 16-byte aligned", what); return TOAD_EALIGN; }
    const int L0 = kDims.L0, L = kDims.L, T = kDims.T;
    // parameter slots (same order for params and grads): w1 b1 w2 b2 wab bab wc bc wcls bcls wsite bsite
    const float *w1 = params[0], *b1 = params[1], *w2 = params[2], *b2 = params[3], *wab = params[4], *bab = params[5],
                *wc = params[6], *bc = params[7], *wcls = params[8], *bcls = params[9], *wsite = params[10], *bsite = params[11];
    for (int i = 0; i < 12; ++i) if (!params[i] || !grads[i]) { set_error("%s: null parameter/gradient slot %d", what, i); return TOAD_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    Arena a{reinterpret_cast<char *>(ws), 0, ws_bytes};
    a.off = align256(reinterpret_cast<uintptr_t>(ws)) - reinterpret_cast<uintptr_t>(ws);     // 2 MiB-align the first buffer
    float *H1 = a.take<float>((size_t)N * L), *H = a.take<float>((size_t)N * L), *P = a.take<float>((size_t)N * 2 * D);
    float *A_raw = a.take<float>((size_t)N * T), *stats = a.take<float>(T * 2), *M = a.take<float>(T * L), *Mcat = a.take<float>(T * (L + 1));
    float *logits = a.take<float>(C), *yprob = a.take<float>(C);
    int64_t *yhat = a.take<int64_t>(1);
    float *slog = a.take<float>(2), *sprob = a.take<float>(2);
    int64_t *shat = a.take<int64_t>(1);
    (void)a.take<float>(2);
    float *dlogits = a.take<float>(C), *dsite = a.take<float>(2), *dM = a.take<float>(T * L);
    float *dP = a.take<float>((size_t)N * 2 * D), *dH = a.take<float>((size_t)N * L), *dZ1 = a.take<float>((size_t)N * L);
    float *WabT = a.take<float>((size_t)L * 2 * D), *W2T = a.take<float>((size_t)L * L);
    const size_t pws = toad_gated_pool_ws_bytes(N, L, D, T), pbws = toad_gated_pool_bwd_ws_bytes(N, L, D, T);
    void *pool_ws = a.take<char>(pws), *poolb_ws = a.take<char>(pbws);
    const size_t gws = toad_linear_ws_bytes(N, L, L0);
    void *gemm_ws = a.take<char>(gws);
    void *wgrad_ws = a.base + a.off;
    const size_t wgrad_cap = ws_bytes - a.off;

    auto ev = [&](int i) { if (events && events[i]) (void)hipEventRecord((hipEvent_t)events[i], st); };
    // four dropout streams from one seed (must match toad_amd.functional.drop_seeds)
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    const uint64_t s1 = seed + 1 * G, s2 = seed + 2 * G, sa = seed + 3 * G, sb = seed + 4 * G;
    const float mscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    int rc;
#define TOAD_TRY(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- forward
    ev(2); TOAD_TRY(toad_linear_act_fwd_f32(X, w1, b1, H1, N, L0, L, TOAD_ACT_RELU, drop_p, s1, gemm_ws, gws, st)); ev(3);
    ev(4); TOAD_TRY(toad_linear_act_fwd_f32(H1, w2, b2, H, N, L, L, TOAD_ACT_RELU, drop_p, s2, gemm_ws, gws, st)); ev(5);
    ev(6); TOAD_TRY(toad_linear_act_fwd_f32(H, wab, bab, P, N, L, 2 * D, TOAD_ACT_NONE, 0.f, 0, gemm_ws, gws, st)); ev(7);
    ev(0); TOAD_TRY(toad_gated_pool_fwd_f32(P, P + D, 2 * D, H, wc, bc, A_raw, M, stats, pool_ws, pws, N, L, D, T, drop_p, sa, sb, st)); ev(1);
    TOAD_TRY(toad_heads_fwd_f32(M, sex, wcls, bcls, wsite, bsite, Mcat, logits, yprob, yhat, slog, sprob, shat, L, C, st));
    TOAD_TRY(toad_mtl_ce_fwd_bwd_f32(logits, slog, label, site, w_cls, w_site, loss_out, dlogits, dsite, C, st));
    if (logits_out) (void)hipMemcpyAsync(logits_out, logits, C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (site_logits_out) (void)hipMemcpyAsync(site_logits_out, slog, 2 * sizeof(float), hipMemcpyDeviceToDevice, st);
    // ---- backward
    TOAD_TRY(toad_heads_bwd_f32(Mcat, dlogits, dsite, wcls, wsite, nullptr, grads[8], grads[9], grads[10], grads[11], dM, beta, L, C, st));
    TOAD_TRY(toad_gated_pool_bwd_f32(P, P + D, 2 * D, H, wc, A_raw, stats, M, dM, nullptr, dP, dP + D, 2 * D, dH, grads[6], grads[7],
                                     beta, poolb_ws, pbws, N, L, D, T, drop_p, sa, sb, st));
    ev(8); TOAD_TRY(toad_linear_wgrad_f32(dP, H, grads[4], grads[5], N, 2 * D, L, beta, wgrad_ws, wgrad_cap, st)); ev(9);
    TOAD_TRY(toad_transpose_f32(wab, WabT, 2 * D, L, st));
    ev(10); TOAD_TRY(toad_linear_dgrad_f32(dP, WabT, dH, H, mscale, dH, N, 2 * D, L, gemm_ws, gws, st)); ev(11);
    ev(12); TOAD_TRY(toad_linear_wgrad_f32(dH, H1, grads[2], grads[3], N, L, L, beta, wgrad_ws, wgrad_cap, st)); ev(13);
    TOAD_TRY(toad_transpose_f32(w2, W2T, L, L, st));
    ev(14); TOAD_TRY(toad_linear_dgrad_f32(dH, W2T, nullptr, H1, mscale, dZ1, N, L, L, gemm_ws, gws, st)); ev(15);
    ev(16); TOAD_TRY(toad_linear_wgrad_f32(dZ1, X, grads[0], grads[1], N, L, L0, beta, wgrad_ws, wgrad_cap, st)); ev(17);
#undef TOAD_TRY
    return TOAD_OK;
}

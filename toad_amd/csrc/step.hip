// step.hip — whole-slide entry points of TOAD's MIL path: forward, backward and the fused training step.
//
// The reference drives the path with `results = model(data, sex)` and `loss.backward()` (utils/core_utils_mtl_concat.py:206,231;
// forward = models/model_toad.py:90-116). Here each of those is ONE C-ABI call that sequences the kernels of this library in C++
// over caller-owned memory: a forward ARENA (saved activations, outputs, abs-max arrays - what the backward and the caller read)
// and a reusable SCRATCH (GEMM slabs, split weight planes, gradients of activations). What disappears against the per-op path
// is ~30 host round trips and ~40 allocations per slide, per-call weight re-splitting (all weight operands of a pass are split
// by one launch; the dgrad operands are read transposed in place, no transpose launches), the [N,512] dH_pool round trip (the
// dgrad epilogue recomputes it), and three dependent tail launches (heads + CE + heads backward are one single-workgroup kernel
// in the fused step).
#include "common.h"

namespace toad {

struct MilShape { int64_t N; int C, D; };
constexpr int kL0 = 1024, kL = 512, kT = 2;          // size_arg "big"/"small": [1024, 512, D] (models/model_toad.py:56)

// Sub-buffers of a large bag start on 2 MiB boundaries (like separate large device allocations do): with 256-B packing the
// pool kernels, which stream P, H and dP concurrently, ran 9-17 % slower (HBM channel aliasing between the streams; measured
// with rocprofv3 on the same kernels, profiles/). Small bags pack at 4 KiB so that a 256-patch slide does not hold 30 MB.
static inline size_t big_align(int64_t N) { return N >= 8192 ? ((size_t)1 << 21) : ((size_t)1 << 12); }
static inline size_t up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes, size_t align) { off = up(off, align); const size_t o = off; off += bytes; return o; }
};

// ---- forward arena -------------------------------------------------------------------------------------------------
enum { A_H1 = 0, A_H, A_P, A_ARAW, A_STATS, A_M, A_MCAT, A_LOGITS, A_YPROB, A_YHAT, A_SLOG, A_SPROB, A_SHAT, A_AMAX_X, A_AMAX_H1, A_AMAX_H,
       A_BITS_H1, A_BITS_H };
static size_t arena_layout(const MilShape &s, int64_t *o) {
    const size_t big = big_align(s.N), N = (size_t)s.N;
    const size_t nb = toad_amax_floats(s.N) * sizeof(float);
    Carver c;
    o[A_H1] = c.take(N * kL * 4, big);
    o[A_H] = c.take(N * kL * 4, big);
    o[A_P] = c.take(N * 2 * s.D * 4, big);
    o[A_ARAW] = c.take(N * kT * 4, big);
    o[A_STATS] = c.take(kT * 2 * 4, 256);
    o[A_M] = c.take(kT * kL * 4, 256);
    o[A_MCAT] = c.take(kT * (kL + 1) * 4, 256);
    o[A_LOGITS] = c.take((size_t)s.C * 4, 256);
    o[A_YPROB] = c.take((size_t)s.C * 4, 256);
    o[A_YHAT] = c.take(8, 256);
    o[A_SLOG] = c.take(8, 256);
    o[A_SPROB] = c.take(8, 256);
    o[A_SHAT] = c.take(8, 256);
    o[A_AMAX_X] = c.take(nb, 256);        // the three arrays are adjacent: the weight-split launch zeroes them in one go
    o[A_AMAX_H1] = c.take(nb, 4);
    o[A_AMAX_H] = c.take(nb, 4);
    o[A_BITS_H1] = c.take(toad_relu_bits_bytes(s.N, kL), 4096);      // one-bit ReLU images of H1 and H (8 KB per 256 x 256 tile)
    o[A_BITS_H] = c.take(toad_relu_bits_bytes(s.N, kL), 4096);
    return up(c.off, 256);
}

// ---- scratch ---------------------------------------------------------------------------------------------------------
enum { W_1 = 0, W_2, W_AB, W_ABT, W_2T, W_1T, W_COUNT };      // split weight operands: forward x3, dgrad (transposed in place) x3
struct Scratch {
    float *slabs; void *gemm_ws; size_t gemm_ws_bytes;
    unsigned short *planes[W_COUNT]; float *binv[W_COUNT];
    float *t_abT, *t_2T, *t_1T;                               // materialised transposes (legacy path only)
    void *pool_ws, *poolb_ws; size_t pool_ws_bytes, poolb_ws_bytes;
    float *dlogits, *dsite, *dM, *loss;
    float *amax_dP, *amax_dZ2, *amax_dZ1;
    int *slab_ke;                                             // exponents of the first Linear's K-split slabs (the GEMM measures a raw fp32 bag itself)
    float *dP, *dZ2, *dZ1;
    void *wgrad_ws, *wgrad_ws2, *wgrad_ws3; size_t wgrad_ws_bytes;
    size_t total;
};
static Scratch scratch_layout(const MilShape &s, char *base) {
    const size_t big = big_align(s.N), N = (size_t)s.N;
    const int D2 = 2 * s.D;
    const int wn[W_COUNT] = {kL, kL, D2, kL, kL, kL0}, wk[W_COUNT] = {kL0, kL, kL, D2, kL, kL};
    Scratch r{};
    Carver c;
    auto P = [&](size_t off) { return base ? base + off : nullptr; };
    r.gemm_ws_bytes = toad_linear_ws_bytes(s.N, kL0, kL0);            // slabs first, then room for the legacy per-op calls
    r.gemm_ws = P(c.take(r.gemm_ws_bytes, big));
    r.slabs = reinterpret_cast<float *>(r.gemm_ws);
    for (int i = 0; i < W_COUNT; ++i) {
        r.planes[i] = reinterpret_cast<unsigned short *>(P(c.take(h2_planes_bytes(wn[i], wk[i]), 4096)));
        r.binv[i] = reinterpret_cast<float *>(P(c.take(h2_binv_bytes(wn[i]), 256)));
    }
    r.t_abT = reinterpret_cast<float *>(P(c.take((size_t)kL * D2 * 4, 256)));
    r.t_2T = reinterpret_cast<float *>(P(c.take((size_t)kL * kL * 4, 256)));
    r.t_1T = reinterpret_cast<float *>(P(c.take((size_t)kL0 * kL * 4, 256)));
    r.pool_ws_bytes = toad_gated_pool_ws_bytes(s.N, kL, s.D, kT);
    r.poolb_ws_bytes = toad_gated_pool_bwd_ws_bytes(s.N, kL, s.D, kT);
    r.pool_ws = P(c.take(r.pool_ws_bytes, 4096));
    r.poolb_ws = P(c.take(r.poolb_ws_bytes, 4096));
    r.dlogits = reinterpret_cast<float *>(P(c.take((size_t)s.C * 4, 256)));
    r.dsite = reinterpret_cast<float *>(P(c.take(8, 256)));
    r.dM = reinterpret_cast<float *>(P(c.take(kT * kL * 4, 256)));
    r.loss = reinterpret_cast<float *>(P(c.take(16, 256)));
    const size_t nb = toad_amax_floats(s.N) * sizeof(float);
    r.amax_dP = reinterpret_cast<float *>(P(c.take(nb, 256)));        // three adjacent arrays: one memset per backward
    r.amax_dZ2 = reinterpret_cast<float *>(P(c.take(nb, 4)));
    r.amax_dZ1 = reinterpret_cast<float *>(P(c.take(nb, 4)));
    r.slab_ke = reinterpret_cast<int *>(P(c.take(256 * sizeof(int), 256)));
    r.dP = reinterpret_cast<float *>(P(c.take(N * D2 * 4, big)));
    r.dZ2 = reinterpret_cast<float *>(P(c.take(N * kL * 4, big)));
    r.dZ1 = reinterpret_cast<float *>(P(c.take(N * kL * 4, big)));
    size_t wg = toad_linear_wgrad_ws_bytes(s.N, D2, kL);
    const size_t w2 = toad_linear_wgrad_ws_bytes(s.N, kL, kL), w1 = toad_linear_wgrad_ws_bytes(s.N, kL, kL0);
    if (w2 > wg) wg = w2;
    if (w1 > wg) wg = w1;
    r.wgrad_ws_bytes = wg;
    r.wgrad_ws = P(c.take(wg, big));           // one area per weight gradient: their slab reductions are deferred to ONE launch at the end
    r.wgrad_ws2 = P(c.take(wg, big));
    r.wgrad_ws3 = P(c.take(wg, big));
    r.total = up(c.off, 256);
    return r;
}

struct Fwd {                       // typed view of the arena
    float *H1, *H, *P, *A_raw, *stats, *M, *Mcat, *logits, *yprob; int64_t *yhat; float *slog, *sprob; int64_t *shat;
    float *amax_x, *amax_h1, *amax_h;
    unsigned long long *bits_h1, *bits_h;
};
static Fwd arena_view(const MilShape &s, char *base) {
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    arena_layout(s, o);
    Fwd f;
    f.H1 = (float *)(base + o[A_H1]); f.H = (float *)(base + o[A_H]); f.P = (float *)(base + o[A_P]); f.A_raw = (float *)(base + o[A_ARAW]);
    f.stats = (float *)(base + o[A_STATS]); f.M = (float *)(base + o[A_M]); f.Mcat = (float *)(base + o[A_MCAT]);
    f.logits = (float *)(base + o[A_LOGITS]); f.yprob = (float *)(base + o[A_YPROB]); f.yhat = (int64_t *)(base + o[A_YHAT]);
    f.slog = (float *)(base + o[A_SLOG]); f.sprob = (float *)(base + o[A_SPROB]); f.shat = (int64_t *)(base + o[A_SHAT]);
    f.amax_x = (float *)(base + o[A_AMAX_X]); f.amax_h1 = (float *)(base + o[A_AMAX_H1]); f.amax_h = (float *)(base + o[A_AMAX_H]);
    f.bits_h1 = (unsigned long long *)(base + o[A_BITS_H1]); f.bits_h = (unsigned long long *)(base + o[A_BITS_H]);
    return f;
}

struct Params { const float *w1, *b1, *w2, *b2, *wab, *bab, *wc, *bc, *wcls, *bcls, *wsite, *bsite; };
static bool load_params(const float *const *p, Params &q, const char *what) {
    for (int i = 0; i < 12; ++i) if (!p[i]) { set_error("%s: null parameter slot %d", what, i); return false; }
    q = Params{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]};
    return true;
}
static bool shape_ok(const MilShape &s) { return s.N > 0 && s.N < INT32_MAX - 4096 && s.C > 0 && s.C <= 512 && (s.D == 256 || s.D == 384); }

struct DropSeeds { uint64_t s1, s2, sa, sb; float mscale; };
static DropSeeds drop_seeds(float drop_p, uint64_t seed) {          // must match toad_amd.functional.drop_seeds
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    return DropSeeds{seed + 1 * G, seed + 2 * G, seed + 3 * G, seed + 4 * G, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
}

#define TOAD_TRY(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

// ---- bags beyond the NT kernels' 32-bit row offsets (more than ~1.05 M patches): the same kernels over ROW CHUNKS -----------------------------
// gemm_nt_h2_big_kernel addresses its A operand with 32-bit byte offsets (h2_nt_ok: M * lda * 4 < 2^32, i.e. 1,048,575 rows of the 1024-wide
// bag). Rows of an NT product are independent, so a longer bag runs as consecutive launches over chunks of kChunkRows rows (a multiple of the
// 256-row blocks the abs-max arrays, the one-bit ReLU images and the per-tile scales are indexed by): every per-row pointer advances by the
// chunk's first row, the weight planes are shared. Until round 5 such bags fell back to the exact-fp32 128 x 128 kernels (5x slower).
// A chunk of at most kChunkRows rows is ONE launch with exactly the arguments of the unchunked call, so nothing changes for ordinary bags.
// Train-mode dropout in the epilogue hashes the element index INSIDE a launch: chunk j > 0 therefore draws its masks from the stream
// seed + j * kChunkSeedStep at the chunk-local index (toad_dropout_mask_f32 reproduces them chunk by chunk); the backward never recomputes
// trunk masks (the saved outputs carry them), so the two passes cannot disagree. The weight gradients reduce over rows inside ONE launch of the
// TN kernels, which take any M.
constexpr int64_t kChunkRows = 4092 * 256;                  // 1,047,552 rows: 4,290,772,992 bytes of a 1024-wide fp32 operand
constexpr uint64_t kChunkSeedStep = 0xD1B54A32D192ED03ull;
static bool nt_rows_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
    return h2_nt_ok(M < kChunkRows ? M : kChunkRows, N, K, lda, ldc);
}
static int nt_rows(const float *A, int64_t lda, const float *a_amax, const unsigned short *planes, const float *binv, float *C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const float *bias, EpiScalars es, const float *addend, const float *mask_src,
                   const unsigned long long *mask_bits, H2Pool pool, float *slabs, float *y_amax, unsigned long long *bits_out, hipStream_t st,
                   const char *what, int a_mode = TOAD_X_F32, float *a_amax_out = nullptr, int *slab_ke = nullptr) {
    if (M <= kChunkRows)
        return launch_nt_h2(A, lda, a_amax, planes, binv, C, ldc, M, N, K, bias, es, addend, mask_src, mask_bits, pool, slabs, y_amax, bits_out, st, what,
                            a_mode, 1, 1, a_amax_out, slab_ke);
    if (a_mode != TOAD_X_F32) { set_error("%s: an fp16 / prepared bag is limited to the kernels' 32-bit row offsets", what); return TOAD_ESHAPE; }
    const int64_t bits_per_blk = (int64_t)((N + 255) / 256) * 8 * 2 * 64;          // 64-bit words of the ReLU image per 256-row block (toad_relu_bits_bytes)
    int j = 0;
    for (int64_t c0 = 0; c0 < M; c0 += kChunkRows, ++j) {
        const int64_t m = M - c0 < kChunkRows ? M - c0 : kChunkRows, blk = c0 / H2_ROWBLK;
        EpiScalars e = es;
        if (e.drop.thresh) e.drop.seed += (uint64_t)j * kChunkSeedStep;
        H2Pool pl = pool;
        if (pl.T > 0) pl.a_raw += c0 * pl.T;
        TOAD_TRY(launch_nt_h2(A + c0 * lda, lda, a_amax ? a_amax + blk : nullptr, planes, binv, C + c0 * ldc, ldc, m, N, K, bias, e, addend ? addend + c0 * ldc : nullptr,
                              mask_src ? mask_src + c0 * ldc : nullptr, mask_bits ? mask_bits + blk * bits_per_blk : nullptr, pl, slabs, y_amax ? y_amax + blk : nullptr,
                              bits_out ? bits_out + blk * bits_per_blk : nullptr, st, what, a_mode, 1, 1, a_amax_out ? a_amax_out + blk : nullptr, slab_ke));
    }
    return TOAD_OK;
}

// forward up to the pooled features: trunk, stacked attention GEMM, fused gated pool. `ev` records bench events (or nothing).
template <typename Ev>
static int forward_body(const MilShape &s, const Params &p, const float *X, const float *x_amax, float drop_p, uint64_t seed,
                        bool attention_only, const Fwd &f, const Scratch &w, hipStream_t st, Ev ev, const char *what, int x_mode,
                        bool with_backward_operands = false) {
    const int64_t N = s.N;
    const int D2 = 2 * s.D;
    const DropSeeds ds = drop_seeds(drop_p, seed);
    const bool h2 = x_mode == TOAD_X_F32 ? nt_rows_ok(N, kL, kL0, kL0, kL) : h2_nt_ok(N, kL, kL0, kL0, kL);     // fp32 bags of any length: row chunks (nt_rows)
    if (x_mode != TOAD_X_F32 && !h2) { set_error("%s: an fp16 / prepared bag needs the fp16 two-piece kernels (N * 1024 * 4 < 2^32)", what); return TOAD_ESHAPE; }
    if (x_mode == TOAD_X_PT && !x_amax) { set_error("%s: a prepared bag comes with its abs-max array", what); return TOAD_EINVAL; }
    const EpiScalars relu1{1, 1.f, make_drop(drop_p, ds.s1)}, relu2{1, 1.f, make_drop(drop_p, ds.s2)}, lin{0, 1.f, make_drop(0.f, 0)};
    const H2Pool nopool{nullptr, nullptr, nullptr, 0};
    if (h2) {
        const H2Operand ops[3] = {{p.w1, kL0, 1, kL, kL0, w.planes[W_1], w.binv[W_1]},
                                  {p.w2, kL, 1, kL, kL, w.planes[W_2], w.binv[W_2]},
                                  {p.wab, kL, 1, D2, kL, w.planes[W_AB], w.binv[W_AB]}};
        // one launch splits the three forward weight operands AND zeroes the three adjacent abs-max arrays (x, h1, h): no memsets
        const int nz = (int)(((char *)f.amax_h - (char *)f.amax_x) / sizeof(float) + toad_amax_floats(N));
        if (with_backward_operands) {
            // the fused step knows its backward follows with the same weights: the two dgrad operands (transposed in place) and the
            // backward's abs-max arrays go into the same launch (one launch less per slide; backward_body is told `presplit`)
            const H2Operand ops5[5] = {ops[0], ops[1], ops[2], {p.wab, 1, kL, kL, D2, w.planes[W_ABT], w.binv[W_ABT]},
                                       {p.w2, 1, kL, kL, kL, w.planes[W_2T], w.binv[W_2T]}};
            const int nzb = (int)(((char *)w.amax_dZ1 - (char *)w.amax_dP) / sizeof(float) + toad_amax_floats(N));
            TOAD_TRY(launch_split_h2(ops5, 5, f.amax_x, nz, st, what, w.amax_dP, nzb));
        } else {
            TOAD_TRY(launch_split_h2(ops, 3, f.amax_x, nz, st, what));
        }
        // an fp16 bag needs no abs-max array: its elements are first pieces with scale 1 (gemm_nt_h2_big_kernel, A16)
        const float *ax = x_mode == TOAD_X_PT ? x_amax : f.amax_x;    // a prepared bag carries its own array: nothing to copy or measure
        // a raw fp32 bag without an abs-max array is measured INSIDE the first GEMM (running block maximum, gemm_h2.inc AMODE 3), which fills
        // f.amax_x (zeroed by the split launch above) for the weight gradient of this layer: no pass over the bag of its own
        const bool self_measure = x_mode == TOAD_X_F32 && !x_amax && nt_run_ok(N < kChunkRows ? N : kChunkRows, kL, kL0);
        if (x_mode == TOAD_X_F32 && !x_amax && !self_measure) TOAD_TRY(launch_absmax(X, kL0, N, kL0, f.amax_x, false, st, what));
        if (x_mode == TOAD_X_F32 && x_amax) (void)hipMemcpyAsync(f.amax_x, x_amax, toad_amax_floats(N) * sizeof(float), hipMemcpyDeviceToDevice, st);
        ev(2); TOAD_TRY(nt_rows(X, kL0, self_measure ? nullptr : ax, w.planes[W_1], w.binv[W_1], f.H1, kL, N, kL, kL0, p.b1, relu1, nullptr, nullptr, nullptr, nopool,
                                w.slabs, f.amax_h1, f.bits_h1, st, what, x_mode, self_measure ? f.amax_x : nullptr, self_measure ? w.slab_ke : nullptr)); ev(3);
        ev(4); TOAD_TRY(nt_rows(f.H1, kL, f.amax_h1, w.planes[W_2], w.binv[W_2], f.H, kL, N, kL, kL, p.b2, relu2, nullptr, nullptr, nullptr, nopool, w.slabs, f.amax_h, f.bits_h, st, what)); ev(5);
        ev(6); TOAD_TRY(nt_rows(f.H, kL, f.amax_h, w.planes[W_AB], w.binv[W_AB], f.P, D2, N, D2, kL, p.bab, lin, nullptr, nullptr, nullptr, nopool, w.slabs, nullptr, nullptr, st, what)); ev(7);
    } else {       // (unreachable for the TOAD shapes since round 5; kept for operands the NT kernels refuse for another reason)
        ev(2); TOAD_TRY(toad_linear_act_fwd_f32(X, p.w1, p.b1, f.H1, N, kL0, kL, TOAD_ACT_RELU, drop_p, ds.s1, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st)); ev(3);
        ev(4); TOAD_TRY(toad_linear_act_fwd_f32(f.H1, p.w2, p.b2, f.H, N, kL, kL, TOAD_ACT_RELU, drop_p, ds.s2, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st)); ev(5);
        ev(6); TOAD_TRY(toad_linear_act_fwd_f32(f.H, p.wab, p.bab, f.P, N, kL, D2, TOAD_ACT_NONE, 0.f, 0, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st)); ev(7);
    }
    if (attention_only) {
        TOAD_TRY(toad_gated_pool_fwd_f32(f.P, f.P + s.D, D2, nullptr, p.wc, p.bc, f.A_raw, nullptr, nullptr, nullptr, 0, N, kL, s.D, kT, drop_p, ds.sa, ds.sb, st));
        return TOAD_OK;
    }
    ev(0); TOAD_TRY(toad_gated_pool_fwd_f32(f.P, f.P + s.D, D2, f.H, p.wc, p.bc, f.A_raw, f.M, f.stats, w.pool_ws, w.pool_ws_bytes, N, kL, s.D, kT, drop_p, ds.sa, ds.sb, st)); ev(1);
    return TOAD_OK;
}

// backward from dM (gradient of the pooled features) down to the trunk weights (and dX)
template <typename Ev>
static int backward_body(const MilShape &s, const Params &p, float *const *grads, float beta, const float *X, const float *x_amax_pt, float drop_p, uint64_t seed,
                         const Fwd &f, const float *dM, const float *dA_ext, float *dX, const Scratch &w, hipStream_t st, Ev ev, const char *what,
                         int x_mode, bool presplit = false) {
    const int64_t N = s.N;
    const int D2 = 2 * s.D;
    const DropSeeds ds = drop_seeds(drop_p, seed);
    const bool h2 = x_mode == TOAD_X_F32 ? nt_rows_ok(N, kL, kL0, kL0, kL) : h2_nt_ok(N, kL, kL0, kL0, kL);
    if (x_mode != TOAD_X_F32 && !h2) { set_error("%s: an fp16 / prepared bag needs the fp16 two-piece kernels", what); return TOAD_ESHAPE; }
    const EpiScalars msk{0, ds.mscale, make_drop(0.f, 0)}, plain{0, 1.f, make_drop(0.f, 0)};
    const H2Pool nopool{nullptr, nullptr, nullptr, 0};
    if (h2) {
        // dgrad operands B[n,k] = W[k,n], read transposed in place: no transpose launches; the same launch zeroes the three
        // adjacent abs-max arrays of this pass (dP, dZ2, dZ1)
        const H2Operand ops[3] = {{p.wab, 1, kL, kL, D2, w.planes[W_ABT], w.binv[W_ABT]},
                                  {p.w2, 1, kL, kL, kL, w.planes[W_2T], w.binv[W_2T]},
                                  {p.w1, 1, kL0, kL0, kL, w.planes[W_1T], w.binv[W_1T]}};
        const int nz = (int)(((char *)w.amax_dZ1 - (char *)w.amax_dP) / sizeof(float) + toad_amax_floats(N));
        if (!(presplit && !dX)) TOAD_TRY(launch_split_h2(ops, dX ? 3 : 2, w.amax_dP, nz, st, what));
        TOAD_TRY(launch_pool_bwd(f.P, f.P + s.D, D2, f.H, p.wc, f.A_raw, f.stats, f.M, dM, dA_ext, w.dP, w.dP + s.D, D2, nullptr,
                                 grads[6], grads[7], beta, w.amax_dP, false, w.poolb_ws, w.poolb_ws_bytes, N, kL, s.D, kT, drop_p, ds.sa, ds.sb, st));
        WgradDeferred dw[3];
        // Bags of at most kTnBatchMaxRows rows (raw fp32): the three weight gradients wait for the end of the pass and run as ONE launch
        // (launch_wgrad_batch: dP / dZ2 / dZ1 and the saved activations all still exist then) - a third of the slab traffic and two launches less
        // than three launches of the per-XCD plan: 10k patches 142 -> 86 us, 100k patches 724 -> 661 us of the step's weight-gradient time
        // (profiles/r06*); at 500k concatenated rows the two orders measure the same, and longer calls keep the interleaved one.
        const WgradJob wj[3] = {{w.dP, w.amax_dP, f.H, f.amax_h, grads[4], grads[5], D2, kL, w.wgrad_ws},
                                {w.dZ2, w.amax_dZ2, f.H1, f.amax_h1, grads[2], grads[3], kL, kL, w.wgrad_ws2},
                                {w.dZ1, w.amax_dZ1, X, f.amax_x, grads[0], grads[1], kL, kL0, w.wgrad_ws3}};
        const bool wbatch = x_mode == TOAD_X_F32 && wgrad_batch_ok(N, wj, 3, w.wgrad_ws_bytes);
        if (!wbatch) { ev(8); TOAD_TRY(launch_wgrad(w.dP, w.amax_dP, f.H, f.amax_h, grads[4], grads[5], N, D2, kL, beta, w.wgrad_ws, st, what, TOAD_X_F32, &dw[0])); ev(9); }
        // dZ2 = (dP Wab + dH_pool) * (H > 0): the pooling gradient dH_pool is recomputed in the epilogue from A_raw, stats, dM
        ev(10); TOAD_TRY(nt_rows(w.dP, D2, w.amax_dP, w.planes[W_ABT], w.binv[W_ABT], w.dZ2, kL, N, kL, D2, nullptr, msk, nullptr, f.H, f.bits_h,
                                 H2Pool{f.A_raw, f.stats, dM, kT}, w.slabs, w.amax_dZ2, nullptr, st, what)); ev(11);
        if (!wbatch) { ev(12); TOAD_TRY(launch_wgrad(w.dZ2, w.amax_dZ2, f.H1, f.amax_h1, grads[2], grads[3], N, kL, kL, beta, w.wgrad_ws2, st, what, TOAD_X_F32, &dw[1])); ev(13); }
        // (an fp16 / prepared bag ran its first Linear on 256-row tiles, whose K-split remainder tiles leave no bits; where this dgrad would run on
        //  half-height tiles - which read the bit image for EVERY tile - it takes the fp32 activations instead)
        const unsigned long long *bits1 = (x_mode == TOAD_X_F32 || !nt_half_tiles(N < kChunkRows ? N : kChunkRows, kL)) ? f.bits_h1 : nullptr;
        ev(14); TOAD_TRY(nt_rows(w.dZ2, kL, w.amax_dZ2, w.planes[W_2T], w.binv[W_2T], w.dZ1, kL, N, kL, kL, nullptr, msk, nullptr, f.H1, bits1, nopool,
                                 w.slabs, w.amax_dZ1, nullptr, st, what)); ev(15);
        if (wbatch) {         // (bench events: the one launch + the reduction are booked under the first weight gradient's pair, the other two pairs are empty)
            ev(8); TOAD_TRY(launch_wgrad_batch(wj, 3, N, beta, st, what, dw));
            TOAD_TRY(launch_wgrad_reduce(dw, 3, st, what)); ev(9);
            ev(12); ev(13); ev(16); ev(17);
        } else {
            ev(16); TOAD_TRY(launch_wgrad(w.dZ1, w.amax_dZ1, X, x_mode == TOAD_X_PT ? x_amax_pt : f.amax_x, grads[0], grads[1], N, kL, kL0, beta, w.wgrad_ws3, st, what, x_mode, &dw[2]));
            TOAD_TRY(launch_wgrad_reduce(dw, 3, st, what)); ev(17);
        }
        if (dX) TOAD_TRY(nt_rows(w.dZ1, kL, w.amax_dZ1, w.planes[W_1T], w.binv[W_1T], dX, kL0, N, kL0, kL, nullptr, plain, nullptr, nullptr, nullptr, nopool,
                                 w.slabs, nullptr, nullptr, st, what));
        return TOAD_OK;
    }
    // legacy sequence (per-op entry points, materialised dH_pool in the dZ2 buffer, explicit transposes)
    float *dH = w.dZ2;
    TOAD_TRY(toad_gated_pool_bwd_f32(f.P, f.P + s.D, D2, f.H, p.wc, f.A_raw, f.stats, f.M, dM, dA_ext, w.dP, w.dP + s.D, D2, dH, grads[6], grads[7],
                                     beta, nullptr, w.poolb_ws, w.poolb_ws_bytes, N, kL, s.D, kT, drop_p, ds.sa, ds.sb, st));
    ev(8); TOAD_TRY(toad_linear_wgrad_f32(w.dP, f.H, grads[4], grads[5], N, D2, kL, beta, nullptr, nullptr, w.wgrad_ws, w.wgrad_ws_bytes, st)); ev(9);
    TOAD_TRY(toad_transpose_f32(p.wab, w.t_abT, D2, kL, st));
    ev(10); TOAD_TRY(toad_linear_dgrad_f32(w.dP, w.t_abT, dH, f.H, ds.mscale, dH, N, D2, kL, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st)); ev(11);
    ev(12); TOAD_TRY(toad_linear_wgrad_f32(dH, f.H1, grads[2], grads[3], N, kL, kL, beta, nullptr, nullptr, w.wgrad_ws, w.wgrad_ws_bytes, st)); ev(13);
    TOAD_TRY(toad_transpose_f32(p.w2, w.t_2T, kL, kL, st));
    ev(14); TOAD_TRY(toad_linear_dgrad_f32(dH, w.t_2T, nullptr, f.H1, ds.mscale, w.dZ1, N, kL, kL, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st)); ev(15);
    ev(16); TOAD_TRY(toad_linear_wgrad_f32(w.dZ1, X, grads[0], grads[1], N, kL, kL0, beta, nullptr, nullptr, w.wgrad_ws, w.wgrad_ws_bytes, st)); ev(17);
    if (dX) {
        TOAD_TRY(toad_transpose_f32(p.w1, w.t_1T, kL, kL0, st));
        TOAD_TRY(toad_linear_dgrad_f32(w.dZ1, w.t_1T, nullptr, nullptr, 1.f, dX, N, kL, kL0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, w.gemm_ws, w.gemm_ws_bytes, st));
    }
    return TOAD_OK;
}

struct NoEvents { void operator()(int) const {} };
struct StreamEvents {
    void **events; hipStream_t st;
    void operator()(int i) const { if (events && events[i]) (void)hipEventRecord((hipEvent_t)events[i], st); }
};

}  // namespace toad

using namespace toad;

extern "C" size_t toad_mil_arena_bytes(int64_t N, int C, int D) {
    const MilShape s{N, C, D};
    if (!shape_ok(s)) return 0;
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    return arena_layout(s, o) + big_align(N);          // slack to align the base
}
extern "C" int toad_mil_arena_layout(int64_t N, int C, int D, int64_t *offsets) {
    const MilShape s{N, C, D};
    if (!shape_ok(s) || !offsets) { set_error("toad_mil_arena_layout: bad shape"); return TOAD_ESHAPE; }
    arena_layout(s, offsets);
    return TOAD_OK;
}
extern "C" size_t toad_mil_scratch_bytes(int64_t N, int C, int D) {
    const MilShape s{N, C, D};
    if (!shape_ok(s)) return 0;
    return scratch_layout(s, nullptr).total + big_align(N);
}
extern "C" size_t toad_mil_buffer_align(int64_t N) { return big_align(N); }
extern "C" size_t toad_mil_step_ws_bytes(int64_t N, int C, int D) {
    const size_t a = toad_mil_arena_bytes(N, C, D), b = toad_mil_scratch_bytes(N, C, D);
    return (a && b) ? a + b : 0;
}

// The arena base is used as given when it is aligned to the bag's buffer granularity, else rounded up inside the allocation.
// toad_mil_arena_layout offsets are relative to that aligned base: callers pass an aligned pointer (torch allocations of this
// size are 2 MiB aligned) or compute the same round-up.
static char *align_base(void *p, int64_t N) {
    const uintptr_t a = big_align(N);
    return reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + a - 1) & ~(a - 1));
}

static int mil_fwd_impl(const float *const *params, const float *X, const float *sex, int64_t N, int C, int D, float drop_p,
                        uint64_t seed, const float *x_amax, int attention_only, void *arena, size_t arena_bytes,
                        void *scratch, size_t scratch_bytes, void *stream, int x_mode, const char *what) {
    const MilShape s{N, C, D};
    if (!params || !X || !arena || !scratch || (!attention_only && !sex)) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (!shape_ok(s)) { set_error("%s: unsupported shape N=%lld C=%d D=%d", what, (long long)N, C, D); return TOAD_ESHAPE; }
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    if (!aligned16(X)) { set_error("%s: X must be 16-byte aligned", what); return TOAD_EALIGN; }
    char *ab = align_base(arena, N), *sb = align_base(scratch, N);
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    if ((size_t)(ab - (char *)arena) + arena_layout(s, o) > arena_bytes) { set_error("%s: arena too small", what); return TOAD_EWORKSPACE; }
    const Scratch w = scratch_layout(s, sb);
    if ((size_t)(sb - (char *)scratch) + w.total > scratch_bytes) { set_error("%s: scratch too small", what); return TOAD_EWORKSPACE; }
    Params p;
    if (!load_params(params, p, what)) return TOAD_EINVAL;
    const Fwd f = arena_view(s, ab);
    hipStream_t st = (hipStream_t)stream;
    TOAD_TRY(forward_body(s, p, X, x_amax, drop_p, seed, attention_only != 0, f, w, st, NoEvents{}, what, x_mode));
    if (attention_only) return TOAD_OK;
    return toad_heads_fwd_f32(f.M, sex, p.wcls, p.bcls, p.wsite, p.bsite, f.Mcat, f.logits, f.yprob, f.yhat, f.slog, f.sprob, f.shat, kL, C, st);
}

extern "C" int toad_mil_fwd_f32(const float *const *params, const float *X, const float *sex, int64_t N, int C, int D, float drop_p,
                                 uint64_t seed, const float *x_amax, int attention_only, void *arena, size_t arena_bytes,
                                 void *scratch, size_t scratch_bytes, void *stream) {
    return mil_fwd_impl(params, X, sex, N, C, D, drop_p, seed, x_amax, attention_only, arena, arena_bytes, scratch, scratch_bytes, stream, TOAD_X_F32,
                        "toad_mil_fwd_f32");
}
extern "C" int toad_mil_fwd_x16_f32(const float *const *params, const void *X16, const float *sex, int64_t N, int C, int D, float drop_p,
                                     uint64_t seed, int attention_only, void *arena, size_t arena_bytes, void *scratch, size_t scratch_bytes,
                                     void *stream) {
    return mil_fwd_impl(params, reinterpret_cast<const float *>(X16), sex, N, C, D, drop_p, seed, nullptr, attention_only, arena, arena_bytes, scratch,
                        scratch_bytes, stream, TOAD_X_F16, "toad_mil_fwd_x16_f32");
}

static int mil_bwd_impl(const float *const *params, float *const *grads, float beta, const float *X, const float *x_amax_pt, int64_t N, int C, int D,
                        float drop_p, uint64_t seed, const void *arena, size_t arena_bytes, const float *dlogits,
                        const float *dsite, const float *dA_ext, const float *dMcat_ext, float *dX, float *dsex,
                        void *scratch, size_t scratch_bytes, void *stream, int x_mode, const char *what) {
    const MilShape s{N, C, D};
    if (!params || !grads || !X || !arena || !scratch || !dlogits || !dsite) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (!shape_ok(s)) { set_error("%s: unsupported shape N=%lld C=%d D=%d", what, (long long)N, C, D); return TOAD_ESHAPE; }
    if (dX && !aligned16(dX)) { set_error("%s: dX must be 16-byte aligned", what); return TOAD_EALIGN; }
    char *ab = align_base(const_cast<void *>(arena), N), *sb = align_base(scratch, N);
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    if ((size_t)(ab - (const char *)arena) + arena_layout(s, o) > arena_bytes) { set_error("%s: arena too small", what); return TOAD_EWORKSPACE; }
    const Scratch w = scratch_layout(s, sb);
    if ((size_t)(sb - (char *)scratch) + w.total > scratch_bytes) { set_error("%s: scratch too small", what); return TOAD_EWORKSPACE; }
    Params p;
    if (!load_params(params, p, what)) return TOAD_EINVAL;
    for (int i = 0; i < 12; ++i) if (!grads[i]) { set_error("%s: null gradient slot %d", what, i); return TOAD_EINVAL; }
    const Fwd f = arena_view(s, ab);
    hipStream_t st = (hipStream_t)stream;
    TOAD_TRY(toad_heads_bwd_f32(f.Mcat, dlogits, dsite, p.wcls, p.wsite, dMcat_ext, grads[8], grads[9], grads[10], grads[11], w.dM, dsex, beta, kL, C, st));
    return backward_body(s, p, grads, beta, X, x_amax_pt, drop_p, seed, f, w.dM, dA_ext, dX, w, st, NoEvents{}, what, x_mode);
}

extern "C" int toad_mil_bwd_f32(const float *const *params, float *const *grads, float beta, const float *X, int64_t N, int C, int D,
                                 float drop_p, uint64_t seed, const void *arena, size_t arena_bytes, const float *dlogits,
                                 const float *dsite, const float *dA_ext, const float *dMcat_ext, float *dX, float *dsex,
                                 void *scratch, size_t scratch_bytes, void *stream) {
    return mil_bwd_impl(params, grads, beta, X, nullptr, N, C, D, drop_p, seed, arena, arena_bytes, dlogits, dsite, dA_ext, dMcat_ext, dX, dsex, scratch,
                        scratch_bytes, stream, TOAD_X_F32, "toad_mil_bwd_f32");
}
extern "C" int toad_mil_bwd_x16_f32(const float *const *params, float *const *grads, float beta, const void *X16, int64_t N, int C, int D,
                                     float drop_p, uint64_t seed, const void *arena, size_t arena_bytes, const float *dlogits,
                                     const float *dsite, const float *dA_ext, const float *dMcat_ext, float *dsex,
                                     void *scratch, size_t scratch_bytes, void *stream) {
    return mil_bwd_impl(params, grads, beta, reinterpret_cast<const float *>(X16), nullptr, N, C, D, drop_p, seed, arena, arena_bytes, dlogits, dsite, dA_ext,
                        dMcat_ext, nullptr, dsex, scratch, scratch_bytes, stream, TOAD_X_F16, "toad_mil_bwd_x16_f32");
}

// events: NULL, or 18 hipEvent_t: [0,1] bracket the fused pool forward, [2+2i, 3+2i] bracket GEMM call i
// (fwd1, fwd2, fwd_ab, wgrad_ab, dgrad_ab, wgrad_2, dgrad_2, wgrad_1) - for bench.py's roofline figures.
static int mil_step_impl(const float *const *params, float *const *grads, float beta, const float *X,
                         const float *sex, const int64_t *label, const int64_t *site, float w_cls,
                         float w_site, int64_t N, int C, int D, float drop_p, uint64_t seed, const float *x_amax,
                         float *loss_out, float *logits_out, float *site_logits_out, void *ws,
                         size_t ws_bytes, void **events, void *stream, int x_mode, const char *what) {
    const MilShape s{N, C, D};
    if (!params || !grads || !X || !sex || !label || !site || !loss_out || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (!shape_ok(s)) { set_error("%s: unsupported shape N=%lld C=%d D=%d", what, (long long)N, C, D); return TOAD_ESHAPE; }
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    if (!aligned16(X)) { set_error("%s: X must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_mil_step_ws_bytes(N, C, D)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    Params p;
    if (!load_params(params, p, what)) return TOAD_EINVAL;
    for (int i = 0; i < 12; ++i) if (!grads[i]) { set_error("%s: null gradient slot %d", what, i); return TOAD_EINVAL; }
    char *ab = align_base(ws, N);
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    char *sb = align_base(ab + arena_layout(s, o), N);
    const Fwd f = arena_view(s, ab);
    const Scratch w = scratch_layout(s, sb);
    hipStream_t st = (hipStream_t)stream;
    const StreamEvents ev{events, st};
    const bool presplit = x_mode == TOAD_X_F32 ? nt_rows_ok(N, kL, kL0, kL0, kL) : h2_nt_ok(N, kL, kL0, kL0, kL);
    TOAD_TRY(forward_body(s, p, X, x_amax, drop_p, seed, false, f, w, st, ev, what, x_mode, presplit));
    // heads + weighted CE + heads backward: one single-workgroup launch
    TOAD_TRY(toad_heads_ce_fused_f32(f.M, sex, p.wcls, p.bcls, p.wsite, p.bsite, label, site, w_cls, w_site, f.Mcat, f.logits, f.yprob, f.yhat,
                                     f.slog, f.sprob, f.shat, loss_out, nullptr, nullptr, grads[8], grads[9], grads[10], grads[11], w.dM, beta, kL, C, st));
    if (logits_out) (void)hipMemcpyAsync(logits_out, f.logits, C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (site_logits_out) (void)hipMemcpyAsync(site_logits_out, f.slog, 2 * sizeof(float), hipMemcpyDeviceToDevice, st);
    return backward_body(s, p, grads, beta, X, x_amax, drop_p, seed, f, w.dM, nullptr, nullptr, w, st, ev, what, x_mode, presplit);
}
extern "C" int toad_mil_step_f32(const float *const *params, float *const *grads, float beta, const float *X,
                                  const float *sex, const int64_t *label, const int64_t *site, float w_cls,
                                  float w_site, int64_t N, int C, int D, float drop_p, uint64_t seed, const float *x_amax,
                                  float *loss_out, float *logits_out, float *site_logits_out, void *ws,
                                  size_t ws_bytes, void **events, void *stream) {
    return mil_step_impl(params, grads, beta, X, sex, label, site, w_cls, w_site, N, C, D, drop_p, seed, x_amax, loss_out, logits_out,
                         site_logits_out, ws, ws_bytes, events, stream, TOAD_X_F32, "toad_mil_step_f32");
}
extern "C" int toad_mil_step_x16_f32(const float *const *params, float *const *grads, float beta, const void *X16,
                                      const float *sex, const int64_t *label, const int64_t *site, float w_cls,
                                      float w_site, int64_t N, int C, int D, float drop_p, uint64_t seed,
                                      float *loss_out, float *logits_out, float *site_logits_out, void *ws,
                                      size_t ws_bytes, void **events, void *stream) {
    return mil_step_impl(params, grads, beta, reinterpret_cast<const float *>(X16), sex, label, site, w_cls, w_site, N, C, D, drop_p, seed, nullptr,
                         loss_out, logits_out, site_logits_out, ws, ws_bytes, events, stream, TOAD_X_F16, "toad_mil_step_x16_f32");
}

// ---- prepared bags (ABI 9) --------------------------------------------------------------------------------------------
// The bag of a slide is constant across epochs, and the two GEMMs that read it (the first Linear, models/model_toad.py:59, and its
// weight gradient) are a third of the step's flops. toad_bag_prepare_f32 converts it ONCE - at ingest, next to the host-to-device
// copy - into the plane-tiled two-piece form those GEMMs consume by LDS-DMA without any conversion (csrc/gemm_pt.inc): the same
// 4 bytes per element as fp32, so the caller may drop the fp32 copy. `amax` receives the bag's abs-max array (toad_amax_floats(N)
// floats), which the *_xp_* calls take back together with the planes.
extern "C" size_t toad_bag_planes_bytes(int64_t N, int64_t K) { return (N > 0 && K > 0) ? pt_bytes_host(N, K) : 0; }
extern "C" int toad_bag_prepare_f32(const float *X, int64_t N, int64_t K, void *planes, float *amax, void *stream) {
    const char *what = "toad_bag_prepare_f32";
    if (!X || !planes || !amax || N <= 0 || K <= 0 || K % 8 != 0) { set_error("%s: bad argument (K must be a multiple of 8)", what); return TOAD_EINVAL; }
    if (N > INT32_MAX - 4096) { set_error("%s: N too large", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(planes)) { set_error("%s: X and planes must be 16-byte aligned", what); return TOAD_EALIGN; }
    hipStream_t st = (hipStream_t)stream;
    TOAD_TRY(launch_absmax(X, K, N, K, amax, true, st, what));
    return launch_pt_split(X, K, N, K, amax, reinterpret_cast<unsigned short *>(planes), st, what);
}
extern "C" int toad_mil_fwd_xp_f32(const float *const *params, const void *Xp, const float *x_amax, const float *sex, int64_t N, int C, int D,
                                    float drop_p, uint64_t seed, int attention_only, void *arena, size_t arena_bytes, void *scratch,
                                    size_t scratch_bytes, void *stream) {
    return mil_fwd_impl(params, reinterpret_cast<const float *>(Xp), sex, N, C, D, drop_p, seed, x_amax, attention_only, arena, arena_bytes, scratch,
                        scratch_bytes, stream, TOAD_X_PT, "toad_mil_fwd_xp_f32");
}
extern "C" int toad_mil_bwd_xp_f32(const float *const *params, float *const *grads, float beta, const void *Xp, const float *x_amax, int64_t N, int C,
                                    int D, float drop_p, uint64_t seed, const void *arena, size_t arena_bytes, const float *dlogits,
                                    const float *dsite, const float *dA_ext, const float *dMcat_ext, float *dsex, void *scratch,
                                    size_t scratch_bytes, void *stream) {
    if (!x_amax) { set_error("toad_mil_bwd_xp_f32: null x_amax"); return TOAD_EINVAL; }
    return mil_bwd_impl(params, grads, beta, reinterpret_cast<const float *>(Xp), x_amax, N, C, D, drop_p, seed, arena, arena_bytes, dlogits, dsite, dA_ext,
                        dMcat_ext, nullptr, dsex, scratch, scratch_bytes, stream, TOAD_X_PT, "toad_mil_bwd_xp_f32");
}
extern "C" int toad_mil_step_xp_f32(const float *const *params, float *const *grads, float beta, const void *Xp, const float *x_amax,
                                     const float *sex, const int64_t *label, const int64_t *site, float w_cls, float w_site, int64_t N,
                                     int C, int D, float drop_p, uint64_t seed, float *loss_out, float *logits_out,
                                     float *site_logits_out, void *ws, size_t ws_bytes, void **events, void *stream) {
    if (!x_amax) { set_error("toad_mil_step_xp_f32: null x_amax"); return TOAD_EINVAL; }
    return mil_step_impl(params, grads, beta, reinterpret_cast<const float *>(Xp), sex, label, site, w_cls, w_site, N, C, D, drop_p, seed, x_amax,
                         loss_out, logits_out, site_logits_out, ws, ws_bytes, events, stream, TOAD_X_PT, "toad_mil_step_xp_f32");
}

// ---- ragged multi-slide training step (ABI 9) ---------------------------------------------------------------------------
// The reference steps once per slide with batch size 1 (utils/utils.py:51-55, utils/core_utils_mtl_concat.py:200-234); its real bags
// are a few hundred to a few thousand patches, where one slide cannot fill 256 CUs: a 256-patch step is ~23 dependent launches
// of mostly idle kernels. Under slide-sharded data-parallel semantics (ONE optimiser step per batch of B slides, the gradient is
// the sum over the batch - toad_amd/dp.py, BASELINE config 4's contract) the rows of different slides never interact before the
// pooling, so the five trunk / attention GEMMs of the forward AND the backward run ONCE over the concatenated bags
// Xcat [sum N_b, 1024]; only the softmax pooling, the heads and the loss run per slide, on row ranges of the shared activations.
//   grads = beta * grads + sum_b d( w_cls * CE(logits_b, label_b) + w_site * CE(site_logits_b, site_b) ) / d params
// (the caller folds 1/B into w_cls, w_site). Per-slide results: loss_out [B][3], logits_out [B][C], site_logits_out [B][2].
// offsets: HOST array of B + 1 row offsets (offsets[0] = 0, offsets[B] = sum N_b); sex / label / site: DEVICE arrays of B.
// Differences from B calls of toad_mil_step_f32, all at fp32 round-off level: GEMM operand scales are taken per 256-row block of
// the CONCATENATION. Train-mode dropout masks hash the element index in the concatenation. (Until round 5 the pooling gradient dH_pool was
// materialised here; the attention dgrad now recomputes it from per-row records, gemm_h2_epilogue.inc PBATCH.)
namespace toad {
// Per-row records of the batched pooled addend (gemm_h2_epilogue.inc PBATCH): rec[row] = {w0, w1, slide index as a bit pattern, 0} with
// w_t = softmax weight of the row inside ITS slide = exp(A_raw[row, t] - max_t) / sum_t (models/model_toad.py:97). blockIdx.y = slide.
__global__ __launch_bounds__(256) void pool_rowrec_kernel(const float *__restrict__ A_raw, const float *__restrict__ stats, int s_stride,
                                                          const int64_t *__restrict__ seg, float *__restrict__ rec) {
    const int b = blockIdx.y;
    const int64_t r0 = seg[b], r1 = seg[b + 1];
    const float *st = stats + (int64_t)b * s_stride;
    const float m0 = st[0], i0 = 1.f / st[1], m1 = st[2], i1 = 1.f / st[3];
    for (int64_t r = r0 + (int64_t)blockIdx.x * 256 + threadIdx.x; r < r1; r += (int64_t)gridDim.x * 256) {
        const f32x2 a = *reinterpret_cast<const f32x2 *>(A_raw + 2 * r);
        f32x4 v;
        v[0] = __builtin_amdgcn_exp2f((a[0] - m0) * 1.4426950408889634f) * i0;
        v[1] = __builtin_amdgcn_exp2f((a[1] - m1) * 1.4426950408889634f) * i1;
        v[2] = __builtin_bit_cast(float, b);
        v[3] = 0.f;
        *reinterpret_cast<f32x4 *>(rec + 4 * r) = v;
    }
}

struct MultiSmall { float *stats, *M, *Mcat, *logits, *yprob, *slog, *sprob, *dM, *dl, *dsv; int64_t *yhat, *shat, *seg; void *pool_ws; size_t total; };
constexpr size_t kSlideRec = 8192;                  // bytes reserved per slide and per small array (>= T*(L+1)*4 = 4104, 256-B multiple)
static MultiSmall multi_small_layout(int B, char *base) {
    MultiSmall m{};
    Carver c;
    auto P = [&](size_t off) { return base ? base + off : nullptr; };
    const size_t n = (size_t)B * kSlideRec;
    m.stats = (float *)P(c.take(n, 256)); m.M = (float *)P(c.take(n, 256)); m.Mcat = (float *)P(c.take(n, 256));
    m.logits = (float *)P(c.take(n, 256)); m.yprob = (float *)P(c.take(n, 256)); m.slog = (float *)P(c.take(n, 256));
    m.sprob = (float *)P(c.take(n, 256)); m.dM = (float *)P(c.take(n, 256));
    m.yhat = (int64_t *)P(c.take(n, 256)); m.shat = (int64_t *)P(c.take(n, 256));
    m.dl = (float *)P(c.take(n, 256)); m.dsv = (float *)P(c.take(n, 256));
    m.seg = (int64_t *)P(c.take((size_t)(B + 1) * sizeof(int64_t), 256));
    m.pool_ws = P(c.take(pool_batch_ws_bytes(B, kL, 384, kT), 4096));
    m.total = up(c.off, 256);
    return m;
}
}  // namespace toad

extern "C" size_t toad_mil_multi_ws_bytes(int64_t Ntot, int B, int C, int D) {
    if (B <= 0 || B > 4096 || C > 512) return 0;
    const size_t a = toad_mil_step_ws_bytes(Ntot, C, D);
    return a ? a + multi_small_layout(B, nullptr).total + 4096 : 0;
}

extern "C" int toad_mil_multi_step_f32(const float *const *params, float *const *grads, float beta, const float *Xcat,
                                        const int64_t *offsets, int B, const float *sex, const int64_t *label, const int64_t *site,
                                        float w_cls, float w_site, int C, int D, float drop_p, uint64_t seed,
                                        float *loss_out, float *logits_out, float *site_logits_out, void *ws, size_t ws_bytes, void **events,
                                        void *stream) {
    const char *what = "toad_mil_multi_step_f32";
    if (!params || !grads || !Xcat || !offsets || !sex || !label || !site || !loss_out || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || B > 4096) { set_error("%s: B must be in [1, 4096]", what); return TOAD_EINVAL; }
    if (offsets[0] != 0) { set_error("%s: offsets[0] must be 0", what); return TOAD_EINVAL; }
    for (int b = 0; b < B; ++b) if (offsets[b + 1] <= offsets[b]) { set_error("%s: slide %d is empty or offsets decrease", what, b); return TOAD_EINVAL; }
    const int64_t N = offsets[B];
    const MilShape s{N, C, D};
    if (!shape_ok(s) || !h2_nt_ok(N, kL, kL0, kL0, kL)) { set_error("%s: unsupported shape sum N=%lld C=%d D=%d", what, (long long)N, C, D); return TOAD_ESHAPE; }
    if (!(drop_p >= 0.f && drop_p < 1.f)) { set_error("%s: drop_p must be in [0,1)", what); return TOAD_EINVAL; }
    if (!aligned16(Xcat)) { set_error("%s: Xcat must be 16-byte aligned", what); return TOAD_EALIGN; }
    if (ws_bytes < toad_mil_multi_ws_bytes(N, B, C, D)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    Params p;
    if (!load_params(params, p, what)) return TOAD_EINVAL;
    for (int i = 0; i < 12; ++i) if (!grads[i]) { set_error("%s: null gradient slot %d", what, i); return TOAD_EINVAL; }
    char *ab = align_base(ws, N);
    int64_t o[TOAD_MIL_ARENA_SLOTS];
    char *sb = align_base(ab + arena_layout(s, o), N);
    const Fwd f = arena_view(s, ab);
    const Scratch w = scratch_layout(s, sb);
    char *mb = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(sb + w.total) + 255) & ~(uintptr_t)255);
    const MultiSmall ms = multi_small_layout(B, mb);
    hipStream_t st = (hipStream_t)stream;
    const StreamEvents ev{events, st};
    const int D2 = 2 * D;
    const DropSeeds ds = drop_seeds(drop_p, seed);
    const EpiScalars relu1{1, 1.f, make_drop(drop_p, ds.s1)}, relu2{1, 1.f, make_drop(drop_p, ds.s2)}, lin{0, 1.f, make_drop(0.f, 0)};
    const EpiScalars msk{0, ds.mscale, make_drop(0.f, 0)};
    const H2Pool nopool{nullptr, nullptr, nullptr, 0};
    // the slides' row offsets go to the device FIRST: the source is the caller's pageable array, so the runtime stages the copy before it
    // returns and orders it behind whatever the stream already holds - in front of the GEMMs that wait is nothing, behind them it was the
    // three forward GEMMs on every call (the host lost its launch run-ahead)
    if (hipMemcpyAsync(ms.seg, offsets, (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
        set_error("%s: copying the slide offsets to the device failed: %s", what, hipGetErrorString(hipGetLastError()));
        return TOAD_EINVAL;
    }
    // ---- forward: one launch splits the five weight operands and zeroes both groups of abs-max arrays; three GEMMs over all rows
    {
        const H2Operand ops5[5] = {{p.w1, kL0, 1, kL, kL0, w.planes[W_1], w.binv[W_1]}, {p.w2, kL, 1, kL, kL, w.planes[W_2], w.binv[W_2]},
                                   {p.wab, kL, 1, D2, kL, w.planes[W_AB], w.binv[W_AB]}, {p.wab, 1, kL, kL, D2, w.planes[W_ABT], w.binv[W_ABT]},
                                   {p.w2, 1, kL, kL, kL, w.planes[W_2T], w.binv[W_2T]}};
        const int nz = (int)(((char *)f.amax_h - (char *)f.amax_x) / sizeof(float) + toad_amax_floats(N));
        const int nzb = (int)(((char *)w.amax_dZ1 - (char *)w.amax_dP) / sizeof(float) + toad_amax_floats(N));
        TOAD_TRY(launch_split_h2(ops5, 5, f.amax_x, nz, st, what, w.amax_dP, nzb));
    }
    // the concatenated bags are measured inside the first GEMM (no abs-max pass of its own), which fills f.amax_x for the weight gradient below
    const bool self_measure = nt_run_ok(N, kL, kL0);
    if (!self_measure) TOAD_TRY(launch_absmax(Xcat, kL0, N, kL0, f.amax_x, false, st, what));
    ev(2); TOAD_TRY(launch_nt_h2(Xcat, kL0, self_measure ? nullptr : f.amax_x, w.planes[W_1], w.binv[W_1], f.H1, kL, N, kL, kL0, p.b1, relu1, nullptr, nullptr, nullptr, nopool,
                                 w.slabs, f.amax_h1, f.bits_h1, st, what, TOAD_X_F32, 1, 1, self_measure ? f.amax_x : nullptr, self_measure ? w.slab_ke : nullptr)); ev(3);
    ev(4); TOAD_TRY(launch_nt_h2(f.H1, kL, f.amax_h1, w.planes[W_2], w.binv[W_2], f.H, kL, N, kL, kL, p.b2, relu2, nullptr, nullptr, nullptr, nopool, w.slabs, f.amax_h, f.bits_h, st, what)); ev(5);
    ev(6); TOAD_TRY(launch_nt_h2(f.H, kL, f.amax_h, w.planes[W_AB], w.binv[W_AB], f.P, D2, N, D2, kL, p.bab, lin, nullptr, nullptr, nullptr, nopool, w.slabs, nullptr, nullptr, st, what)); ev(7);
    // ---- all slides at once (blockIdx.y = slide): fused pool forward on each row range + merge; heads + weighted CE + heads backward with
    // one workgroup per slide, then the head-weight gradients summed over the batch; pooling backward (dP rows, the pooling gradient
    // dH_pool rows into the dZ2 buffer, dWc / dbc summed over the batch). Five launches + two small copies, whatever B is - a 64-slide
    // batch of 256-patch bags was 320 launches of mostly idle kernels when this ran slide by slide.
    int64_t max_n = 0;
    for (int b = 0; b < B; ++b) if (offsets[b + 1] - offsets[b] > max_n) max_n = offsets[b + 1] - offsets[b];
    const int rec_f = (int)(kSlideRec / sizeof(float));
    ev(0); TOAD_TRY(launch_pool_fwd_batch(f.P, f.P + D, D2, f.H, p.wc, p.bc, f.A_raw, ms.M, rec_f, ms.stats, rec_f, ms.pool_ws, ms.seg, B, max_n, kL, D, kT, drop_p,
                                   ds.sa, ds.sb, st)); ev(1);
    const HeadsBatch hb{ms.M, ms.Mcat, ms.logits, ms.yprob, ms.yhat, ms.slog, ms.sprob, ms.shat, ms.dM, ms.dl, ms.dsv, kSlideRec};
    TOAD_TRY(launch_heads_batch(hb, sex, p.wcls, p.bcls, p.wsite, p.bsite, label, site, w_cls, w_site, loss_out, grads[8], grads[9], grads[10], grads[11], beta,
                                B, kL, C, st));
    if (logits_out && hipMemcpy2DAsync(logits_out, (size_t)C * sizeof(float), ms.logits, kSlideRec, (size_t)C * sizeof(float), B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("%s: copying the per-slide logits failed: %s", what, hipGetErrorString(hipGetLastError()));
        return TOAD_EINVAL;
    }
    if (site_logits_out && hipMemcpy2DAsync(site_logits_out, 2 * sizeof(float), ms.slog, kSlideRec, 2 * sizeof(float), B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("%s: copying the per-slide site logits failed: %s", what, hipGetErrorString(hipGetLastError()));
        return TOAD_EINVAL;
    }
    // (round 5: no dH_pool output any more - the attention dgrad below recomputes the pooling gradient per row from the records written here, like the
    // one-slide step does from A_raw and its statistics: 410 MB less HBM traffic per 100k rows. The records live in dZ1's buffer behind the row bounds.)
    float *rowrec = w.dZ1 + ((N + 3) & ~(int64_t)3);
    TOAD_TRY(launch_pool_bwd_batch(f.P, f.P + D, D2, f.H, p.wc, f.A_raw, ms.stats, rec_f, ms.M, ms.dM, rec_f, w.dP, w.dP + D, D2, nullptr, grads[6], grads[7], beta,
                                   ms.pool_ws, ms.seg, B, max_n, kL, D, kT, drop_p, ds.sa, ds.sb, st, w.amax_dP, w.dZ1, N));
    {
        int64_t gx = (max_n + 255) / 256;
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(pool_rowrec_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, st, (const float *)f.A_raw, (const float *)ms.stats, rec_f,
                           (const int64_t *)ms.seg, rowrec);
        TOAD_TRY(check_launch(what));
    }
    // (the slides' row ranges do not line up with the 256-row blocks of the concatenation: the pool backward leaves one upper bound of |dP| per ROW -
    // in dZ1's buffer, which nobody has written yet - and its reduce kernel folds 256 of them into each slot of dP's abs-max array. Round 3 measured
    // the array in a pass of its own over dP: 307 MB read per 100k rows.)
    // ---- backward GEMMs over all rows
    WgradDeferred dw[3];
    const WgradJob wj[3] = {{w.dP, w.amax_dP, f.H, f.amax_h, grads[4], grads[5], D2, kL, w.wgrad_ws},
                            {w.dZ2, w.amax_dZ2, f.H1, f.amax_h1, grads[2], grads[3], kL, kL, w.wgrad_ws2},
                            {w.dZ1, w.amax_dZ1, Xcat, f.amax_x, grads[0], grads[1], kL, kL0, w.wgrad_ws3}};
    const bool wbatch = wgrad_batch_ok(N, wj, 3, w.wgrad_ws_bytes);          // a short batch: the three weight gradients as one launch at the end (backward_body)
    if (!wbatch) { ev(8); TOAD_TRY(launch_wgrad(w.dP, w.amax_dP, f.H, f.amax_h, grads[4], grads[5], N, D2, kL, beta, w.wgrad_ws, st, what, TOAD_X_F32, &dw[0])); ev(9); }
    // dZ2 = (dP Wab + dH_pool) * (H > 0): dH_pool[row] = sum_t w_t(row) dM_t[slide(row)] recomputed in the epilogue (batched pooled addend)
    ev(10); TOAD_TRY(launch_nt_h2(w.dP, D2, w.amax_dP, w.planes[W_ABT], w.binv[W_ABT], w.dZ2, kL, N, kL, D2, nullptr, msk, nullptr, f.H, f.bits_h,
                                  H2Pool{rowrec, nullptr, ms.dM, kT | (rec_f << 8)}, w.slabs, w.amax_dZ2, nullptr, st, what)); ev(11);
    if (!wbatch) { ev(12); TOAD_TRY(launch_wgrad(w.dZ2, w.amax_dZ2, f.H1, f.amax_h1, grads[2], grads[3], N, kL, kL, beta, w.wgrad_ws2, st, what, TOAD_X_F32, &dw[1])); ev(13); }
    ev(14); TOAD_TRY(launch_nt_h2(w.dZ2, kL, w.amax_dZ2, w.planes[W_2T], w.binv[W_2T], w.dZ1, kL, N, kL, kL, nullptr, msk, nullptr, f.H1, f.bits_h1, nopool, w.slabs,
                                  w.amax_dZ1, nullptr, st, what)); ev(15);
    if (wbatch) {
        ev(8); TOAD_TRY(launch_wgrad_batch(wj, 3, N, beta, st, what, dw));
        TOAD_TRY(launch_wgrad_reduce(dw, 3, st, what)); ev(9);
        ev(12); ev(13); ev(16); ev(17);
    } else {
        ev(16); TOAD_TRY(launch_wgrad(w.dZ1, w.amax_dZ1, Xcat, f.amax_x, grads[0], grads[1], N, kL, kL0, beta, w.wgrad_ws3, st, what, TOAD_X_F32, &dw[2]));
        TOAD_TRY(launch_wgrad_reduce(dw, 3, st, what)); ev(17);
    }
    return TOAD_OK;
}

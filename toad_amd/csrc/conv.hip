// conv.hip — the truncated ResNet-50 feature extractor (models/resnet_custom.py:19-119 of the reference) in
// inference form, as the producer of the [N,1024] bags the MIL path consumes (BASELINE config 5).
//
// Design for gfx950: activations live in HBM as NHWC fp32, so every convolution is the NT product the MIL trunk
// already runs — Y[M, Cout] = act(cols[M, K] . Wf[Cout, K]^T + bf (+ residual)) with M = B*Ho*Wo pixels — in the
// same fp16 two-piece MFMA arithmetic (gemm_f32.hip; fp32-accurate, so the extractor keeps the 1e-4 parity bar):
// Cout >= 256 on the persistent 256x256 kernel, narrower layers, short-K residual expansions, the stem and the 3x3
// convolutions on the kernels of gemm_stream.inc (A streamed through registers; stride-1 3x3 with the activation halo
// in LDS) - those never materialise cols. Batch-norm (eval form) is folded into Wf / bf on the host; ReLU and the
// bottleneck's residual add ride in the GEMM epilogue. 1x1 stride-1 convolutions need no data movement at all (cols ==
// the NHWC activation); the explicit gathers below (16-B lanes, coalesced on the channel dimension) remain for strided
// 1x1 convolutions, for small batches of the 256-channel 3x3, and as API entry points.
#include "common.h"

namespace toad {

// ---- gathers ---------------------------------------------------------------------------------------------------
// cols[m, (ky*kw + kx)*C + c] = X[b, oy*s - p + ky, ox*s - p + kx, c]  (0 outside), m = (b*Ho + oy)*Wo + ox.
// One thread per 16-B chunk of a cols row; C % 4 == 0 so a chunk never straddles a tap.
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float *__restrict__ X, float *__restrict__ cols, int B, int H,
                                                          int W, int C, int Ho, int Wo, int kh, int kw, int stride, int pad) {
    const uint32_t kq = (uint32_t)(kh * kw * C) >> 2, cq = (uint32_t)C >> 2;
    const uint64_t total = (uint64_t)B * Ho * Wo * kq;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t m = (uint32_t)(i / kq), j = (uint32_t)(i - (uint64_t)m * kq);
        const uint32_t tap = j / cq, c4 = j - tap * cq;
        const int ky = (int)(tap / (uint32_t)kw), kx = (int)(tap - (uint32_t)ky * kw);
        const uint32_t ox = m % (uint32_t)Wo, t = m / (uint32_t)Wo;
        const uint32_t oy = t % (uint32_t)Ho, b = t / (uint32_t)Ho;
        const int iy = (int)oy * stride - pad + ky, ix = (int)ox * stride - pad + kx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(X + (((uint64_t)b * H + iy) * W + ix) * C + 4 * c4);
        st4(cols + i * 4, v);
    }
}

// The stem: cols[m, c*49 + ky*7 + kx] (K = 147, zero-padded to Kp = 160) from NCHW tiles, 7x7 / stride 2 / pad 3.
// K keeps the reference's own weight order [Cout, Cin, kh, kw] (resnet_custom.py:62), so kx runs along W in memory.
__global__ __launch_bounds__(256) void im2col_stem_nchw_kernel(const float *__restrict__ X, float *__restrict__ cols, int B, int H,
                                                               int W, int Ho, int Wo) {
    constexpr uint32_t KQ = 40;        // 160 / 4
    const uint64_t total = (uint64_t)B * Ho * Wo * KQ;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t m = (uint32_t)(i / KQ), j = (uint32_t)(i - (uint64_t)m * KQ);
        const uint32_t ox = m % (uint32_t)Wo, t = m / (uint32_t)Wo;
        const uint32_t oy = t % (uint32_t)Ho, b = t / (uint32_t)Ho;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = 4 * j + e;
            float x = 0.f;
            if (k < 147) {
                const uint32_t c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
                const int iy = (int)oy * 2 - 3 + (int)ky, ix = (int)ox * 2 - 3 + (int)kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) x = X[(((uint64_t)b * 3 + c) * H + iy) * W + ix];
            }
            v[e] = x;
        }
        st4(cols + i * 4, v);
    }
}

// Space-to-depth image of the stem: Xs[b, Y, X, (ry*2 + rx)*3 + c] = x[b, c, 2Y + ry - 4, 2X + rx - 4] (0 outside), Y < Ho + 3,
// X < Wo + 3. With it the 7x7 / stride-2 / pad-3 convolution (resnet_custom.py:62) becomes a 4x4 / stride-1 one whose K row
// for output pixel (oy, ox) is 4 window rows of 48 CONTIGUOUS floats (Xs[b, oy+qy, ox..ox+3, :]), i.e. something the GEMM's
// LDS-DMA can gather by itself: tap ky = 2*qy + ry - 1, kx = 2*qx + rx - 1 (the -1 slots carry zero weights). One thread per
// (b, Y, X): 12 reads that are pairwise adjacent in NCHW memory, 48 contiguous bytes written.
// gmax (optional): receives max |X| by atomic max (zeroed by the caller) - the image is the tiles' values plus zeros, so this IS the abs-max of the
// stem GEMM's operand and the separate pass over the tiles (74 us per 512 tiles) is not needed.
__global__ __launch_bounds__(256) void stem_s2d_kernel(const float *__restrict__ X, float *__restrict__ Xs, int B, int H, int W, int Hs, int Ws, float *gmax) {
    const uint64_t total = (uint64_t)B * Hs * Ws;
    float mx = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t xs = (uint32_t)(i % (uint32_t)Ws), t = (uint32_t)(i / (uint32_t)Ws);
        const uint32_t ys = t % (uint32_t)Hs, b = t / (uint32_t)Hs;
        float v[12];
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            const int iy = 2 * (int)ys + ry - 4;
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
                const int ix = 2 * (int)xs + rx - 4;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[(ry * 2 + rx) * 3 + c] = ok ? X[(((uint64_t)b * 3 + c) * H + iy) * W + ix] : 0.f;
            }
        }
        float *dst = Xs + i * 12;
        st4(dst, f32x4{v[0], v[1], v[2], v[3]}); st4(dst + 4, f32x4{v[4], v[5], v[6], v[7]}); st4(dst + 8, f32x4{v[8], v[9], v[10], v[11]});
#pragma unroll
        for (int e = 0; e < 12; ++e) mx = __builtin_fmaxf(mx, __builtin_fabsf(v[e]));
    }
    if (gmax) {
        mx = h2_wave_max(mx);
        if ((threadIdx.x & 63) == 0 && mx > 0.f) h2_atomic_amax(gmax, mx);
    }
}

// nn.MaxPool2d(3, stride 2, padding 1) on NHWC (resnet_custom.py:66): padding never wins the max.
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float *__restrict__ X, float *__restrict__ Y, int B, int H, int W,
                                                                int C, int Ho, int Wo) {
    const uint32_t cq = (uint32_t)C >> 2;
    const uint64_t total = (uint64_t)B * Ho * Wo * cq;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t m = (uint32_t)(i / cq), c4 = (uint32_t)(i - (uint64_t)m * cq);
        const uint32_t ox = m % (uint32_t)Wo, t = m / (uint32_t)Wo;
        const uint32_t oy = t % (uint32_t)Ho, b = t / (uint32_t)Ho;
        const float ninf = -__builtin_huge_valf();
        f32x4 best = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = (int)oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = (int)ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = ld4(X + (((uint64_t)b * H + iy) * W + ix) * C + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) best[e] = v[e] > best[e] ? v[e] : best[e];
            }
        }
        st4(Y + i * 4, best);
    }
}

// nn.AdaptiveAvgPool2d(1) + flatten (resnet_custom.py:70,104-105): feat[b, c] = mean over the HW pixels of X[b, :, c].
// Block = (tile b, 256 channels): 4 pixel groups x 64 lanes x 16 B; fixed-order LDS combine -> deterministic.
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float *__restrict__ X, float *__restrict__ feat, int HW, int C) {
    __shared__ f32x4 part[4][64];
    const int b = blockIdx.y, c0 = blockIdx.x * 256;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const bool live = c0 + 4 * lane < C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float *p = X + (uint64_t)b * HW * C + c0 + 4 * lane;
        for (int px = grp; px < HW; px += 4) acc += ld4(p + (uint64_t)px * C);
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && live) {
        const f32x4 s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        const float inv = 1.f / (float)HW;
        st4(feat + (uint64_t)b * C + c0 + 4 * lane, s * inv);
    }
}

static int grid_for(uint64_t threads) {
    uint64_t g = (threads + 255) / 256;
    const uint64_t cap = 256 * 32;                     // 32 blocks per CU is plenty for a streaming gather
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static inline int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

// ---- the network plan (shared by the workspace query and the sequencer) -----------------------------------------
struct ConvSpec { int cin, cout, k, stride, pad; };
constexpr int kNumConvs = 43;
struct NetPlan {
    ConvSpec conv[kNumConvs];
    size_t act_max, cols_max, gemm_ws;    // floats, floats, bytes
    int Hs, Ws, Hp, Wp;                   // stem and max-pool output sizes
};
static const int kPlanes[3] = {64, 128, 256}, kBlocks[3] = {3, 4, 6}, kStride[3] = {1, 2, 2};

static bool make_plan(int B, int H, int W, NetPlan &p) {
    if (B <= 0 || H < 1 || W < 1) return false;
    int n = 0;
    p.conv[n++] = {3, 64, 7, 2, 3};
    p.Hs = conv_out(H, 7, 2, 3); p.Ws = conv_out(W, 7, 2, 3);
    p.Hp = conv_out(p.Hs, 3, 2, 1); p.Wp = conv_out(p.Ws, 3, 2, 1);
    if (p.Hs < 1 || p.Ws < 1 || p.Hp < 1 || p.Wp < 1) return false;
    size_t act = (size_t)B * p.Hs * p.Ws * 64, cols = (size_t)B * (p.Hs + 3) * (p.Ws + 3) * 12, gws = toad_linear_ws_bytes((int64_t)B * p.Hs * p.Ws, 64, 192);
    auto upd = [&](size_t M, int cout, int K, bool gathered) {
        if (M * cout > act) act = M * cout;
        if (gathered && M * K > cols) cols = M * K;
        const size_t w = toad_linear_ws_bytes((int64_t)M, cout, K);
        if (w > gws) gws = w;
    };
    int h = p.Hp, w = p.Wp, inpl = 64;
    for (int l = 0; l < 3; ++l)
        for (int b = 0; b < kBlocks[l]; ++b) {
            const int s = b == 0 ? kStride[l] : 1, pl = kPlanes[l];
            const int ho = conv_out(h, 3, s, 1), wo = conv_out(w, 3, s, 1);
            p.conv[n++] = {inpl, pl, 1, 1, 0};     upd((size_t)B * h * w, pl, inpl, false);
            p.conv[n++] = {pl, pl, 3, s, 1};       upd((size_t)B * ho * wo, pl, 9 * pl, true);
            p.conv[n++] = {pl, 4 * pl, 1, 1, 0};   upd((size_t)B * ho * wo, 4 * pl, pl, false);
            if (b == 0) { p.conv[n++] = {inpl, 4 * pl, 1, s, 0}; upd((size_t)B * ho * wo, 4 * pl, inpl, s != 1); }
            inpl = 4 * pl; h = ho; w = wo;
        }
    p.act_max = act; p.cols_max = cols; p.gemm_ws = gws;
    return n == kNumConvs;
}

static int implicit_conv_enabled() {
    return 1;
}

static inline size_t align2m(size_t x) { return (x + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1); }

}  // namespace toad

using namespace toad;

extern "C" int toad_im2col_nhwc_f32(const float *X, float *cols, int B, int H, int W, int C, int kh, int kw, int stride,
                                     int pad, void *stream) {
    const char *what = "toad_im2col_nhwc_f32";
    if (!X || !cols) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) { set_error("%s: bad shape (C must be a multiple of 4)", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(cols)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    if (Ho < 1 || Wo < 1 || (uint64_t)B * Ho * Wo >= (1ull << 32)) { set_error("%s: empty or oversized output", what); return TOAD_ESHAPE; }
    const uint64_t total = (uint64_t)B * Ho * Wo * (uint64_t)(kh * kw * C / 4);
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, X, cols, B, H, W, C, Ho, Wo, kh, kw, stride, pad);
    return check_launch(what);
}

extern "C" int toad_im2col_stem_nchw_f32(const float *X, float *cols, int B, int H, int W, void *stream) {
    const char *what = "toad_im2col_stem_nchw_f32";
    if (!X || !cols) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0) { set_error("%s: bad shape", what); return TOAD_ESHAPE; }
    if (!aligned16(cols)) { set_error("%s: cols must be 16-byte aligned", what); return TOAD_EALIGN; }
    const int Ho = conv_out(H, 7, 2, 3), Wo = conv_out(W, 7, 2, 3);
    if (Ho < 1 || Wo < 1 || (uint64_t)B * Ho * Wo >= (1ull << 32)) { set_error("%s: empty or oversized output", what); return TOAD_ESHAPE; }
    hipLaunchKernelGGL(im2col_stem_nchw_kernel, dim3(grid_for((uint64_t)B * Ho * Wo * 40)), dim3(256), 0, (hipStream_t)stream, X, cols, B, H, W, Ho, Wo);
    return check_launch(what);
}

extern "C" int toad_stem_s2d_nchw_f32(const float *X, float *Xs, int B, int H, int W, void *stream) {
    const char *what = "toad_stem_s2d_nchw_f32";
    if (!X || !Xs) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0) { set_error("%s: bad shape", what); return TOAD_ESHAPE; }
    if (!aligned16(Xs)) { set_error("%s: Xs must be 16-byte aligned", what); return TOAD_EALIGN; }
    const int Hs = conv_out(H, 7, 2, 3) + 3, Ws = conv_out(W, 7, 2, 3) + 3;
    hipLaunchKernelGGL(stem_s2d_kernel, dim3(grid_for((uint64_t)B * Hs * Ws)), dim3(256), 0, (hipStream_t)stream, X, Xs, B, H, W, Hs, Ws, (float *)nullptr);
    return check_launch(what);
}

extern "C" int toad_maxpool3x3s2_nhwc_f32(const float *X, float *Y, int B, int H, int W, int C, void *stream) {
    const char *what = "toad_maxpool3x3s2_nhwc_f32";
    if (!X || !Y) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) { set_error("%s: bad shape (C must be a multiple of 4)", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(Y)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    const int Ho = conv_out(H, 3, 2, 1), Wo = conv_out(W, 3, 2, 1);
    if ((uint64_t)B * Ho * Wo >= (1ull << 32)) { set_error("%s: oversized output", what); return TOAD_ESHAPE; }
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3(grid_for((uint64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, X, Y, B, H, W, C, Ho, Wo);
    return check_launch(what);
}

extern "C" int toad_avgpool_nhwc_f32(const float *X, float *feat, int B, int HW, int C, void *stream) {
    const char *what = "toad_avgpool_nhwc_f32";
    if (!X || !feat) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (B <= 0 || B > 65535 || HW <= 0 || C <= 0 || C % 4) { set_error("%s: bad shape", what); return TOAD_ESHAPE; }
    if (!aligned16(X) || !aligned16(feat)) { set_error("%s: pointers must be 16-byte aligned", what); return TOAD_EALIGN; }
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, X, feat, HW, C);
    return check_launch(what);
}

extern "C" size_t toad_resnet50_trunc_ws_bytes(int B, int H, int W) {
    NetPlan p;
    if (!make_plan(B, H, W, p)) return 0;
    return 4 * align2m(p.act_max * 4) + align2m(p.cols_max * 4) + align2m(p.gemm_ws) + align2m(4096) + ((size_t)1 << 21);
}

// weights[i] : folded conv i as [Cout, K] fp32 (K = kh*kw*Cin in (ky, kx, c) order; the stem is [64, 192] in the space-to-depth
//              order of toad_stem_conv_s2d_f32),
// biases[i]  : folded BN shift [Cout]; i runs in execution order (conv1, then per block conv1, conv2, conv3[, downsample]).
extern "C" int toad_resnet50_trunc_fwd_f32(const float *tiles_nchw, const float *const *weights, const float *const *biases,
                                            float *feat, int B, int H, int W, void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_resnet50_trunc_fwd_f32";
    if (!tiles_nchw || !weights || !biases || !feat || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    NetPlan p;
    if (!make_plan(B, H, W, p)) { set_error("%s: bad shape B=%d H=%d W=%d", what, B, H, W); return TOAD_ESHAPE; }
    if (ws_bytes < toad_resnet50_trunc_ws_bytes(B, H, W)) { set_error("%s: workspace too small", what); return TOAD_EWORKSPACE; }
    if (!aligned16(ws) || !aligned16(feat)) { set_error("%s: workspace / output must be 16-byte aligned", what); return TOAD_EALIGN; }
    for (int i = 0; i < kNumConvs; ++i) if (!weights[i] || !biases[i]) { set_error("%s: null weight/bias slot %d", what, i); return TOAD_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>(ws);
    size_t off = align2m(reinterpret_cast<uintptr_t>(ws)) - reinterpret_cast<uintptr_t>(ws);
    auto take = [&](size_t bytes) { char *q = base + off; off += align2m(bytes); return q; };
    float *act[4];
    for (int i = 0; i < 4; ++i) act[i] = reinterpret_cast<float *>(take(p.act_max * 4));
    float *cols = reinterpret_cast<float *>(take(p.cols_max * 4));
    void *gws = take(p.gemm_ws);
    const size_t gcap = p.gemm_ws;
    // Tensor-wide abs-max scalars, one per activation the network produces (round 3): every GEMM epilogue maxes |output| into the slot of
    // its output, the consumer derives ONE power-of-two scale for its fp16 two-piece A operand from it (gemm_narrow.inc). Pools and the
    // im2col / space-to-depth gathers only move values (plus zeros), so their outputs share the producer's slot. All slots are zeroed
    // by one memset; nothing in the call measures an activation a second time.
    float *gm = reinterpret_cast<float *>(take(4096));
    (void)hipMemsetAsync(gm, 0, 4096, st);
    int gi = 0;
    auto slot = [&]() { return gm + (gi++); };
    int rc;
#define TOAD_TRY(call) do { rc = (call); if (rc) return rc; } while (0)
    // stem: 7x7/2 conv + BN + ReLU (resnet_custom.py:96-98), 3x3/2 max-pool (:99)
    float *g_in = slot();
    float *gx = slot();
    if (stem_nchw_pool_ok(H, W) && aligned16(tiles_nchw)) {
        // 256-wide tiles: stem + ReLU + max-pool as ONE kernel reading the NCHW tiles themselves (stem_halo.inc); per-tile operand scales, so not even
        // max |tiles| is measured
        TOAD_TRY(ext_stem_nchw_pool(tiles_nchw, weights[0], biases[0], act[1], gx, B, H, W, gws, gcap, st, what));
    } else {
    // 12-channel space-to-depth image (53 MB per 64 tiles); the gather also emits max |tiles| - the only measured tensor - into its slot
    if (!aligned16(cols)) { set_error("%s: internal: cols not aligned", what); return TOAD_EALIGN; }
    hipLaunchKernelGGL(stem_s2d_kernel, dim3(grid_for((uint64_t)B * (p.Hs + 3) * (p.Ws + 3))), dim3(256), 0, st, tiles_nchw, cols, B, H, W, p.Hs + 3, p.Ws + 3, g_in);
    TOAD_TRY(check_launch(what));
    if (stem_pool_ok(p.Hs, p.Ws, TOAD_ACT_RELU)) {      // tiles 256 wide: the max-pool rides in the stem's epilogue, the stem's own output is never stored
        TOAD_TRY(ext_stem_conv(cols, g_in, weights[0], biases[0], act[1], gx, B, p.Hs, p.Ws, TOAD_ACT_RELU, gws, gcap, st, what, true));
    } else {
        TOAD_TRY(ext_stem_conv(cols, g_in, weights[0], biases[0], act[0], gx, B, p.Hs, p.Ws, TOAD_ACT_RELU, gws, gcap, st, what));
        TOAD_TRY(toad_maxpool3x3s2_nhwc_f32(act[0], act[1], B, p.Hs, p.Ws, 64, st));      // max-pooling keeps the maximum: same slot
    }
    }
    float *x = act[1];                          // block input (abs-max scalar: gx)
    auto other = [&](float *a0, float *a1, float *a2) {          // a buffer different from the (up to) three in use
        for (int i = 0; i < 4; ++i) if (act[i] != a0 && act[i] != a1 && act[i] != a2) return act[i];
        return (float *)nullptr;
    };
    int h = p.Hp, w = p.Wp, inpl = 64, ci = 1;
    for (int l = 0; l < 3; ++l)
        for (int b = 0; b < kBlocks[l]; ++b) {
            const int s = b == 0 ? kStride[l] : 1, pl = kPlanes[l];
            const int ho = conv_out(h, 3, s, 1), wo = conv_out(w, 3, s, 1);
            const int64_t Mi = (int64_t)B * h * w, Mo = (int64_t)B * ho * wo;
            float *t1 = other(x, nullptr, nullptr);
            // conv1 1x1 + BN + ReLU (:38-40): cols == x
            float *g1 = slot();
            TOAD_TRY(ext_linear(x, gx, weights[ci], biases[ci], nullptr, t1, g1, Mi, inpl, pl, TOAD_ACT_RELU, gws, gcap, st, what));
            // conv2 3x3 stride s + BN + ReLU (:42-44)
            float *t2 = other(x, t1, nullptr);
            // implicit convolution (halo in LDS for stride 1, streamed taps for stride 2), no cols buffer: always for the narrow
            // layers; for 256 output channels (two 128-column tiles per 256 pixels) only when there are enough tiles to fill
            // the chip's 512 workgroup slots - otherwise im2col + the 256x256 kernel with its K-split is faster (measured at B = 64 vs 512)
            const bool implicit = implicit_conv_enabled() && (pl <= 128 || (pl <= 256 && (Mo / 256) * ((pl + 127) / 128) >= 512));
            float *g2 = slot();
            if (implicit) {
                TOAD_TRY(ext_conv_nhwc(t1, g1, weights[ci + 1], biases[ci + 1], nullptr, t2, g2, B, h, w, pl, 3, 3, s, 1, pl, TOAD_ACT_RELU, gws, gcap, st, what));
            } else {
                TOAD_TRY(toad_im2col_nhwc_f32(t1, cols, B, h, w, pl, 3, 3, s, 1, st));
                TOAD_TRY(ext_linear(cols, g1, weights[ci + 1], biases[ci + 1], nullptr, t2, g2, Mo, 9 * pl, pl, TOAD_ACT_RELU, gws, gcap, st, what));
            }
            // residual: identity, or downsample = strided 1x1 conv + BN (:49-50, :79-85)
            const float *res = x;
            float *rbuf = nullptr;
            if (b == 0) {
                rbuf = t1;                       // t1 is dead once conv2 has consumed it
                const float *src = x;
                if (s != 1) { TOAD_TRY(toad_im2col_nhwc_f32(x, cols, B, h, w, inpl, 1, 1, s, 0, st)); src = cols; }
                TOAD_TRY(ext_linear(src, gx, weights[ci + 3], biases[ci + 3], nullptr, rbuf, nullptr, Mo, inpl, 4 * pl, TOAD_ACT_NONE, gws, gcap, st, what));
                res = rbuf;                      // (only ever an epilogue addend: nobody needs its abs-max)
            }
            // conv3 1x1 + BN, + residual, ReLU (:46-47, :52-53) in one epilogue
            float *y = other(x, t2, rbuf);
            float *gy = slot();
            TOAD_TRY(ext_linear(t2, g2, weights[ci + 2], biases[ci + 2], res, y, gy, Mo, pl, 4 * pl, TOAD_ACT_RELU, gws, gcap, st, what));
            ci += b == 0 ? 4 : 3;
            x = y; gx = gy; inpl = 4 * pl; h = ho; w = wo;
        }
#undef TOAD_TRY
    return toad_avgpool_nhwc_f32(x, feat, B, h * w, inpl, st);
}

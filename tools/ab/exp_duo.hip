// exp_duo.hip — experiment translation unit: the shipped GEMM sources + the duo kernel (gemm_h2_duo.inc) + one entry point with the
// signature of toad_linear_act_fwd_f32 / toad_linear_dgrad_f32's h2 paths, so that tools/ab/duo_bench.py can run both kernels on the
// same operands. Built into tools/ab/libtoad_duo.so by tools/ab/build_duo.py; never part of libtoad_hip.so.
#include "../../toad_amd/csrc/gemm_f32.hip"

namespace toad {
#include "gemm_h2_duo.inc"
}

using namespace toad;

// Y = act(X W^T + b) (mask_bits == NULL) or dX = (dY W^T [+ pooling addend]) * mask (mask_bits != NULL) on the duo kernel.
// ws as for toad_linear_act_fwd_f32; x_amax may be NULL (measured here).
extern "C" int toad_exp_nt_duo_f32(const float *X, const float *W, int64_t wsn, int64_t wsk, const float *bias, float *Y, int64_t M, int64_t K, int64_t N,
                                   int act, const float *x_amax, float *y_amax, uint64_t *relu_bits_out, const float *relu_src,
                                   const uint64_t *relu_bits_in, const float *pool_a_raw, const float *pool_stats, const float *pool_dM, int pool_T,
                                   void *ws, size_t ws_bytes, void *stream, long long *trace) {
    const char *what = "toad_exp_nt_duo_f32";
    hipStream_t st = (hipStream_t)stream;
    if (!X || !W || !Y || !ws) { set_error("%s: null pointer", what); return TOAD_EINVAL; }
    if (ws_bytes < toad_linear_ws_bytes(M, N, K) || !h2_nt_ok(M, N, K, K, N)) { set_error("%s: workspace / shape", what); return TOAD_ESHAPE; }
    char *w = reinterpret_cast<char *>(ws);
    w += (size_t)PB_GRID * PB * PB * sizeof(float);
    unsigned short *planes = reinterpret_cast<unsigned short *>(w);
    w += h2_planes_bytes(N, K);
    float *binv = reinterpret_cast<float *>(w);
    w += h2_binv_bytes(N);
    float *amax_ws = reinterpret_cast<float *>(w);
    if (y_amax) (void)hipMemsetAsync(y_amax, 0, (size_t)h2_nblk(M) * sizeof(float), st);
    if (!x_amax) { if (int rc = launch_absmax(X, K, M, K, amax_ws, true, st, what)) return rc; x_amax = amax_ws; }
    const H2Operand op{W, wsn, wsk, N, K, planes, binv};          // B[n, k] = W[n * wsn + k * wsk] (forward: K, 1; dgrad: 1, ld of W)
    if (int rc = launch_split_h2(&op, 1, nullptr, 0, st, what)) return rc;
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(0.f, 0)};
    return launch_nt_h2_duo(X, K, x_amax, planes, binv, Y, N, M, N, K, bias, es, nullptr, relu_src,
                            reinterpret_cast<const unsigned long long *>(relu_bits_in), H2Pool{pool_a_raw, pool_stats, pool_dM, pool_T}, y_amax,
                            reinterpret_cast<unsigned long long *>(relu_bits_out), st, what, trace);
}

// The extractor's wide GEMM (ext_linear's "big" branch: tensor-wide scalar scales, optional residual) on the shipped 8-wave kernel (which = 0)
// or on the duo kernel (which = 1). x_gmax: device scalar max |X| (given); y_gmax: device scalar receiving max |Y| (zeroed by the caller) or NULL.
extern "C" int toad_exp_ext_f32(int which, const float *X, const float *x_gmax, const float *W, const float *bias, const float *residual, float *Y, float *y_gmax,
                                int64_t M, int64_t K, int64_t N, int act, void *ws, size_t ws_bytes, void *stream) {
    const char *what = "toad_exp_ext_f32";
    hipStream_t st = (hipStream_t)stream;
    if (ws_bytes < toad_linear_ws_bytes(M, N, K) || !h2_nt_ok(M, N, K, K, N)) { set_error("%s: workspace / shape", what); return TOAD_ESHAPE; }
    char *w = reinterpret_cast<char *>(ws);
    float *slabs = reinterpret_cast<float *>(w);
    w += (size_t)PB_GRID * PB * PB * sizeof(float);
    unsigned short *planes = reinterpret_cast<unsigned short *>(w);
    w += h2_planes_bytes(N, K);
    float *binv = reinterpret_cast<float *>(w);
    const H2Operand op{W, K, 1, N, K, planes, binv};
    if (int rc = launch_split_h2(&op, 1, nullptr, 0, st, what)) return rc;
    EpiScalars es{act == TOAD_ACT_RELU, 1.f, make_drop(0.f, 0)};
    const H2Pool nopool{nullptr, nullptr, nullptr, 0};
    if (which >= 2 || which <= -2) es.stagger = which < 0 ? -which : which;      // |which| >= 2: a staggered start over that many 10 ns ticks; negative: on the duo kernel
    if (which == 0 || which >= 2)
        return launch_nt_h2(X, K, x_gmax, planes, binv, Y, N, M, N, K, bias, es, residual, nullptr, nullptr, nopool, slabs, y_gmax, nullptr, st, what, TOAD_X_F32, 0, 0);
    return launch_nt_h2_duo(X, K, x_gmax, planes, binv, Y, N, M, N, K, bias, es, residual, nullptr, nullptr, nopool, y_gmax, nullptr, st, what, nullptr, 0, 0);
}

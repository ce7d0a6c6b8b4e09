/*
 * toad_hip.h — C ABI of libtoad_hip.so: the MI355X (gfx950) kernels behind TOAD's
 * gated-attention MIL hot path (reference: models/model_toad.py, mahmoodlab/TOAD).
 *
 * The reference has no native code and no FFI of its own: every op on this path is a
 * stock PyTorch op called from Python.  Each entry point below therefore cites the
 * reference *Python call site(s)* it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - All tensors are dense row-major fp32 in device memory, 16-byte aligned.
 *    Weights are [out_features, in_features] exactly as nn.Linear stores them.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no
 *    allocation, no host synchronisation and keeps no global mutable state (re-entrant;
 *    callable from PyTorch's autograd thread).  Workspaces are caller-owned; their sizes
 *    come from the *_ws_bytes() queries.
 *  - Return value: 0 on success; a negative TOAD_E* code for argument errors; a positive
 *    hipError_t for launch failures.  toad_last_error() returns a thread-local message.
 *  - `beta` arguments: out = beta*out + result (beta = 0 overwrites and never reads out).
 */
#ifndef TOAD_HIP_H
#define TOAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOAD_ABI_VERSION 6

enum { TOAD_OK = 0, TOAD_EINVAL = -1, TOAD_ESHAPE = -2, TOAD_EWORKSPACE = -3, TOAD_EALIGN = -4 };
enum { TOAD_ACT_NONE = 0, TOAD_ACT_RELU = 1 };

int toad_abi_version(void);
const char *toad_last_error(void);

/* ---- Linear layers (exact-fp32 MFMA GEMMs) ------------------------------------------ */

/* Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]).   bias may be NULL.
 * Replaces nn.Linear(+nn.ReLU): models/model_toad.py:59 and :62 (trunk, act=RELU) and the
 * attention_a / attention_b pre-activations :21,:25 (act=NONE, W = [Wa;Wb] stacked).
 * drop_p > 0 applies train-mode nn.Dropout(drop_p) after the activation (models/model_toad.py:61,64):
 * element (row, col) is kept iff hash(drop_seed, row*N+col) >= drop_p*2^32 and then scaled by
 * 1/(1-drop_p); the mask is never stored (toad_dropout_mask_f32 reproduces it).
 * Requires K % 4 == 0. `ws` (toad_linear_ws_bytes, also used by toad_linear_dgrad_f32) holds the
 * fp32 slabs of K-split remainder tiles of the persistent 256x256 kernel; with ws == NULL, or
 * K % 32 != 0, the generic 128x128 kernel runs instead. */
size_t toad_linear_ws_bytes(int64_t M, int64_t N, int64_t K);
int toad_linear_act_fwd_f32(const float *X, const float *W, const float *bias, float *Y,
                            int64_t M, int64_t K, int64_t N, int act,
                            float drop_p, uint64_t drop_seed,
                            void *ws, size_t ws_bytes, void *stream);

/* dX[M,K] = (dY[M,N] W[N,K] + addend[M,K]) * (relu_src[M,K] > 0) * mask_scale
 * mask_scale = 1, or 1/(1-p) when relu_src is a ReLU+Dropout(p) output (its zeros already encode the
 * dropout mask).  `WT` is W transposed, [K,N] row-major (see toad_transpose_f32).  addend and relu_src may be
 * NULL (no add / no mask); dX may alias addend.
 * Replaces autograd's mm backward + threshold_backward behind loss.backward()
 * (utils/core_utils_mtl_concat.py:231) for models/model_toad.py:62 and :21,:25.
 * Requires N % 4 == 0. */
int toad_linear_dgrad_f32(const float *dY, const float *WT, const float *addend,
                          const float *relu_src, float mask_scale, float *dX,
                          int64_t M, int64_t N, int64_t K,
                          void *ws, size_t ws_bytes, void *stream);

/* dW[N,K] = beta*dW + dY[M,N]^T X[M,K];  db[N] = beta*db + column sums of dY (db may be NULL).
 * Split over M with a deterministic two-stage reduction through `ws`.
 * Replaces autograd's weight/bias gradient for every nn.Linear on the path
 * (models/model_toad.py:59,62,21,25 via utils/core_utils_mtl_concat.py:231).
 * Requires N % 4 == 0 and K % 4 == 0. */
size_t toad_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K);
int toad_linear_wgrad_f32(const float *dY, const float *X, float *dW, float *db,
                          int64_t M, int64_t N, int64_t K, float beta,
                          void *ws, size_t ws_bytes, void *stream);

/* out[e] = the dropout multiplier (0 or 1/(1-p)) the kernels apply to flat element e under `drop_seed`
 * (all ones when drop_p == 0). Lets a caller or test reproduce the masks, which are never stored. */
int toad_dropout_mask_f32(float *out, int64_t n, float drop_p, uint64_t drop_seed, void *stream);

/* out[cols,rows] = in[rows,cols]^T  (weight transposes for dgrad). */
int toad_transpose_f32(const float *in, float *out, int64_t rows, int64_t cols, void *stream);

/* ---- Fused gated-attention pooling --------------------------------------------------- */

/* One pass over the bag:
 *   g[i,:]   = tanh(Pa[i,:]) * sigmoid(Pb[i,:])            models/model_toad.py:37-39
 *   A_raw[i,t] = g[i,:] . Wc[t,:] + bc[t]                   models/model_toad.py:40
 *   M[t,:]   = sum_i softmax_i(A_raw[:,t])[i] * H[i,:]      models/model_toad.py:92,97-98
 * Pa/Pb are the pre-activation rows (row stride ldp floats; Pb = Pa + D when both halves
 * come from one stacked GEMM).  H may be NULL together with M and stats: then only A_raw
 * is produced (the attention_only path, models/model_toad.py:93-94).
 * Outputs: A_raw[N,T] (row-major; the reference's `A` is its transpose view),
 *          M[T,L], stats[T,2] = (max_i A_raw[i,t], sum_i exp(A_raw[i,t]-max)) for backward.
 * Supported shapes: T in {1,2}; D in {256,384}; L in {512,1024}; N >= 1. */
size_t toad_gated_pool_ws_bytes(int64_t N, int L, int D, int T);
int toad_gated_pool_fwd_f32(const float *Pa, const float *Pb, int64_t ldp, const float *H,
                            const float *Wc, const float *bc,
                            float *A_raw, float *M, float *stats,
                            void *ws, size_t ws_bytes,
                            int64_t N, int L, int D, int T,
                            float drop_p, uint64_t seed_a, uint64_t seed_b, void *stream);
/* drop_p > 0: train-mode Dropout(drop_p) on tanh(Pa) (stream seed_a) and on sigmoid(Pb) (seed_b),
 * models/model_toad.py:27-29; element index = row*D + d. The backward takes the same seeds. */

/* Backward of the above (autograd mirror, utils/core_utils_mtl_concat.py:231):
 *   p[i,t]  = exp(A_raw[i,t]-max_t)/sum_t
 *   dS[i,t] = p[i,t]*(dM[t,:].H[i,:] - dM[t,:].M[t,:]) + dA_ext[i,t]   (dA_ext may be NULL)
 *   dH[i,:] = sum_t p[i,t]*dM[t,:]
 *   dPa = (dS Wc) * b*(1-a^2),  dPb = (dS Wc) * a*b*(1-b)   with a=tanh(Pa), b=sigmoid(Pb)
 *   dWc = beta*dWc + dS^T g,  dbc = beta*dbc + column sums of dS
 * dPa/dPb have row stride ldd floats. */
size_t toad_gated_pool_bwd_ws_bytes(int64_t N, int L, int D, int T);
int toad_gated_pool_bwd_f32(const float *Pa, const float *Pb, int64_t ldp, const float *H,
                            const float *Wc, const float *A_raw, const float *stats,
                            const float *M, const float *dM, const float *dA_ext,
                            float *dPa, float *dPb, int64_t ldd, float *dH,
                            float *dWc, float *dbc, float beta,
                            void *ws, size_t ws_bytes,
                            int64_t N, int L, int D, int T,
                            float drop_p, uint64_t seed_a, uint64_t seed_b, void *stream);

/* ---- Classifier heads ---------------------------------------------------------------- */

/* models/model_toad.py:99-107:
 *   Mcat[t,:] = [M[t,:], sex];  logits = Mcat[0] Wcls^T + bcls;  site_logits = Mcat[1] Wsite^T + bsite
 *   Y_prob/site_prob = softmax;  Y_hat/site_hat = argmax (first maximal index, as torch.topk).
 * Wcls [C,L+1], Wsite [2,L+1]; sex points at ONE device float. C <= 1024. */
int toad_heads_fwd_f32(const float *M, const float *sex,
                       const float *Wcls, const float *bcls, const float *Wsite, const float *bsite,
                       float *Mcat, float *logits, float *Y_prob, int64_t *Y_hat,
                       float *site_logits, float *site_prob, int64_t *site_hat,
                       int L, int C, void *stream);

/* Backward of the heads: dWcls = beta*dWcls + dlogits^T Mcat[0], dbcls, dWsite, dbsite likewise;
 * dM[t,:] = (d{logits,site}[.] W{cls,site})[:L] + dMcat_ext[t,:L]  (dMcat_ext [2,L+1] may be NULL). */
int toad_heads_bwd_f32(const float *Mcat, const float *dlogits, const float *dsite,
                       const float *Wcls, const float *Wsite, const float *dMcat_ext,
                       float *dWcls, float *dbcls, float *dWsite, float *dbsite, float *dM,
                       float beta, int L, int C, void *stream);

/* Fused caller-side loss (utils/core_utils_mtl_concat.py:213-215) and its gradient:
 *   loss = w_cls*CE(logits,label) + w_site*CE(site_logits,site);  dlogits, dsite = d loss/d logits.
 * label/site point at ONE device int64 each. loss_out[3] = (loss, cls_loss, site_loss). */
int toad_mtl_ce_fwd_bwd_f32(const float *logits, const float *site_logits,
                            const int64_t *label, const int64_t *site,
                            float w_cls, float w_site,
                            float *loss_out, float *dlogits, float *dsite,
                            int C, void *stream);

/* Adam over a flat fp32 buffer (n % 4 == 0), identical update to torch.optim.Adam(lr, betas, eps, weight_decay)
 * as built by the reference's get_optim (utils/utils.py:63-70); `step` counts from 1. One launch. */
int toad_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t step, void *stream);

/* ---- Feature extractor: truncated ResNet-50 (models/resnet_custom.py) -------------------- */
/* Inference form of the reference's `resnet50_baseline` (models/resnet_custom.py:111-119): the producer of the
 * [N,1024] bags. Activations are NHWC fp32 in HBM, so each convolution is Y[M,Cout] = act(cols[M,K] Wf[Cout,K]^T + bf
 * (+ residual)) on the same MFMA GEMM as the MIL trunk, with eval-mode BatchNorm folded into Wf / bf by the host. */

/* Y[M,N] = act(X[M,K] W[N,K]^T + bias[N] + residual[M,N]);  bias / residual may be NULL.
 * = conv (as GEMM) + folded BN [+ `out += residual`] + ReLU: Bottleneck_Baseline.forward, resnet_custom.py:38-53. */
int toad_linear_act_res_fwd_f32(const float *X, const float *W, const float *bias, const float *residual, float *Y,
                                int64_t M, int64_t K, int64_t N, int act,
                                void *ws, size_t ws_bytes, void *stream);

/* Implicit-GEMM convolution on an NHWC activation, no im2col buffer:
 *   Y[b,oy,ox,:] = act(sum_{ky,kx,c} X[b, oy*stride-pad+ky, ox*stride-pad+kx, c] * Wf[:, (ky*kw+kx)*Cin + c] + bias (+ residual))
 * = nn.Conv2d(Cin, Cout, (kh,kw), stride, pad, bias=False) + folded BN [+ skip] + ReLU (resnet_custom.py:26-27,42-44).
 * The gather happens inside the GEMM's LDS-DMA (per k-stage tap offset, zero fill outside the image).
 * Needs Cin % 32 == 0 and Cout <= 512 (the narrow-tile kernels, 64 / 128 output columns per tile: every further 128 columns
 * gather and split the activation again, so wide layers are usually faster through toad_im2col_nhwc_f32 + the GEMM). */
int toad_conv_nhwc_f32(const float *X, const float *Wf, const float *bias, const float *residual, float *Y,
                       int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, int act,
                       void *ws, size_t ws_bytes, void *stream);

/* cols[m, (ky*kw+kx)*C + c] = X[b, oy*stride-pad+ky, ox*stride-pad+kx, c] (0 outside), m = (b*Ho+oy)*Wo+ox,
 * Ho = (H+2*pad-kh)/stride+1: the gather that turns nn.Conv2d(C, ., (kh,kw), stride, pad) on an NHWC activation into
 * the GEMM above (3x3 convs :26-27, strided 1x1 downsample :81-82). C % 4 == 0. */
int toad_im2col_nhwc_f32(const float *X, float *cols, int B, int H, int W, int C,
                         int kh, int kw, int stride, int pad, void *stream);

/* The stem's gather, straight from the caller's NCHW tiles [B,3,H,W] (what the reference model is fed):
 * cols[m, c*49+ky*7+kx] for nn.Conv2d(3,64,7,stride 2,pad 3) (:62), K = 147 zero-padded to 160 columns. */
int toad_im2col_stem_nchw_f32(const float *X, float *cols, int B, int H, int W, void *stream);

/* The stem without a cols buffer. toad_stem_s2d_nchw_f32 writes the space-to-depth image
 *   Xs[b, Y, X, (ry*2+rx)*3 + c] = x[b, c, 2Y+ry-4, 2X+rx-4] (0 outside),  Y < Ho+3, X < Wo+3,  Ho = (H-1)/2+1, Wo = (W-1)/2+1
 * and toad_stem_conv_s2d_f32 computes nn.Conv2d(3,64,7,stride 2,pad 3) + folded BN (+ReLU) (:62-64,:96-98) from it as a 4x4/1
 * convolution gathered inside the GEMM:  Y[b,oy,ox,:] = act(sum_{qy,qx<4; j<12} Xs[b,oy+qy,ox+qx,j] * Wf[:, qy*48+qx*12+j] + bias),
 * Wf[:, qy*48 + qx*12 + (ry*2+rx)*3 + c] = w[:, c, 2qy+ry-1, 2qx+rx-1] (0 where an index is -1): [64,192]. Y is NHWC [B,Ho,Wo,64]. */
int toad_stem_s2d_nchw_f32(const float *X, float *Xs, int B, int H, int W, void *stream);
int toad_stem_conv_s2d_f32(const float *Xs, const float *Wf, const float *bias, float *Y, int B, int Ho, int Wo, int act,
                           void *ws, size_t ws_bytes, void *stream);

/* nn.MaxPool2d(kernel 3, stride 2, padding 1) (:66) on NHWC; C % 4 == 0. Y is [B, Ho, Wo, C]. */
int toad_maxpool3x3s2_nhwc_f32(const float *X, float *Y, int B, int H, int W, int C, void *stream);

/* nn.AdaptiveAvgPool2d(1) + view(B,-1) (:70,:104-105): feat[b,c] = mean_p X[b,p,c], X = [B, HW, C]. Deterministic. */
int toad_avgpool_nhwc_f32(const float *X, float *feat, int B, int HW, int C, void *stream);

/* ResNet_Baseline.forward (:95-108) for layers [3,4,6]: tiles [B,3,H,W] NCHW fp32 -> feat [B,1024], one call, no host
 * round trips. weights[43] / biases[43]: BN-folded convolutions in execution order (conv1; per block conv1, conv2,
 * conv3 and, for the first block of a layer, downsample), each [Cout, K] with K = kh*kw*Cin in (ky,kx,c) order - the
 * stem as the [64,192] space-to-depth operand of toad_stem_conv_s2d_f32. `ws` from toad_resnet50_trunc_ws_bytes (0 = unsupported shape). */
size_t toad_resnet50_trunc_ws_bytes(int B, int H, int W);
int toad_resnet50_trunc_fwd_f32(const float *tiles_nchw, const float *const *weights, const float *const *biases,
                                float *feat, int B, int H, int W, void *ws, size_t ws_bytes, void *stream);

/* ---- Whole per-slide training step ------------------------------------------------------ */

/* One call = model(data, sex) + weighted CE + loss.backward() of the reference train loop
 * (utils/core_utils_mtl_concat.py:206,213-215,231) for TOAD_fc_mtl_concat(size_arg="big"), sequenced in
 * C++ over a caller-owned arena (toad_mil_step_ws_bytes) with the kernels above, in the same order as the
 * per-op path: no host round trips or allocations between launches.
 *   params / grads : 12 device pointers each, slots w1 b1 w2 b2 wab bab wc bc wcls bcls wsite bsite
 *                    (wab = [Wa;Wb] stacked [2D,512], bab = [ba;bb]); grads = beta*grads + d loss/d param.
 *   sex / label / site : one device float / int64 / int64.  loss_out[3] = (loss, cls CE, site CE).
 *   logits_out [C], site_logits_out [2] : optional copies of the logits.
 *   drop_p, seed : train-mode Dropout(drop_p) masks (0 = off), four streams derived from `seed`.
 *   events : NULL, or 18 hipEvent_t recorded around the fused pool forward ([0],[1]) and the eight GEMM
 *            calls ([2+2i],[3+2i]) - used by bench.py for its roofline figures. */
size_t toad_mil_step_ws_bytes(int64_t N, int C, int D);
int toad_mil_step_f32(const float *const *params, float *const *grads, float beta, const float *X,
                      const float *sex, const int64_t *label, const int64_t *site,
                      float w_cls, float w_site, int64_t N, int C, int D,
                      float drop_p, uint64_t seed,
                      float *loss_out, float *logits_out, float *site_logits_out,
                      void *ws, size_t ws_bytes, void **events, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TOAD_HIP_H */

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

#define TOAD_ABI_VERSION 13

enum { TOAD_OK = 0, TOAD_EINVAL = -1, TOAD_ESHAPE = -2, TOAD_EWORKSPACE = -3, TOAD_EALIGN = -4 };
enum { TOAD_ACT_NONE = 0, TOAD_ACT_RELU = 1 };

int toad_abi_version(void);
const char *toad_last_error(void);
/* ABI 13 (diagnostic): launches of the exact-fp32 fallback GEMM kernels (gemm_nt_f32_kernel / gemm_tn_f32_kernel) since the library was loaded.
 * Those kernels serve raw C-ABI callers whose operands the fp16 two-piece kernels cannot take as they are (a reduction that is not a
 * multiple of 32, an output width that is not a multiple of 4, no workspace, an operand of 2^32 bytes or more). The Python host layer
 * (toad_amd/ops.py) zero-pads / chunks such operands instead - exact, the padded products are zero - so that every reference-legal
 * shape (models/model_toad.py:19: Attn_Net_Gated takes any L, D) runs on ONE arithmetic; tests assert this counter stays put. A
 * monotonic process-wide counter: the one piece of global state in the library, never read on any dispatch path. */
int64_t toad_fallback_launches(void);

/* ---- Linear layers (fp32-accurate MFMA GEMMs) ---------------------------------------- */

/* Arithmetic. Products run on the fp16 matrix pipe with every fp32 operand element carried as TWO fp16 pieces,
 * x*s = h + m (s a power of two), and three MFMA terms per product (h.h + h.m + m.h, fp32 accumulation): results are
 * as close to the exact value as an fp32 fma chain (csrc/gemm_h2.inc, tools/split_emulation.py). The power-of-two
 * scales come from ABS-MAX ARRAYS: amax[b] = max |X[r,:]| over rows r of the b-th block of 256 rows,
 * toad_amax_floats(rows) floats per tensor. Every GEMM entry point
 *   - accepts the array of its activation operand(s) (`*_amax` inputs; NULL = measured inside the call, one extra pass), and
 *   - can emit the array of its output (`y_amax` / `dx_amax` outputs; NULL = not wanted) from its epilogue, so a chain of
 *     layers never re-reads a tensor just to measure it. toad_absmax_rows256_f32 measures a tensor that comes from outside. */
size_t toad_amax_floats(int64_t rows);
int toad_absmax_rows256_f32(const float *X, int64_t M, int64_t K, float *amax, void *stream);
/* 1 when the persistent fp16 two-piece kernel serves an [M,K] x [N,K]^T product (K % 32 == 0, N % 4 == 0, M*K*4 < 2^32);
 * other shapes run on the older exact-fp32 kernels (same results to fp32 round-off). */
int toad_linear_h2_ok(int64_t M, int64_t N, int64_t K);
/* One-bit image of a ReLU output Y[M,N] (= the mask of its backward): 8 KB per 256 x 256 tile, written by the forward GEMM's
 * epilogue (relu_bits_out) and read back by the dgrad of the same layer (relu_bits) with the identical tile / lane mapping, so
 * the dgrad epilogue reads 1/32 of the bytes an fp32 relu_src costs. Only whole tiles of the persistent kernel use it; the
 * caller still passes relu_src (remainder tiles, other kernels). Both calls must see the same M and the same N (= dgrad's K). */
size_t toad_relu_bits_bytes(int64_t M, int64_t N);

/* Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]).   bias may be NULL.
 * Replaces nn.Linear(+nn.ReLU): models/model_toad.py:59 and :62 (trunk, act=RELU) and the
 * attention_a / attention_b pre-activations :21,:25 (act=NONE, W = [Wa;Wb] stacked).
 * drop_p > 0 applies train-mode nn.Dropout(drop_p) after the activation (models/model_toad.py:61,64):
 * element (row, col) is kept iff hash(drop_seed, row*N+col) >= drop_p*2^32 and then scaled by
 * 1/(1-drop_p); the mask is never stored (toad_dropout_mask_f32 reproduces it).
 * Requires K % 4 == 0. `ws` (toad_linear_ws_bytes, also used by toad_linear_dgrad_f32) holds the fp32 slabs of
 * K-split remainder tiles of the persistent 256x256 kernel, the split weight planes and, when x_amax == NULL, the
 * measured abs-max array; with ws == NULL, or K % 32 != 0, the generic 128x128 kernel runs instead.
 * x_amax: abs-max array of X or NULL.  y_amax: receives the abs-max array of Y, or NULL.
 *   x_amax == NULL (ABI 10): there is no separate pass over X. The persistent kernel measures X while it converts it (a work item's scale
 *   comes from its first 32 columns with 3 bits of head-room; a running maximum decides at the item's end whether the room sufficed, items
 *   where it did not are repeated with the exact scale). Results equal the x_amax route to <= 5e-6 of max |Y| (both are exact products
 *   under different power-of-two operand scales); the measured array is left in `ws` for the call's own K-split fix-up.
 * relu_bits_out (act = RELU, toad_linear_h2_ok shapes): receives the one-bit image of Y (toad_relu_bits_bytes), or NULL. */
size_t toad_linear_ws_bytes(int64_t M, int64_t N, int64_t K);
int toad_linear_act_fwd_f32(const float *X, const float *W, const float *bias, float *Y,
                            int64_t M, int64_t K, int64_t N, int act,
                            float drop_p, uint64_t drop_seed,
                            const float *x_amax, float *y_amax, uint64_t *relu_bits_out,
                            void *ws, size_t ws_bytes, void *stream);

/* dX[M,K] = (dY[M,N] W[N,K] + addend[M,K] + pool[M,K]) * (relu_src[M,K] > 0) * mask_scale
 * mask_scale = 1, or 1/(1-p) when relu_src is a ReLU+Dropout(p) output (its zeros already encode the
 * dropout mask).  `WT` is W transposed, [K,N] row-major (see toad_transpose_f32).  addend and relu_src may be
 * NULL (no add / no mask); dX may alias addend.
 * pool (pool_T in {1,2}; 0 = none): the gradient of the attention pooling w.r.t. its input rows,
 *   pool[r,c] = sum_t softmax_r(A_raw[:,t])[r] * dM[t,c]   (models/model_toad.py:97-98 backward),
 * recomputed in the epilogue from pool_a_raw [M,pool_T], pool_stats [pool_T,2] (max, sum; toad_gated_pool_fwd_f32) and
 * pool_dM [pool_T,K] instead of being read from an [M,K] buffer (toad_gated_pool_bwd_f32 then runs with dH == NULL).
 * Needs toad_linear_h2_ok(M, K, N) and a workspace.
 * Replaces autograd's mm backward + threshold_backward behind loss.backward()
 * (utils/core_utils_mtl_concat.py:231) for models/model_toad.py:62 and :21,:25.
 * Requires N % 4 == 0.  dy_amax: abs-max array of dY or NULL.  dx_amax: receives the abs-max array of dX, or NULL.
 * relu_bits: the one-bit image of relu_src written by the forward of this layer, or NULL. */
int toad_linear_dgrad_f32(const float *dY, const float *WT, const float *addend,
                          const float *relu_src, float mask_scale, float *dX,
                          int64_t M, int64_t N, int64_t K,
                          const float *pool_a_raw, const float *pool_stats, const float *pool_dM, int pool_T,
                          const float *dy_amax, float *dx_amax, const uint64_t *relu_bits,
                          void *ws, size_t ws_bytes, void *stream);

/* dW[N,K] = beta*dW + dY[M,N]^T X[M,K];  db[N] = beta*db + column sums of dY (db may be NULL).
 * Split over M with a deterministic two-stage reduction through `ws`.
 * Replaces autograd's weight/bias gradient for every nn.Linear on the path
 * (models/model_toad.py:59,62,21,25 via utils/core_utils_mtl_concat.py:231).
 * Requires N % 4 == 0 and K % 4 == 0.  dy_amax / x_amax: abs-max arrays of the operands or NULL. */
size_t toad_linear_wgrad_ws_bytes(int64_t M, int64_t N, int64_t K);
int toad_linear_wgrad_f32(const float *dY, const float *X, float *dW, float *db,
                          int64_t M, int64_t N, int64_t K, float beta,
                          const float *dy_amax, const float *x_amax,
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
 *   dH[i,:] = sum_t p[i,t]*dM[t,:]      (dH == NULL: not written - toad_linear_dgrad_f32 recomputes it in its epilogue)
 *   dPa = (dS Wc) * b*(1-a^2),  dPb = (dS Wc) * a*b*(1-b)   with a=tanh(Pa), b=sigmoid(Pb)
 *   dWc = beta*dWc + dS^T g,  dbc = beta*dbc + column sums of dS
 * dPa/dPb have row stride ldd floats. dp_amax (or NULL): receives an abs-max array for the dP rows: per 256-row block an UPPER
 * BOUND of max |dP| (|dS| . max|Wc| per row - all a consumer needs to pick its power-of-two operand scale), not the exact maximum. */
size_t toad_gated_pool_bwd_ws_bytes(int64_t N, int L, int D, int T);
int toad_gated_pool_bwd_f32(const float *Pa, const float *Pb, int64_t ldp, const float *H,
                            const float *Wc, const float *A_raw, const float *stats,
                            const float *M, const float *dM, const float *dA_ext,
                            float *dPa, float *dPb, int64_t ldd, float *dH,
                            float *dWc, float *dbc, float beta, float *dp_amax,
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
 * dM[t,:] = (d{logits,site}[.] W{cls,site})[:L] + dMcat_ext[t,:L]  (dMcat_ext [2,L+1] may be NULL);
 * dsex (one float, or NULL) = gradient of the `sex` scalar that models/model_toad.py:99 appends to BOTH pooled rows:
 *   sum_c dlogits[c] Wcls[c,L] + sum_c dsite[c] Wsite[c,L] + dMcat_ext[0,L] + dMcat_ext[1,L]. */
int toad_heads_bwd_f32(const float *Mcat, const float *dlogits, const float *dsite,
                       const float *Wcls, const float *Wsite, const float *dMcat_ext,
                       float *dWcls, float *dbcls, float *dWsite, float *dbsite, float *dM, float *dsex,
                       float beta, int L, int C, void *stream);

/* toad_heads_fwd_f32 + toad_mtl_ce_fwd_bwd_f32 + toad_heads_bwd_f32 in ONE single-workgroup launch (bitwise the results of
 * the three calls): the tail of a training step (models/model_toad.py:99-107 + utils/core_utils_mtl_concat.py:213-215,231).
 * dlogits / dsite may be NULL (not exported). */
int toad_heads_ce_fused_f32(const float *M, const float *sex,
                            const float *Wcls, const float *bcls, const float *Wsite, const float *bsite,
                            const int64_t *label, const int64_t *site, float w_cls, float w_site,
                            float *Mcat, float *logits, float *Y_prob, int64_t *Y_hat,
                            float *site_logits, float *site_prob, int64_t *site_hat,
                            float *loss_out, float *dlogits, float *dsite,
                            float *dWcls, float *dbcls, float *dWsite, float *dbsite, float *dM,
                            float beta, int L, int C, void *stream);

/* Fused caller-side loss (utils/core_utils_mtl_concat.py:213-215) and its gradient:
 *   loss = w_cls*CE(logits,label) + w_site*CE(site_logits,site);  dlogits, dsite = d loss/d logits.
 * label/site point at ONE device int64 each. loss_out[3] = (loss, cls_loss, site_loss).
 * A label outside [0, C) (torch's CrossEntropyLoss raises) poisons loss and gradient with NaN instead of reading out of bounds. */
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

/* SGD over a flat fp32 buffer (n % 4 == 0), identical update to torch.optim.SGD(lr, momentum, weight_decay) as built by
 * get_optim's SGD branch (utils/utils.py:66-67: momentum 0.9): g' = g + wd*p; buf = momentum*buf + g' (buf = g' at step 1);
 * p -= lr*buf. momentum_buf may be NULL when momentum == 0. `step` counts from 1. One launch. */
int toad_sgd_step_f32(float *p, const float *g, float *momentum_buf, int64_t n,
                      float lr, float momentum, float weight_decay, int64_t step, void *stream);

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
/* The stem + folded BN + ReLU AND the 3x3/2 max-pool that follows it (models/resnet_custom.py:96-99) as ONE kernel: the pool is formed in the
 * GEMM's epilogue, the stem's own output is never stored. Yp is NHWC [B, Ho/2, Wo/2, 64], bit-identical to toad_maxpool3x3s2_nhwc_f32 of
 * toad_stem_conv_s2d_f32(.., TOAD_ACT_RELU). Shapes: Wo == 128 (tiles 256 wide) and Ho even; TOAD_ESHAPE otherwise. */
int toad_stem_conv_pool_s2d_f32(const float *Xs, const float *Wf, const float *bias, float *Yp, int B, int Ho, int Wo,
                                void *ws, size_t ws_bytes, void *stream);
/* The same result straight from the NCHW tiles X [B,3,H,W] (no space-to-depth image): the window of a tile of two conv rows is loaded into LDS in
 * whole image rows, converted once, and every MFMA operand comes from there. Wf as for toad_stem_conv_s2d_f32. Shapes: W == 256, H % 4 == 0;
 * TOAD_ESHAPE otherwise. Values agree with the two routes above to fp32 round-off (one power-of-two operand scale per tile instead of per call). */
int toad_stem_pool_nchw_f32(const float *X, const float *Wf, const float *bias, float *Yp, int B, int H, int W,
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

/* ---- Whole-slide calls: forward, backward, training step -------------------------------- */

/* The reference drives this path through three Python statements,
 *     results = model(data, sex)      utils/core_utils_mtl_concat.py:206  (also :284,:393, eval_utils_mtl_concat.py:91)
 *     loss.backward()                 :231
 * and the model's forward is models/model_toad.py:90-116. The three entry points below run those statements for
 * TOAD_fc_mtl_concat(size_arg="big" | "small") as ONE library call each, sequencing the kernels above in C++ (no host
 * round trips, no allocations, every weight operand split by one launch, abs-max arrays handed from producer to consumer).
 *
 *   params / grads : 12 device pointers each, slots w1 b1 w2 b2 wab bab wc bc wcls bcls wsite bsite
 *                    (wab = [Wa;Wb] stacked [2D,512], bab = [ba;bb]); grads = beta*grads + d loss/d param.
 *   D in {256, 384}; X [N,1024] fp32, N >= 1 (an empty bag has no kernels to run: handle it in the host). Any N up to 2^31 - 4096: a bag
 *                    beyond the NT kernels' 32-bit row offsets (N > 1,048,575) runs the same kernels over row chunks of 1,047,552 rows
 *                    (csrc/step.hip nt_rows); fp16 / prepared bags keep the single-launch limit (toad_mil_x16_ok).
 *   drop_p, seed   : train-mode Dropout(drop_p) masks (0 = off), four streams derived from `seed`. In a row-chunked call the trunk masks of
 *                    chunk j > 0 hash the chunk-local element index under seed_stream + j * 0xD1B54A32D192ED03 (toad_dropout_mask_f32
 *                    reproduces them chunk by chunk); bags of up to 1,047,552 patches are one chunk and unaffected.
 *   x_amax         : abs-max array of X (toad_absmax_rows256_f32) or NULL = measured inside the call - by the first GEMM itself while it
 *                    converts the bag (no extra pass over X, see toad_linear_act_fwd_f32); arena slot 13 then receives the measured array.
 *
 * Memory. `arena` (toad_mil_arena_bytes) receives everything the backward needs and everything the caller reads:
 * toad_mil_arena_layout() returns the byte offset of each tensor in it, in this order (TOAD_MIL_ARENA_SLOTS entries):
 *   0 H1 [N,512]  1 H [N,512]  2 P [N,2D]  3 A_raw [N,2]  4 stats [2,2]  5 M [2,512]  6 Mcat [2,513]
 *   7 logits [C]  8 Y_prob [C]  9 Y_hat (int64)  10 site_logits [2]  11 site_prob [2]  12 site_hat (int64)
 *   13 x_amax  14 h1_amax  15 h_amax   (abs-max arrays, toad_amax_floats(N) floats each)
 *   16 h1_bits  17 h_bits              (one-bit ReLU images, toad_relu_bits_bytes(N, 512) bytes each)
 * `scratch` (toad_mil_scratch_bytes) is temporary (GEMM slabs, weight planes, gradients of activations): it can be one
 * buffer reused by every call on a stream. */
#define TOAD_MIL_ARENA_SLOTS 18
size_t toad_mil_buffer_align(int64_t N);   /* offsets are relative to `arena` rounded up to this power of two (the byte counts include the slack) */
size_t toad_mil_arena_bytes(int64_t N, int C, int D);
int toad_mil_arena_layout(int64_t N, int C, int D, int64_t *offsets /* [TOAD_MIL_ARENA_SLOTS] */);
size_t toad_mil_scratch_bytes(int64_t N, int C, int D);

/* Forward: models/model_toad.py:90-116. attention_only != 0 stops after A_raw (:93-94; M, heads not computed). */
int toad_mil_fwd_f32(const float *const *params, const float *X, const float *sex, int64_t N, int C, int D,
                     float drop_p, uint64_t seed, const float *x_amax, int attention_only,
                     void *arena, size_t arena_bytes, void *scratch, size_t scratch_bytes, void *stream);

/* Backward of toad_mil_fwd_f32 for the same (params, X, arena, drop_p, seed): given dlogits [C] and dsite [2]
 * (d loss / d logits, from the caller's loss) and optionally dA_ext [N,2] (gradient arriving through results['A']) and
 * dMcat_ext [2,513] (through results['features']), accumulates all parameter gradients; dX [N,1024] and dsex [1] are
 * written when non-NULL. */
int toad_mil_bwd_f32(const float *const *params, float *const *grads, float beta, const float *X, int64_t N, int C, int D,
                     float drop_p, uint64_t seed, const void *arena, size_t arena_bytes,
                     const float *dlogits, const float *dsite, const float *dA_ext, const float *dMcat_ext,
                     float *dX, float *dsex, void *scratch, size_t scratch_bytes, void *stream);

/* One call = model(data, sex) + weighted CE + loss.backward() of the reference train loop
 * (utils/core_utils_mtl_concat.py:206,213-215,231): toad_mil_fwd_f32, the fused heads/CE tail and toad_mil_bwd_f32 over one
 * workspace (toad_mil_step_ws_bytes = arena + scratch).
 *   sex / label / site : one device float / int64 / int64.  loss_out[3] = (loss, cls CE, site CE).
 *   logits_out [C], site_logits_out [2] : optional copies of the logits.
 *   events : NULL, or 18 hipEvent_t recorded around the fused pool forward ([0],[1]) and the eight GEMM
 *            calls ([2+2i],[3+2i]) - used by bench.py for its roofline figures. */
size_t toad_mil_step_ws_bytes(int64_t N, int C, int D);
int toad_mil_step_f32(const float *const *params, float *const *grads, float beta, const float *X,
                      const float *sex, const int64_t *label, const int64_t *site,
                      float w_cls, float w_site, int64_t N, int C, int D,
                      float drop_p, uint64_t seed, const float *x_amax,
                      float *loss_out, float *logits_out, float *site_logits_out,
                      void *ws, size_t ws_bytes, void **events, void *stream);

/* ---- fp16 feature bags (ABI 8) -------------------------------------------------------------------------------------------
 * The same three calls for a bag stored as fp16, X16 [N,1024] halves (16-byte aligned): the reference's data path upcasts whatever the
 * .pt file holds when it reaches nn.Linear (datasets/dataset_mtl_concat.py:358-373 -> models/model_toad.py:91); feature stores kept in
 * fp16 halve the PCIe / disk traffic that bounds streaming training. An fp16 element is exactly a first piece of
 * the fp16 two-piece arithmetic (h = x, m = 0, scale 1), so the first Linear and its weight gradient run with TWO MFMA terms per
 * product instead of three and read half the bytes; the results equal those of the fp32 calls on the up-cast bag (same products,
 * same accumulation order). No abs-max array of X is needed; dX is not available (the bag is data, not a parameter).
 * Needs the fp16 two-piece kernels for both products (toad_mil_x16_ok(N): 64 <= N, N * 4096 < 2^32); TOAD_ESHAPE otherwise. */
int toad_mil_x16_ok(int64_t N);
int toad_mil_fwd_x16_f32(const float *const *params, const void *X16, const float *sex, int64_t N, int C, int D,
                         float drop_p, uint64_t seed, int attention_only,
                         void *arena, size_t arena_bytes, void *scratch, size_t scratch_bytes, void *stream);
int toad_mil_bwd_x16_f32(const float *const *params, float *const *grads, float beta, const void *X16, int64_t N, int C, int D,
                         float drop_p, uint64_t seed, const void *arena, size_t arena_bytes,
                         const float *dlogits, const float *dsite, const float *dA_ext, const float *dMcat_ext,
                         float *dsex, void *scratch, size_t scratch_bytes, void *stream);
int toad_mil_step_x16_f32(const float *const *params, float *const *grads, float beta, const void *X16,
                          const float *sex, const int64_t *label, const int64_t *site,
                          float w_cls, float w_site, int64_t N, int C, int D,
                          float drop_p, uint64_t seed,
                          float *loss_out, float *logits_out, float *site_logits_out,
                          void *ws, size_t ws_bytes, void **events, void *stream);

/* ---- prepared bags (ABI 9) -------------------------------------------------------------------------------------------------
 * A slide's bag is an INPUT, constant across epochs (datasets/dataset_mtl_concat.py:369-373 loads the same .pt file every time
 * the slide comes up), and the two products that read it - the first Linear (models/model_toad.py:59) and its weight gradient
 * (utils/core_utils_mtl_concat.py:231) - are a third of a step's flops. toad_bag_prepare_f32 converts the fp32 bag X [N,K] ONCE,
 * at ingest, into the form those products consume directly: its two fp16 pieces (x * 2^k = h + m, one exponent k per block of 256
 * rows) stored plane-tiled in the GEMMs' LDS stage order (csrc/gemm_pt.inc). Same 4 bytes per element as fp32 (the caller may then
 * drop the fp32 copy), no per-step abs-max pass over the bag. Agreement with the fp32 calls: BITWISE when the fp32 call is given the
 * same abs-max array (x_amax != NULL: the same pieces, products and accumulation order); since ABI 10 an fp32 call with x_amax == NULL
 * scales a raw bag inside the first GEMM from each tile's first 32 columns with 3 bits of head-room (gemm_h2.inc AMODE 3), so the two
 * routes then differ in the operand exponent only: <= 5e-6 of each result tensor's scale (tests/test_gpu_pt.py ROUTE_TOL). Either route
 * is bitwise deterministic run to run.
 *   planes : toad_bag_planes_bytes(N, K) bytes, 16-byte aligned; amax : toad_amax_floats(N) floats (the bag's abs-max array).
 * The *_xp_* calls are toad_mil_{fwd,bwd,step}_f32 with (Xp, x_amax) in place of X; dX is not available (the bag is data). */
size_t toad_bag_planes_bytes(int64_t N, int64_t K);
int toad_bag_prepare_f32(const float *X, int64_t N, int64_t K, void *planes, float *amax, void *stream);
/* toad_linear_wgrad_f32 with the layer input given as a prepared (plane-tiled) operand Xp [M,K] + its abs-max array (required). */
int toad_linear_wgrad_xp_f32(const float *dY, const void *Xp, const float *x_amax, float *dW, float *db, int64_t M, int64_t N,
                             int64_t K, float beta, const float *dy_amax, void *ws, size_t ws_bytes, void *stream);
int toad_mil_fwd_xp_f32(const float *const *params, const void *Xp, const float *x_amax, const float *sex, int64_t N, int C, int D,
                        float drop_p, uint64_t seed, int attention_only,
                        void *arena, size_t arena_bytes, void *scratch, size_t scratch_bytes, void *stream);
int toad_mil_bwd_xp_f32(const float *const *params, float *const *grads, float beta, const void *Xp, const float *x_amax,
                        int64_t N, int C, int D, float drop_p, uint64_t seed, const void *arena, size_t arena_bytes,
                        const float *dlogits, const float *dsite, const float *dA_ext, const float *dMcat_ext,
                        float *dsex, void *scratch, size_t scratch_bytes, void *stream);
int toad_mil_step_xp_f32(const float *const *params, float *const *grads, float beta, const void *Xp, const float *x_amax,
                         const float *sex, const int64_t *label, const int64_t *site,
                         float w_cls, float w_site, int64_t N, int C, int D,
                         float drop_p, uint64_t seed,
                         float *loss_out, float *logits_out, float *site_logits_out,
                         void *ws, size_t ws_bytes, void **events, void *stream);

/* ---- ragged multi-slide training step (ABI 9; `events` since ABI 10) -----------------------------------------------------------
 * One call = forward + weighted CE + backward for a BATCH of B slides whose bags lie concatenated in Xcat [sum N_b, 1024]
 * (reference loop body: utils/core_utils_mtl_concat.py:200-234, one slide per iteration; data-parallel semantics: one optimiser
 * step per batch, toad_amd/dp.py). The five trunk / attention GEMMs of the forward and of the backward run ONCE over all rows;
 * pooling, heads and loss run per slide on row ranges. grads = beta*grads + sum_b d loss_b (fold 1/B into w_cls / w_site).
 *   offsets : HOST array [B+1] of row offsets (offsets[0] = 0, strictly increasing); sex / label / site : DEVICE arrays [B];
 *   loss_out [B][3] (weighted loss, cls CE, site CE); logits_out [B][C], site_logits_out [B][2] (either may be NULL);
 *   ws >= toad_mil_multi_ws_bytes(sum N_b, B, C, D). Agrees with B calls of toad_mil_step_f32 to fp32 round-off (operand scales
 *   are taken per 256-row block of the concatenation), not bitwise.
 *   events : NULL, or 18 hipEvent_t laid out as for toad_mil_step_f32 ([0,1] bracket the batched pool forward + merge, [2+2i, 3+2i] GEMM
 *   call i of fwd1, fwd2, fwd_ab, wgrad_ab, dgrad_ab, wgrad_2, dgrad_2, wgrad_1) - bench.py's roofline figures of the batched configurations. */
size_t toad_mil_multi_ws_bytes(int64_t Ntot, int B, int C, int D);
int toad_mil_multi_step_f32(const float *const *params, float *const *grads, float beta, const float *Xcat,
                            const int64_t *offsets, int B, const float *sex, const int64_t *label, const int64_t *site,
                            float w_cls, float w_site, int C, int D, float drop_p, uint64_t seed,
                            float *loss_out, float *logits_out, float *site_logits_out, void *ws, size_t ws_bytes, void **events,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TOAD_HIP_H */

/* pda_train.h -- C ABI of the "next rows" around the sampling/grouping path:
 * the train-step arithmetic and the target-assignment / post-processing kernels.  Same library
 * (libpda_pointnet2.so), same conventions as pda_pointnet2.h: device pointers, explicit stream,
 * status return + pda_last_error(), nothing allocated, no torch types.
 */
#ifndef PDA_TRAIN_H
#define PDA_TRAIN_H
#include "pda_pointnet2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- adam_onecycle step (tools/train_utils/optimization/fastai_optim.py:138-156 OptimWrapper.step
 * with true_wd=True, bn_wd=True, wrapping torch.optim.Adam(betas=(mom, 0.99)); preceded by
 * clip_grad_norm_ in tools/train_utils/train_utils.py:56) on FLAT fp32 buffers: all trained
 * parameters, their gradients and both Adam moments each live in one contiguous allocation.
 *
 * pda_grad_norm: norm_out[0] = sqrt(sum g^2) in a fixed two-stage order (deterministic);
 *   scratch holds 1024 floats.
 * pda_adam_onecycle_step, per element:
 *   g' = g * min(1, max_norm / (total_norm + 1e-6))         (total_norm == NULL: no clipping)
 *   p  = p * (1 - wd * lr)                                   (decoupled decay, before Adam)
 *   m += (g' - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g' * g'
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` is the 1-based Adam step count.  g is not modified. */
int pda_grad_norm(const float *g, int64_t n, float *norm_out, float *scratch1024, pda_stream_t stream);
int pda_adam_onecycle_step(float *p, const float *g, float *m, float *v, int64_t n, float lr,
                           float beta1, float beta2, float eps, float wd, int step,
                           const float *total_norm, float max_norm, pda_stream_t stream);

/* ---- training-mode BatchNorm + ReLU over the last dimension (MI355X extension) ---------------------
 * The point-major form of the reference's Conv1x1 -> BatchNorm -> ReLU stacks (pointnet2_modules.py:
 * 1605-1611, :628-671): x (rows, C) -> y = relu((x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c) with
 * batch statistics over the rows (biased variance), running statistics updated as nn.BatchNorm does
 * (momentum, unbiased variance; pass NULL for both to skip).  mean_invstd (2, C) is written for the
 * backward pass.  Backward: grad_x (rows, C), grad_gamma (C), grad_beta (C) are fully written.
 * C: power of two in [4, 1024]; anything else returns PDA_ERR_INVALID_ARGUMENT (callers keep the
 * framework's batch_norm + relu).  scratch: pda_bn_relu_scratch_bytes(C) bytes, 16-byte aligned. */
int64_t pda_bn_relu_scratch_bytes(int c);
int pda_bn_relu_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                    float *running_var, float *y, float *mean_invstd, void *scratch, int64_t rows, int c,
                    float eps, float momentum, pda_stream_t stream);
int pda_bn_relu_bwd(const float *x, const float *grad_y, const float *gamma, const float *beta,
                    const float *mean_invstd, float *grad_x, float *grad_gamma, float *grad_beta,
                    void *scratch, int64_t rows, int c, pda_stream_t stream);
/* Dense-bf16 mode variants (same kernels and fp32 / double arithmetic): x may be the bf16 output of a GEMM, y may be
 * written as bf16 (when it only feeds the next GEMM), grad_y may be bf16; grad_x has the element type of x. */
int pda_bn_relu_fwd_mixed(const void *x, int x_is_bf16, const float *gamma, const float *beta, float *running_mean,
                          float *running_var, void *y, int y_is_bf16, float *mean_invstd, void *scratch,
                          int64_t rows, int c, float eps, float momentum, pda_stream_t stream);
int pda_bn_relu_bwd_mixed(const void *x, int x_is_bf16, const void *grad_y, int grad_y_is_bf16, const float *gamma,
                          const float *beta, const float *mean_invstd, void *grad_x, float *grad_gamma,
                          float *grad_beta, void *scratch, int64_t rows, int c, pda_stream_t stream);
/* The last [BN -> ReLU] of a set-abstraction MLP together with the max-pool over the group's samples that follows it
 * (pointnet2_modules.py:1657-1670): x (groups*ns, C), row = group*ns + sample (fp32, or bf16 when x_is_bf16) ->
 * out (groups, C) = max over the ns rows of relu(bn(x)), arg (groups, C) uint8 = first row attaining it; the (rows, C)
 * activation is never written.  Backward takes grad_out (groups, C) and arg: the dense (rows, C) gradient is generated on
 * the fly inside the two BN-backward passes, never stored.  grad_x has the element type of x.  ns <= 255. */
int pda_bn_relu_max_pool_fwd(const void *x, int x_is_bf16, const float *gamma, const float *beta, float *running_mean,
                             float *running_var, float *out, uint8_t *arg, float *mean_invstd, void *scratch,
                             int64_t groups, int ns, int c, float eps, float momentum, pda_stream_t stream);
int pda_bn_relu_max_pool_bwd(const void *x, int x_is_bf16, const float *grad_out, const uint8_t *arg,
                             const float *gamma, const float *beta, const float *mean_invstd, void *grad_x,
                             float *grad_gamma, float *grad_beta, void *scratch, int64_t groups, int ns, int c,
                             pda_stream_t stream);

/* ---- LayerNorm over the last dimension with optional fused residual add (MI355X extension) ----------
 * nn.LayerNorm(D) of TransformerEncoderLayerPreNorm (PointFormer.py:17-18,29,33): x (rows, D) [+ residual
 * (rows, D), the sum also written to sum_out] -> y = (s - mean) / sqrt(var + eps) * gamma + beta per row
 * (biased variance); mean_rstd (rows, 2) kept for the backward pass, whose x argument is the normalised
 * tensor (x, or sum_out when a residual was added); grad_y2 (may be NULL) is a second incoming gradient
 * added to grad_y on the fly.  grad_x / grad_gamma / grad_beta are fully written.
 * D in {256, 512, 1024}.  scratch: pda_layer_norm_scratch_bytes(D) bytes. */
int64_t pda_layer_norm_scratch_bytes(int d);
int pda_layer_norm_fwd(const float *x, const float *residual, const float *gamma, const float *beta,
                       float *sum_out, float *y, float *mean_rstd, int64_t rows, int d, float eps,
                       pda_stream_t stream);
int pda_layer_norm_bwd(const float *x, const float *grad_y, const float *grad_y2, const float *gamma,
                       const float *mean_rstd, float *grad_x, float *grad_gamma, float *grad_beta,
                       void *scratch, int64_t rows, int d, pda_stream_t stream);
/* Dense-bf16 mode variants (same kernels, same fp32 arithmetic; bf16 = raw bit patterns, rounded to nearest even on the
 * store): x may be the bf16 output of a GEMM (x_is_bf16), y (fp32) and y_bf16 (a copy for the next bf16 GEMM) are each
 * optional (at least one); grad_y2 may be bf16; grad_x_bf16 (may be NULL) receives a bf16 copy of grad_x. */
int pda_layer_norm_fwd_mixed(const void *x, int x_is_bf16, const float *residual, const float *gamma,
                             const float *beta, float *sum_out, float *y, uint16_t *y_bf16, float *mean_rstd,
                             int64_t rows, int d, float eps, pda_stream_t stream);
int pda_layer_norm_bwd_mixed(const float *x, const float *grad_y, const void *grad_y2, int grad_y2_is_bf16,
                             const float *gamma, const float *mean_rstd, float *grad_x, uint16_t *grad_x_bf16,
                             float *grad_gamma, float *grad_beta, void *scratch, int64_t rows, int d,
                             pda_stream_t stream);

/* ---- weight / bias gradient of a linear layer over a long token axis (MI355X extension) -------------
 * The backward GEMMs of the point-major 1x1 convolutions and transformer projections: x (tokens, in),
 * grad_out (tokens, out), both row-major -> grad_weight (out, in) = grad_out^T x (the layout of
 * nn.Linear.weight / Conv.weight.flatten(1)) and, when grad_bias != NULL, grad_bias (out) = column sums
 * of grad_out.  Split over the token axis with a fixed-order second stage (deterministic).
 * in, out: multiples of 4.  scratch: pda_linear_wgrad_scratch_bytes(tokens, in, out) bytes.
 * pda_linear_wgrad_form: which kernel a call of this shape runs -- 0 the f32-MFMA split-K form, 1 the streaming form of
 * narrow layers (in, out <= 64), 2 the split-bf16 form (in, out multiples of 256, tokens * in * out >= 2e9). */
int64_t pda_linear_wgrad_scratch_bytes(int64_t tokens, int in_features, int out_features);
int pda_linear_wgrad_form(int64_t tokens, int in_features, int out_features);
int pda_linear_wgrad(const float *x, const float *grad_out, float *grad_weight, float *grad_bias,
                     void *scratch, int64_t tokens, int in_features, int out_features, pda_stream_t stream);
/* The same with x = the PRE-BatchNorm tensor of the layer: the weight gradient is taken against relu(bn(x)) formed in the
 * operand load (x_mean_invstd (2, in) = mean | invstd of the batch, x_gamma, x_beta (in)); the layer's activation need not
 * exist in memory.  Only shapes with pda_linear_wgrad_form == 2; otherwise PDA_ERR_UNSUPPORTED. */
int pda_linear_wgrad_bn(const float *x, const float *grad_out, float *grad_weight, float *grad_bias, void *scratch,
                        int64_t tokens, int in_features, int out_features, const float *x_mean_invstd,
                        const float *x_gamma, const float *x_beta, pda_stream_t stream);

/* ---- a [conv1x1 -> BatchNorm(batch statistics) -> ReLU] chain with the BatchNorm passes folded into the contractions
 * (MI355X extension; the group MLP of pointnet2_modules.py:1657-1662 in training).
 * pda_gemm_split_bn: y (tokens, n_out) = X' W^T on the 256 x 256 tile split-bf16 kernel (wf = pda_linear_split_pack planes,
 * K a multiple of 32 <= 1024), where X' = x, or relu(bn(x)) when in_mean_invstd (2, K) / in_gamma / in_beta (K) are given.
 * stats_mode 1: partial [pda_gemm_split_bn_tiles(tokens)][2][n_out] doubles = per-column sum and sum of squares of y over each
 * tile of 256 tokens (the statistics pass of the BatchNorm behind this contraction; finish with pda_bn_finalize_fwd).
 * Fixed summation order. */
int64_t pda_gemm_split_bn_tiles(int64_t tokens);
int pda_gemm_split_bn(const float *x, const void *wf, float *y, int64_t tokens, int k, int n_out,
                      const float *in_mean_invstd, const float *in_gamma, const float *in_beta, int stats_mode,
                      double *partial, pda_stream_t stream);
/* Inference: out (tokens / ns, n_out) = max over every group of ns consecutive token rows of relu?(x W^T + bias) -- the last layer
 * of an SA scale with the max over nsample in the epilogue; the (tokens, n_out) tensor is never written.  ns in {16, 32, 64},
 * tokens a multiple of ns, K a multiple of 32.  Same arithmetic as pda_gemm_split followed by the max. */
int pda_gemm_split_maxpool(const float *x, const void *wf, const float *bias, float *out, int64_t tokens, int k, int n_out,
                           int ns, int relu, pda_stream_t stream);
/* Inference: the first two layers of a wide SA scale in one launch.  y (b*m*ns, n_out) = relu?(A W2^T + bias2) with
 * A[token] = relu(point_rows[idx[token]] + W1[:, 0:3] (xyz[idx[token]] - new_xyz[group]) + bias1) -- what pda_sa_point_gather would
 * write -- formed in the operand load.  point_rows (b*n, K): the per-point projection of the features by W1[:, 3:]; w1 (K, ldw1);
 * wf: pda_linear_split_pack planes of W2 (n_out, K).  K a multiple of 32 <= 1024. */
int pda_gemm_split_gather(const float *point_rows, const float *xyz, const float *new_xyz, const int32_t *idx,
                          const float *w1, int ldw1, const float *bias1, const void *wf, const float *bias2, float *y,
                          int b, int n, int m, int ns, int k, int n_out, int relu, pda_stream_t stream);
/* The passes of pda_bn_relu_fwd / pda_bn_relu_max_pool_fwd one at a time.  pda_bn_stats_fwd: statistics of x only
 * (mean_invstd (2, C), running statistics updated); scratch: pda_bn_relu_scratch_bytes(c).  pda_bn_finalize_fwd: the same
 * from `nblocks` rows of per-block sums [nblocks][2][C] (count = the number of rows they cover).
 * pda_bn_relu_max_pool_apply: normalise + ReLU + max over ns with given statistics. */
int pda_bn_stats_fwd(const float *x, float *running_mean, float *running_var, float *mean_invstd, void *scratch,
                     int64_t rows, int c, float eps, float momentum, pda_stream_t stream);
int pda_bn_finalize_fwd(const double *partial, int nblocks, int c, int64_t count, float eps, float momentum,
                        float *mean_invstd, float *running_mean, float *running_var, pda_stream_t stream);
int pda_bn_relu_max_pool_apply(const float *x, const float *gamma, const float *beta, const float *mean_invstd,
                               float *out, uint8_t *arg, int64_t groups, int ns, int c, pda_stream_t stream);

/* Dense-bf16 mode: the bias gradient alone, column sums of the bf16 gradient g (rows, cols) -> out (cols) fp32 (fixed
 * summation order).  cols: multiple of 8, <= 2048; scratch: pda_colsum_scratch_bytes(cols) bytes. */
int64_t pda_colsum_scratch_bytes(int cols);
int pda_colsum_bf16(const uint16_t *g, float *out, void *scratch, int64_t rows, int cols, pda_stream_t stream);

/* Token assembly of a PDA scale (MI355X extension; pointnet2_modules.py:879-922), point-major:
 * out (B,M,ns,4C) = [rppe (B,M,ns,C) | f * dscale | f | glob (B,M,C) broadcast over ns] with f = feats (B,N,C)
 * gathered by idx (B,M,ns) and dscale (B,M,ns) the density score.  The gradient entry writes grad_rppe,
 * grad_dscale and grad_glob fully and ADDS into grad_feats (B,N,C), which the caller zero-fills.
 * C in {16, 32, 64, 128, 256}. */
int pda_assemble_tokens(const float *rppe, const float *dscale, const float *feats, const int32_t *idx,
                        const float *glob, float *out, int b, int n, int m, int nsample, int c,
                        pda_stream_t stream);
int pda_assemble_tokens_grad(const float *grad_out, const float *dscale, const float *feats, const int32_t *idx,
                             float *grad_rppe, float *grad_dscale, float *grad_feats, float *grad_glob,
                             int b, int n, int m, int nsample, int c, pda_stream_t stream);

/* Residual add + max-pool over the tokens of a group (MI355X extension; pointnet2_modules.py:929-931 on the
 * encoder layer's output): a, b (groups, seq, D) -> out (groups, D) = max over seq of a + b, arg (groups, D)
 * uint8 = token of the first maximum; pda_max_pool_scatter writes the dense (groups, seq, D) gradient
 * (grad_out routed to the arg-max token, zeros elsewhere).  seq <= 255, D multiple of 4. */
int pda_add_max_pool(const float *a, const float *b, float *out, uint8_t *arg, int64_t groups, int seq, int d,
                     pda_stream_t stream);
int pda_max_pool_scatter(const float *grad_out, const uint8_t *arg, float *grad_x, int64_t groups, int seq,
                         int d, pda_stream_t stream);
/* Dense-bf16 mode: b is the bf16 output of a GEMM; the scatter also writes a bf16 copy of the gradient. */
int pda_add_max_pool_bf16(const float *a, const uint16_t *b, float *out, uint8_t *arg, int64_t groups, int seq,
                          int d, pda_stream_t stream);
int pda_max_pool_scatter_bf16(const float *grad_out, const uint8_t *arg, float *grad_x, uint16_t *grad_x_bf16,
                              int64_t groups, int seq, int d, pda_stream_t stream);

/* Multiplicity-weighted training-mode BatchNorm + ReLU (rows stand for row_weight[r] identical rows of a dense tensor of
 * `count` rows): same results as pda_bn_relu_fwd/bwd on the dense tensor, with grad_y / grad_x the SUMS over the copies. */
int pda_bn_relu_fwd_weighted(const float *x, const float *gamma, const float *beta, float *running_mean,
                             float *running_var, float *y, float *mean_invstd, void *scratch, int64_t rows, int c,
                             float eps, float momentum, const float *row_weight, int64_t count, pda_stream_t stream);
int pda_bn_relu_bwd_weighted(const float *x, const float *grad_y, const float *gamma, const float *beta,
                             const float *mean_invstd, float *grad_x, float *grad_gamma, float *grad_beta,
                             void *scratch, int64_t rows, int c, const float *row_weight, int64_t count,
                             pda_stream_t stream);

/* ---- unique-token ("ragged") execution of a PDA scale (MI355X extension; csrc/ragged.hip) -----------------
 * ball_query pads a short neighbour list with repeats of its first entry (ball_query_gpu.cu:35-41), and the PDA
 * layer runs its transformer encoder over all nsample tokens of every centre (pointnet2_modules.py:924-931).  A
 * repeated neighbour is an identical token, so the encoder is evaluated on the DISTINCT tokens only: compact row
 * u in [off[g], off[g] + cnt[g]) is slot u - off[g] of group g.  Same results as the dense form in exact arithmetic
 * (key 0 of a group enters the softmax with weight nsample - cnt + 1; the max over a group ignores repeats; a
 * compact token receives the summed gradient of its copies).
 * pda_ragged_plan: idx (groups, nsample) -> cnt (groups), off (groups + 1; off[groups] = U = number of distinct
 *   tokens), rowmap (>= U; compact row -> dense row g * nsample + s), row_weight (>= U, optional: the multiplicity of
 *   each compact token, nsample - cnt + 1 for slot 0 and 1 otherwise).  No synchronisation: the caller reads U.
 * rppe_compact != 0: the position-encoding rows (and their gradient) are compact (U, C) as well -- the position MLP
 *   ran on the distinct tokens with pda_bn_relu_{fwd,bwd}_weighted (statistics of the dense tensor from weighted sums;
 *   a compact row's gradient is the sum over its copies).
 * pda_assemble_tokens_ragged(_grad): pda_assemble_tokens writing / reading compact rows (out (U, 4C)); the grid
 *   covers max_tokens >= U rows and reads U on the device.  The gradient entry writes the DENSE grad_rppe /
 *   grad_dscale (zero at the repeat slots) and grad_glob, and ADDS into grad_feats (zero-filled by the caller);
 *   with rowmap (and max_tokens >= U) its per-token part runs over (token, column) threads, with rowmap NULL one
 *   thread walks the tokens of a centre (same results; grad_feats is a float-atomic sum either way).
 * pda_add_max_pool_ragged / pda_max_pool_scatter_ragged: the add + max-pool tail on compact rows; arg = slot.
 * pda_group_attention_ragged_fwd/bwd (include/pda_pointnet2.h layout with compact rows): qkv (U, 3, H, hd),
 *   out / grad_out (U, H * hd), lse (groups, H, seq).  tokens = U (the caller has read it to size qkv): one wave
 *   packs the distinct tokens of several consecutive groups into 32-row tiles, and U / groups decides how many
 *   groups a wave takes (a wrong value costs speed, never results). */
int pda_ragged_plan(const int32_t *idx, int32_t *cnt, int32_t *off, int32_t *rowmap, float *row_weight, int64_t groups,
                    int nsample, pda_stream_t stream);
int pda_assemble_tokens_ragged(const float *rppe, const float *dscale, const float *feats, const int32_t *idx,
                               const float *glob, const int32_t *rowmap, const int32_t *off, float *out,
                               int64_t max_tokens, int b, int n, int m, int nsample, int c, int rppe_compact,
                               pda_stream_t stream);
int pda_assemble_tokens_ragged_grad(const float *grad_out, const float *dscale, const float *feats,
                                    const int32_t *idx, const int32_t *cnt, const int32_t *off, const int32_t *rowmap,
                                    float *grad_rppe, float *grad_dscale, float *grad_feats, float *grad_glob,
                                    int64_t max_tokens, int b, int n, int m, int nsample, int c, int rppe_compact,
                                    pda_stream_t stream);
int pda_add_max_pool_ragged(const float *a, const float *b, const int32_t *cnt, const int32_t *off, float *out,
                            uint8_t *arg, int64_t groups, int d, pda_stream_t stream);
int pda_max_pool_scatter_ragged(const float *grad_out, const uint8_t *arg, const int32_t *rowmap,
                                const int32_t *off, float *grad_x, int64_t max_tokens, int64_t groups, int nsample,
                                int d, pda_stream_t stream);
int pda_group_attention_ragged_fwd(const float *qkv, const int32_t *cnt, const int32_t *off, float *out, float *lse,
                                   int64_t tokens, int64_t num_groups, int seq, int heads, int head_dim,
                                   pda_stream_t stream);
int pda_group_attention_ragged_bwd(const float *qkv, const float *grad_out, const float *lse, const int32_t *cnt,
                                   const int32_t *off, float *grad_qkv, int64_t tokens, int64_t num_groups, int seq,
                                   int heads, int head_dim, pda_stream_t stream);

/* ---- DensityNet in training mode (MI355X extension) ----------------------------------------------
 * pointnet2_modules.py:958-981: y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) with 1x1 convs
 * 1 -> 16 -> 8 -> 1 (with bias), batch statistics over the n tokens; x, y (n) fp32.
 * params (pda_densitynet_param_count() = 227 floats): w1[16] b1[16] gamma1[16] beta1[16] W2[8][16] b2[8]
 * gamma2[8] beta2[8] w3[8] b3 gamma3 beta3; grad_params has the same layout.  stats (46 floats) carries the
 * batch statistics to the backward pass; running_mean/var k (sizes 16, 8, 1) are updated like nn.BatchNorm
 * (all six NULL: skip).  The input receives no gradient (it is a function of coordinates only).
 * scratch: pda_densitynet_scratch_bytes() bytes. */
int pda_densitynet_param_count(void);
int64_t pda_densitynet_scratch_bytes(void);
int pda_densitynet_fwd(const float *x, const float *params, float *y, float *stats, void *scratch,
                       float *running_mean1, float *running_var1, float *running_mean2, float *running_var2,
                       float *running_mean3, float *running_var3, int64_t n, float eps, float momentum,
                       pda_stream_t stream);
int pda_densitynet_bwd(const float *x, const float *grad_y, const float *params, const float *stats,
                       float *grad_params, void *scratch, int64_t n, float eps, pda_stream_t stream);
/* The same on the DISTINCT slots of padded neighbour lists (pda_ragged_plan above): x, y, grad_y keep the dense
 * (groups, nsample) layout, n = groups * nsample, but only the *n_unique slots rowmap[0..) are evaluated, slot 0 of a
 * group standing for its row_weight = nsample - cnt + 1 identical tokens (ball_query's repeats, slots cnt..nsample-1,
 * hold the same x).  n_unique is DEVICE memory (off + groups of the plan): no host read.  Batch statistics, running
 * statistics and parameter gradients are those of the n dense tokens; forward writes every slot of y (a group's
 * repeats together with its slot 0); backward takes the gradient of a distinct slot as the sum over its copies.
 * Results equal pda_densitynet_fwd / _bwd up to the order of the floating-point additions. */
int pda_densitynet_fwd_unique(const float *x, const float *params, float *y, float *stats, void *scratch,
                              float *running_mean1, float *running_var1, float *running_mean2, float *running_var2,
                              float *running_mean3, float *running_var3, int64_t n, const int32_t *rowmap,
                              const float *row_weight, const int32_t *n_unique, int nsample, float eps, float momentum,
                              pda_stream_t stream);
int pda_densitynet_bwd_unique(const float *x, const float *grad_y, const float *params, const float *stats,
                              float *grad_params, void *scratch, int64_t n, const int32_t *rowmap,
                              const float *row_weight, const int32_t *n_unique, int nsample, float eps,
                              pda_stream_t stream);
/* Several independent DensityNet problems in ONE set of launches (the scales of a PDA layer: each of the nine passes is
 * ~10 us of dependency latency on <= 128 workgroups, so two problems per launch cost what one does).  Per problem the
 * arguments of the entries above; rowmap / row_weight / n_unique all NULL: every token (pda_densitynet_fwd), else the
 * distinct slots (pda_densitynet_fwd_unique).  Forward reads x, params, writes y, stats, running (all six or none);
 * backward reads x, grad_y, params, stats, writes grad_params.  scratch: pda_densitynet_scratch_bytes() each. */
#define PDA_DENSITYNET_MAX_SCALES 4
typedef struct pda_densitynet_scale {
    const float *x, *grad_y, *params;
    float *y, *stats;
    void *scratch;
    float *running[6]; /* mean1, var1, mean2, var2, mean3, var3 */
    float *grad_params;
    int64_t n;
    const int32_t *rowmap;
    const float *row_weight;
    const int32_t *n_unique;
    int nsample;
    float eps, momentum;
} pda_densitynet_scale_t;
int pda_densitynet_fwd_multi(const pda_densitynet_scale_t *scales, int nscales, pda_stream_t stream);
int pda_densitynet_bwd_multi(const pda_densitynet_scale_t *scales, int nscales, pda_stream_t stream);

/* Inference: the three layers with their BatchNorms folded in (running statistics), one launch, one thread per token.
 * folded (pda_densitynet_eval_param_count() = 177 floats): w1[16] b1[16] W2[8][16] b2[8] w3[8] b3 with
 * W' = W * gamma / sqrt(running_var + eps), b' = (bias - running_mean) * gamma / sqrt(running_var + eps) + beta. */
int pda_densitynet_eval_param_count(void);
int pda_densitynet_eval(const float *x, const float *folded, float *y, int64_t n, pda_stream_t stream);

/* PDA grouper geometry (MI355X extension; pointnet2_utils.py:590-607, pointnet2_modules.py:905-913,:1000-1001),
 * point-major: xyz (B,N,3), new_xyz (B,M,3), idx (B,M,nsample) -> rppe (B,M,nsample,12) = [centre, neighbour,
 * centre - neighbour, (neighbour - centre) / radius] and dscale (B,M,nsample) = gaussian density
 * exp(-|d|^2 / (2 r^2)) / (2.5 r) divided by its maximum over the group.  nsample: power of two <= 64. */
int pda_pda_geometry(const float *xyz, const float *new_xyz, const int32_t *idx, float *rppe, float *dscale,
                     int b, int n, int m, int nsample, float radius, pda_stream_t stream);

/* ---- target assignment ------------------------------------------------------------------------
 * replaces points_in_boxes_gpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:98-118 ->
 * roiaware_pool3d_kernel.cu:313-359; test :16-36): boxes (B,T,7) [x,y,z,dx,dy,dz,heading],
 * pts (B,M,3) -> box_idx_of_points (B,M) int32 = lowest k whose box contains the point
 * (|z-cz| <= dz/2, |local x| < dx/2 + 1e-5, |local y| < dy/2 + 1e-5 after rotating by -heading);
 * entries of points in no box are NOT written (the caller pre-fills -1, roiaware_pool3d_utils.py:41). */
int pda_points_in_boxes(const float *boxes, const float *pts, int32_t *box_idx_of_points, int b, int t,
                        int m, pda_stream_t stream);
/* The point-wise remainder of the IA-SSD head's target assignment (IASSD_head.py:132-277 after the two points_in_boxes
 * queries), one launch per point set: from in_box / in_ext (B, N) = index of the ground-truth box / enlarged box each point
 * lies in (-1: none) and gt_boxes (B, T, 8) [x y z dx dy dz heading class]:
 *   mode 0 (set_ignore_flag, :207-217)   foreground = in a box; points only inside the enlarged box get label -1
 *   mode 1 (use_ex_gt_assign, :190-205)  foreground = in an enlarged box; instance points keep their own box index
 *   mode 2 (mode 1 + fg_pc_ignore)       foreground = in the enlarged box only; box index -1 for the instance points
 * labels (B*N) int64: 0 background, -1 ignored, else the class of the box (1 when single_class); box_idx (B*N) int64;
 * gt_of_points (B*N, 8) = gt_boxes[scene][box index], index -1 wrapping to the last row as the reference's indexing does. */
int pda_assign_point_targets(const float *gt_boxes, const int32_t *in_box, const int32_t *in_ext, int64_t *labels,
                             int64_t *box_idx, float *gt_of_points, int b, int n, int t, int mode, int single_class,
                             pda_stream_t stream);
/* Soft instance labels of gauss_fun_once_topk_GT_add_same_size (IASSD_head.py:889-963): out[p] = exp(-0.5 |S d|^2) where
 * labels[p] > 0, else 0; d = offset of point p (coords + p*stride + offset: x, y, z) in the frame of its box
 * gt_of_points[p], S = diag(4/(w^2+l^2), 4/(w^2+h^2), 4/(h^2+l^2)) times 4 / 6 / 5 for classes 1 / 2 / 3. */
/* assign_stack_targets_IASSD (IASSD_head.py:132-277) for one point set in one launch: both box queries (boxes, and boxes
 * enlarged by extra_width[3] -- a HOST pointer), labels / box index / gathered box, and (box_labels != NULL) the box-coder
 * targets of the foreground points.  points: rows of point_stride floats, xyz at point_offset (b * n rows, scene-major). */
int pda_head_assign_targets(const float *points, int point_stride, int point_offset, const float *gt_boxes, const float *extra_width,
                            int64_t *labels, int64_t *box_idx, float *gt_of_points, float *box_labels, const float *mean_size,
                            int bins, int b, int n, int t, int mode, int single_class, pda_stream_t stream);
int pda_sa_gaussian_mask(const float *coords, int stride, int offset, const float *gt_of_points, const int64_t *labels,
                         float *out, int64_t points, pda_stream_t stream);

/* ---- rotated BEV overlap / IoU / NMS (pcdet/ops/iou3d_nms) --------------------------------------
 * replace boxes_overlap_bev_gpu / boxes_iou_bev_gpu (src/iou3d_nms.cpp:40-63 / :65-87 ->
 * iou3d_nms_kernel.cu:236-264): boxes_a (num_a,7), boxes_b (num_b,7) [x,y,z,dx,dy,dz,heading] ->
 * (num_a,num_b) BEV intersection area / IoU. */
int pda_boxes_overlap_bev(const float *boxes_a, const float *boxes_b, float *ans_overlap, int num_a,
                          int num_b, pda_stream_t stream);
int pda_boxes_iou_bev(const float *boxes_a, const float *boxes_b, float *ans_iou, int num_a, int num_b,
                      pda_stream_t stream);
/* replaces nms_gpu / nms_normal_gpu (src/iou3d_nms.cpp:90-138 / :141-188 -> kernels :266-369), for a
 * whole batch and without the reference's device->host mask copy: boxes (B,N,7), each scene already
 * sorted by descending score; num_valid (B) int32 or NULL = only the first num_valid[s] boxes of a
 * scene take part; keep (B,N) int64 = kept indices in score order, padded with -1; num_keep (B) int32.
 * normal != 0: axis-aligned IoU (nms_normal_gpu).  mask_scratch: B * pda_nms_mask_words(N) uint64.
 * N <= 32768. */
int64_t pda_nms_mask_words(int n);
int pda_nms_bev(const float *boxes, const int32_t *num_valid, int64_t *keep, int32_t *num_keep,
                uint64_t *mask_scratch, int b, int n, float thresh, int normal, pda_stream_t stream);

/* ---- loss terms of the IA-SSD head, one launch each: the term AND its gradient (csrc/head_loss.hip) --------------------------
 * Replace the elementwise torch chains of IASSD_head.py:525-735, :1239-1321 and loss_utils.py:75-194, :340-363.  All tensors
 * row-major f32 unless noted; labels int64; every `grad` output has the layout of the prediction it belongs to and is written
 * completely; the autograd node multiplies it by the incoming scalar.
 *   pda_head_cls_loss:    out2 = {scale * sum_rows w_row * mean_c bce(x, t), #positives}; w_row = [label >= 0] / max(#pos, 1),
 *                         t = [label == c + 1] * soft_row (soft may be NULL = 1); the C logits sit in columns col0 .. col0+C-1
 *                         of rows of row_stride floats.
 *   pda_head_centerness:  generate_center_ness_mask (:795-817); centers (n, 4) [bs, x, y, z], gt (n, 8).
 *   pda_head_box_loss:    get_center_box_binori_layer_loss; preds (n, 6 + 2 bins), labels (n, 8); out4 = {total, xyzwhl,
 *                         ori_bin * dir_weight, ori_res}; code_weights 6 floats or NULL.
 *   pda_head_vote_loss:   mode 0 get_contextual_vote_loss (key = class label of the point), mode 1 _ver2 (key = box index, -1
 *                         none; b scenes of n / b points, `boxes` boxes per scene); origin, offsets, grad (n, 4) [bs, x, y, z].
 *   pda_head_corner_loss: get_corner_layer_loss incl. the decode of PointResidual_BinOri_Coder (mean_size (num_class, 3) or
 *                         NULL); NaN without positive centres, like the reference. */
int pda_head_cls_loss(const float *preds, int row_stride, int col0, int num_class, const int64_t *labels, const float *soft,
                      int64_t n, float scale, float *out2, float *grad, pda_stream_t stream);
int pda_head_centerness(const float *centers, const float *gt, const int64_t *labels, float *out, int64_t n, pda_stream_t stream);
int pda_head_box_loss(const float *preds, const float *labels, const int64_t *cls_labels, const float *code_weights, float beta,
                      int bins, float dir_weight, float box_weight, int64_t n, float *out4, float *grad, pda_stream_t stream);
int pda_head_vote_loss(int mode, const float *origin, const float *offsets, const int64_t *key, const float *gt, int b, int boxes,
                       int num_class, float weight, int64_t n, float *out1, float *grad, pda_stream_t stream);
int pda_head_corner_loss(const float *box_preds, const float *centers, const float *cls_preds, int num_class, const float *gt,
                         const int64_t *cls_labels, const float *mean_size, int bins, float weight, int64_t n, float *out1,
                         float *grad_box, float *grad_centers, pda_stream_t stream);

/* ---- the NARROW vanilla set-abstraction scale in training form (csrc/sa_train_small.hip; MI355X extension) -------------
 * QueryAndGroup -> [Conv2d 1x1 (no bias) -> BatchNorm2d (batch statistics) -> ReLU] x 3 -> max over nsample
 * (pointnet2_modules.py:1657-1670, pointnet2_utils.py:671-704) for chains 3 + c -> c1 -> c2 -> c3 with c <= 5,
 * c1, c2 in {16, 32} and (nsample, c3) in {(16, 32), (32, 64)}: ONCE / KITTI layer 0.  Forward and backward are families
 * of recompute passes over the neighbour lists: no (B, M, ns, C) tensor exists in HBM in the forward pass, the backward
 * pass keeps dz2 (tokens, c2) and dz1 (tokens, c1) only (tokens = b * m * nsample, a multiple of 32).
 *   fwd: out (b*m, c3) = the pooled activation, zmax = the layer-3 pre-activation at the arg-max, arg = the arg-max slot
 *        (lowest slot on ties); running statistics of the three BatchNorms updated (entries of running_* may be NULL);
 *        `workspace` (pda_sa_small_train_workspace_bytes(), 256-byte aligned) receives the packed weights and the batch
 *        statistics and must reach the backward call unchanged.
 *   bwd: grad_out (b*m, c3) -> dw1 (c1, 3 + c), dw2 (c2, c1), dw3 (c3, c2), dgamma[l] / dbeta[l] (l = 0..2).  The
 *        gradient wrt the gathered inputs (xyz, features) is NOT produced: layer 0's inputs are the raw points.
 * feat_pm is point-major (b, n, c); with c == 1 that is the memory of the reference's (b, 1, n).
 * Returns PDA_ERR_UNSUPPORTED for any other shape (pda_sa_small_train_supported tells without a call). */
int64_t pda_sa_small_train_workspace_bytes(void);
int pda_sa_small_train_supported(int c, int nsample, int c1, int c2, int c3, int64_t tokens);
int pda_sa_small_train_fwd(const float *xyz, const float *new_xyz, const float *feat_pm, const int32_t *idx,
                           const float *w1, const float *w2, const float *w3, const float *const *gamma,
                           const float *const *beta, float *const *running_mean, float *const *running_var,
                           const float *eps, const float *momentum, void *workspace, float *out, float *zmax,
                           uint8_t *arg, int b, int n, int m, int c, int nsample, int c1, int c2, int c3,
                           pda_stream_t stream);
int pda_sa_small_train_bwd(const float *xyz, const float *new_xyz, const float *feat_pm, const int32_t *idx,
                           const float *grad_out, const float *zmax, const uint8_t *arg, void *workspace, float *dz2,
                           float *dz1, float *dw1, float *dw2, float *dw3, float *const *dgamma, float *const *dbeta,
                           int b, int n, int m, int c, int nsample, int c1, int c2, int c3, pda_stream_t stream);

/* ---- the coordinate columns of a vanilla SA scale's first layer, backward (csrc/sa_xyz_grad.hip; MI355X extension) ----
 * z1 = [xyz[idx] - new_xyz | features[idx]] W1^T, W1 (c1, ldw >= 3 + C).  From grad_z1 (b*m*nsample, c1):
 * dw[o][0:3] (row stride lddw) = sum over tokens of grad_z1[t][o] * (xyz[idx[t]] - new_xyz[centre(t)]) and, when
 * grad_new_xyz (b, m, 3) is not NULL, grad_new_xyz[centre] = -(sum over the centre's samples of grad_z1) W1[:, 0:3].
 * One pass over grad_z1; the feature columns go through pda_linear_wgrad / pda_gemm_split. */
int64_t pda_sa_xyz_grad_scratch_bytes(int c1);
/* Forward of the same layer without a per-token contraction (a linear layer commutes with the gather):
 * z (b*m*nsample, c1) = point_rows[idx] + W1[:, 0:3] (xyz[idx] - new_xyz[centre]), with point_rows (b*n, c1) = features W1[:, 3:]^T
 * computed once per point by the caller (pda_gemm_split); optional bias (c1) and ReLU (inference: BatchNorm folded into W1).
 * c1 in {128, 256, 512, 1024}. */
int pda_sa_point_gather(const float *point_rows, const float *xyz, const float *new_xyz, const int32_t *idx, const float *w,
                        int ldw, const float *bias, int relu, float *z, int b, int n, int m, int nsample, int c1,
                        pda_stream_t stream);
int pda_sa_xyz_grad(const float *grad_z1, const float *xyz, const float *new_xyz, const int32_t *idx, const float *w, int ldw,
                    float *dw, int lddw, float *grad_new_xyz, void *scratch, int b, int n, int m, int nsample, int c1,
                    pda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

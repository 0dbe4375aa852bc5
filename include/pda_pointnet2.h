/*
 * pda_pointnet2.h -- C ABI of libpda_pointnet2.so, the MI355X (gfx950) implementation of
 * the PDA-SSD point-sampling / grouping hot path.
 *
 * Drop-in boundary: these entry points are what a binding of the reference's extension
 * module `pointnet2_batch_cuda` would call.  The reference binds 14 functions with pybind11
 * (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:12-33); each
 * declaration below cites the reference wrapper it replaces.  Differences from the
 * reference wrappers, by design of a C ABI:
 *   - raw device pointers + sizes instead of at::Tensor; all buffers are caller-allocated,
 *     contiguous, fp32 / int32, resident on the device that `stream` belongs to;
 *   - an explicit stream (hipStream_t passed as void*; NULL = the null stream) instead of
 *     the legacy default stream; every call is asynchronous and captures into hipGraphs
 *     (no allocation, no synchronisation inside);
 *   - a status code is returned (PDA_OK == 0) instead of the reference's
 *     fprintf(stderr)+exit(-1) (e.g. sampling_gpu.cu:248-252); pda_last_error() gives the
 *     message for the calling thread.  The Python mirror (pdanet_amd/pointnet2_batch_cuda.py)
 *     raises on non-zero status and returns the reference's own return values (1 / 2 / None).
 *
 * Initialisation contracts are the reference's (pointnet2_utils.py:26,95,174,218,246):
 * FPS `temp` arrives filled with 1e10 and is clobbered; ball-query `idx` arrives zeroed and
 * rows without any neighbour are left untouched; every `*_grad` output arrives zeroed and is
 * accumulated into.
 */
#ifndef PDA_POINTNET2_H
#define PDA_POINTNET2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDA_OK 0
#define PDA_ERR_INVALID_ARGUMENT 1 /* null pointer / negative or inconsistent size */
#define PDA_ERR_LAUNCH 2           /* HIP reported a launch error */
#define PDA_ERR_UNSUPPORTED 3      /* shape outside what the kernels were built for */

typedef void *pda_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
#define PDA_POINTNET2_ABI_VERSION 20
int pda_abi_version(void);
/* Message of the last non-PDA_OK status returned on the calling thread ("" if none). */
const char *pda_last_error(void);
/* 1 if the library was compiled with the nvcc-style contraction fma(dz,dz,fma(dy,dy,dx*dx))
 * of the squared distance (default), 0 for the uncontracted form.  Must match the oracle. */
int pda_fp_contract_mode(void);
/* Block size the reference launcher would choose for `work_size` (cuda_utils.h:10-14);
 * it fixes the FPS tie-breaking and is exported so hosts/tests can state it. */
int pda_opt_n_threads(int work_size);
/* Diagnostics of the multi-workgroup FPS form (16384 < n <= 65536: K workgroups per scene exchange their
 * row records / round winners through device memory with BOUNDED polls).  A timed-out exchange is recovered on the device
 * (the scene is recomputed by a follow-up kernel on the same stream: results stay exact) and counted;
 * *total = workgroups that timed out since load (or the last reset).  Synchronises the device.
 * The reference has no counterpart: its kernel is one block per scene (sampling_gpu.cu:211-253). */
int pda_fps_coop_timeouts(unsigned long long *total, int reset);
/* Test hook: polls per exchange before a workgroup gives up (0 forces the recovery path whenever a partner
 * is late; negative restores the default 1 << 20).  Synchronises the device. */
int pda_debug_fps_spin_limit(int polls);
/* Test hook: the number of non-zero exchange granules once everything queued has run (synchronises the device).  The
 * follow-up kernel of every multi-workgroup launch wipes the granules of its scenes, so this is 0 between launches. */
long long pda_debug_fps_exchange_nonzero(void);

/* ---- furthest point sampling ------------------------------------------------------- */
/* replaces farthest_point_sampling_wrapper (sampling.cpp:34-43, pointnet2_api.cpp:27;
 * kernel sampling_gpu.cu:93-253).  xyz (b,n,3), temp (b,n) pre-filled 1e10 (clobbered:
 * holds the final min squared distances on return), idx (b,m).  Index-exact including the
 * reference's shared-memory-tree tie-breaking. */
int pda_furthest_point_sampling(const float *xyz, float *temp, int32_t *idx, int b, int n, int m,
                                pda_stream_t stream);
/* replaces furthest_point_sampling_with_dist_wrapper (sampling.cpp:46-56,
 * pointnet2_api.cpp:28; kernel sampling_gpu.cu:256-416).  dist (b,n,n). */
int pda_furthest_point_sampling_with_dist(const float *dist, float *temp, int32_t *idx, int b,
                                          int n, int m, pda_stream_t stream);

/* ---- gather ------------------------------------------------------------------------ */
/* replaces gather_points_wrapper_fast (sampling.cpp:11-19; kernel sampling_gpu.cu:8-44):
 * out[b,c,j] = points[b,c,idx[b,j]];  points (b,c,n), idx (b,m), out (b,c,m). */
int pda_gather_points(const float *points, const int32_t *idx, float *out, int b, int c, int n,
                      int m, pda_stream_t stream);
/* replaces gather_points_grad_wrapper_fast (sampling.cpp:22-31; kernel sampling_gpu.cu:46-83):
 * grad_points[b,c,idx[b,j]] += grad_out[b,c,j];  grad_points (b,c,n) pre-zeroed. */
int pda_gather_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b,
                           int c, int n, int m, pda_stream_t stream);

/* ---- ball query -------------------------------------------------------------------- */
/* replaces ball_query_wrapper_fast (ball_query.cpp:32-42; kernel ball_query_gpu.cu:9-67).
 * new_xyz (b,m,3), xyz (b,n,3), idx (b,m,nsample) pre-zeroed.  First `nsample` points in
 * ascending index with d2 < radius*radius; the first hit pre-fills the row. */
int pda_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int b, int n, int m,
                   float radius, int nsample, pda_stream_t stream);
/* replaces ball_query_dilated_wrapper_fast (ball_query.cpp:45-56; kernel
 * ball_query_gpu.cu:70-139), including the d2==0 double append. */
int pda_ball_query_dilated(const float *new_xyz, const float *xyz, int32_t *idx, int b, int n,
                           int m, float max_radius, float min_radius, int nsample,
                           pda_stream_t stream);
/* replaces `ellipsoid_query` (pointnet2_api.cpp:16; ellipsoid_query.cpp:13-76, kernel ellipsoid_query_gpu.cu:311-498
 * with the Jacobi eigen-decomposition :58-298).  No PDA-SSD yaml reaches it (its only call on the PDA path is commented
 * out, pointnet2_utils.py:586-587; the "Ellipsoid" SA class uses the spherical ball query); built so that
 * every name of the boundary has a kernel.  idx (b, m, nsample) is caller-allocated and ZERO-filled (the reference's
 * wrapper allocates it with torch::zeros); on return a row holds the ball query of radius e3 extended, for centres with
 * >= 3 hits, by the unlisted points inside the ellipsoid (e1, e2, e3) oriented along the eigenvectors of the hits'
 * covariance, in index order, up to nsample.  The reference's work tensors (ingroup_*, v, d) are not materialised.
 * Parity vs a CUDA build is unpinned for this operator (float expression contraction inside the Jacobi rotations). */
int pda_ellipsoid_query(const float *new_xyz, const float *xyz, int32_t *idx, int b, int n, int m,
                        float e1, float e2, float e3, int nsample, pda_stream_t stream);
/* MI355X extension: pda_ball_query_multi through a uniform cell list (csrc/ball_query_cells.hip) -- the same rows,
 * bit for bit, computed from the points of the 27 cells around a centre instead of all n (bin -> one wave per centre
 * -> the nsample smallest indices of its hits; a dense ball falls back to the reference's ascending scan with early
 * exit).  Worth it for n >= 8192.  scratch: pda_ball_query_cells_scratch_bytes(b, n) bytes, 256-byte aligned, owned
 * by the caller (nothing is allocated inside); nsample <= 128, radius > 0. */
int64_t pda_ball_query_cells_scratch_bytes(int b, int n);
int pda_ball_query_cells(const float *new_xyz, const float *xyz, int32_t *const *idx, int b, int n, int m, int nr,
                         const float *radii, const int32_t *nsamples, void *scratch, int64_t scratch_bytes,
                         pda_stream_t stream);
/* MI355X extension (no reference counterpart): up to 3 radii over the SAME centres and
 * points in one pass -- the multi-scale groupers call ball_query once per scale
 * (pointnet2_modules.py:1657).  idx[i] is (b,m,nsamples[i]); results are identical to
 * `nr` separate pda_ball_query calls. */
int pda_ball_query_multi(const float *new_xyz, const float *xyz, int32_t *const *idx, int b, int n,
                         int m, int nr, const float *radii, const int32_t *nsamples,
                         pda_stream_t stream);

/* ---- grouping ---------------------------------------------------------------------- */
/* replaces group_points_wrapper_fast (group_points.cpp:30-41; kernel
 * group_points_gpu.cu:53-92): out[b,c,p,s] = points[b,c,idx[b,p,s]]. */
int pda_group_points(const float *points, const int32_t *idx, float *out, int b, int c, int n,
                     int npoints, int nsample, pda_stream_t stream);
/* replaces group_points_grad_wrapper_fast (group_points.cpp:18-28; kernel
 * group_points_gpu.cu:14-50): grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s]. */
int pda_group_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b,
                          int c, int n, int npoints, int nsample, pda_stream_t stream);

/* MI355X extension (no reference counterpart): gather in the point-major layout,
 * out[b,e,:] = rows[b, idx[b,e], :] with rows (b,n,c), idx (b,num_idx), out (b,num_idx,c): a
 * neighbour's features are one contiguous row, so gather and scatter-add are coalesced.  Same
 * values as pda_group_points on the transposed tensors. */
int pda_group_rows(const float *rows, const int32_t *idx, float *out, int b, int n, int c,
                   int64_t num_idx, pda_stream_t stream);
/* grad_rows[b, idx[b,e], :] += grad_out[b,e,:]; grad_rows (b,n,c) pre-zeroed. */
int pda_group_rows_grad(const float *grad_out, const int32_t *idx, float *grad_rows, int b, int n,
                        int c, int64_t num_idx, pda_stream_t stream);

/* ---- three-NN + interpolation ------------------------------------------------------ */
/* replaces three_nn_wrapper_fast (interpolate.cpp:21-30; kernel interpolate_gpu.cu:16-81).
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared distances, idx (b,n,3). */
int pda_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx, int b,
                 int n, int m, pda_stream_t stream);
/* replaces three_interpolate_wrapper_fast (interpolate.cpp:32-44; kernel
 * interpolate_gpu.cu:84-124): out[b,c,j] = sum_t weight[b,j,t] * points[b,c,idx[b,j,t]]. */
int pda_three_interpolate(const float *points, const int32_t *idx, const float *weight,
                          float *out, int b, int c, int m, int n, pda_stream_t stream);
/* replaces three_interpolate_grad_wrapper_fast (interpolate.cpp:46-58; kernel
 * interpolate_gpu.cu:127-168): grad_points (b,c,m) pre-zeroed. */
int pda_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                               float *grad_points, int b, int c, int n, int m,
                               pda_stream_t stream);

/* ---- Chamfer 1-NN distance ------------------------------------ */
/* replaces chamfer_forward (chamfer_cuda.cpp:22-25 -> chamferthreed.cu:136-153; kernel :12-134).
 * xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n) squared distance to / idx1 (b,n) index of the nearest
 * point of xyz2 (lowest index on ties), and dist2/idx2 (b,m) the other way round. */
int pda_chamfer_forward(const float *xyz1, const float *xyz2, float *dist1, float *dist2,
                        int32_t *idx1, int32_t *idx2, int b, int n, int m, pda_stream_t stream);
/* replaces chamfer_backward (chamfer_cuda.cpp:27-31 -> chamferthreed.cu:176-195; kernel :155-174).
 * gradxyz1 (b,n,3) / gradxyz2 (b,m,3) pre-zeroed, accumulated into. */
int pda_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                         const float *graddist1, const float *graddist2, const int32_t *idx1,
                         const int32_t *idx2, int b, int n, int m, pda_stream_t stream);

/* ---- per-group self-attention (MI355X extension) ------------------------------------ */
/* The attention of the PDA layer's TransformerEncoderLayerPreNorm (PointFormer.py:30-33,
 * nn.MultiheadAttention without mask/dropout) over sequences of `seq` = nsample tokens:
 *   qkv (num_groups, seq, 3, heads, head_dim) = the in_proj output; out (num_groups, seq,
 *   heads*head_dim) = softmax(Q K^T / sqrt(head_dim)) V, ready for out_proj;
 *   lse (num_groups, heads, seq) = log-sum-exp per query, kept for the backward pass.
 * seq in {8,16,32}, head_dim in {32,64,128}; anything else returns PDA_ERR_UNSUPPORTED (callers
 * then use the framework's scaled_dot_product_attention).  fp32 MFMA, fp32 accumulate. */
int pda_group_attention_fwd(const float *qkv, float *out, float *lse, int64_t num_groups, int seq,
                            int heads, int head_dim, pda_stream_t stream);
/* grad_qkv (num_groups, seq, 3, heads, head_dim) is fully written (no pre-zeroing needed). */
int pda_group_attention_bwd(const float *qkv, const float *grad_out, const float *lse,
                            float *grad_qkv, int64_t num_groups, int seq, int heads, int head_dim,
                            pda_stream_t stream);
/* The same kernels with bf16 tensors at the HBM boundary (raw bf16 bit patterns; lse stays fp32): for the dense-bf16
 * mode, where qkv / grad_out come straight out of bf16 GEMMs and out / grad_qkv go straight into them.  All arithmetic
 * (MFMA operands, softmax, accumulation) is fp32 as above; outputs are rounded to nearest even once, on the store. */
int pda_group_attention_fwd_bf16(const uint16_t *qkv, uint16_t *out, float *lse, int64_t num_groups, int seq,
                                 int heads, int head_dim, pda_stream_t stream);
int pda_group_attention_bwd_bf16(const uint16_t *qkv, const uint16_t *grad_out, const float *lse,
                                 uint16_t *grad_qkv, int64_t num_groups, int seq, int heads, int head_dim,
                                 pda_stream_t stream);

/* ---- fused set-abstraction scale (MI355X extension) --------------------------------- */
/* One scale of a vanilla SA layer in inference form, fused into one kernel:
 *   QueryAndGroup (pointnet2_utils.py:671-704) -> [Conv2d 1x1 no bias -> BatchNorm2d (folded
 *   into scale/shift) -> ReLU] x 3 -> max over nsample   (pointnet2_modules.py:1657-1670).
 * No reference extension entry corresponds to it: the reference runs this chain as ~12 torch /
 * cuDNN kernels over materialised (B,C,npoint,nsample) tensors.  f32 MFMA, f32 accumulate.
 *   dims[4] = {3 + c, c1, c2, c3};  wf[l] = weights of layer l packed by
 *   pda_sa_mlp_pack_weights (layer 0: first_layer = 1);  scale[l], shift[l]: folded BN, padded
 *   with zeros to a multiple of 32 entries;  out (b, c3, m).
 * Returns PDA_ERR_UNSUPPORTED for chains no kernel was built for (callers then run the unfused
 * operator sequence) -- see the case table in csrc/sa_mlp.hip. */
int pda_sa_mlp_maxpool(const float *xyz, const float *new_xyz, const float *features,
                       const int32_t *idx, float *out, int b, int n, int m, int c, int nsample,
                       const int32_t *dims, const float *const *wf, const float *const *scale,
                       const float *const *shift, pda_stream_t stream);
/* Training form of the group MLP (csrc/sa_mlp.hip, lin_cols_kernel): training-mode BatchNorm separates the layers, so
 * the chain runs one contraction per call on the same f32 MFMA code -- forward and input gradient; the statistics,
 * ReLU and max-pool are pda_bn_relu_* (include/pda_train.h), the weight gradient pda_linear_wgrad.
 *   pda_linear_cols:       y (tokens, n_out) = x (tokens, k) W^T;  k in {256, 512}, n_out a multiple of 128 <= 1024;
 *                          wf = W (n_out, k) packed by pda_linear_cols_pack(..., transposed_source = 0, gather_order = 0).
 *                          dX = dY W is the same call with wf packed from the transposed source (W given as stored,
 *                          (k_out = rows of W, n = its columns): pack(w, wf, n_out = cols of W, k = rows of W, 1, 0)).
 *   pda_sa_gather_linear:  layer 1 with the grouping fused in (QueryAndGroup, pointnet2_utils.py:671-704): y (b, m, ns,
 *                          n_out) = [xyz[idx] - new_xyz | feats_pm[idx]] W1^T with feats_pm (b, n, c) POINT-major, c = 256,
 *                          W1 (n_out, 3 + c) packed with gather_order = 1.  The grouped input is never materialised.
 * PDA_ERR_UNSUPPORTED for other shapes (callers use the library GEMM). */
int pda_linear_cols_packed_size(int n_out, int k);
int pda_linear_cols_pack(const float *w, float *wf, int n_out, int k, int transposed_source, int gather_order,
                         pda_stream_t stream);
int pda_linear_cols(const float *x, const float *wf, float *y, int64_t tokens, int k, int n_out, pda_stream_t stream);
int pda_sa_gather_linear(const float *xyz, const float *new_xyz, const float *feats_pm, const int32_t *idx,
                         const float *wf, float *y, int b, int n, int m, int c, int nsample, int n_out,
                         pda_stream_t stream);
/* The same contraction on the bf16 matrix cores with f32 operands split into three bf16 terms each (csrc/gemm_split.hip:
 * x = h + m + l exactly; six of the nine cross products are kept, the dropped ones are below 2^-24 of the product; f32
 * accumulate): f32-grade results at 2.67x the f32 MFMA rate.  Replaces the same cuDNN/cuBLAS f32 calls as pda_linear_cols.
 *   pda_linear_split:  y (tokens, n_out) = x (tokens, k) W^T [+ bias] [relu];  k a multiple of 32 <= 512, n_out a multiple
 *                      of 128;  wf = pda_linear_split_pack(W (n_out, k)); transposed_source = 1: w holds W^T (k, n_out).
 * PDA_ERR_UNSUPPORTED for other shapes. */
int64_t pda_linear_split_packed_bytes(int n_out, int k);
int pda_linear_split_pack(const float *w, void *wf, int n_out, int k, int transposed_source, pda_stream_t stream);
/* Both forms a training step needs of one weight W (n_out, k), packed in one launch: wf = the planes of W (as
 * pda_linear_split_pack(w, wf, n_out, k, 0)), wft = the planes of W^T from the same source (as
 * pda_linear_split_pack(w, wft, k, n_out, 1): the weights of dX = dY W). */
int pda_linear_split_pack_both(const float *w, void *wf, void *wft, int n_out, int k, pda_stream_t stream);
int pda_linear_split(const float *x, const void *wf, const float *bias, float *y, int64_t tokens, int k, int n_out,
                     int relu, pda_stream_t stream);
/* The same arithmetic as an LDS-tiled GEMM (gemm_split_kernel): any k that is a multiple of 32, any n_out (the planes of
 * pda_linear_split_pack are padded to 128 outputs), y [+]= x W^T [+ bias] [relu]; accumulate = 1 adds to y. */
int pda_gemm_split(const float *x, const void *wf, const float *bias, float *y, int64_t tokens, int k, int n_out,
                   int relu, int accumulate, pda_stream_t stream);
/* Number of floats pda_sa_mlp_pack_weights writes for a (rows x cols) layer. */
int pda_sa_mlp_packed_size(int rows, int cols, int first_layer);
/* Re-orders a row-major (rows x cols) fp32 weight matrix into MFMA A-fragment order. */
int pda_sa_mlp_pack_weights(const float *w, float *wf, int rows, int cols, int first_layer,
                            pda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PDA_POINTNET2_H */

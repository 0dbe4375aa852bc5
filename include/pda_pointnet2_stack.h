/* pda_pointnet2_stack.h -- C ABI of the pointnet2_stack operator set
 * (/root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31): scenes of different
 * sizes stacked along the point axis and described by int32 *_batch_cnt arrays (device memory), features
 * point-major (N, C).  Same library and conventions as pda_pointnet2.h.  voxel_query and vector_pool
 * (PV-RCNN++ / Voxel-RCNN only, :14,27-31) are not provided.  The batch-layout
 * farthest_point_sampling_wrapper of that module (:16) is pda_furthest_point_sampling.
 */
#ifndef PDA_POINTNET2_STACK_H
#define PDA_POINTNET2_STACK_H
#include "pda_pointnet2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces ball_query_wrapper_stack (ball_query.cpp:22-40 -> ball_query_gpu.cu:16-66): new_xyz (M,3),
 * xyz (N,3) -> idx (M,nsample) indices LOCAL to the centre's scene, first hit pre-fills the row, a ball
 * with no hit gets idx[0] = -1 and is otherwise untouched (caller zero-fills, pointnet2_utils.py:32). */
int pda_stack_ball_query(const float *new_xyz, const int32_t *new_xyz_batch_cnt, const float *xyz,
                         const int32_t *xyz_batch_cnt, int32_t *idx, int b, int m, float radius,
                         int nsample, pda_stream_t stream);
/* replace group_points_wrapper_stack / group_points_grad_wrapper_stack (group_points.cpp:41-68 / :19-38 ->
 * group_points_gpu.cu:71-102 / :15-45): features (N,C), idx (M,nsample) local -> out (M,C,nsample);
 * grad_features (N,C) pre-zeroed, accumulated with atomics. */
int pda_stack_group_points(const float *features, const int32_t *features_batch_cnt, const int32_t *idx,
                           const int32_t *idx_batch_cnt, float *out, int b, int m, int c, int nsample,
                           pda_stream_t stream);
int pda_stack_group_points_grad(const float *grad_out, const int32_t *idx, const int32_t *idx_batch_cnt,
                                const int32_t *features_batch_cnt, float *grad_features, int b, int m,
                                int c, int n, int nsample, pda_stream_t stream);
/* replaces stack_farthest_point_sampling_wrapper (sampling.cpp:38-58 -> sampling_gpu.cu:188-345): xyz (N,3),
 * temp (N) pre-filled 1e10, num_sampled_points (B) -> idx (sum of num_sampled_points) GLOBAL indices;
 * the reference's fixed block size 1024 fixes the tie-break order. */
int pda_stack_furthest_point_sampling(const float *xyz, float *temp, const int32_t *xyz_batch_cnt,
                                      int32_t *idx, const int32_t *num_sampled_points, int b,
                                      pda_stream_t stream);
/* replaces three_nn_wrapper_stack (interpolate.cpp:24-47 -> interpolate_gpu.cu:16-99): unknown (N,3),
 * known (M,3) -> dist2 (N,3) squared distances, idx (N,3) GLOBAL known indices. */
int pda_stack_three_nn(const float *unknown, const int32_t *unknown_batch_cnt, const float *known,
                       const int32_t *known_batch_cnt, float *dist2, int32_t *idx, int b, int n,
                       pda_stream_t stream);
/* replace three_interpolate_wrapper_stack / three_interpolate_grad_wrapper_stack (interpolate.cpp:50-107 ->
 * interpolate_gpu.cu:107-189): features (M,C), idx / weight (N,3) -> out (N,C); grad_features (M,C) pre-zeroed. */
int pda_stack_three_interpolate(const float *features, const int32_t *idx, const float *weight, float *out,
                                int n, int c, pda_stream_t stream);
int pda_stack_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                                     float *grad_features, int n, int c, pda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

"""Drop-in for the reference's `pointnet2_stack_cuda` extension
(pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31): same entry-point names, positional
arguments and return conventions, on libpda_pointnet2.so (include/pda_pointnet2_stack.h).
voxel_query / vector_pool entry points are not provided (PV-RCNN++ / Voxel-RCNN only)."""
from . import pointnet2_batch_cuda as _batch
from .pointnet2_batch_cuda import F32, I32, _call, _chk, _numel_ok


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    _numel_ok(new_xyz, M * 3, "new_xyz"); _numel_ok(idx, M * nsample, "idx")
    _numel_ok(new_xyz_batch_cnt, B, "new_xyz_batch_cnt"); _numel_ok(xyz_batch_cnt, B, "xyz_batch_cnt")
    _call("pda_stack_ball_query", xyz, _chk(new_xyz, "new_xyz", F32), _chk(new_xyz_batch_cnt, "new_xyz_batch_cnt", I32),
          _chk(xyz, "xyz", F32), _chk(xyz_batch_cnt, "xyz_batch_cnt", I32), _chk(idx, "idx", I32), B, M, float(radius), nsample)
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _numel_ok(idx, M * nsample, "idx"); _numel_ok(out, M * C * nsample, "out")
    _numel_ok(features_batch_cnt, B, "features_batch_cnt"); _numel_ok(idx_batch_cnt, B, "idx_batch_cnt")
    _call("pda_stack_group_points", features, _chk(features, "features", F32), _chk(features_batch_cnt, "features_batch_cnt", I32),
          _chk(idx, "idx", I32), _chk(idx_batch_cnt, "idx_batch_cnt", I32), _chk(out, "out", F32), B, M, C, nsample)
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    _numel_ok(grad_out, M * C * nsample, "grad_out"); _numel_ok(idx, M * nsample, "idx"); _numel_ok(grad_features, N * C, "grad_features")
    _call("pda_stack_group_points_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(idx, "idx", I32),
          _chk(idx_batch_cnt, "idx_batch_cnt", I32), _chk(features_batch_cnt, "features_batch_cnt", I32),
          _chk(grad_features, "grad_features", F32), B, M, C, N, nsample)
    return 1


farthest_point_sampling_wrapper = _batch.farthest_point_sampling_wrapper      # (B, N, 3) batch layout (:16)


def stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, idx, num_sampled_points):
    B = xyz_batch_cnt.shape[0]
    _numel_ok(temp, xyz.shape[0], "temp"); _numel_ok(num_sampled_points, B, "num_sampled_points")
    _call("pda_stack_furthest_point_sampling", xyz, _chk(xyz, "xyz", F32), _chk(temp, "temp", F32),
          _chk(xyz_batch_cnt, "xyz_batch_cnt", I32), _chk(idx, "idx", I32), _chk(num_sampled_points, "num_sampled_points", I32), B)
    return 1


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    B, N = unknown_batch_cnt.shape[0], unknown.shape[0]
    _numel_ok(dist2, N * 3, "dist2"); _numel_ok(idx, N * 3, "idx"); _numel_ok(known_batch_cnt, B, "known_batch_cnt")
    _call("pda_stack_three_nn", unknown, _chk(unknown, "unknown", F32), _chk(unknown_batch_cnt, "unknown_batch_cnt", I32),
          _chk(known, "known", F32), _chk(known_batch_cnt, "known_batch_cnt", I32), _chk(dist2, "dist2", F32),
          _chk(idx, "idx", I32), B, N)


def three_interpolate_wrapper(features, idx, weight, out):
    N, C = idx.shape[0], features.shape[1]
    _numel_ok(weight, N * 3, "weight"); _numel_ok(out, N * C, "out")
    _call("pda_stack_three_interpolate", features, _chk(features, "features", F32), _chk(idx, "idx", I32),
          _chk(weight, "weight", F32), _chk(out, "out", F32), N, C)


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    N, C = grad_out.shape[0], grad_out.shape[1]
    _numel_ok(idx, N * 3, "idx"); _numel_ok(weight, N * 3, "weight")
    _call("pda_stack_three_interpolate_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(idx, "idx", I32),
          _chk(weight, "weight", F32), _chk(grad_features, "grad_features", F32), N, C)

"""ctypes binding of oracle/libpda_oracle.so (test infrastructure; see package docstring)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def lib_path():
    # PDA_ORACLE_LIB=libpda_oracle_c0.so: the build with the uncontracted distance expression (tests/test_contract0.py)
    return os.path.join(_HERE, os.environ.get("PDA_ORACLE_LIB") or "libpda_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (recipe: oracle/Makefile): the default build and the uncontracted one."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    for name in ("libpda_oracle.so", "libpda_oracle_c0.so"):
        so = os.path.join(_HERE, name)
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return lib_path()


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            build()
        _LIB = ctypes.CDLL(lib_path())
    return _LIB


def _f(a, shape=None):
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous, \
        "oracle expects C-contiguous float32 numpy arrays"
    if shape is not None:
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
    return a.ctypes.data_as(_f32p)


def _i(a, shape=None):
    assert isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous, \
        "oracle expects C-contiguous int32 numpy arrays"
    if shape is not None:
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
    return a.ctypes.data_as(_i32p)


def opt_n_threads(n):
    return int(_lib().pda_oracle_opt_n_threads(int(n)))


def contract_mode():
    return int(_lib().pda_oracle_contract_mode())


def num_threads():
    return int(_lib().pda_oracle_num_threads())


def set_num_threads(n):
    _lib().pda_oracle_set_num_threads(int(n))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    return _lib().pda_oracle_ball_query(b, n, m, ctypes.c_float(radius), nsample,
                                        _f(new_xyz, (b, m, 3)), _f(xyz, (b, n, 3)),
                                        _i(idx, (b, m, nsample)))


def ellipsoid_query(new_xyz, xyz, e1, e2, e3, nsample):
    """pointnet2_api.cpp:16 / ellipsoid_query.cpp:13-76: allocates (zero-filled) and returns idx (b, m, nsample)."""
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    _lib().pda_oracle_ellipsoid_query(b, n, m, ctypes.c_float(e1), ctypes.c_float(e2), ctypes.c_float(e3), nsample,
                                      _f(new_xyz, (b, m, 3)), _f(xyz, (b, n, 3)), _i(idx, (b, m, nsample)))
    return idx


def ball_query_dilated_wrapper(b, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx):
    return _lib().pda_oracle_ball_query_dilated(b, n, m, ctypes.c_float(max_radius),
                                                ctypes.c_float(min_radius), nsample,
                                                _f(new_xyz, (b, m, 3)), _f(xyz, (b, n, 3)),
                                                _i(idx, (b, m, nsample)))


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    return _lib().pda_oracle_group_points(b, c, n, npoints, nsample, _f(points, (b, c, n)),
                                          _i(idx, (b, npoints, nsample)),
                                          _f(out, (b, c, npoints, nsample)))


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    return _lib().pda_oracle_group_points_grad(b, c, n, npoints, nsample,
                                               _f(grad_out, (b, c, npoints, nsample)),
                                               _i(idx, (b, npoints, nsample)),
                                               _f(grad_points, (b, c, n)))


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    return _lib().pda_oracle_gather_points(b, c, n, npoints, _f(points, (b, c, n)),
                                           _i(idx, (b, npoints)), _f(out, (b, c, npoints)))


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    return _lib().pda_oracle_gather_points_grad(b, c, n, npoints, _f(grad_out, (b, c, npoints)),
                                                _i(idx, (b, npoints)),
                                                _f(grad_points, (b, c, n)))


def farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx):
    return _lib().pda_oracle_furthest_point_sampling(b, n, m, _f(xyz, (b, n, 3)),
                                                     _f(temp, (b, n)), _i(idx, (b, m)))


def furthest_point_sampling_with_dist_wrapper(b, n, m, dist, temp, idx):
    return _lib().pda_oracle_furthest_point_sampling_with_dist(b, n, m, _f(dist, (b, n, n)),
                                                               _f(temp, (b, n)),
                                                               _i(idx, (b, m)))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _lib().pda_oracle_three_nn(b, n, m, _f(unknown, (b, n, 3)), _f(known, (b, m, 3)),
                               _f(dist2, (b, n, 3)), _i(idx, (b, n, 3)))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _lib().pda_oracle_three_interpolate(b, c, m, n, _f(points, (b, c, m)), _i(idx, (b, n, 3)),
                                        _f(weight, (b, n, 3)), _f(out, (b, c, n)))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _lib().pda_oracle_three_interpolate_grad(b, c, n, m, _f(grad_out, (b, c, n)),
                                             _i(idx, (b, n, 3)), _f(weight, (b, n, 3)),
                                             _f(grad_points, (b, c, m)))


def chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda.cpp:22-25; tensors only, sizes from the shapes as in the reference."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    return _lib().pda_oracle_chamfer_forward(b, n, m, _f(xyz1, (b, n, 3)), _f(xyz2, (b, m, 3)), _f(dist1, (b, n)),
                                             _f(dist2, (b, m)), _i(idx1, (b, n)), _i(idx2, (b, m)))


def chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    return _lib().pda_oracle_chamfer_backward(b, n, m, _f(xyz1, (b, n, 3)), _f(xyz2, (b, m, 3)),
                                              _f(gradxyz1, (b, n, 3)), _f(gradxyz2, (b, m, 3)),
                                              _f(graddist1, (b, n)), _f(graddist2, (b, m)),
                                              _i(idx1, (b, n)), _i(idx2, (b, m)))


def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
    """roiaware_pool3d.cpp:98-118 (argument order of the extension: boxes, points, out)."""
    b, t, _ = boxes.shape
    m = pts.shape[1]
    return _lib().pda_oracle_points_in_boxes(b, t, m, _f(boxes, (b, t, 7)), _f(pts, (b, m, 3)),
                                             _i(box_idx_of_points, (b, m)))


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans):
    """iou3d_nms.cpp:40-63."""
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    return _lib().pda_oracle_boxes_bev(na, nb, _f(boxes_a, (na, 7)), _f(boxes_b, (nb, 7)), _f(ans, (na, nb)), 0)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans):
    """iou3d_nms.cpp:65-87."""
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    return _lib().pda_oracle_boxes_bev(na, nb, _f(boxes_a, (na, 7)), _f(boxes_b, (nb, 7)), _f(ans, (na, nb)), 1)


def _nms(boxes, keep, thresh, normal):
    n = boxes.shape[0]
    assert keep.dtype == np.int64 and keep.flags.c_contiguous and keep.shape[0] >= n
    return int(_lib().pda_oracle_nms(n, _f(boxes, (n, 7)), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                     ctypes.c_float(thresh), normal))


def nms_gpu(boxes, keep, thresh):
    """iou3d_nms.cpp:90-138: returns num_to_keep."""
    return _nms(boxes, keep, thresh, 0)


def nms_normal_gpu(boxes, keep, thresh):
    """iou3d_nms.cpp:141-188."""
    return _nms(boxes, keep, thresh, 1)


# ---- pointnet2_stack entry points (pointnet2_stack/src/pointnet2_api.cpp:12-31), same positional order ----
def stack_ball_query_wrapper(b, m, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    return _lib().pda_oracle_ball_query_stack(b, m, ctypes.c_float(radius), nsample, _f(new_xyz, (m, 3)),
                                              _i(new_xyz_batch_cnt, (b,)), _f(xyz), _i(xyz_batch_cnt, (b,)),
                                              _i(idx, (m, nsample)))


def stack_group_points_wrapper(b, m, c, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    return _lib().pda_oracle_group_points_stack(b, m, c, nsample, _f(features), _i(features_batch_cnt, (b,)),
                                                _i(idx, (m, nsample)), _i(idx_batch_cnt, (b,)), _f(out, (m, c, nsample)))


def stack_group_points_grad_wrapper(b, m, c, n, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    return _lib().pda_oracle_group_points_grad_stack(b, m, c, n, nsample, _f(grad_out, (m, c, nsample)), _i(idx, (m, nsample)),
                                                     _i(idx_batch_cnt, (b,)), _i(features_batch_cnt, (b,)),
                                                     _f(grad_features, (n, c)))


def stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, idx, num_sampled_points):
    b = xyz_batch_cnt.shape[0]
    return _lib().pda_oracle_stack_furthest_point_sampling(b, _f(xyz), _f(temp), _i(xyz_batch_cnt, (b,)), _i(idx),
                                                           _i(num_sampled_points, (b,)))


def stack_three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    b, n = unknown_batch_cnt.shape[0], unknown.shape[0]
    _lib().pda_oracle_three_nn_stack(b, n, _f(unknown, (n, 3)), _i(unknown_batch_cnt, (b,)), _f(known),
                                     _i(known_batch_cnt, (b,)), _f(dist2, (n, 3)), _i(idx, (n, 3)))


def stack_three_interpolate_wrapper(features, idx, weight, out):
    n, c = idx.shape[0], features.shape[1]
    _lib().pda_oracle_three_interpolate_stack(n, c, _f(features), _i(idx, (n, 3)), _f(weight, (n, 3)), _f(out, (n, c)))


def stack_three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    n, c = grad_out.shape
    _lib().pda_oracle_three_interpolate_grad_stack(n, c, _f(grad_out, (n, c)), _i(idx, (n, 3)), _f(weight, (n, 3)),
                                                   _f(grad_features))

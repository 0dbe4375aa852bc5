"""Mirror of the part of pcdet/ops/roiaware_pool3d the IA-SSD head uses
(roiaware_pool3d_utils.py:30-43 points_in_boxes_gpu over roiaware_pool3d_cuda.points_in_boxes_gpu,
src/roiaware_pool3d.cpp:98-118) on libpda_pointnet2.so (include/pda_train.h)."""
import torch

from .pointnet2_batch_cuda import F32, I32, _call, _chk, _numel_ok


class roiaware_pool3d_cuda:  # noqa: N801  (the reference's extension module name)
    @staticmethod
    def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
        """(B,T,7) boxes, (B,M,3) points, (B,M) int32 pre-filled with -1 -> 1."""
        b, t, m = boxes.shape[0], boxes.shape[1], pts.shape[1]
        _numel_ok(boxes, b * t * 7, "boxes"); _numel_ok(pts, b * m * 3, "pts")
        _numel_ok(box_idx_of_points, b * m, "box_idx_of_points")
        _call("pda_points_in_boxes", pts, _chk(boxes, "boxes", F32), _chk(pts, "pts", F32),
              _chk(box_idx_of_points, "box_idx_of_points", I32), b, t, m)
        return 1


def points_in_boxes_gpu(points, boxes):
    """
    :param points: (B, M, 3)
    :param boxes: (B, T, 7), num_valid_boxes <= T
    :return box_idxs_of_pts: (B, M), default background = -1
    """
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    batch_size, num_points, _ = points.shape
    box_idxs_of_pts = points.new_zeros((batch_size, num_points), dtype=torch.int).fill_(-1)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), box_idxs_of_pts)
    return box_idxs_of_pts


def assign_point_targets(gt_boxes, in_box, in_ext, mode, single_class):
    """MI355X extension (csrc/head_targets.hip): the point-wise part of IASSD_head.assign_stack_targets_IASSD after the two
    points_in_boxes queries, one launch.  gt_boxes (B,T,8), in_box / in_ext (B,N) int32 -> labels (B*N) int64,
    box index (B*N) int64, gt_of_points (B*N,8)."""
    B, T = gt_boxes.shape[0], gt_boxes.shape[1]
    N = in_box.shape[1]
    _numel_ok(gt_boxes, B * T * 8, "gt_boxes"); _numel_ok(in_ext, B * N, "in_ext")
    labels = torch.empty((B * N,), dtype=torch.int64, device=gt_boxes.device)
    idx = torch.empty((B * N,), dtype=torch.int64, device=gt_boxes.device)
    gt_of = torch.empty((B * N, 8), dtype=torch.float32, device=gt_boxes.device)
    _call("pda_assign_point_targets", gt_boxes, _chk(gt_boxes, "gt_boxes", F32), _chk(in_box, "in_box", I32), _chk(in_ext, "in_ext", I32),
          _chk(labels, "labels", torch.int64), _chk(idx, "box_idx", torch.int64), _chk(gt_of, "gt_of_points", F32), B, N, T, int(mode),
          int(bool(single_class)))
    return labels, idx, gt_of


def sa_gaussian_mask(coords, gt_of_points, labels):
    """MI355X extension: soft instance labels (IASSD_head.py:889-963) for the points coords (P, 1+3[+...]) [bs, x, y, z, ...]
    against their boxes gt_of_points (P, 8); 0 where labels <= 0.  One launch."""
    P = gt_of_points.shape[0]
    _numel_ok(coords, P * coords.shape[-1], "coords"); _numel_ok(labels, P, "labels")
    out = torch.empty((P,), dtype=torch.float32, device=gt_of_points.device)
    _call("pda_sa_gaussian_mask", gt_of_points, _chk(coords, "coords", F32), int(coords.shape[-1]), 1, _chk(gt_of_points, "gt_of_points", F32),
          _chk(labels, "labels", torch.int64), _chk(out, "out", F32), P)
    return out

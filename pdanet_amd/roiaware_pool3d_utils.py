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

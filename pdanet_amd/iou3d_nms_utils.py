"""Mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py (boxes_iou_bev :31-46, boxes_iou3d_gpu :49-83,
nms_gpu :86-101, nms_normal_gpu :104-116) on libpda_pointnet2.so (include/pda_train.h), plus the
batched, sync-free form the detector's post-processing uses (`nms_batched`)."""
import torch

from . import _lib
from .pointnet2_batch_cuda import F32, I32, _call, _chk, _numel_ok


class iou3d_nms_cuda:  # noqa: N801  (the reference's extension module name; src/iou3d_nms_api.cpp:11-16)
    @staticmethod
    def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
        na, nb = boxes_a.shape[0], boxes_b.shape[0]
        _numel_ok(boxes_a, na * 7, "boxes_a"); _numel_ok(boxes_b, nb * 7, "boxes_b"); _numel_ok(ans_overlap, na * nb, "ans_overlap")
        _call("pda_boxes_overlap_bev", boxes_a, _chk(boxes_a, "boxes_a", F32), _chk(boxes_b, "boxes_b", F32),
              _chk(ans_overlap, "ans_overlap", F32), na, nb)
        return 1

    @staticmethod
    def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
        na, nb = boxes_a.shape[0], boxes_b.shape[0]
        _numel_ok(boxes_a, na * 7, "boxes_a"); _numel_ok(boxes_b, nb * 7, "boxes_b"); _numel_ok(ans_iou, na * nb, "ans_iou")
        _call("pda_boxes_iou_bev", boxes_a, _chk(boxes_a, "boxes_a", F32), _chk(boxes_b, "boxes_b", F32),
              _chk(ans_iou, "ans_iou", F32), na, nb)
        return 1


def nms_batched(boxes, thresh, num_valid=None, normal=False):
    """boxes (B, N, 7), every scene sorted by descending score; num_valid (B) int32 device tensor or None.
    Returns keep (B, N) int64 (kept indices in score order, -1 padded) and num_keep (B) int32, on the
    device, without synchronising."""
    B, N = boxes.shape[0], boxes.shape[1]
    _numel_ok(boxes, B * N * 7, "boxes")
    keep = torch.empty((B, N), dtype=torch.int64, device=boxes.device)
    num_keep = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    words = int(_lib.load().pda_nms_mask_words(N))
    scratch = torch.empty((B * max(words, 1),), dtype=torch.int64, device=boxes.device)
    nv = None if num_valid is None else _chk(num_valid, "num_valid", I32)
    _call("pda_nms_bev", boxes, _chk(boxes, "boxes", F32), nv, _chk(keep, "keep", torch.int64), _chk(num_keep, "num_keep", I32),
          _chk(scratch, "scratch", torch.int64), B, N, float(thresh), int(bool(normal)))
    return keep, num_keep


def boxes_iou_bev(boxes_a, boxes_b):
    """(N, 7), (M, 7) -> (N, M) rotated BEV IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans = boxes_a.new_zeros((boxes_a.shape[0], boxes_b.shape[0]))
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N, 7), (M, 7) -> (N, M) 3-D IoU = BEV overlap x height overlap / union volume (:49-83)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max, a_min = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max, b_min = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1), (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_a.new_zeros((boxes_a.shape[0], boxes_b.shape[0]))
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms_one(boxes, scores, thresh, pre_maxsize, normal):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, num = nms_batched(boxes[order].contiguous().unsqueeze(0), thresh, normal=normal)
    return order[keep[0, :int(num[0])]].contiguous(), None     # the reference's return shape needs the count on the host


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """(N, 7), (N) -> indices kept by rotated-IoU NMS, in score order (:86-101).  Synchronises (variable-length
    result); `nms_batched` is the sync-free form."""
    return _nms_one(boxes, scores, thresh, pre_maxsize, False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned variant (:104-116)."""
    return _nms_one(boxes, scores, thresh, None, True)

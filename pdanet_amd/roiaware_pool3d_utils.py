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


def head_assign_targets(points, gt_boxes, extra_width, mode, single_class, mean_size=None, bins=0, ret_box_labels=False):
    """MI355X extension (csrc/head_targets.hip): assign_stack_targets_IASSD for one point set in ONE launch.  points (B*N, 4)
    [bs, x, y, z] scene-major, gt_boxes (B, T, 8), extra_width 3 python floats -> labels (B*N) int64, box index (B*N) int64,
    gt_of_points (B*N, 8), box-coder targets (B*N, 8) or None."""
    import ctypes
    B, T = gt_boxes.shape[0], gt_boxes.shape[1]
    P = points.shape[0]
    N = P // B
    dev = gt_boxes.device
    labels = torch.empty((P,), dtype=torch.int64, device=dev)
    idx = torch.empty((P,), dtype=torch.int64, device=dev)
    gt_of = torch.empty((P, 8), dtype=torch.float32, device=dev)
    box_labels = torch.empty((P, 8), dtype=torch.float32, device=dev) if ret_box_labels else None
    ew = (ctypes.c_float * 3)(*[float(w) for w in extra_width])
    _call("pda_head_assign_targets", gt_boxes, _chk(points, "points", F32), int(points.shape[1]), 1, _chk(gt_boxes, "gt_boxes", F32), ew,
          _chk(labels, "labels", torch.int64), _chk(idx, "box_idx", torch.int64), _chk(gt_of, "gt_of_points", F32),
          None if box_labels is None else _chk(box_labels, "box_labels", F32), None if mean_size is None else _chk(mean_size, "mean_size", F32),
          int(bins), B, N, T, int(mode), int(bool(single_class)))
    return labels, idx, gt_of, box_labels


def sa_gaussian_mask(coords, gt_of_points, labels):
    """MI355X extension: soft instance labels (IASSD_head.py:889-963) for the points coords (P, 1+3[+...]) [bs, x, y, z, ...]
    against their boxes gt_of_points (P, 8); 0 where labels <= 0.  One launch."""
    P = gt_of_points.shape[0]
    _numel_ok(coords, P * coords.shape[-1], "coords"); _numel_ok(labels, P, "labels")
    out = torch.empty((P,), dtype=torch.float32, device=gt_of_points.device)
    _call("pda_sa_gaussian_mask", gt_of_points, _chk(coords, "coords", F32), int(coords.shape[-1]), 1, _chk(gt_of_points, "gt_of_points", F32),
          _chk(labels, "labels", torch.int64), _chk(out, "out", F32), P)
    return out


# ---- loss terms of the IA-SSD head, one launch each (csrc/head_loss.hip) ------------------------------------------------------
I64 = torch.int64


def head_centerness(centers, gt_of_points, labels):
    """generate_center_ness_mask (IASSD_head.py:795-817): centers (n, 4) [bs, x, y, z], gt (n, 8), labels (n) -> (n)."""
    n = gt_of_points.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=gt_of_points.device)
    _call("pda_head_centerness", gt_of_points, _chk(centers, "centers", F32), _chk(gt_of_points, "gt_of_points", F32),
          _chk(labels, "labels", I64), _chk(out, "out", F32), n)
    return out


class _ScaledGrad(torch.autograd.Function):
    """Shared shape of the fused loss nodes: forward(prediction tensors..., run) calls `run()`, which launches the kernel and
    returns (outputs, gradients of the term wrt each prediction tensor); backward multiplies the stored gradients by the
    incoming scalar of output 0 (the other outputs are logged values without gradient)."""

    @staticmethod
    def forward(ctx, run, *preds):
        outs, grads = run()
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, *unused):
        return (None,) + tuple(g * g0 for g in ctx.saved_tensors)


def head_cls_loss(preds, col0, num_class, labels, soft, scale):
    """(loss, #positives) of WeightedClassificationLoss with soft one-hot targets over the rows of preds (n, cols); the class
    logits are columns col0 .. col0 + num_class - 1.  Differentiable wrt preds."""
    p2 = preds.reshape(-1, preds.shape[-1])
    p2 = p2 if p2.is_contiguous() else p2.contiguous()
    n, cols = p2.shape

    def run():
        out = torch.empty((2,), dtype=torch.float32, device=p2.device)
        grad = torch.empty_like(p2)
        _call("pda_head_cls_loss", p2, _chk(p2.detach(), "preds", F32), cols, int(col0), int(num_class), _chk(labels, "labels", I64),
              None if soft is None else _chk(soft, "soft", F32), n, float(scale), _chk(out, "out", F32), _chk(grad, "grad", F32))
        return (out[0], out[1]), (grad.view(preds.shape),)
    return _ScaledGrad.apply(run, preds)


def head_box_loss(preds, labels, cls_labels, code_weights, beta, bins, dir_weight, box_weight):
    """(total, xyzwhl, ori_bin * dir_weight, ori_res) of get_center_box_binori_layer_loss; differentiable wrt preds (n, 6 + 2 bins)."""
    p2 = preds if preds.is_contiguous() else preds.contiguous()
    n = p2.shape[0]

    def run():
        out = torch.empty((4,), dtype=torch.float32, device=p2.device)
        grad = torch.empty_like(p2)
        _call("pda_head_box_loss", p2, _chk(p2.detach(), "preds", F32), _chk(labels, "labels", F32), _chk(cls_labels, "cls_labels", I64),
              None if code_weights is None else _chk(code_weights, "code_weights", F32), float(beta), int(bins), float(dir_weight),
              float(box_weight), n, _chk(out, "out", F32), _chk(grad, "grad", F32))
        return (out[0], out[1], out[2], out[3]), (grad,)
    return _ScaledGrad.apply(run, preds)


def head_vote_loss(mode, origin, offsets, key, gt, batch_size, boxes, num_class, weight):
    """The vote loss (mode 0: LOSS_VOTE_TYPE none, mode 1: ver2); differentiable wrt offsets (n, 4) [bs, dx, dy, dz]."""
    o2 = offsets if offsets.is_contiguous() else offsets.contiguous()
    n = o2.shape[0]

    def run():
        out = torch.empty((1,), dtype=torch.float32, device=o2.device)
        grad = torch.empty_like(o2)
        _call("pda_head_vote_loss", o2, int(mode), _chk(origin, "origin", F32), _chk(o2.detach(), "offsets", F32), _chk(key, "key", I64),
              _chk(gt, "gt", F32), int(batch_size), int(boxes), int(num_class), float(weight), n, _chk(out, "out", F32), _chk(grad, "grad", F32))
        return (out[0],), (grad,)
    return _ScaledGrad.apply(run, offsets)[0]


def head_corner_loss(box_preds, centers, cls_preds, gt, cls_labels, mean_size, bins, weight):
    """get_corner_layer_loss incl. the box decode; differentiable wrt box_preds (n, 6 + 2 bins) and centers (n, 4)."""
    b2 = box_preds if box_preds.is_contiguous() else box_preds.contiguous()
    c2 = centers if centers.is_contiguous() else centers.contiguous()
    k2 = cls_preds.detach().reshape(b2.shape[0], -1).contiguous()
    n = b2.shape[0]

    def run():
        out = torch.empty((1,), dtype=torch.float32, device=b2.device)
        gb, gc = torch.empty_like(b2), torch.empty_like(c2)
        _call("pda_head_corner_loss", b2, _chk(b2.detach(), "box_preds", F32), _chk(c2.detach(), "centers", F32), _chk(k2, "cls_preds", F32),
              int(k2.shape[1]), _chk(gt, "gt", F32), _chk(cls_labels, "cls_labels", I64),
              None if mean_size is None else _chk(mean_size, "mean_size", F32), int(bins), float(weight), n, _chk(out, "out", F32),
              _chk(gb, "grad_box", F32), _chk(gc, "grad_centers", F32))
        return (out[0],), (gb, gc)
    return _ScaledGrad.apply(run, box_preds, centers)[0]

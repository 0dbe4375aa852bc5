"""The element-wise losses the PDA-SSD head is configured with, in torch (pcdet/utils/loss_utils.py:75-130
`WeightedClassificationLoss`, :133-194 `WeightedSmoothL1Loss`, :340-363 `get_corner_loss_lidar`).  On the GPU the training
path computes these terms inside csrc/head_loss.hip (one launch per term, gradient included); this module is the formulation
those kernels are checked against (tests/test_iassd_head.py) and what runs when the fused path is switched off."""
import math

import torch
import torch.nn as nn

from . import box_utils


def _stable_bce_logits(logits, targets):
    """max(x, 0) - x t + log(1 + exp(-|x|)): sigmoid cross-entropy on logits without overflow (:90-108)."""
    return logits.clamp(min=0) - logits * targets + torch.log1p(torch.exp(-logits.abs()))


def _huber(x, beta):
    """0.5 x^2 / beta below beta, |x| - 0.5 beta above; plain |x| for a vanishing beta (:157-165)."""
    a = x.abs()
    if beta < 1e-5:
        return a
    return torch.where(a < beta, 0.5 * a ** 2 / beta, a - 0.5 * beta)


class WeightedClassificationLoss(nn.Module):
    sigmoid_cross_entropy_with_logits = staticmethod(_stable_bce_logits)

    def forward(self, input, target, weights=None, reduction='none'):
        loss = _stable_bce_logits(input, target)
        if weights is not None:
            w = weights.unsqueeze(-1) if weights.dim() == loss.dim() - 1 else weights      # per-anchor weights broadcast over classes
            assert w.dim() == loss.dim()
            loss = w * loss
        if reduction == 'none':
            return loss
        return loss.sum(dim=-1) if reduction == 'sum' else loss.mean(dim=-1)


class WeightedSmoothL1Loss(nn.Module):
    smooth_l1_loss = staticmethod(_huber)

    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        # a non-persistent buffer follows the module across devices (the reference keeps a bare .cuda() tensor, :153-155)
        self.register_buffer("code_weights", None if code_weights is None else torch.tensor(code_weights, dtype=torch.float32),
                             persistent=False)

    def forward(self, input, target, weights=None):
        """(B, N, C) codes.  A NaN target means "no target for this code": it contributes neither loss nor gradient."""
        diff = input - torch.where(torch.isnan(target), input, target)
        if self.code_weights is not None:
            diff = diff * self.code_weights.view(1, 1, -1)
        loss = _huber(diff, self.beta)
        if weights is not None:
            assert weights.shape[:2] == loss.shape[:2]
            loss = loss * weights.unsqueeze(-1)
        return loss


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """(N, 7), (N, 7) -> (N): Huber(1) of the eight corner distances, per corner the smaller of the distances to the box
    and to its heading-flipped twin, averaged over the corners (:340-363)."""
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    corners = box_utils.boxes_to_corners_3d(pred_bbox3d)
    twin = gt_bbox3d.clone()
    twin[:, 6] += math.pi
    d_box = (corners - box_utils.boxes_to_corners_3d(gt_bbox3d)).norm(dim=2)
    d_twin = (corners - box_utils.boxes_to_corners_3d(twin)).norm(dim=2)
    return _huber(torch.minimum(d_box, d_twin), 1.0).mean(dim=1)

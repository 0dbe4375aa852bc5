"""The loss functions PDA-SSD's head is configured with (pcdet/utils/loss_utils.py:75-130
WeightedClassificationLoss, :133-194 WeightedSmoothL1Loss, :340-363 get_corner_loss_lidar)."""
import numpy as np
import torch
import torch.nn as nn

from . import box_utils


class WeightedClassificationLoss(nn.Module):
    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights=None, reduction='none'):
        loss = self.sigmoid_cross_entropy_with_logits(input, target)
        if weights is not None:
            if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
                weights = weights.unsqueeze(-1)
            assert weights.dim() == loss.dim()
            loss = weights * loss
        if reduction == 'sum':
            loss = loss.sum(dim=-1)
        elif reduction == 'mean':
            loss = loss.mean(dim=-1)
        return loss


class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        # the reference keeps a plain .cuda() tensor (:153-155); a non-persistent buffer follows the module
        self.register_buffer("code_weights", None if code_weights is None else
                             torch.from_numpy(np.array(code_weights, dtype=np.float32)), persistent=False)

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights.view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """loss_utils.py:340-363: (N, 7), (N, 7) -> (N) smooth-L1 (beta 1) of the corner distances, the
    better of the box and its heading-flipped twin."""
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    pred = box_utils.boxes_to_corners_3d(pred_bbox3d)
    gt = box_utils.boxes_to_corners_3d(gt_bbox3d)
    flip = gt_bbox3d.clone()
    flip[:, 6] += np.pi
    gt_flip = box_utils.boxes_to_corners_3d(flip)
    dist = torch.min(torch.norm(pred - gt, dim=2), torch.norm(pred - gt_flip, dim=2))
    return WeightedSmoothL1Loss.smooth_l1_loss(dist, beta=1.0).mean(dim=1)

"""Box coder of both PDA-SSD yamls (`PointResidual_BinOri_Coder`, pcdet/utils/box_coder_utils.py:224-319): a box is coded
against the point that predicts it as

    [ (centre - point) / (d, d, h) | log(size / (l, w, h)) | heading bin | heading residual in [-1, 1] | extras ]

with (l, w, h) the mean size of the box's class and d = sqrt(l^2 + w^2) (or, without mean sizes, plain differences and
log sizes).  The heading is one of `bin_size` equal sectors of [-pi, pi) plus the offset from the sector's middle in
units of half a sector.  Written on (N, 3) blocks; element for element the reference's arithmetic (the head's golden
tensors pin it, tests/test_iassd_head.py).  On the GPU the training path never decodes (csrc/head_loss.hip carries the
decode inside the corner loss); this module is the eval decode and the reference formulation the kernels are checked against."""
import math

import torch


class PointResidual_BinOri_Coder(object):  # noqa: N801 (the yaml names the class)
    def __init__(self, code_size=8, use_mean_size=True, **kwargs):
        self.bin_size = kwargs.get('bin_size', 12)       # (the yaml's 'angle_bin_num' key is not read by the reference, :227)
        self.code_size = 6 + 2 * self.bin_size
        self.bin_inter = 2 * math.pi / self.bin_size
        self.use_mean_size = use_mean_size
        self.mean_size = None
        if use_mean_size:
            self.mean_size = torch.tensor(kwargs['mean_size'], dtype=torch.float32)
            assert self.mean_size.min() > 0

    def _mean(self, like):
        """The (num_class, 3) mean sizes on `like`'s device (the reference moves them to the GPU at construction, :234)."""
        if self.mean_size.device != like.device:
            self.mean_size = self.mean_size.to(like.device)
        return self.mean_size

    def _scales(self, classes, like):
        """Per-row divisors of the centre residual (d, d, h) and of the size ratio (l, w, h), or None without mean sizes."""
        if not self.use_mean_size:
            return None, None
        anchor = self._mean(like)[classes - 1]                                  # classes are 1-based
        diag = torch.sqrt(anchor[..., 0:1] ** 2 + anchor[..., 1:2] ** 2)
        return torch.cat([diag, diag, anchor[..., 2:3]], dim=-1), anchor

    def encode_torch(self, gt_boxes, points, gt_classes=None):
        """(N, 7 + C) boxes, (N, 3) points, (N) classes in [1, num_class] -> (N, 8 + C) codes."""
        size = torch.clamp_min(gt_boxes[:, 3:6], min=1e-5)
        ctr_div, anchor = self._scales(gt_classes, gt_boxes)
        offset = gt_boxes[:, 0:3] - points
        if anchor is not None:
            offset, size = offset / ctr_div, size / anchor
        heading = torch.clamp(gt_boxes[:, 6:7], max=math.pi - 1e-5, min=-math.pi + 1e-5) + math.pi     # in (0, 2 pi)
        sector = torch.floor(heading / self.bin_inter)
        within = (heading - (sector * self.bin_inter + self.bin_inter / 2)) / (self.bin_inter / 2)
        return torch.cat([offset, torch.log(size), sector, within, gt_boxes[:, 7:]], dim=-1)

    def decode_torch(self, box_encodings, points, pred_classes=None):
        """(N, 6 + 2 bin_size) codes, (N, 3) points -> (N, 7) boxes; the heading bin is the arg-max of the bin scores."""
        nb = self.bin_size
        ctr_div, anchor = self._scales(pred_classes, box_encodings)
        centre, size = box_encodings[..., 0:3], torch.exp(box_encodings[..., 3:6])
        if anchor is not None:
            centre, size = centre * ctr_div, size * anchor
        centre = centre + points
        sector = torch.argmax(box_encodings[..., 6:6 + nb], dim=-1, keepdim=True)
        within = torch.gather(box_encodings[..., 6 + nb:6 + 2 * nb], -1, sector)
        heading = sector.float() * self.bin_inter - math.pi + self.bin_inter / 2
        return torch.cat([centre, size, heading + within * (self.bin_inter / 2)], dim=-1)

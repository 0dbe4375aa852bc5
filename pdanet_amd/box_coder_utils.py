"""PointResidual_BinOri_Coder (pcdet/utils/box_coder_utils.py:224-319), the box coder of both
PDA-SSD yamls: centre/size residuals (optionally relative to a per-class mean size) + heading as
one of `bin_size` bins and a residual in [-1, 1]."""
import numpy as np
import torch


class PointResidual_BinOri_Coder(object):  # noqa: N801 (reference name, looked up from the yaml)
    def __init__(self, code_size=8, use_mean_size=True, **kwargs):
        self.bin_size = kwargs.get('bin_size', 12)       # the yaml's 'angle_bin_num' key is not read (:227)
        self.code_size = 6 + 2 * self.bin_size
        self.bin_inter = 2 * np.pi / self.bin_size
        self.use_mean_size = use_mean_size
        self.mean_size = None
        if self.use_mean_size:
            self.mean_size = torch.from_numpy(np.array(kwargs['mean_size'])).float()
            assert self.mean_size.min() > 0

    def _mean(self, like):
        if self.mean_size.device != like.device:
            self.mean_size = self.mean_size.to(like.device)  # the reference does .cuda() at construction (:234)
        return self.mean_size

    def encode_torch(self, gt_boxes, points, gt_classes=None):
        """(N, 7+C) boxes, (N, 3) points, (N) classes in [1, num_class] -> (N, 8+C)."""
        gt_boxes = gt_boxes.clone()
        gt_boxes[:, 3:6] = torch.clamp_min(gt_boxes[:, 3:6], min=1e-5)
        xg, yg, zg, dxg, dyg, dzg, rg, *cgs = torch.split(gt_boxes, 1, dim=-1)
        xa, ya, za = torch.split(points, 1, dim=-1)
        if self.use_mean_size:
            anchor = self._mean(gt_boxes)[gt_classes - 1]
            dxa, dya, dza = torch.split(anchor, 1, dim=-1)
            diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
            xt, yt, zt = (xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / dza
            dxt, dyt, dzt = torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza)
        else:
            xt, yt, zt = xg - xa, yg - ya, zg - za
            dxt, dyt, dzt = torch.log(dxg), torch.log(dyg), torch.log(dzg)
        rg = torch.clamp(rg, max=np.pi - 1e-5, min=-np.pi + 1e-5)
        bin_id = torch.floor((rg + np.pi) / self.bin_inter)
        bin_res = ((rg + np.pi) - (bin_id * self.bin_inter + self.bin_inter / 2)) / (self.bin_inter / 2)
        return torch.cat([xt, yt, zt, dxt, dyt, dzt, bin_id, bin_res, *cgs], dim=-1)

    def decode_torch(self, box_encodings, points, pred_classes=None):
        """(N, 6 + 2*bin_size) encodings, (N, 3) points -> (N, 7) boxes."""
        xt, yt, zt, dxt, dyt, dzt = torch.split(box_encodings[..., :6], 1, dim=-1)
        xa, ya, za = torch.split(points, 1, dim=-1)
        if self.use_mean_size:
            anchor = self._mean(box_encodings)[pred_classes - 1]
            dxa, dya, dza = torch.split(anchor, 1, dim=-1)
            diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
            xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * dza + za
            dxg, dyg, dzg = torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza
        else:
            xg, yg, zg = xt + xa, yt + ya, zt + za
            dxg, dyg, dzg = torch.split(torch.exp(box_encodings[..., 3:6]), 1, dim=-1)
        bin_id = box_encodings[..., 6:6 + self.bin_size]
        bin_res = box_encodings[..., 6 + self.bin_size:]
        _, bin_id = torch.max(bin_id, dim=-1)
        one_hot = torch.nn.functional.one_hot(bin_id.long(), self.bin_size)
        bin_res = torch.sum(bin_res * one_hot.float(), dim=-1)
        rg = bin_id.float() * self.bin_inter - np.pi + self.bin_inter / 2
        rg = (rg + bin_res * (self.bin_inter / 2)).unsqueeze(-1)
        return torch.cat([xg, yg, zg, dxg, dyg, dzg, rg], dim=-1)

"""IASSD detector = IASSD_Backbone + IASSD_Head (pcdet/models/detectors/IASSD.py:3-27,
detector3d_template.py:45-49 module names `backbone_3d` / `point_head`): training forward (loss) and
inference forward (batched NMS post-processing; recall bookkeeping is eval tooling and not built)."""
import torch
import torch.nn as nn

from . import config, model_nms_utils
from .backbone import IASSD_Backbone
from .iassd_head import IASSD_Head


class _HeadLoss(nn.Module):
    """IASSD_Head.forward + get_loss (training) as a function of TENSORS only: the form torch.cuda.make_graphed_callables
    captures.  Argument order: centers_features, centers, centers_origin, ctr_offsets, gt_boxes, then the non-empty
    sa_ins_preds entries, then encoder_coords; returns (loss, *log values in `self.keys` order)."""

    def __init__(self, head, batch_size, sa_slots, n_coords):
        super().__init__()
        self.head, self.batch_size, self.sa_slots, self.n_coords = head, batch_size, list(sa_slots), n_coords
        self.keys = None

    def forward(self, centers_features, centers, centers_origin, ctr_offsets, gt_boxes, *rest):
        n_sa = sum(self.sa_slots)
        it = iter(rest[:n_sa])
        bd = {'batch_size': self.batch_size, 'centers_features': centers_features, 'centers': centers,
              'centers_origin': centers_origin, 'ctr_offsets': ctr_offsets, 'gt_boxes': gt_boxes,
              'sa_ins_preds': [next(it) if has else [] for has in self.sa_slots],
              'encoder_coords': list(rest[n_sa:n_sa + self.n_coords])}
        self.head(bd)
        loss, tb = self.head.get_loss()
        if self.keys is None:
            self.keys = [k for k, v in tb.items() if isinstance(v, torch.Tensor)]
        return (loss,) + tuple(tb[k] for k in self.keys)


class IASSD(nn.Module):
    def __init__(self, model_cfg, num_class, num_point_features):
        super().__init__()
        self.backbone_3d = IASSD_Backbone(model_cfg["BACKBONE_3D"], num_class=num_class, input_channels=num_point_features)
        self.point_head = IASSD_Head(num_class=num_class, input_channels=self.backbone_3d.num_point_features,
                                     model_cfg=model_cfg["POINT_HEAD"])
        self.module_list = [self.backbone_3d, self.point_head]
        self.model_cfg, self.num_class = model_cfg, num_class
        # graph_head: replay the head + losses (target assignment, ~400 small launches forward and backward, static
        # shapes, no host synchronisation) as two hipGraphs instead of enqueueing them from Python every iteration.
        # Off by default: capture runs a few warm-up iterations of the head first (its BatchNorm running statistics
        # move, `num_batches_tracked` is not advanced by replays), so exact-trajectory tests use the eager form.
        self.graph_head = False
        self._graphed = None

    def _head_loss_graphed(self, bd):
        sa = bd['sa_ins_preds']
        args = [bd['centers_features'], bd['centers'], bd['centers_origin'], bd['ctr_offsets'], bd['gt_boxes']]
        args += [p for p in sa if isinstance(p, torch.Tensor)] + list(bd['encoder_coords'])
        args = [a.contiguous() for a in args]
        key = tuple((tuple(a.shape), a.requires_grad) for a in args)
        if self._graphed is None or self._graphed[0] != key:
            fn = _HeadLoss(self.point_head, bd['batch_size'], [isinstance(p, torch.Tensor) for p in sa], len(bd['encoder_coords']))
            sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
            # the warm-up iterations run on a side stream: AccumulateGrad nodes created there trigger a (harmless)
            # stream-mismatch warning on the first real backward; silence it for the capture only
            warn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn is not None:
                warn(False)
            graphed = torch.cuda.make_graphed_callables(fn, sample, allow_unused_input=True)
            self._graphed = (key, graphed, fn)
        _, graphed, fn = self._graphed
        out = graphed(*args)
        return out[0], dict(zip(fn.keys, (o.detach() for o in out[1:])))

    def forward(self, batch_dict):
        if self.training and self.graph_head and torch.is_grad_enabled():
            batch_dict = self.backbone_3d(batch_dict)
            loss, tb_dict = self._head_loss_graphed(batch_dict)
            return {'loss': loss}, tb_dict, {}
        for m in self.module_list:
            batch_dict = m(batch_dict)
        if self.training:
            loss, tb_dict = self.point_head.get_loss()
            return {'loss': loss}, tb_dict, {}
        padded = model_nms_utils.post_processing(batch_dict, self.model_cfg["POST_PROCESSING"], self.num_class)
        batch_dict['final_padded'] = padded            # device tensors, no synchronisation so far
        return model_nms_utils.to_pred_dicts(padded), {}   # (pred_dicts, recall_dicts): detectors/IASSD.py:20-22


def build_detector(cfg_path="once_pda_ssd.yaml"):
    cfg = config.load_yaml(cfg_path)
    model = IASSD(cfg["MODEL"], num_class=len(cfg["CLASS_NAMES"]),
                  num_point_features=cfg["DATA_CONFIG"]["NUM_POINT_FEATURES"])
    return model, cfg

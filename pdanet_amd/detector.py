"""IASSD detector = IASSD_Backbone + IASSD_Head (pcdet/models/detectors/IASSD.py:3-27,
detector3d_template.py:45-49 module names `backbone_3d` / `point_head`): training forward (loss) and
inference forward (batched NMS post-processing; recall bookkeeping is eval tooling and not built)."""
import torch.nn as nn

from . import config, model_nms_utils
from .backbone import IASSD_Backbone
from .iassd_head import IASSD_Head


class IASSD(nn.Module):
    def __init__(self, model_cfg, num_class, num_point_features):
        super().__init__()
        self.backbone_3d = IASSD_Backbone(model_cfg["BACKBONE_3D"], num_class=num_class, input_channels=num_point_features)
        self.point_head = IASSD_Head(num_class=num_class, input_channels=self.backbone_3d.num_point_features,
                                     model_cfg=model_cfg["POINT_HEAD"])
        self.module_list = [self.backbone_3d, self.point_head]
        self.model_cfg, self.num_class = model_cfg, num_class

    def forward(self, batch_dict):
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

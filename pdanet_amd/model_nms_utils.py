"""Post-processing of the detector (pcdet/models/detectors/detector3d_template.py:179-290 with
MULTI_CLASSES_NMS False -> model_utils/model_nms_utils.py:6-27 class_agnostic_nms), for all scenes
of a batch at once and without host synchronisation until the final compaction.

Reference, per scene: sigmoid -> max over classes -> `scores >= SCORE_THRESH` compaction -> topk
(NMS_PRE_MAXSIZE) -> nms_gpu (sort, mask kernel, device->host copy, host scan) -> first
NMS_POST_MAXSIZE -> index back.  Here: one masked sort for the batch, one gather, `nms_batched`
(csrc/iou3d_nms.hip) with per-scene valid counts, one gather."""
import torch

from . import iou3d_nms_utils


def class_agnostic_nms_batched(box_scores, box_preds, nms_config, score_thresh=None):
    """box_scores (B, N), box_preds (B, N, 7+C).  Returns selected (B, K) int64 indices into N (-1 padded),
    their scores (B, K) (0 padded) and num_selected (B) int32, K = min(N, NMS_POST_MAXSIZE)."""
    B, N = box_scores.shape
    if nms_config["NMS_TYPE"] not in ("nms_gpu", "nms_normal_gpu"):
        raise NotImplementedError(nms_config["NMS_TYPE"])
    valid = box_scores >= score_thresh if score_thresh is not None else torch.ones_like(box_scores, dtype=torch.bool)
    masked = torch.where(valid, box_scores, torch.full_like(box_scores, float("-inf")))
    sorted_scores, order = masked.sort(dim=1, descending=True)
    num_valid = valid.sum(dim=1).clamp(max=int(nms_config["NMS_PRE_MAXSIZE"])).to(torch.int32)
    boxes = torch.gather(box_preds[..., 0:7], 1, order.unsqueeze(-1).expand(B, N, 7)).contiguous()
    keep, num_keep = iou3d_nms_utils.nms_batched(boxes, nms_config["NMS_THRESH"], num_valid=num_valid,
                                                 normal=nms_config["NMS_TYPE"] == "nms_normal_gpu")
    K = min(N, int(nms_config["NMS_POST_MAXSIZE"]))
    keep = keep[:, :K]
    ok = keep >= 0
    safe = keep.clamp(min=0)
    selected = torch.where(ok, torch.gather(order, 1, safe), torch.full_like(keep, -1))
    scores = torch.where(ok, torch.gather(sorted_scores, 1, safe), torch.zeros_like(sorted_scores[:, :K]))
    return selected, scores, num_keep.clamp(max=K)


def post_processing(batch_dict, post_process_cfg, num_class):
    """detector3d_template.py:179-290 for point heads (`batch_index` layout, equal points per scene).
    Returns padded device tensors: pred_boxes (B, K, 7+C), pred_scores (B, K), pred_labels (B, K) int64
    (0 = padding) and num_pred (B) int32."""
    if post_process_cfg["NMS_CONFIG"]["MULTI_CLASSES_NMS"]:
        raise NotImplementedError("MULTI_CLASSES_NMS (not used by PDA-SSD.yaml)")
    B = batch_dict['batch_size']
    box_preds = batch_dict['batch_box_preds'].view(B, -1, batch_dict['batch_box_preds'].shape[-1])
    cls_preds = batch_dict['batch_cls_preds'].view(B, box_preds.shape[1], -1)
    assert cls_preds.shape[-1] in (1, num_class)
    src_cls_preds = cls_preds
    if not batch_dict['cls_preds_normalized']:
        cls_preds = torch.sigmoid(cls_preds)
    scores, labels = torch.max(cls_preds, dim=-1)
    labels = labels + 1
    selected, sel_scores, num = class_agnostic_nms_batched(scores, box_preds, post_process_cfg["NMS_CONFIG"],
                                                           score_thresh=post_process_cfg["SCORE_THRESH"])
    ok = selected >= 0
    safe = selected.clamp(min=0)
    if post_process_cfg.get("OUTPUT_RAW_SCORE", False):
        raw, _ = torch.max(src_cls_preds, dim=-1)
        sel_scores = torch.where(ok, torch.gather(raw, 1, safe), torch.zeros_like(sel_scores))
    pred_labels = torch.where(ok, torch.gather(labels, 1, safe), torch.zeros_like(safe))
    pred_boxes = torch.gather(box_preds, 1, safe.unsqueeze(-1).expand(-1, -1, box_preds.shape[-1])) * ok.unsqueeze(-1)
    return {'pred_boxes': pred_boxes, 'pred_scores': sel_scores, 'pred_labels': pred_labels, 'num_pred': num}


def to_pred_dicts(padded):
    """The reference's return value: a list (one per scene) of {'pred_boxes','pred_scores','pred_labels'}
    with variable-length tensors.  One host synchronisation for the whole batch."""
    counts = padded['num_pred'].tolist()
    return [{k: padded[k][s, :n] for k in ('pred_boxes', 'pred_scores', 'pred_labels')} for s, n in enumerate(counts)]

#!/usr/bin/env python
"""Generates tests/golden/head_{once,kitti}.npz: the REFERENCE's IASSD_Head
(/root/reference/pcdet/models/dense_heads/IASSD_head.py, built from the POINT_HEAD section of the
reference's own PDA-SSD.yaml) run on CPU in training mode on seeded synthetic backbone outputs:
forward -> assign_targets -> get_loss -> backward.  The CUDA extensions it reaches are stubbed with
this repo's CPU oracle (points_in_boxes, chamfer); absent third-party imports (SharedArray,
torch_scatter, open3d) are empty stub modules; `.cuda()` is made the identity.  Only inputs and
outputs are stored.  Run here only:  python tests/golden/make_head_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stub infrastructure: oracle-backed pointnet2_batch_cuda, _pkg, to_ad)
from detweights import fill_deterministic  # noqa: E402
from head_inputs import synth_inputs  # noqa: E402

import oracle  # noqa: E402

REF = mg.REF


def import_head():
    mg.import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.set_device = lambda d: None
    for name in ("SharedArray", "torch_scatter"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torch_scatter"].scatter_mean = sys.modules["torch_scatter"].scatter_max = None
    # chamfer entry points of pointnet2_batch_cuda (chamfer_cuda.cpp:22-31)
    # (only reached by the logged-only CD metric; its inputs are non-contiguous slices there)
    mg.stub.chamfer_forward = lambda x1, x2, *a: oracle.chamfer_forward(mg._np(x1.contiguous()), mg._np(x2.contiguous()), *[mg._np(x) for x in a])
    mg.stub.chamfer_backward = lambda x1, x2, *a: oracle.chamfer_backward(mg._np(x1.contiguous()), mg._np(x2.contiguous()), *[mg._np(x) for x in a])
    roi = types.ModuleType("roiaware_pool3d_cuda")
    roi.points_in_boxes_gpu = lambda boxes, pts, out: oracle.points_in_boxes_gpu(mg._np(boxes), mg._np(pts), mg._np(out))
    mg._pkg("pcdet.utils", REF + "/pcdet/utils")
    mg._pkg("pcdet.ops.roiaware_pool3d", REF + "/pcdet/ops/roiaware_pool3d")
    sys.modules["pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda"] = roi
    sys.modules["pcdet.ops.roiaware_pool3d"].roiaware_pool3d_cuda = roi
    mg._pkg("pcdet.models.dense_heads", REF + "/pcdet/models/dense_heads")
    import importlib
    return importlib.import_module("pcdet.models.dense_heads.IASSD_head")


def run(head_mod, yaml_path, tag, seed):
    cfg = yaml.safe_load(open(yaml_path))
    num_class = len(cfg["CLASS_NAMES"])
    head_cfg = mg.to_ad(cfg["MODEL"]["POINT_HEAD"])
    torch.manual_seed(0)
    head = head_mod.IASSD_Head(num_class=num_class, input_channels=512, model_cfg=head_cfg)
    fill_deterministic(head, salt="head.")
    head.train()
    inp = synth_inputs(num_class, seed)
    B = inp["gt_boxes"].shape[0]
    t = lambda a: torch.from_numpy(a.copy())
    feats = t(inp["feats"]).requires_grad_(True)
    offs = t(inp["offsets"].reshape(-1, 3)).requires_grad_(True)
    bidx = t(inp["centers"].reshape(-1, 4)[:, :1])
    sa_raw = [None if p is None else t(p).requires_grad_(True) for p in inp["sa_preds"]]
    sa_preds = [[] if p is None else torch.cat([t(inp["coords"][i + 1][..., :1]), p], dim=-1) for i, p in enumerate(sa_raw)]
    bd = {"batch_size": B, "gt_boxes": t(inp["gt_boxes"]), "centers_features": feats,
          "centers": t(inp["centers"].reshape(-1, 4)), "centers_origin": t(inp["origin"].reshape(-1, 4)),
          "ctr_offsets": torch.cat([bidx, offs], dim=1), "sa_ins_preds": sa_preds,
          "encoder_coords": [t(c) for c in inp["coords"]], "sample_list_id": []}
    bd = head(bd)
    loss, tb = head.get_loss()
    loss.backward()
    r = head.forward_ret_dict
    out = {"seed": np.array(seed), "num_class": np.array(num_class), "loss": loss.detach().numpy()}
    for k, v in tb.items():
        out["tb/" + k] = np.array(v, np.float64)
    out["grad/feats"], out["grad/offsets"] = feats.grad.numpy(), offs.grad.numpy()
    for i, p in enumerate(sa_raw):
        if p is not None:
            out["grad/sa%d" % i] = p.grad.numpy()
    for k in ("center_cls_labels", "center_box_labels", "center_gt_box_of_fg_points", "center_gt_box_of_points",
              "center_origin_cls_labels", "center_origin_box_idxs_of_pts", "gt_box_of_center_origin",
              "center_origin_gt_box_of_fg_points", "center_cls_preds", "center_box_preds", "point_box_preds"):
        out["ret/" + k] = r[k].detach().numpy()
    for i, (l, gbox) in enumerate(zip(r["sa_ins_labels"], r["sa_gt_box_of_fg_points"])):
        out["ret/sa_ins_labels/%d" % i] = l.numpy()
        out["ret/sa_gt_box_of_fg_points/%d" % i] = gbox.numpy()
    out["ret/origin_class_label"] = r["get_origin_class_label"][0].numpy()
    out["ret/centerness"] = head.generate_center_ness_mask().numpy()
    masks, _ = head.gauss_fun_once_topk_GT_add_same_size()
    for i, m in enumerate(masks):
        out["ret/sa_gauss/%d" % i] = m.numpy()
    for bn_name, mod in head.named_modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            out["bn/%s.running_mean" % bn_name] = mod.running_mean.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "head_%s.npz" % tag), **out)
    print(tag, "loss", float(loss), {k: round(float(v), 5) for k, v in tb.items()})
    print("   fg centres:", int((r["center_cls_labels"] > 0).sum()), "ignored:", int((r["center_cls_labels"] < 0).sum()),
          "sa fg:", [int((l > 0).sum()) for l in r["sa_ins_labels"]])


if __name__ == "__main__":
    hm = import_head()
    run(hm, REF + "/tools/cfgs/once_models/PDA-SSD.yaml", "once", seed=11)
    run(hm, REF + "/tools/cfgs/kitti_models/PDA-SSD.yaml", "kitti", seed=12)

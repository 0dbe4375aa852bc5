"""IASSD_Head against tests/golden/head_{once,kitti}.npz, which
tests/golden/make_head_golden.py produced by running the REFERENCE head on CPU.

CPU leg: this repo's head with `points_in_boxes_gpu` swapped for the CPU oracle (host logic only).
GPU leg: the product path (HIP points_in_boxes, everything on the device, no host sync)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from detweights import fill_deterministic  # noqa: E402
from head_inputs import synth_inputs  # noqa: E402


def _prepare(tag, device):
    from pdanet_amd import config
    from pdanet_amd.iassd_head import IASSD_Head
    gold = np.load(os.path.join(HERE, "golden", "head_%s.npz" % tag))
    cfg = config.load_yaml("%s_pda_ssd.yaml" % tag)
    nc = int(gold["num_class"])
    assert nc == len(cfg.CLASS_NAMES)
    head = IASSD_Head(num_class=nc, input_channels=512, model_cfg=cfg.MODEL.POINT_HEAD)
    fill_deterministic(head, salt="head.")
    head = head.to(device).train()
    inp = synth_inputs(nc, int(gold["seed"]))
    t = lambda a: torch.from_numpy(a.copy()).to(device)  # noqa: E731
    B = inp["gt_boxes"].shape[0]
    feats = t(inp["feats"]).requires_grad_(True)
    offs = t(inp["offsets"].reshape(-1, 3)).requires_grad_(True)
    bidx = t(inp["centers"].reshape(-1, 4)[:, :1])
    sa_raw = [None if p is None else t(p).requires_grad_(True) for p in inp["sa_preds"]]
    sa_preds = [[] if p is None else torch.cat([t(inp["coords"][i + 1][..., :1]), p], dim=-1) for i, p in enumerate(sa_raw)]
    bd = {"batch_size": B, "gt_boxes": t(inp["gt_boxes"]), "centers_features": feats,
          "centers": t(inp["centers"].reshape(-1, 4)), "centers_origin": t(inp["origin"].reshape(-1, 4)),
          "ctr_offsets": torch.cat([bidx, offs], dim=1), "sa_ins_preds": sa_preds,
          "encoder_coords": [t(c) for c in inp["coords"]], "sample_list_id": []}
    leaves = dict(feats=feats, offsets=offs, **{"sa%d" % i: p for i, p in enumerate(sa_raw) if p is not None})
    return gold, head, bd, leaves


def _step(head, bd):
    head(bd)
    loss, tb = head.get_loss()
    loss.backward()
    return loss, tb


def _run(tag, device):
    gold, head, bd, leaves = _prepare(tag, device)
    loss, tb = _step(head, bd)
    return gold, head, loss, tb, {k: v.grad for k, v in leaves.items()}


def _check(gold, head, loss, tb, grads):
    n = lambda x: x.detach().cpu().numpy()  # noqa: E731
    r = head.forward_ret_dict
    # target assignment: integer results exact
    for k in ("center_cls_labels", "center_origin_cls_labels", "center_origin_box_idxs_of_pts"):
        assert np.array_equal(n(r[k]), gold["ret/" + k]), k
    assert np.array_equal(n(r["get_origin_class_label"][0]), gold["ret/origin_class_label"])
    for i, l in enumerate(r["sa_ins_labels"]):
        assert np.array_equal(n(l), gold["ret/sa_ins_labels/%d" % i])
    np.testing.assert_allclose(n(r["center_gt_box_of_points"]), gold["ret/center_gt_box_of_points"], atol=0)
    np.testing.assert_allclose(n(r["gt_box_of_center_origin"]), gold["ret/gt_box_of_center_origin"], atol=0)
    np.testing.assert_allclose(n(r["center_box_labels"]), gold["ret/center_box_labels"], rtol=1e-5, atol=1e-6)
    comp = head.compact_targets()
    np.testing.assert_allclose(n(comp["center_gt_box_of_fg_points"]), gold["ret/center_gt_box_of_fg_points"], atol=1e-6)
    np.testing.assert_allclose(n(comp["center_origin_gt_box_of_fg_points"]), gold["ret/center_origin_gt_box_of_fg_points"], atol=0)
    for i, gbox in enumerate(comp["sa_gt_box_of_fg_points"]):
        np.testing.assert_allclose(n(gbox), gold["ret/sa_gt_box_of_fg_points/%d" % i], atol=0)
    # predictions, soft labels
    np.testing.assert_allclose(n(r["center_cls_preds"]), gold["ret/center_cls_preds"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(n(r["point_box_preds"]), gold["ret/point_box_preds"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(n(head.generate_center_ness_mask()), gold["ret/centerness"], rtol=1e-4, atol=1e-6)
    for i, m in enumerate(head.sa_gaussian_masks()):
        np.testing.assert_allclose(n(m), gold["ret/sa_gauss/%d" % i], rtol=1e-4, atol=1e-6)
    # every logged loss term except the reference's logged-only CD metric, then the total and gradients
    for k in gold.files:
        if k.startswith("tb/") and k != "tb/CD_loss":
            assert k[3:] in tb, k
            assert float(tb[k[3:]]) == pytest.approx(float(gold[k]), rel=2e-5, abs=1e-6), k
    assert float(loss) == pytest.approx(float(gold["loss"]), rel=2e-5)
    for k, g in grads.items():
        ref = gold["grad/" + k]
        assert np.abs(n(g) - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("tag", ["once", "kitti"])
def test_head_host_logic_matches_reference_cpu(tag, oracle, monkeypatch):
    from pdanet_amd import roiaware_pool3d_utils as ru

    def pib(points, boxes):
        out = np.full(points.shape[:2], -1, np.int32)
        oracle.points_in_boxes_gpu(boxes.contiguous().numpy(), points.contiguous().numpy(), out)
        return torch.from_numpy(out)
    monkeypatch.setattr(ru, "points_in_boxes_gpu", pib)
    _check(*_run(tag, "cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["once", "kitti"])
def test_head_matches_reference_gpu(tag):
    _check(*_run(tag, "cuda"))


@pytest.mark.gpu
def test_loss_path_has_no_host_sync():
    """forward -> targets -> losses -> backward must not synchronise (the reference syncs dozens of times)."""
    gold, head, bd, leaves = _prepare("once", "cuda")
    _step(head, dict(bd))                                    # warm-up (allocator, library handles)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss, tb = _step(head, dict(bd))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.isfinite(loss)


@pytest.mark.gpu
@pytest.mark.parametrize("kwargs", [dict(set_ignore_flag=True), dict(set_ignore_flag=True, ret_box_labels=True),
                                    dict(use_ex_gt_assign=True, set_ignore_flag=False),
                                    dict(use_ex_gt_assign=True, fg_pc_ignore=True, ret_box_labels=True),
                                    dict(set_ignore_flag=True, binary_label=True)])
def test_fused_target_kernels_equal_the_torch_formulation(kwargs):
    """csrc/head_targets.hip (one launch per point set) against the elementwise torch formulation of the same assignment:
    labels, box indices, gathered boxes and box-coder targets identical; the soft instance masks within 1e-6."""
    from pdanet_amd import iassd_head
    _, head, bd, _ = _prepare("once", "cuda")
    B = bd["batch_size"]
    gt = bd["gt_boxes"]
    gt8 = torch.cat((gt[..., 0:7], gt[..., -1:]), dim=-1) if gt.shape[-1] == 10 else gt
    from pdanet_amd import box_utils
    ext = box_utils.enlarge_box3d(gt8.view(-1, 8), extra_width=[0.5, 0.5, 0.5]).view(B, -1, 8)
    pts = bd["encoder_coords"][1].reshape(-1, 4)
    outs = []
    for fused in (True, False):
        iassd_head.FUSED_HEAD_TARGETS = fused
        try:
            outs.append(head.assign_stack_targets_IASSD(pts, gt8, ext, **kwargs))
        finally:
            iassd_head.FUSED_HEAD_TARGETS = True
    a, b = outs
    assert (a["point_cls_labels"] != 0).any()
    for k in ("point_cls_labels", "box_idxs_labels", "gt_box_of_points"):
        assert torch.equal(a[k], b[k]), k
    if kwargs.get("ret_box_labels"):
        assert torch.equal(a["point_box_labels"], b["point_box_labels"])
    # the single-launch form (both box queries, assignment and box-coder targets in one kernel): same integers, same boxes,
    # box-coder targets to f32 rounding of the logs / divisions
    c = head.assign_stack_targets_IASSD(pts, gt8, None, extra_width=[0.5, 0.5, 0.5], **kwargs)
    for k in ("point_cls_labels", "box_idxs_labels", "gt_box_of_points"):
        assert torch.equal(c[k], b[k]), k
    if kwargs.get("ret_box_labels"):
        assert torch.allclose(c["point_box_labels"], b["point_box_labels"], rtol=1e-6, atol=1e-6)
        assert torch.equal(c["point_box_labels"][:, 6], b["point_box_labels"][:, 6])          # heading bins
    # soft masks
    head.forward_ret_dict = {"sa_ins_labels": [a["point_cls_labels"]], "sa_gt_box_of_points": [a["gt_box_of_points"]],
                             "sa_xyz_coords": [bd["encoder_coords"][1]]}
    masks = []
    for fused in (True, False):
        iassd_head.FUSED_HEAD_TARGETS = fused
        try:
            masks.append(head.sa_gaussian_masks()[0])
        finally:
            iassd_head.FUSED_HEAD_TARGETS = True
    assert masks[0].shape == masks[1].shape and (masks[0] > 0).any()
    assert torch.allclose(masks[0], masks[1], atol=1e-6, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["once", "kitti"])
def test_fused_loss_kernels_equal_the_torch_formulation(tag):
    """csrc/head_loss.hip (one launch per loss term, gradient included) against the elementwise torch formulation of the same
    terms: every logged value, the total, and the gradients of every input that carries one -- including the vote centres,
    which the golden inputs hold as constants (the corner loss decodes boxes around them)."""
    from pdanet_amd import iassd_head
    res = {}
    for fused in (True, False):
        iassd_head.FUSED_HEAD_LOSS = fused
        try:
            gold, head, bd, leaves = _prepare(tag, "cuda")
            bd["centers"] = bd["centers"].clone().requires_grad_(True)
            leaves = dict(leaves, centers=bd["centers"])
            loss, tb = _step(head, bd)
            res[fused] = (float(loss), {k: float(v) for k, v in tb.items()}, {k: v.grad.clone() for k, v in leaves.items()})
        finally:
            iassd_head.FUSED_HEAD_LOSS = True
    (lf, tf, gf), (lt, tt, gt_) = res[True], res[False]
    assert lf == pytest.approx(lt, rel=2e-5)
    assert set(tf) == set(tt)
    for k in tt:
        assert tf[k] == pytest.approx(tt[k], rel=2e-5, abs=1e-6), k
    for k in gt_:
        assert (gf[k] - gt_[k]).abs().max().item() <= 2e-5 * max(1.0, gt_[k].abs().max().item()), k
    assert gf["centers"].abs().max().item() > 0


@pytest.mark.gpu
def test_box_loss_kernel_nan_targets_follow_the_input():
    """WeightedSmoothL1Loss replaces NaN targets by the input (loss_utils.py:166): zero loss and zero gradient on that
    code, finite everywhere else.  The library is built with -fno-honor-nans, so the kernel tests the bit pattern."""
    import torch.nn.functional as F
    from pdanet_amd import loss_utils, roiaware_pool3d_utils
    g = torch.Generator().manual_seed(5)
    n, nb = 257, 12
    preds = torch.randn(n, 6 + 2 * nb, generator=g).cuda().requires_grad_(True)
    labels = torch.randn(n, 8, generator=g)
    labels[:, 6] = torch.randint(0, nb, (n,), generator=g).float()
    labels[::3, 1] = float("nan")
    labels[5::7, 4] = float("nan")
    labels = labels.cuda()
    cls = (torch.rand(n, generator=g) > 0.4).long().cuda()
    cw = torch.ones(6).cuda()
    loss, l_xyz, l_bin, l_res = roiaware_pool3d_utils.head_box_loss(preds, labels, cls, cw, 1.0 / 9.0, nb, 1.0, 1.0)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(preds.grad).all()
    pos = cls > 0
    nan_rows = torch.isnan(labels[:, 1])
    assert (preds.grad[nan_rows, 1] == 0).all() and preds.grad[pos & ~nan_rows, 1].abs().max() > 0
    p2 = preds.detach().clone().requires_grad_(True)
    w = pos.float() / torch.clamp(pos.sum().float(), min=1.0)
    ref = loss_utils.WeightedSmoothL1Loss(beta=1.0 / 9.0, code_weights=[1.0] * 6).cuda()(p2[None, :, :6], labels[None, :, :6], weights=w[None]).sum()
    assert float(l_xyz.detach()) == pytest.approx(float(ref.detach()), rel=2e-5)
    ref.backward()
    assert (preds.grad[:, :6] - p2.grad[:, :6]).abs().max().item() <= 2e-6

"""The whole training iteration (backbone + head + adam_onecycle) on a small synthetic batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cfg, dataset, B=2, N=4096):
    from pdanet_amd import detector, optimization, synth
    torch.manual_seed(7)
    model, c = detector.build_detector(cfg)
    model = model.cuda().train()
    pts = synth.batch_points(B, N, config_id=2, dist="L", dataset=dataset)
    gt = synth.gt_boxes(pts, B, config_id=2, dataset=dataset)
    opt = optimization.build_optimizer(model, c.OPTIMIZATION)
    sched = optimization.build_scheduler(opt, 100, 2, c.OPTIMIZATION)
    bd = lambda: {'batch_size': B, 'points': torch.from_numpy(pts).cuda(), 'gt_boxes': torch.from_numpy(gt).cuda()}  # noqa: E731
    return model, opt, sched, bd


# Iteration-0 loss of the setup above, identical to the last bit on four different MI355X boxes in rounds 1-2
# (GPUTEST_r01.json, gpurun_out/hang.log, gpurun_out/r02_trace.log): the forward pass has no float atomics and
# is deterministic.  rel=1e-4 leaves room for a different library-GEMM heuristic on another ROCm build.
# Re-pinned late in round 3 (were 29.190820693969727 / 161.17782592773438): DensityNet now adds its batch statistics
# over the distinct slots with multiplicities (csrc/densitynet.hip, DnRows) -- another order of the same float sums,
# 3e-6 on its outputs -- and on this UNTRAINED model (all confidence scores within 1e-3 of each other) that moves
# near-tied entries across the top-k boundary of the ctr-aware sampling of layer 2: other points, another loss.
# test_densitynet_on_distinct_slots_moves_only_near_ties_of_the_top_k below holds the old numbers for the dense form
# and checks that nothing but such ties moves.
PINNED_FIRST_LOSS = {"once": 29.138832092285156, "kitti": 161.17782592773438}
PINNED_FIRST_LOSS_DENSE_DENSITYNET = {"once": 29.190820693969727, "kitti": 161.17782592773438}
CASES = [("once_pda_ssd.yaml", "once"), ("kitti_pda_ssd.yaml", "kitti")]


def _iteration(model, opt, sched, bd, it):
    """tools/train_utils/train_utils.py:34-64: scheduler step, zero_grad, forward, backward, clip + optimizer step."""
    sched.step(it)
    opt.zero_grad()
    ret, tb, _ = model(bd())
    ret['loss'].backward()
    return ret, tb


@pytest.mark.parametrize("cfg,dataset", CASES)
def test_densitynet_on_distinct_slots_moves_only_near_ties_of_the_top_k(cfg, dataset):
    """Iteration 0 with DensityNet on the distinct slots (default) against the dense form (round 1-2's arithmetic): the
    dense form still gives the loss pinned in rounds 1-2; the first PDA layer's outputs agree to float re-association
    noise; and where layer 2's ctr-aware top-k (pointnet2_modules.py:1541-1546 of the reference: torch.topk over the
    sigmoid of the largest class score) picks other points, every point that entered or left sits within that noise of
    the k-th score."""
    from pdanet_amd import pointnet2_utils as pu
    runs = {}
    try:
        for unique in (False, True):
            pu.DENSITYNET_UNIQUE = unique
            model, opt, sched, bd = _setup(cfg, dataset)
            got = {}
            mods = model.backbone_3d.SA_modules
            hooks = [mods[1].register_forward_hook(lambda m, i, o: got.update(l1=[t.detach().clone() for t in o[:3]])),
                     mods[2].register_forward_hook(lambda m, i, o: got.update(idx2=o[3].detach().clone()))]
            ret, _ = _iteration(model, opt, sched, bd, 0)
            for h in hooks:
                h.remove()
            runs[unique] = (float(ret['loss'].detach()), got)
    finally:
        pu.DENSITYNET_UNIQUE = True
    assert runs[False][0] == pytest.approx(PINNED_FIRST_LOSS_DENSE_DENSITYNET[dataset], rel=1e-4)
    assert runs[True][0] == pytest.approx(runs[False][0], rel=2e-2)
    (xyz_d, feat_d, cls_d), (xyz_u, feat_u, cls_u) = runs[False][1]['l1'], runs[True][1]['l1']
    assert torch.equal(xyz_d, xyz_u)
    assert (feat_d - feat_u).abs().max().item() < 5e-5 and (cls_d - cls_u).abs().max().item() < 2e-5
    score = torch.sigmoid(cls_d.max(dim=-1)[0])                              # (B, M) as sample_points ranks them
    idx_d, idx_u = runs[False][1]['idx2'].long(), runs[True][1]['idx2'].long()
    k = idx_d.shape[1]
    kth = torch.topk(score, k, dim=-1)[0][:, -1]
    for b in range(score.shape[0]):
        a, c = set(idx_d[b].tolist()), set(idx_u[b].tolist())
        moved = sorted(a ^ c)
        if moved:
            gap = (score[b, moved] - kth[b]).abs().max().item()
            assert gap < 2e-5, (b, len(moved), gap)


@pytest.mark.parametrize("cfg,dataset", CASES)
def test_first_training_iteration_is_pinned(cfg, dataset):
    """Deterministic properties of ONE iteration (train_utils.py:34-64): pinned loss, every parameter has a finite
    gradient, the target assignment found positives, the clipped step moved the trained parameters and left the
    never-trained in_proj_* alone."""
    model, opt, sched, bd = _setup(cfg, dataset)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    ret, tb = _iteration(model, opt, sched, bd, 0)
    missing = [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    assert float(tb['center_pos_num']) > 0 and float(tb['sa1_pos_num']) > 0
    assert float(ret['loss'].detach()) == pytest.approx(PINNED_FIRST_LOSS[dataset], rel=1e-4)
    gnorm = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in model.parameters()))
    opt.step()
    assert float(opt.total_norm) == pytest.approx(float(gnorm), rel=1e-4)
    moved = {n: (p.detach() - before[n]).abs().max().item() for n, p in model.named_parameters()}
    for n, d in moved.items():
        if "in_proj" in n:
            assert d == 0.0, n
        else:
            assert np.isfinite(d), n
            # Adam's first step moves a parameter by at most lr (plus the decoupled decay lr*wd*|p|)
            assert d <= opt.lr * (1.0 + 0.01 * before[n].abs().max().item()) * 1.001, (n, d)
    assert sum(d > 0 for d in moved.values()) > 200


@pytest.mark.parametrize("cfg,dataset", CASES)
def test_training_learns_over_40_iterations(cfg, dataset):
    """Round 1 asserted `loss[5] < loss[0]`, which is a coin flip: from the first backward on the trajectories of
    two runs differ (float-atomic gradient sums -> Adam's sign-like first steps -> a different top-k centre set), and
    `center_loss_cls` -- weight 1, normalised by a positive count of a few dozen centres -- swings 129 -> 28 -> 47 ->
    60 -> 34 in the first iterations (KITTI; gpurun_out/r02_trace.log, recorded in DESIGN.md).  Over 40 iterations
    every recorded run falls 161 -> <4 (KITTI) and 29 -> <3.5 (ONCE); assert a property with a wide margin."""
    model, opt, sched, bd = _setup(cfg, dataset)
    losses, terms, norms = [], [], []
    for it in range(40):
        ret, tb = _iteration(model, opt, sched, bd, it)
        opt.step()
        losses.append(ret['loss'].detach())
        terms.append({k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()})   # no synchronisation here
        norms.append(opt.total_norm)
    losses = [float(x) for x in torch.stack(losses).cpu()]
    # The reference's corner loss is the MEAN over the positive centres (IASSD_head.py:1307-1317): an iteration whose
    # 2-scene synthetic batch assigns no positive centre (center_pos_num is 0-2 here) has a NaN loss VALUE and zero
    # gradient from that term -- in the reference as in this repo; training goes on.  Anything else must be finite.
    for i, x in enumerate(losses):
        if not np.isfinite(x):
            t = {k: float(v) for k, v in terms[i].items()}
            assert t['center_pos_num'] == 0 and all(np.isfinite(v) for k, v in t.items() if k != 'corner_loss_reg'), (i, t)
            assert np.isfinite(float(norms[i])), (i, float(norms[i]))
    finite = [x for x in losses if np.isfinite(x)]
    assert len(finite) >= 30, losses
    assert min(finite[-10:]) < 0.5 * losses[0], losses
    assert float(np.mean(finite[-10:])) < float(np.mean(finite[:10])), losses


def test_graphed_tail_in_dense_bf16_mode_follows_the_weights():
    """The graphed tail in dense-bf16 mode (the KITTI configs[2] workload runs with it).  The replay
    must use the CURRENT weights (the bf16 parameter copies are made inside the graph): three iterations, graphed and eager,
    stay together (bf16 GEMMs + different trajectories after the first step: a loose bar), and zeroing the head's output
    weights changes the replayed loss."""
    from pdanet_amd import pointnet2_utils as pu
    keep = pu.DENSE_BF16
    res = {}
    try:
        pu.DENSE_BF16 = True
        for graphed in (False, True):
            model, opt, sched, bd = _setup("kitti_pda_ssd.yaml", "kitti")
            model.graph_tail = graphed
            if graphed:
                assert model.tail_start() == 3
            out = []
            for it in range(3):
                ret, _ = _iteration(model, opt, sched, bd, it)
                opt.step()
                out.append(float(ret['loss'].detach()))
            if graphed:
                with torch.no_grad():
                    for n, p in model.named_parameters():
                        if "cls_center_layers.6" in n or "box_center_layers.6" in n:
                            p.zero_()
                ret, _ = _iteration(model, opt, sched, bd, 3)
                out.append(float(ret['loss'].detach()))
            res[graphed] = out
    finally:
        pu.DENSE_BF16 = keep
    e, g = res[False], res[True]
    assert g[0] == pytest.approx(e[0], rel=2e-2)
    # iteration 1 replays the graphs with the weights of one optimizer step; from iteration 2 on the two runs are different
    # trajectories of a chaotic start (losses swing 76 -> 200..360 here), so only iteration 1 is compared
    assert np.isfinite(g[1]) and g[1] == pytest.approx(e[1], rel=0.3)
    assert g[3] != g[2]                                # the replay saw the zeroed weights


@pytest.mark.parametrize("segment", ["head", "tail"])
def test_graphed_head_equals_eager_head(segment):
    """detector.IASSD.graph_head / graph_tail: the head + losses (tail: and the backbone layers behind the last
    unique-token plan) replayed as hipGraphs (forward and backward) give the eager iteration: same loss, same
    clipped-gradient norm, same parameter update; log values come back as tensors."""
    res = {}
    for graphed in (False, True):
        model, opt, sched, bd = _setup("once_pda_ssd.yaml", "once")
        setattr(model, "graph_" + segment, graphed)
        if segment == "tail":
            assert model.tail_start() == 3
        out = []
        for it in range(2):      # the second iteration replays the captured graphs
            ret, tb = _iteration(model, opt, sched, bd, it)
            if it == 0:
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()
                         if p.grad is not None and ("point_head" in n or segment == "tail")}
            opt.step()
            out.append((float(ret['loss'].detach()), float(opt.total_norm), {k: float(v) for k, v in tb.items()}))
        res[graphed] = (out, grads)
    (eo, eg), (go, gg) = res[False], res[True]
    assert go[0][0] == pytest.approx(PINNED_FIRST_LOSS["once"], rel=1e-4)
    assert go[0][0] == pytest.approx(eo[0][0], rel=1e-5) and go[0][1] == pytest.approx(eo[0][1], rel=1e-3)
    assert set(go[0][2]) == set(eo[0][2])
    for k in eo[0][2]:
        assert go[0][2][k] == pytest.approx(eo[0][2][k], rel=1e-4, abs=1e-6), k
    assert set(eg) == set(gg) and len(gg) >= 12
    gmax = max(float(v.abs().max()) for v in eg.values())
    for n in eg:       # (a conv bias in front of a BatchNorm has an exactly-zero gradient: rounding noise of the largest one)
        assert (eg[n] - gg[n]).abs().max().item() <= 1e-4 * max(1e-3, eg[n].abs().max().item()) + 3e-6 * gmax, n
    # after one (Adam, sign-like) step the two runs are different trajectories in the last bits; the replayed iteration
    # must still be the same computation: finite and close
    assert np.isfinite(go[1][0]) and go[1][0] == pytest.approx(eo[1][0], rel=0.2)


def test_state_dict_keys_follow_the_reference_detector():
    from pdanet_amd import detector
    model, _ = detector.build_detector("once_pda_ssd.yaml")
    keys = list(model.state_dict())
    assert any(k.startswith("backbone_3d.SA_modules.0.") for k in keys)
    head = [k for k in keys if k.startswith("point_head.")]
    assert head[0] == "point_head.cls_center_layers.0.weight"
    assert "point_head.box_center_layers.6.bias" in head and model.state_dict()["point_head.box_center_layers.6.bias"].shape == (30,)
    assert model.state_dict()["point_head.cls_center_layers.6.weight"].shape == (5, 256)


def test_inference_post_processing_matches_per_scene_reference_algorithm(oracle):
    """Detector in eval mode: padded batched NMS output == the reference's per-scene algorithm
    (score threshold -> topk -> rotated NMS -> first NMS_POST_MAXSIZE) evaluated with the CPU oracle."""
    from pdanet_amd import detector, synth
    torch.manual_seed(3)
    model, cfg = detector.build_detector("once_pda_ssd.yaml")
    model = model.cuda().eval()
    B, N = 2, 4096
    pts = torch.from_numpy(synth.batch_points(B, N, config_id=2, dist="L")).cuda()
    with torch.no_grad():
        bd = {'batch_size': B, 'points': pts}
        pred_dicts, _ = model(bd)
    pp = cfg.MODEL.POST_PROCESSING
    boxes_all = bd['batch_box_preds'].view(B, -1, 7).cpu().numpy()
    scores_all = torch.sigmoid(bd['batch_cls_preds']).view(B, boxes_all.shape[1], -1).max(-1)
    assert len(pred_dicts) == B
    for s in range(B):
        sc, lab = scores_all[0][s].cpu().numpy(), scores_all[1][s].cpu().numpy() + 1
        idx = np.nonzero(sc >= pp.SCORE_THRESH)[0]
        order = idx[np.argsort(-sc[idx], kind="stable")][: pp.NMS_CONFIG.NMS_PRE_MAXSIZE]
        keep = np.zeros(max(1, len(order)), np.int64)
        k = oracle.nms_gpu(np.ascontiguousarray(boxes_all[s][order]), keep, pp.NMS_CONFIG.NMS_THRESH) if len(order) else 0
        sel = order[keep[:k]][: pp.NMS_CONFIG.NMS_POST_MAXSIZE]
        got = pred_dicts[s]
        assert got['pred_boxes'].shape[0] == len(sel) > 0
        np.testing.assert_array_equal(got['pred_boxes'].cpu().numpy(), boxes_all[s][sel])
        np.testing.assert_array_equal(got['pred_scores'].cpu().numpy(), sc[sel])
        np.testing.assert_array_equal(got['pred_labels'].cpu().numpy(), lab[sel])


def test_fused_and_unfused_training_steps_agree():
    """All of this repo's fused training kernels (attention, LayerNorm, BN+ReLU, weight gradients, DensityNet, token
    assembly, add+max, one-node transformer) against the op-by-op torch execution of the same layers: identical
    initial weights and data; outputs, loss and gradients of layers 0-1 within fp32 re-association noise.  (The
    loss stops at layer 1: from layer 2 on the centres are a top-k over learned scores, and a random-init model's
    near-equal scores turn 1e-6 differences into different point sets.)"""
    from pdanet_amd import synth, pointnet2_modules as pm, pointnet2_utils as pu
    from pdanet_amd.backbone import build_backbone
    flags = [(pm, "GROUP_ATTENTION_KERNEL"), (pm, "FUSED_BN_RELU"), (pm, "FUSED_LAYER_NORM"), (pm, "FUSED_TRANSFORMER_BLOCK"),
             (pm, "FUSED_GEOMETRY"), (pm, "FUSED_BN_RELU_MAX_POOL"), (pu, "LINEAR_WGRAD_KERNEL"), (pu, "FUSED_DENSITYNET"), (pu, "FUSED_ASSEMBLE"), (pu, "RAGGED_TOKENS"), (pu, "RAGGED_POSITION_MLP"), (pu, "SA_MFMA_TRAIN")]
    B, N = 2, 4096
    pts = torch.from_numpy(synth.batch_points(B, N, config_id=2, dist="L")).cuda()
    results = []
    for on in (True, False):
        for mod, name in flags:
            setattr(mod, name, on)
        try:
            torch.manual_seed(21)
            model, _ = build_backbone("once_pda_ssd.yaml")
            model = model.cuda().train()
            bd = model({'batch_size': B, 'points': pts})
            feat, cls = bd['encoder_features'][2], bd['sa_ins_preds'][1][..., 1:]
            loss = feat.pow(2).mean() + cls.pow(2).mean()
            loss.backward()
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            results.append((float(loss), feat.detach().clone(), grads,
                            {n: b.detach().clone() for n, b in model.named_buffers() if 'running_var' in n and ('SA_modules.0' in n or 'SA_modules.1' in n)}))
        finally:
            for mod, name in flags:
                setattr(mod, name, True)
    (la, fa, ga, ba), (lb, fb, gb, bb) = results
    assert la == pytest.approx(lb, rel=1e-4)
    assert (fa - fb).abs().max().item() < 2e-3 * max(1.0, fb.abs().max().item())
    assert set(ga) == set(gb) and len(ga) > 60
    for n in ga:
        # floor 1e-3: gradients of convs that feed a BatchNorm are O(eps) remainders of cancelling sums (BN is
        # invariant to the scale / shift of its input): ~1e-4 in magnitude and fp32-noise dominated on either path
        scale = max(gb[n].abs().max().item(), 1e-3)
        assert (ga[n] - gb[n]).abs().max().item() < 2e-2 * scale, n
    for n in ba:
        assert torch.allclose(ba[n], bb[n], atol=1e-5, rtol=1e-3), n


def test_dense_bf16_mode_tracks_fp32():
    """pointnet2_utils.DENSE_BF16 (DESIGN.md "Dense-bf16 mode"): bf16 GEMMs with fp32 accumulation, GEMM-adjacent tensors
    stored as bf16, everything else fp32.  Layers 0-1 of the backbone against the fp32 run: bf16-level agreement."""
    from pdanet_amd import synth, pointnet2_utils as pu
    from pdanet_amd.backbone import build_backbone
    B, N = 2, 4096
    pts = torch.from_numpy(synth.batch_points(B, N, config_id=2, dist="L")).cuda()
    res = []
    for flag in (True, False):
        pu.DENSE_BF16 = flag
        try:
            torch.manual_seed(21)
            model, _ = build_backbone("once_pda_ssd.yaml")
            model = model.cuda().train()
            bd = model({'batch_size': B, 'points': pts})
            feat = bd['encoder_features'][2]
            loss = feat.pow(2).mean()
            loss.backward()
            res.append((float(loss), feat.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            pu.DENSE_BF16 = False
    (la, fa, ga), (lb, fb, gb) = res
    assert la == pytest.approx(lb, rel=2e-2)
    assert (fa - fb).abs().mean().item() < 2e-2 * fb.abs().mean().item()
    big = [n for n in gb if gb[n].abs().max().item() > 2e-3]
    assert len(big) >= 4
    for n in big:
        cos = torch.nn.functional.cosine_similarity(ga[n].flatten(), gb[n].flatten(), dim=0).item()
        assert cos > 0.98, (n, cos)


# Iteration-0 loss of the HEADLINE configuration (ONCE yaml, B = 2, N = 16384: bench.py's default workload shape);
# the forward has no float atomics and is deterministic (same argument as PINNED_FIRST_LOSS).
PINNED_FIRST_LOSS_16K = 138.3255157470703


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_headline_size_model_forward_against_the_oracle(mode, oracle):
    """Model-level parity at the size the bench is quoted on: the detector's forward on 2 x 16384
    points in training-mode and in eval-mode BatchNorm.  The D-FPS centres of layer 1 (16384 -> 4096, chain kernel on the
    side stream) must be the oracle's samples of the same points, the layer-1 neighbour lists around them (both radii,
    cell list) the oracle's rows, the layer-0 'sampling' the identity; training mode: the first-iteration loss is pinned."""
    from pdanet_amd import pointnet2_utils as pu
    B, N = 2, 16384
    model, opt, sched, bd = _setup("once_pda_ssd.yaml", "once", B=B, N=N)
    batch = bd()
    bb = model.backbone_3d
    if mode == "train":
        sched.step(0)
        opt.zero_grad()
        feats = {}
        hook = bb.register_forward_hook(lambda m, i, o: feats.update(o))
        ret, tb, _ = model(batch)
        hook.remove()
        loss = float(ret['loss'].detach())
        assert np.isfinite(loss) and float(tb['center_pos_num']) > 0
        if PINNED_FIRST_LOSS_16K is not None:
            assert loss == pytest.approx(PINNED_FIRST_LOSS_16K, rel=1e-4)
        print("first-iteration loss at 2 x 16384:", repr(loss))
        out = feats
    else:
        model.eval()
        with torch.no_grad():
            out = bb(dict(batch))
        assert torch.isfinite(out['centers_features']).all()
    xyz = batch['points'][:, 1:4].reshape(B, N, 3).contiguous()
    enc = out['encoder_xyz']
    assert torch.equal(enc[1], xyz)                                   # layer 0: N <= npoint, sampling is the identity
    xyz_np = xyz.cpu().numpy()
    temp = np.full((B, N), 1e10, np.float32)
    idx_o = np.zeros((B, 4096), np.int32)
    oracle.farthest_point_sampling_wrapper(B, N, 4096, xyz_np, temp, idx_o)
    centres_o = np.stack([xyz_np[i][idx_o[i]] for i in range(B)])
    assert np.array_equal(centres_o, enc[2].cpu().numpy())             # the model's layer-1 centres == oracle D-FPS picks
    ids = out['sample_list_id'][1]
    assert np.array_equal(idx_o, ids.cpu().numpy().astype(np.int32))
    m1 = bb.SA_modules[1]
    rows = pu.ball_query_multi(list(m1.radii) if hasattr(m1, "radii") else [g.radius for g in m1.groupers],
                               list(m1.nsamples), xyz, enc[2].contiguous())
    for g, got in zip(m1.groupers, rows):
        exp = np.zeros((B, 4096, g.nsample), np.int32)
        oracle.ball_query_wrapper(B, N, 4096, g.radius, g.nsample, centres_o, xyz_np, exp)
        assert np.array_equal(exp, got.cpu().numpy()), g.radius


@pytest.mark.parametrize("segment", ["head", "tail"])
def test_graph_replay_uses_the_current_weights_f32(segment):
    """Default (f32) mode: the packed bf16 planes of the split GEMMs are cached per parameter; inside a captured region the
    packing must be part of the graph, or a replay would keep computing with the weights of capture time.  Four graphed
    iterations (three optimizer steps behind the capture), then the same batch forward through the replay and through the
    eager path on the SAME weights: equal losses (a stale weight set is ~3 Adam steps off: orders of magnitude more)."""
    model, opt, sched, bd = _setup("once_pda_ssd.yaml", "once")
    setattr(model, "graph_" + segment, True)
    for it in range(4):
        _iteration(model, opt, sched, bd, it)
        opt.step()
    batch = bd()
    with torch.enable_grad():
        loss_g = float(model(batch)[0]['loss'].detach())
    setattr(model, "graph_" + segment, False)
    loss_e = float(model(bd())[0]['loss'].detach())
    assert np.isfinite(loss_g) and loss_g == pytest.approx(loss_e, rel=2e-5)

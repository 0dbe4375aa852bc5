"""-m gpu: unique-token ("ragged") execution of a PDA scale (csrc/ragged.hip, RAGGED mode of csrc/group_attention.hip)
against the dense execution it replaces (pointnet2_modules.py:879-933, PointFormer.py:28-38 of the reference).

ball_query pads a short neighbour list with repeats of its first entry (ball_query_gpu.cu:35-41).  The ragged form
evaluates the encoder on the distinct tokens only; in exact arithmetic its results equal the dense ones.  Floating
point: the softmax sums `w * exp(s)` instead of w equal terms, so the bar is fp32 re-association noise (1e-5 on O(1)
attention outputs, 1e-4 of the largest gradient), as in tests/test_group_attention.py."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from detweights import fill_deterministic  # noqa: E402

pytestmark = pytest.mark.gpu


def padded_idx(G, S, n, rng, p_single=0.3, p_full=0.1, empty_rows=2):
    """Neighbour lists as ball_query leaves them: cnt distinct ascending indices, then repeats of the first."""
    idx = np.zeros((G, S), np.int32)
    cnt = np.zeros(G, np.int32)
    for g in range(G):
        u = rng.random()
        c = 1 if u < p_single else (S if u > 1 - p_full else int(rng.integers(1, S + 1)))
        hits = np.sort(rng.choice(n, size=c, replace=False)).astype(np.int32)
        idx[g, :c] = hits
        idx[g, c:] = hits[0]
        cnt[g] = c
    idx[:empty_rows] = 0        # a ball with no hit: the caller's zeros = point 0 repeated
    cnt[:empty_rows] = 1
    return idx, cnt


@pytest.mark.parametrize("G,S", [(1, 8), (77, 16), (1000, 32), (4099, 16)])
def test_plan_matches_numpy(G, S):
    from pdanet_amd import pointnet2_utils as pu
    rng = np.random.default_rng(G + S)
    idx, cnt = padded_idx(G, S, 5000, rng)
    (plan,) = pu.ragged_plans([torch.from_numpy(idx).cuda().view(1, G, S)])
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    assert plan.tokens == int(off[-1]) and plan.groups == G and plan.nsample == S
    assert np.array_equal(plan.cnt.cpu().numpy(), cnt)
    assert np.array_equal(plan.off.cpu().numpy(), off)
    rowmap = np.concatenate([g * S + np.arange(c) for g, c in enumerate(cnt)]).astype(np.int32)
    assert np.array_equal(plan.rowmap.cpu().numpy(), rowmap)


# The ragged kernel packs the distinct tokens of consecutive groups into 32-row tiles, several groups (chosen from the mean
# count) per wave: the count distributions below make a wave's groups fit one tile exactly (all single: 32 groups of 1),
# never share a tile (all full at S = 32), overflow into a second tile, end in a ragged last block (G not a multiple of
# the groups per wave), and -- the large case -- walk several tiles per wave.
@pytest.mark.parametrize("S,hd,heads,G,p_single,p_full", [
    (8, 32, 4, 203, 0.3, 0.1), (16, 64, 4, 203, 0.3, 0.1), (32, 64, 4, 203, 0.3, 0.1), (32, 128, 4, 203, 0.3, 0.1),
    (16, 128, 2, 203, 0.3, 0.1), (32, 64, 4, 333, 1.0, 0.0), (32, 64, 4, 131, 0.0, 1.0), (16, 32, 4, 131, 0.0, 1.0),
    (8, 32, 4, 1001, 0.9, 0.1), (16, 32, 4, 20011, 0.0, 0.0)])
def test_ragged_attention_equals_dense_attention_on_padded_groups(S, hd, heads, G, p_single, p_full):
    """Dense kernel on the (G, S) layout with token 0 repeated == ragged kernel on the compact rows."""
    from pdanet_amd import pointnet2_utils as pu, pointnet2_batch_cuda as ext
    D = heads * hd
    rng = np.random.default_rng(S * 10 + hd + G)
    idx, cnt = padded_idx(G, S, 4000, rng, p_single, p_full, empty_rows=2 if p_full < 1.0 else 0)
    (plan,) = pu.ragged_plans([torch.from_numpy(idx).cuda().view(1, G, S)])
    U = plan.tokens
    torch.manual_seed(S + hd)
    qkv_c = torch.randn(U, 3 * D, device="cuda") * 0.7
    go_c = torch.randn(U, D, device="cuda")
    # dense twin: slot s of group g = compact row off[g] + (s if s < cnt else 0)
    off = plan.off.long()[:-1]
    slot = torch.arange(S, device="cuda").view(1, S)
    src = off.view(G, 1) + torch.where(slot < plan.cnt.long().view(G, 1), slot, torch.zeros_like(slot))
    qkv_d = qkv_c[src].contiguous()                                         # (G, S, 3D)
    valid = (slot < plan.cnt.long().view(G, 1))
    go_d = torch.where(valid.unsqueeze(-1), go_c[src], torch.zeros((), device="cuda")).contiguous()   # copies carry no gradient
    out_d = torch.empty(G, S, D, device="cuda"); lse_d = torch.empty(G, heads, S, device="cuda")
    ext.group_attention_fwd(qkv_d, out_d, lse_d, G, S, heads, hd)
    dq_d = torch.empty_like(qkv_d)
    ext.group_attention_bwd(qkv_d, go_d, lse_d, dq_d, G, S, heads, hd)
    out_c = torch.empty(U, D, device="cuda"); lse_c = torch.empty(G, heads, S, device="cuda")
    ext.group_attention_ragged_fwd(qkv_c, plan.cnt, plan.off, out_c, lse_c, U, G, S, heads, hd)
    dq_c = torch.empty_like(qkv_c)
    ext.group_attention_ragged_bwd(qkv_c, go_c, lse_c, plan.cnt, plan.off, dq_c, U, G, S, heads, hd)
    torch.cuda.synchronize()
    assert (out_d[valid] - out_c).abs().max().item() < 1e-5
    # every copy of a token has the same output as the token itself
    assert (out_d - out_c[src]).abs().max().item() < 1e-5
    # the compact token's gradient is the SUM over its copies
    ref = torch.zeros_like(dq_c).index_add_(0, src.reshape(-1), dq_d.reshape(-1, 3 * D))
    scale = max(1.0, ref.abs().max().item())
    assert (ref - dq_c).abs().max().item() < 1e-4 * scale
    assert torch.isfinite(out_c).all() and torch.isfinite(dq_c).all()


@pytest.mark.parametrize("G,S,C,compact", [(203, 16, 64, False), (1000, 32, 128, True), (77, 8, 16, True), (4099, 32, 256, False)])
def test_assembly_backward_token_parallel_equals_centre_form(G, S, C, compact):
    """pda_assemble_tokens_ragged_grad with rowmap (per-token work over (token, column) threads + a per-centre kernel) against
    the form without (one thread walks a centre's tokens): grad_rppe / grad_dscale / grad_glob bit for bit -- including the
    zeros on the repeat slots --, grad_feats (float atomics in both) to 1e-5."""
    from pdanet_amd import pointnet2_utils as pu
    rng = np.random.default_rng(G + S + C)
    N = 3000
    idx, cnt = padded_idx(G, S, N, rng)
    idx_t = torch.from_numpy(idx).cuda().view(1, G, S)
    (plan,) = pu.ragged_plans([idx_t])
    torch.manual_seed(C)
    rppe = torch.randn((plan.tokens, C) if compact else (1, G, S, C), device="cuda", requires_grad=True)
    dscale = torch.rand(1, G, S, 1, device="cuda", requires_grad=True)
    feats = torch.randn(1, N, C, device="cuda", requires_grad=True)
    glob = torch.randn(1, G, C, device="cuda", requires_grad=True)
    go = torch.randn(plan.tokens, 4 * C, device="cuda")
    res = {}
    try:
        for flag in (True, False):
            pu.ASSEMBLE_BWD_TOKEN_PARALLEL = flag
            x = pu.AssembleTokensRagged.apply(rppe, dscale, feats, idx_t, glob, plan)
            res[flag] = torch.autograd.grad(x, [rppe, dscale, feats, glob], go)
    finally:
        pu.ASSEMBLE_BWD_TOKEN_PARALLEL = True
    for k, name in ((0, "rppe"), (1, "dscale"), (3, "glob")):
        assert torch.equal(res[True][k], res[False][k]), name
    assert (res[True][2] - res[False][2]).abs().max().item() <= 1e-5 * max(1.0, res[False][2].abs().max().item())
    valid = torch.arange(S, device="cuda").view(1, 1, S) < plan.cnt.view(1, G, 1)
    assert float(res[True][1].view(1, G, S)[~valid].abs().sum()) == 0.0


def _pda_layer():
    """ONCE layer 1 in small: C = 64 (encoder width D = 256, the width the fused encoder path is built for)."""
    from pdanet_amd import pointnet2_modules as pm
    layer = pm.PointnetSAModuleMSG_WithSampling_Ellipsoid(
        npoint_list=[1024], sample_range_list=[-1], sample_type_list=["D-FPS"], radii=[0.8, 1.6], nsamples=[16, 32],
        mlps=[[64, 96, 128], [64, 96, 128]], use_xyz=True, dilated_group=False, aggregation_mlp=[128],
        confidence_mlp=[128], num_class=5)
    return fill_deterministic(layer).cuda()


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pda_layer_ragged_equals_dense(mode):
    """The whole PDA layer (both scales) with RAGGED_TOKENS on / off: outputs, input gradient, every parameter gradient."""
    from pdanet_amd import pointnet2_utils as pu, synth
    xyz = torch.from_numpy(synth.batch_xyz(2, 4096, config_id=2)).cuda()
    feats0 = torch.randn(2, 64, 4096, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    res = {}
    used = []
    orig = pu.ragged_transformer_block
    try:
        for flag in (True, False):
            pu.RAGGED_TOKENS = flag
            pu.ragged_transformer_block = (lambda *a, **k: (used.append(a[2].fraction), orig(*a, **k))[1])
            layer = _pda_layer().train(mode == "train")
            feats = feats0.clone().requires_grad_(True)
            nx, nf, cf, _ = layer(xyz, feats, None)
            (nf.pow(2).mean() + cf.pow(2).mean()).backward()
            res[flag] = (nf.detach(), cf.detach(), feats.grad.clone(),
                         {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    finally:
        pu.RAGGED_TOKENS = True
        pu.ragged_transformer_block = orig
    assert len(used) == 2 and max(used) < 0.85, used     # both scales took the ragged path
    a, b = res[True], res[False]
    for i in (0, 1):
        assert (a[i] - b[i]).abs().max().item() <= 2e-4 * max(1.0, b[i].abs().max().item())
    # input gradient: the two forms run different token counts through the f32 GEMMs, so last-bit differences move near-tied
    # max-pool winners and ReLU kinks of single units (valid subgradients either way): bound the tensor, and every entry loosely
    assert (a[2] - b[2]).norm().item() <= 2e-3 * b[2].norm().item() + 1e-9
    assert (a[2] - b[2]).abs().max().item() <= 2e-2 * b[2].abs().max().item() + 1e-7
    assert set(a[3]) == set(b[3])
    gmax = max(float(v.abs().max()) for v in b[3].values())
    for k in b[3]:
        s = float(b[3][k].abs().max())
        assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-2 * s + 1e-3 * gmax, k


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_count_read_overlap_changes_no_result(mode):
    """pointnet2_modules.PLAN_OVERLAP (the host waits for the token counts through an event, after enqueuing the work that needs
    no count, and runs the larger scale first) against the blocking read in the original scale order: the same kernels on the
    same operands -- forward outputs bit for bit (gradients sum float atomics, so those are compared to 1e-5)."""
    from pdanet_amd import pointnet2_modules as pm, synth
    xyz = torch.from_numpy(synth.batch_xyz(2, 4096, config_id=2)).cuda()
    feats0 = torch.randn(2, 64, 4096, device="cuda", generator=torch.Generator("cuda").manual_seed(6))
    res = {}
    try:
        for flag in (True, False):
            pm.PLAN_OVERLAP = flag
            layer = _pda_layer().train(mode == "train")
            feats = feats0.clone().requires_grad_(mode == "train")
            with torch.set_grad_enabled(mode == "train"):
                nx, nf, cf, _ = layer(xyz, feats, None)
            g = None
            if mode == "train":
                (nf.pow(2).mean() + cf.pow(2).mean()).backward()
                g = feats.grad.clone()
            res[flag] = (nx, nf.detach(), cf.detach(), g)
    finally:
        pm.PLAN_OVERLAP = True
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    if mode == "train":
        assert (a[3] - b[3]).abs().max().item() <= 1e-5 * b[3].abs().max().item()


def test_dense_scale_is_left_alone_above_the_fraction_threshold():
    from pdanet_amd import pointnet2_utils as pu
    rng = np.random.default_rng(3)
    idx, _ = padded_idx(64, 16, 2000, rng, p_single=0.0, p_full=1.0, empty_rows=0)
    (plan,) = pu.ragged_plans([torch.from_numpy(idx).cuda().view(1, 64, 16)])
    assert plan.tokens == 64 * 16 and plan.fraction == 1.0 > pu.RAGGED_MAX_FRACTION


@pytest.mark.parametrize("G,S,C", [(300, 16, 32), (1000, 32, 64), (77, 8, 128)])
def test_weighted_batch_norm_equals_dense_batch_norm_on_expanded_rows(G, S, C):
    """pda_bn_relu_{fwd,bwd}_weighted on the compact rows == the plain kernels on the dense (G*S, C) tensor in which token 0
    of every group is repeated: same y, same running statistics, and the compact gradient is the sum over the copies."""
    from pdanet_amd import pointnet2_utils as pu
    rng = np.random.default_rng(G + C)
    idx, cnt = padded_idx(G, S, 4000, rng)
    (plan,) = pu.ragged_plans([torch.from_numpy(idx).cuda().view(1, G, S)])
    U = plan.tokens
    assert float(plan.roww.sum()) == G * S
    g = torch.Generator("cuda").manual_seed(C)
    xc = torch.randn(U, C, device="cuda", generator=g).requires_grad_(True)
    slot = torch.arange(S, device="cuda").view(1, S)
    src = plan.off.long()[:-1].view(G, 1) + torch.where(slot < plan.cnt.long().view(G, 1), slot, torch.zeros_like(slot))
    xd = xc.detach()[src].reshape(G * S, C).requires_grad_(True)
    bn_c, bn_d = torch.nn.BatchNorm1d(C).cuda().train(), torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        for b in (bn_c, bn_d):
            b.weight.copy_(torch.linspace(0.5, 1.5, C)); b.bias.copy_(torch.linspace(-0.3, 0.3, C))
    yc = pu.BatchNormReLUWeighted.apply(xc, bn_c.weight, bn_c.bias, bn_c.running_mean, bn_c.running_var, bn_c.eps, bn_c.momentum,
                                        plan.roww, G * S)
    yd = pu.batch_norm_relu(bn_d, xd)
    assert (yd.view(G, S, C) - yc[src]).abs().max().item() < 2e-5
    assert torch.allclose(bn_c.running_mean, bn_d.running_mean, atol=1e-6) and torch.allclose(bn_c.running_var, bn_d.running_var, rtol=1e-5)
    go_c = torch.randn(U, C, device="cuda", generator=g)
    valid = slot < plan.cnt.long().view(G, 1)
    go_d = torch.where(valid.unsqueeze(-1), go_c[src], torch.zeros((), device="cuda")).reshape(G * S, C)    # total on one copy
    gc = torch.autograd.grad(yc, (xc, bn_c.weight, bn_c.bias), go_c)
    gd = torch.autograd.grad(yd, (xd, bn_d.weight, bn_d.bias), go_d)
    ref = torch.zeros_like(gc[0]).index_add_(0, src.reshape(-1), gd[0])
    assert (ref - gc[0]).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    for a, b in zip(gc[1:], gd[1:]):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())

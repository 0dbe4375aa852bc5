"""csrc/densitynet.hip against the torch module (float64): forward, running statistics, parameter gradients."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 3, 5, 1), (2, 300, 16, 1), (2, 4096, 32, 1), (3, 1111, 8, 1)])
def test_fused_densitynet_matches_module(shape):
    from pdanet_amd import pointnet2_modules as pm, pointnet2_utils as pu
    torch.manual_seed(sum(shape))
    dn = pm.DensityNet().cuda().train()
    with torch.no_grad():
        for b in dn.mlp_bns:
            b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.3); b.running_mean.normal_(); b.running_var.uniform_(0.5, 2)
    ref = pm.DensityNet().cuda().double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in dn.state_dict().items()})
    x = torch.rand(shape, device="cuda")
    assert pu.DensityNetFused.supported(x, dn)
    y = pu.densitynet(dn, x)
    # module works channel-major (B, 1, M, ns)
    yr = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert (y.double() - yr).abs().max().item() < 2e-5
    go = torch.randn_like(y)
    params = [p for p in dn.parameters()]
    g = torch.autograd.grad(y, params, go)
    gr = torch.autograd.grad(yr, [p for p in ref.parameters()], go.double())
    ntok = x.numel()
    for (n, _), a, b in zip(dn.named_parameters(), g, gr):
        assert a.shape == b.shape
        tol = 3e-5 * max(1.0, b.abs().max().item())
        if "mlp_convs" in n:
            # every conv feeds a BatchNorm, which is invariant to the scale and shift of its input: these gradients
            # are the O(eps) remainder of two sums of magnitude ~ntok that cancel, so fp32 noise grows with ntok
            tol += 2e-8 * ntok
        assert (a.double() - b).abs().max().item() < tol, n
    for b1, b2 in zip(dn.mlp_bns, ref.mlp_bns):
        assert torch.allclose(b1.running_mean.double(), b2.running_mean, atol=1e-6)
        assert torch.allclose(b1.running_var.double(), b2.running_var, atol=1e-6, rtol=1e-5)
        assert int(b1.num_batches_tracked) == 1


@pytest.mark.parametrize("G,ns,p_single,p_full,sparse_grad", [(7, 8, 0.3, 0.1, True), (1000, 16, 0.7, 0.05, True),
                                                               (8192, 32, 0.3, 0.1, True), (4099, 32, 0.0, 1.0, True),
                                                               (1500, 16, 1.0, 0.0, False), (8192, 32, 0.5, 0.1, False)])
def test_densitynet_on_distinct_slots_equals_dense(G, ns, p_single, p_full, sparse_grad):
    """pda_densitynet_fwd_unique / _bwd_unique (the passes walk only the distinct slots of padded neighbour lists, slot 0
    weighted by its multiplicity) against the dense passes on the same (groups, ns) tensor whose repeat slots hold slot 0's
    value: y in EVERY slot, running statistics, parameter gradients -- with the gradient the ragged consumer leaves (zero
    on the repeats) and with a gradient on every slot (a dense consumer: the unique backward sums a token's copies)."""
    import numpy as np
    from pdanet_amd import pointnet2_modules as pm, pointnet2_utils as pu
    from test_ragged_tokens import padded_idx
    rng = np.random.default_rng(G + ns)
    idx, cnt = padded_idx(G, ns, 5000, rng, p_single, p_full, empty_rows=min(2, G - 1) if p_full < 1.0 else 0)
    idx_t = torch.from_numpy(idx).cuda().view(1, G, ns)
    parts, _ = pu.ragged_plan_parts([idx_t])
    torch.manual_seed(G)
    # x per DISTINCT neighbour; a repeat slot holds what slot 0 holds (the same neighbour of the same centre)
    val = torch.rand(5000, device="cuda")
    x = val[idx_t.long()].view(1, G, ns, 1).contiguous()
    models = []
    for _ in range(2):
        torch.manual_seed(3)
        dn = pm.DensityNet().cuda().train()
        with torch.no_grad():
            for b in dn.mlp_bns:
                b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.3); b.running_mean.normal_(); b.running_var.uniform_(0.5, 2)
        models.append(dn)
    y_u = pu.densitynet(models[0], x, parts[0])
    y_d = pu.densitynet(models[1], x, None)
    assert (y_u - y_d).abs().max().item() < 2e-6
    go = torch.randn_like(y_d)
    if sparse_grad:
        slot = torch.arange(ns, device="cuda").view(1, 1, ns, 1)
        go = torch.where(slot < torch.from_numpy(cnt).cuda().view(1, G, 1, 1), go, torch.zeros_like(go))
    g_u = torch.autograd.grad(y_u, list(models[0].parameters()), go)
    g_d = torch.autograd.grad(y_d, list(models[1].parameters()), go)
    ntok = x.numel()
    for (n, _), a, b in zip(models[0].named_parameters(), g_u, g_d):
        tol = 3e-5 * max(1.0, b.abs().max().item()) + (4e-8 * ntok if "mlp_convs" in n else 0.0)   # see the test above
        assert (a - b).abs().max().item() < tol, n
    for b1, b2 in zip(models[0].mlp_bns, models[1].mlp_bns):
        assert torch.allclose(b1.running_mean, b2.running_mean, atol=1e-6)
        assert torch.allclose(b1.running_var, b2.running_var, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("shape", [(1, 3, 5, 1), (2, 4096, 32, 1), (3, 1111, 8, 1)])
def test_inference_kernel_matches_module_in_eval_mode(shape):
    """pda_densitynet_eval (BatchNorm folded into the three layers, one launch) against the torch module in eval mode (float64),
    and the cached parameter block follows a change of the running statistics."""
    from pdanet_amd import pointnet2_modules as pm
    torch.manual_seed(sum(shape))
    dn = pm.DensityNet().cuda().eval()
    with torch.no_grad():
        for b in dn.mlp_bns:
            b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.3); b.running_mean.normal_(0, 0.2); b.running_var.uniform_(0.5, 2)
    x = torch.rand(shape, device="cuda")
    for _ in range(2):
        ref = pm.DensityNet().cuda().double().eval()
        ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in dn.state_dict().items()})
        with torch.no_grad():
            assert pm._densitynet_eval_ok(dn, x)
            y = pm._densitynet_eval(dn, x)
            yr = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        assert y.shape == x.shape and (y.double() - yr).abs().max().item() < 1e-5
        with torch.no_grad():                      # second round: other statistics, the cached block must not survive them
            dn.mlp_bns[1].running_mean.add_(0.5)
            dn.mlp_bns[2].running_var.mul_(1.7)


def test_scales_sharing_their_launches_equal_one_scale_at_a_time():
    """pda_densitynet_{fwd,bwd}_multi (the scales of a layer as blockIdx.y of one set of launches) against one call per scale:
    three problems of different sizes -- distinct slots, every token, distinct slots -- give the same y, running statistics and
    parameter gradients bit for bit (per problem the same blocks add the same numbers in the same order)."""
    import numpy as np
    from pdanet_amd import pointnet2_modules as pm, pointnet2_utils as pu
    from test_ragged_tokens import padded_idx
    rng = np.random.default_rng(5)
    shapes = [(3000, 16), (700, 32), (9000, 8)]
    xs, parts = [], []
    val = torch.rand(6000, device="cuda")
    for k, (G, ns) in enumerate(shapes):
        idx, _ = padded_idx(G, ns, 6000, rng)
        idx_t = torch.from_numpy(idx).cuda().view(1, G, ns)
        xs.append(val[idx_t.long()].view(1, G, ns, 1).contiguous())
        parts.append(None if k == 1 else pu.ragged_plan_parts([idx_t])[0][0])

    def models():
        out = []
        for k in range(3):
            torch.manual_seed(10 + k)
            dn = pm.DensityNet().cuda().train()
            with torch.no_grad():
                for b in dn.mlp_bns:
                    b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.3); b.running_mean.normal_(); b.running_var.uniform_(0.5, 2)
            out.append(dn)
        return out
    gos = [torch.randn_like(x) for x in xs]
    res = {}
    try:
        for multi in (True, False):
            pu.DENSITYNET_MULTI = multi
            dns = models()
            ys = pu.densitynet_multi(dns, xs, parts)
            grads = torch.autograd.grad(ys, [p for dn in dns for p in dn.parameters()], gos)
            res[multi] = (ys, grads, [t.clone() for dn in dns for b in dn.mlp_bns for t in (b.running_mean, b.running_var, b.num_batches_tracked)])
    finally:
        pu.DENSITYNET_MULTI = True
    for a, b in zip(res[True][0] + list(res[True][1]) + res[True][2], res[False][0] + list(res[False][1]) + res[False][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,M,ns,r", [(1, 50, 7, 16, 0.8), (2, 4096, 1000, 32, 1.6), (3, 300, 129, 8, 4.8)])
def test_pda_geometry_kernel_matches_torch_expression(B, N, M, ns, r):
    from pdanet_amd import pointnet2_batch_cuda as ext, pointnet2_utils as pu
    torch.manual_seed(N + ns)
    xyz = torch.randn(B, N, 3, device="cuda") * 2
    new_xyz = xyz[:, :M].contiguous() + 0.01
    idx = torch.randint(0, N, (B, M, ns), device="cuda", dtype=torch.int32)
    rppe = torch.empty(B, M, ns, 12, device="cuda")
    dscale = torch.empty(B, M, ns, 1, device="cuda")
    ext.pda_geometry(xyz, new_xyz, idx, rppe, dscale, B, N, M, ns, r)
    nbr = pu.group_rows(xyz, idx)
    centre = new_xyz.unsqueeze(2)
    diff = nbr - centre
    dist = torch.norm(diff, dim=-1, keepdim=True)
    density = torch.exp(-dist ** 2 / (2 * r ** 2)) / (2.5 * r)
    want_d = density / density.max(dim=2, keepdim=True)[0]
    want_r = torch.cat([centre.expand(B, M, ns, 3), nbr, -diff, diff / r], dim=-1)
    assert torch.allclose(rppe, want_r, rtol=1e-6, atol=1e-6)
    assert torch.allclose(dscale, want_d, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("B,N,M,ns,C", [(1, 40, 5, 16, 16), (2, 4096, 1000, 32, 64), (2, 500, 129, 8, 128)])
def test_token_assembly_matches_torch_chain(B, N, M, ns, C):
    from pdanet_amd import pointnet2_utils as pu
    torch.manual_seed(N + C)
    rppe = torch.randn(B, M, ns, C, device="cuda", requires_grad=True)
    dscale = torch.rand(B, M, ns, 1, device="cuda", requires_grad=True)
    feats = torch.randn(B, N, C, device="cuda", requires_grad=True)
    glob = torch.randn(B, M, C, device="cuda", requires_grad=True)
    idx = torch.randint(0, N, (B, M, ns), device="cuda", dtype=torch.int32)
    assert pu.AssembleTokens.supported(rppe, feats)
    x = pu.AssembleTokens.apply(rppe, dscale, feats, idx, glob)
    g = pu.group_rows(feats, idx)
    ref = torch.cat([rppe, g * dscale, g, glob.unsqueeze(2).expand(-1, -1, ns, -1)], dim=-1)
    assert torch.equal(x, ref)
    go = torch.randn_like(x)
    got = torch.autograd.grad(x, [rppe, dscale, feats, glob], go)
    want = torch.autograd.grad(ref, [rppe, dscale, feats, glob], go)
    for a, b, name in zip(got, want, ["rppe", "dscale", "feats", "glob"]):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item()), name

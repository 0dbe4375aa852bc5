"""csrc/densitynet.hip against the torch module (float64): forward, running statistics, parameter gradients."""
import pytest
import torch

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

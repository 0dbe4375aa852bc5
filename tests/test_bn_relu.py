"""csrc/bn_relu.hip against torch (float64 reference): training-mode BatchNorm + ReLU over the last dim.
Floating point: outputs within 2e-5, gradients within 2e-5 of the largest gradient, running statistics 1e-6."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 7, 4), (3, 50, 16, 8), (2, 1000, 32, 16), (1, 4099, 64), (2, 300, 8, 512), (5, 1024), (2, 128, 16, 256)])
def test_forward_backward_running_stats(shape):
    from pdanet_amd import pointnet2_utils as pu
    torch.manual_seed(sum(shape))
    c = shape[-1]
    bn = nn.BatchNorm1d(c).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    ref = nn.BatchNorm1d(c).cuda().double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    x = (torch.randn(shape, device="cuda") * 1.7 + 0.4).requires_grad_(True)
    assert pu.BatchNormReLU.supported(x, bn)
    y = pu.batch_norm_relu(bn, x)
    xr = x.detach().double().requires_grad_(True)
    yr = F.relu(ref(xr.reshape(-1, c))).reshape(shape)
    assert (y.double() - yr).abs().max().item() < 2e-5
    go = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, [x, bn.weight, bn.bias], go)
    rx, rw, rb = torch.autograd.grad(yr, [xr, ref.weight, ref.bias], go.double())
    for a, b in ((gx, rx), (gw, rw), (gb, rb)):
        assert (a.double() - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref.running_var, atol=1e-6, rtol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_unsupported_cases_use_torch():
    from pdanet_amd import pointnet2_modules as pm, pointnet2_utils as pu
    bn = nn.BatchNorm1d(96).cuda().train()
    x = torch.randn(4, 10, 96, device="cuda")
    assert not pu.BatchNormReLU.supported(x, bn)                      # not a power of two
    assert not pu.BatchNormReLU.supported(torch.randn(4, 10, 64, device="cuda"), nn.BatchNorm1d(64).cuda().eval())
    y = pm._bn_relu_lastdim(bn, x)
    assert y.shape == x.shape and float(y.min()) >= 0


def test_mlp_helper_fused_equals_unfused():
    from pdanet_amd import pointnet2_modules as pm
    torch.manual_seed(0)
    mlp = nn.Sequential(nn.Conv2d(12, 32, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(),
                        nn.Conv2d(32, 64, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU()).cuda().train()
    x = torch.randn(2, 500, 16, 12, device="cuda", requires_grad=True)
    outs = []
    for flag in (True, False):
        pm.FUSED_BN_RELU = flag
        y = pm._mlp_lastdim(mlp, x)
        (g,) = torch.autograd.grad(y.square().mean(), x)
        outs.append((y.detach(), g))
    pm.FUSED_BN_RELU = True
    assert torch.allclose(outs[0][0], outs[1][0], atol=2e-5, rtol=1e-5)
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-8 + 2e-5 * outs[1][1].abs().max().item())


@pytest.mark.parametrize("rows,c", [(7, 4), (4099, 64), (32768, 256), (1000, 512)])
def test_dense_bf16_boundary_variants(rows, c):
    """pda_bn_relu_{fwd,bwd}_mixed: bf16 tensors on the GEMM side (x, grad_y in; y, grad_x out), same arithmetic.  On
    bf16-valued inputs: statistics, fp32 outputs and parameter gradients bit-identical to the fp32 entry points, bf16
    outputs = the fp32 results rounded to nearest even."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(rows + c)
    dev = "cuda"
    xb = (torch.randn(rows, c, device=dev) * 1.3 + 0.2).bfloat16()
    gyb = torch.randn(rows, c, device=dev).bfloat16()
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    scratch = torch.empty(ext.bn_relu_scratch_bytes(c), dtype=torch.uint8, device=dev)

    def fwd(x, out_dtype):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        y, st = torch.empty(rows, c, device=dev, dtype=out_dtype), torch.empty(2, c, device=dev)
        ext.bn_relu_fwd(x, g, b, rm, rv, y, st, scratch, rows, c, 1e-5, 0.1)
        return y, st, rm, rv
    y0, st0, rm0, rv0 = fwd(xb.float(), torch.float32)
    for x, od in ((xb, torch.float32), (xb, torch.bfloat16), (xb.float(), torch.bfloat16)):
        y, st, rm, rv = fwd(x, od)
        assert torch.equal(st, st0) and torch.equal(rm, rm0) and torch.equal(rv, rv0)
        assert torch.equal(y, y0 if od == torch.float32 else y0.bfloat16())

    def bwd(x, gy):
        gx, gg, gb_ = torch.empty_like(x), torch.empty(c, device=dev), torch.empty(c, device=dev)
        ext.bn_relu_bwd(x, gy, g, b, st0, gx, gg, gb_, scratch, rows, c)
        return gx, gg, gb_
    gx0, gg0, gb0 = bwd(xb.float(), gyb.float())
    for x, gy in ((xb, gyb), (xb, gyb.float()), (xb.float(), gyb)):
        gx, gg, gb_ = bwd(x, gy)
        assert gx.dtype == x.dtype
        assert torch.equal(gg, gg0) and torch.equal(gb_, gb0)
        assert torch.equal(gx, gx0 if x.dtype == torch.float32 else gx0.bfloat16())


@pytest.mark.parametrize("shape", [(2, 5, 16, 64), (1, 300, 32, 512), (3, 7, 1, 8), (2, 64, 64, 256)])
@pytest.mark.parametrize("bf16_x", [False, True])
def test_bn_relu_max_pool_fused(shape, bf16_x):
    """BatchNormReLUMaxPool == max over nsample of BatchNormReLU: same forward values and running statistics bit for bit,
    gradients within fp32 re-association noise (the dense gradient of the max-pool is generated on the fly)."""
    from pdanet_amd import pointnet2_utils as pu
    torch.manual_seed(sum(shape))
    c = shape[-1]
    bns = [nn.BatchNorm2d(c).cuda().train() for _ in range(2)]
    with torch.no_grad():
        bns[0].weight.uniform_(0.5, 1.5); bns[0].bias.normal_(0, 0.3)
        bns[1].load_state_dict(bns[0].state_dict())
    x0 = torch.randn(shape, device="cuda") * 1.5 + 0.3
    if bf16_x:
        x0 = x0.bfloat16()
    xs = [x0.clone().requires_grad_(True) for _ in range(2)]
    assert pu.BatchNormReLUMaxPool.supported(xs[0], bns[0])
    y_f = pu.batch_norm_relu_max_pool(bns[0], xs[0])
    y_u = pu.batch_norm_relu(bns[1], xs[1]).max(dim=-2)[0]
    assert y_f.dtype == torch.float32 and torch.equal(y_f, y_u)
    assert torch.equal(bns[0].running_mean, bns[1].running_mean) and torch.equal(bns[0].running_var, bns[1].running_var)
    go = torch.randn_like(y_f)
    gf = torch.autograd.grad(y_f, [xs[0], bns[0].weight, bns[0].bias], go)
    gu = torch.autograd.grad(y_u, [xs[1], bns[1].weight, bns[1].bias], go)
    assert gf[0].dtype == x0.dtype
    tol = 2e-2 if bf16_x else 2e-5
    for a, b in zip(gf, gu):
        assert (a.double() - b.double()).abs().max().item() <= tol * max(1e-3, b.double().abs().max().item())


def test_eval_bn_folding_matches_unfolded_mlp():
    """Inference: [Conv1x1 -> BN(eval) -> ReLU]* with the BatchNorm folded into the convolution weights (and the ReLU in
    the GEMM epilogue) against the op-by-op path; with and without the max-pool tail."""
    from pdanet_amd import pointnet2_modules as pm
    torch.manual_seed(5)
    layers = []
    c_in = 19
    for c_out in (32, 64, 128):
        layers += [nn.Conv2d(c_in, c_out, 1, bias=(c_out == 64)), nn.BatchNorm2d(c_out), nn.ReLU()]
        c_in = c_out
    mlp = nn.Sequential(*layers).cuda().eval()
    with torch.no_grad():
        for m in mlp:
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    x = torch.randn(2, 50, 16, 19, device="cuda")
    outs = {}
    with torch.no_grad():
        for fold in (True, False):
            pm.FOLD_EVAL_BN = fold
            try:
                outs[fold] = (pm._mlp_lastdim(mlp, x), pm._mlp_lastdim(mlp, x, pool=True))
            finally:
                pm.FOLD_EVAL_BN = True
        ref = mlp(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and torch.allclose(a, b, atol=2e-5, rtol=1e-5)
    assert torch.allclose(outs[True][0], ref, atol=2e-5, rtol=1e-5)
    assert torch.allclose(outs[True][1], ref.max(dim=2)[0], atol=2e-5, rtol=1e-5)
    # a parameter update invalidates the cached folded weights
    with torch.no_grad():
        mlp[0].weight.mul_(2.0)
        again = pm._mlp_lastdim(mlp, x)
        ref2 = mlp(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.allclose(again, ref2, atol=4e-5, rtol=1e-5)

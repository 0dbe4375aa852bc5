"""-m gpu: training form of the vanilla-SA group MLP on this repo's f32 MFMA kernels (csrc/sa_mlp.hip lin_cols_kernel:
`pda_linear_cols`, `pda_sa_gather_linear`) against a plain PyTorch fp32 reference of the same op
(pointnet2_modules.py:1657-1662: Conv2d 1x1 over grouped rows; pointnet2_utils.py:671-704: QueryAndGroup).
Tolerance: both sides are fp32 with different summation orders (k-ordered fmaf chain vs the library's tiles): 2e-5 of
the output scale forward, 1e-4 of the gradient scale backward."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from detweights import fill_deterministic  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,K,N", [(1000, 256, 256), (4099, 256, 512), (257, 512, 512), (300, 512, 1024), (64, 256, 128)])
def test_linear_cols_forward_backward(T, K, N):
    from pdanet_amd import pointnet2_utils as pu
    g = torch.Generator("cuda").manual_seed(T + K)
    x = torch.randn(T, K, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).requires_grad_(True)
    assert pu.LinearColsMFMA.supported(x, w)
    y = pu.LinearColsMFMA.apply(x, w)
    ref = torch.nn.functional.linear(x.double(), w.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    go = torch.randn(T, N, device="cuda", generator=g)
    gx, gw = torch.autograd.grad(y, (x, w), go)
    rx, rw = torch.autograd.grad(ref, (x, w), go.double())
    assert (gx.double() - rx).abs().max().item() <= 1e-4 * rx.abs().max().item()
    assert (gw.double() - rw).abs().max().item() <= 1e-4 * rw.abs().max().item()


@pytest.mark.parametrize("M,ns,n_out", [(100, 16, 256), (257, 32, 256), (33, 64, 128)])
def test_gather_linear_equals_group_then_linear(M, ns, n_out):
    from pdanet_amd import pointnet2_utils as pu, synth
    B, N, C = 2, 2048, 256
    xyz = torch.from_numpy(synth.batch_xyz(B, N, config_id=3)).cuda()
    g = torch.Generator("cuda").manual_seed(M)
    new_xyz = (xyz[:, :M] + 0.1 * torch.randn(B, M, 3, device="cuda", generator=g)).contiguous().requires_grad_(True)
    feats = torch.randn(B, N, C, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(n_out, 3 + C, device="cuda", generator=g) / 16).requires_grad_(True)
    idx = pu.ball_query(6.0, ns, xyz, new_xyz.detach())
    assert pu.SaGatherLinear.supported(xyz, feats, w)
    y = pu.SaGatherLinear.apply(xyz, new_xyz, feats, idx, w)
    x0 = torch.cat([pu.group_rows(xyz, idx) - new_xyz.unsqueeze(2), pu.group_rows(feats, idx)], dim=-1)
    ref = torch.nn.functional.linear(x0.double(), w.double())
    assert y.shape == (B, M, ns, n_out)
    assert (y.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    go = torch.randn_like(y)
    got = torch.autograd.grad(y, (new_xyz, feats, w), go)
    exp = torch.autograd.grad(ref, (new_xyz, feats, w), go.double())
    for a, b, name in zip(got, exp, ("new_xyz", "feats", "weight")):
        assert (a.double() - b).abs().max().item() <= 2e-4 * max(1e-6, b.abs().max().item()), name


def _layer5():
    from pdanet_amd.pointnet2_modules import PointnetSAModuleMSG_WithSampling
    layer = PointnetSAModuleMSG_WithSampling(
        npoint_list=[256], sample_range_list=[-1], sample_type_list=["D-FPS"], radii=[4.8, 8.4], nsamples=[16, 32],
        mlps=[[256, 256, 256, 512], [256, 256, 512, 512]], use_xyz=True, dilated_group=False, aggregation_mlp=[512],
        confidence_mlp=None, num_class=5)
    return fill_deterministic(layer).cuda().train()


def test_sa_layer_training_mfma_path_equals_library_path():
    """ONCE layer 5 in small, train-mode BatchNorm: the MFMA training path (gather-fused first contraction, own forward and
    input-gradient GEMMs) against the op-by-op path on library GEMMs -- outputs, input gradients, every parameter gradient,
    running statistics."""
    from pdanet_amd import pointnet2_utils as pu, synth
    xyz = torch.from_numpy(synth.batch_xyz(2, 1024, config_id=4)).cuda()
    feats0 = torch.randn(2, 256, 1024, device="cuda", generator=torch.Generator("cuda").manual_seed(9))
    res = {}
    used = []
    try:
        for flag in (True, False):
            pu.SA_MFMA_TRAIN = flag
            pu.SA_MFMA_EVENTS = used if flag else None
            layer = _layer5()
            feats = feats0.clone().requires_grad_(True)
            ctr = (xyz[:, :256] + 0.05).contiguous().requires_grad_(True)
            _, nf, _, _ = layer(xyz, feats, None, ctr_xyz=ctr)
            nf.pow(2).mean().backward()
            res[flag] = (nf.detach(), feats.grad.clone(), ctr.grad.clone(),
                         {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in layer.state_dict().items() if "running" in k})
    finally:
        pu.SA_MFMA_TRAIN, pu.SA_MFMA_EVENTS = True, None
    torch.cuda.synchronize()
    assert len(used) >= 2 * 5, "the MFMA kernels did not run: %d launches" % len(used)    # 3 forward + 2 input-gradient per scale
    a, b = res[True], res[False]
    assert (a[0] - b[0]).abs().max().item() <= 2e-4 * max(1.0, b[0].abs().max().item())
    for i in (1, 2):
        assert (a[i] - b[i]).abs().max().item() <= 2e-3 * b[i].abs().max().item() + 1e-8
    gmax = max(float(v.abs().max()) for v in b[3].values())
    assert set(a[3]) == set(b[3])
    for k in b[3]:
        assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-2 * float(b[3][k].abs().max()) + 1e-3 * gmax, k
    for k in b[4]:
        assert torch.allclose(a[4][k], b[4][k], rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("b,n,m,ns,c1", [(2, 512, 96, 16, 256), (1, 300, 50, 13, 128), (2, 256, 33, 64, 512)])
def test_sa_xyz_grad_kernel_against_fp64(b, n, m, ns, c1):
    """csrc/sa_xyz_grad.hip: the coordinate columns of the first SA layer's backward (weight gradient columns 0:3 and the
    centres' gradient) against the same sums in torch fp64."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    g = torch.Generator().manual_seed(b * 100 + ns)
    xyz = (torch.rand(b, n, 3, generator=g) * 10).cuda()
    new_xyz = (torch.rand(b, m, 3, generator=g) * 10).cuda()
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).cuda()
    gz = torch.randn(b * m * ns, c1, generator=g).cuda()
    w = torch.randn(c1, 3 + 7, generator=g).cuda()
    dw = torch.full_like(w, 7.0)
    gnew = torch.empty(b, m, 3, device="cuda")
    ext.sa_xyz_grad(gz, xyz, new_xyz, idx, w, dw, gnew, b, n, m, ns, c1)
    gx = torch.stack([xyz[i][idx[i].long()] for i in range(b)]).double() - new_xyz.double().unsqueeze(2)      # (b, m, ns, 3)
    gz64 = gz.double().view(b, m, ns, c1)
    want_dw = torch.einsum("bmso,bmsd->od", gz64, gx)
    want_new = -(gz64.sum(dim=2) @ w[:, :3].double())
    assert (dw[:, :3].double() - want_dw).abs().max().item() <= 2e-5 * want_dw.abs().max().item()
    assert (dw[:, 3:] == 7.0).all()                       # the feature columns are not this kernel's
    assert (gnew.double() - want_new).abs().max().item() <= 2e-5 * want_new.abs().max().item()


def test_sa_point_linear_equals_gather_linear():
    """SaPointLinear (per-point projection + row gather; backward: scatter first, then (B N)-row products) against
    SaGatherLinear (the per-token contraction): the same z1 and the same gradients for weight, features and centres."""
    from pdanet_amd import pointnet2_utils as pu, synth
    B, N, M, ns, C, c1 = 2, 2048, 1024, 32, 256, 256
    xyz = torch.from_numpy(synth.batch_xyz(B, N, config_id=6)).cuda()
    g = torch.Generator().manual_seed(4)
    ctr0 = (xyz[:, :M] + 0.2 * torch.randn(B, M, 3, generator=g).cuda()).contiguous()
    idx = pu.ball_query(8.4, ns, xyz, ctr0)
    feats0 = torch.randn(B, N, C, generator=g).cuda()
    w0 = (torch.randn(c1, 3 + C, generator=g) * 0.05).cuda()
    gz = torch.randn(B, M, ns, c1, generator=g).cuda()
    res = []
    for fn in (pu.SaPointLinear, pu.SaGatherLinear):
        ctr, feats, w = ctr0.clone().requires_grad_(True), feats0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        assert fn.supported(xyz, feats, w)
        z = fn.apply(xyz, ctr, feats, idx, w)
        z.backward(gz)
        res.append((z.detach(), ctr.grad, feats.grad, w.grad))
    (za, ca, fa, wa), (zb, cb, fb, wb) = res
    assert (za - zb).abs().max().item() <= 2e-5 * zb.abs().max().item()
    for a, b, name in ((ca, cb, "centres"), (fa, fb, "features"), (wa, wb, "weight")):
        assert (a - b).abs().max().item() <= 3e-4 * b.abs().max().item(), name


def test_sa_wide_chain_equals_unfused_path():
    """SaWideChainTrain (BatchNorm statistics in the GEMM epilogues, normalisation + ReLU in the operand loads of the next GEMM
    and of the weight gradient, the BatchNorm backward's reduction in the input-gradient GEMM's epilogue) against the same
    scale op by op (SaPointLinear, bn_relu passes, split GEMMs, fused pool): outputs, input gradients, every parameter
    gradient, running statistics -- at ONCE layer 5's token counts (65536 and 131072)."""
    from pdanet_amd import pointnet2_utils as pu, synth
    from pdanet_amd.pointnet2_modules import PointnetSAModuleMSG_WithSampling
    B, N, M = 2, 2048, 1024
    xyz = torch.from_numpy(synth.batch_xyz(B, N, config_id=6)).cuda()
    feats0 = torch.randn(B, 256, N, device="cuda", generator=torch.Generator("cuda").manual_seed(11))
    res = {}
    used = {}
    try:
        for flag in (True, False):
            pu.FUSED_WIDE_CHAIN_OK = flag
            used[flag] = pu.SA_MFMA_EVENTS = []
            layer = PointnetSAModuleMSG_WithSampling(
                npoint_list=[M], sample_range_list=[-1], sample_type_list=["D-FPS"], radii=[8.4, 12.8], nsamples=[32, 64],
                mlps=[[256, 256, 256, 512], [256, 256, 512, 512]], use_xyz=True, dilated_group=False, aggregation_mlp=[512],
                confidence_mlp=None, num_class=5)
            layer = fill_deterministic(layer).cuda().train()
            feats = feats0.clone().requires_grad_(True)
            ctr = (xyz[:, :M] + 0.05).contiguous().requires_grad_(True)
            for _ in range(2):                     # two iterations: the running statistics move twice
                _, nf, _, _ = layer(xyz, feats, None, ctr_xyz=ctr)
                nf.pow(2).mean().backward()
            res[flag] = (nf.detach(), feats.grad.clone(), ctr.grad.clone(),
                         {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in layer.state_dict().items() if "running" in k or "num_batches" in k})
    finally:
        pu.FUSED_WIDE_CHAIN_OK, pu.SA_MFMA_EVENTS = True, None
    torch.cuda.synchronize()
    assert len(used[True]) == 2 * 2 * 4, len(used[True])       # per scale and iteration: 2 forward + 2 input-gradient launches
    a, b = res[True], res[False]
    assert (a[0] - b[0]).abs().max().item() <= 2e-5 * max(1.0, b[0].abs().max().item())
    for i in (1, 2):
        assert (a[i] - b[i]).abs().max().item() <= 2e-4 * b[i].abs().max().item() + 1e-9
    gmax = max(float(v.abs().max()) for v in b[3].values())
    assert set(a[3]) == set(b[3])
    for k in b[3]:
        assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-3 * float(b[3][k].abs().max()) + 1e-4 * gmax, k
    for k in b[4]:
        assert torch.allclose(a[4][k].float(), b[4][k].float(), rtol=1e-5, atol=1e-6), k


def test_gemm_split_bn_pieces_against_fp64():
    """pda_gemm_split_bn on its own: the input BatchNorm + ReLU in the operand load and the output statistics of the epilogue
    against float64; pda_linear_wgrad_bn against float64."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    T, K, Nn = 33001, 256, 512                     # a ragged last tile
    g = torch.Generator("cuda").manual_seed(5)
    x = torch.randn(T, K, device="cuda", generator=g) * 1.5 + 0.2
    w = torch.randn(Nn, K, device="cuda", generator=g) * 0.05
    gam, bet = torch.rand(K, device="cuda", generator=g) + 0.5, torch.randn(K, device="cuda", generator=g) * 0.3
    mean, var = x.double().mean(0), x.double().var(0, unbiased=False)
    mi = torch.cat([mean, 1.0 / torch.sqrt(var + 1e-5)]).float().contiguous()
    a64 = torch.relu((x.double() - mi[:K].double()) * mi[K:].double() * gam.double() + bet.double())
    wf = ext.linear_split_pack(w, Nn, K)
    tiles = ext.gemm_split_bn_tiles(T)
    y = torch.full((T, Nn), float("nan"), device="cuda")
    part = torch.full((tiles * 2 * Nn,), float("nan"), dtype=torch.float64, device="cuda")
    ext.gemm_split_bn(x, wf, y, T, K, Nn, in_bn=(mi, gam, bet), stats_mode=1, partial=part)
    y64 = a64 @ w.double().t()
    scale = a64.abs() @ w.double().abs().t() + 1e-30
    assert ((y.double() - y64).abs() / scale).max().item() < 3e-6
    p = part.view(tiles, 2, Nn).sum(0)
    assert torch.allclose(p[0], y.double().sum(0), rtol=1e-12, atol=1e-9) and torch.allclose(p[1], (y.double() ** 2).sum(0), rtol=1e-12)
    # weight gradient against relu(bn(x)) formed in the operand load
    T2 = 65536
    x2 = torch.randn(T2, K, device="cuda", generator=g) + 0.1
    g2 = torch.randn(T2, Nn, device="cuda", generator=g)
    mi2 = torch.cat([x2.double().mean(0), 1.0 / torch.sqrt(x2.double().var(0, unbiased=False) + 1e-5)]).float().contiguous()
    dw = torch.empty(Nn, K, device="cuda")
    ext.linear_wgrad_bn(x2, g2, dw, T2, K, Nn, mi2, gam, bet)
    a2 = torch.relu((x2 - mi2[:K]) * mi2[K:] * gam + bet).double()
    want = g2.double().t() @ a2
    assert ((dw.double() - want).abs() / (g2.double().abs().t() @ a2.abs() + 1e-30)).max().item() < 3e-6

"""-m gpu: the narrow vanilla SA scale in training form (csrc/sa_train_small.hip, row a8: ONCE / KITTI layer 0) against
(i) a plain torch fp64 statement of the reference chain QueryAndGroup -> [Conv2d 1x1 -> BatchNorm2d (batch statistics)
-> ReLU] x 3 -> max over nsample (pointnet2_modules.py:1657-1670, pointnet2_utils.py:671-704) and (ii) this repo's
layer-by-layer path (library / MFMA GEMMs + csrc/bn_relu.hip), through the SA module itself.
Tolerances: forward 2e-5 of scale, running statistics 1e-5, gradients 2e-4 of scale (fp32 sums over 0.03-1 M tokens)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _scene(B, N, seed):
    from pdanet_amd import synth
    pts = synth.batch_points(B, N, config_id=seed, dist="L")
    t = torch.from_numpy(pts).cuda()
    xyz = t[:, 1:4].reshape(B, N, 3).contiguous()
    feats = t[:, 4:5].reshape(B, N, 1).contiguous()
    return xyz, feats


def _mlp(dims, seed):
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    layers = []
    for k in range(3):
        conv = nn.Conv2d(dims[k], dims[k + 1], 1, bias=False)
        bn = nn.BatchNorm2d(dims[k + 1])
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.5 / dims[k] ** 0.5))
            bn.weight.copy_(torch.rand(dims[k + 1], generator=g) + 0.5)
            bn.bias.copy_(torch.randn(dims[k + 1], generator=g) * 0.3)
            if k == 1:
                bn.weight[0] = -0.7            # a negative gamma: the arg-max of y is then NOT the arg-max of z
        layers += [conv, bn, nn.ReLU()]
    return nn.Sequential(*layers).cuda().train()


def _reference_fp64(xyz, new_xyz, feats, idx, mlp):
    """The chain in fp64 with plain torch ops; returns out (B, M, c3) and the batch statistics per layer."""
    B, M, ns = idx.shape
    ii = idx.long()
    gx = torch.stack([xyz[b][ii[b]] for b in range(B)]).double() - new_xyz.double().unsqueeze(2)        # (B, M, ns, 3)
    gf = torch.stack([feats[b][ii[b]] for b in range(B)]).double()
    x = torch.cat([gx, gf], dim=-1)
    stats = []
    layers = list(mlp)
    for k in range(3):
        conv, bn = layers[3 * k], layers[3 * k + 1]
        z = x @ conv.weight.flatten(1).double().t()
        mean, var = z.mean(dim=(0, 1, 2)), z.var(dim=(0, 1, 2), unbiased=False)
        stats.append((mean, z.var(dim=(0, 1, 2), unbiased=True)))
        x = torch.relu((z - mean) / torch.sqrt(var + bn.eps) * bn.weight.double() + bn.bias.double())
    return x.max(dim=2)[0], stats


CASES = [((4, 16, 16, 32), 16, 2, 2048, 1024), ((4, 32, 32, 64), 32, 2, 2048, 1024), ((4, 32, 16, 64), 32, 1, 1000, 96),
         ((4, 16, 32, 32), 16, 3, 700, 64)]


@pytest.mark.parametrize("dims,ns,B,N,M", CASES)
def test_fused_chain_against_fp64_reference(dims, ns, B, N, M):
    from pdanet_amd import pointnet2_utils as pu
    xyz, feats = _scene(B, N, seed=11 + ns)
    new_xyz = xyz[:, :M].contiguous()
    idx = pu.ball_query(0.8 if ns == 16 else 1.6, ns, xyz, new_xyz)         # short lists: padded with repeats (ties)
    mlp = _mlp(dims, seed=dims[1] + ns)
    assert pu.SaSmallChainTrain.supported(xyz, new_xyz, feats, idx, mlp)
    run0 = [(l.running_mean.clone(), l.running_var.clone()) for l in list(mlp)[1::3]]
    out = pu.sa_small_chain_train(xyz, new_xyz, feats, idx, mlp)
    g = torch.Generator().manual_seed(3)
    gout = torch.randn(out.shape, generator=g).cuda()
    out.backward(gout)
    grads = [p.grad.clone() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    # fp64 reference through autograd
    ref, stats = _reference_fp64(xyz, new_xyz, feats, idx, mlp)
    scale = ref.abs().max().item()
    assert (out.double() - ref).abs().max().item() <= 2e-5 * scale
    ref.backward(gout.double())
    for (n, p), got in zip(mlp.named_parameters(), grads):
        want = p.grad
        assert (got.double() - want.double()).abs().max().item() <= 2e-4 * max(want.abs().max().item(), 1e-3), n
    for k, bn in enumerate(list(mlp)[1::3]):
        mean, var_u = stats[k]
        rm = 0.9 * run0[k][0].double() + 0.1 * mean
        rv = 0.9 * run0[k][1].double() + 0.1 * var_u
        assert torch.allclose(bn.running_mean.double(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(bn.running_var.double(), rv, rtol=1e-5, atol=1e-6)


def test_sa_layer0_module_fused_equals_layerwise():
    """The ONCE layer-0 module (both scales + aggregation) with the fused passes on and off: same output, same gradients,
    same running statistics; at the headline scale of 2 x 16384 centres."""
    from pdanet_amd import pointnet2_utils as pu
    from pdanet_amd.backbone import build_backbone
    xyz, feats = _scene(2, 16384, seed=2)
    feats_cm = feats.permute(0, 2, 1).contiguous()
    res = {}
    keep = pu.SA_SMALL_TRAIN
    try:
        for fused in (True, False):
            pu.SA_SMALL_TRAIN = fused
            torch.manual_seed(1234)
            model, _ = build_backbone("once_pda_ssd.yaml")
            layer = model.SA_modules[0].cuda().train()
            out = layer(xyz, feats_cm)[1]
            gen = torch.Generator().manual_seed(5)
            out.backward(torch.randn(out.shape, generator=gen).cuda())
            res[fused] = (out.detach(), {n: p.grad.clone() for n, p in layer.named_parameters()},
                          {n: b.clone() for n, b in layer.named_buffers()})
    finally:
        pu.SA_SMALL_TRAIN = keep
    (of, gf, bf), (ol, gl, bl) = res[True], res[False]
    assert (of - ol).abs().max().item() <= 1e-4 * ol.abs().max().item()
    assert set(gf) == set(gl) and len(gf) >= 20
    for n in gl:
        assert (gf[n] - gl[n]).abs().max().item() <= 5e-4 * max(gl[n].abs().max().item(), 1e-3), n
    for n in bl:
        assert torch.allclose(bf[n].float(), bl[n].float(), rtol=1e-4, atol=1e-6), n


def test_unsupported_chain_is_reported():
    from pdanet_amd import pointnet2_batch_cuda as ext
    assert not ext.sa_small_train_supported(1, 64, 16, 16, 32, 1 << 20)      # 64 neighbours: groups span tiles
    assert not ext.sa_small_train_supported(1, 16, 16, 16, 64, 1 << 20)
    assert not ext.sa_small_train_supported(1, 16, 16, 16, 32, 1000)         # tokens not a multiple of 32
    assert ext.sa_small_train_supported(1, 16, 16, 16, 32, 1 << 19) and ext.sa_small_train_supported(1, 32, 32, 32, 64, 1 << 20)

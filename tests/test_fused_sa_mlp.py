"""-m gpu: the fused SA-scale kernel (csrc/sa_mlp.hip: gather -> 3 x [1x1 conv, folded BN,
ReLU] on f32 MFMA -> max-pool) against a plain PyTorch fp32 reference of the same op (the
unfused operator sequence the reference runs, pointnet2_modules.py:1657-1670), eval-mode BN.
Tolerance 2e-4 (relative to max |ref|): both sides are fp32; the MFMA chain and rocBLAS sum
in different orders."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from detweights import fill_deterministic  # noqa: E402

pytestmark = pytest.mark.gpu


def make_layer(c_in, mlps, radii, nsamples, npoint):
    from pdanet_amd.pointnet2_modules import PointnetSAModuleMSG_WithSampling
    layer = PointnetSAModuleMSG_WithSampling(
        npoint_list=[npoint], sample_range_list=[-1], sample_type_list=["D-FPS"], radii=radii,
        nsamples=nsamples, mlps=[[c_in] + m for m in mlps], use_xyz=True, dilated_group=False,
        aggregation_mlp=None, confidence_mlp=None, num_class=3)
    return fill_deterministic(layer).cuda().eval()


@pytest.mark.parametrize("c_in,mlps,radii,nsamples,n,m", [
    (1, [[16, 16, 32], [32, 32, 64]], [0.8, 1.6], [16, 32], 4096, 1024),          # layer-0 chains
    (256, [[256, 256, 512], [256, 256, 512]], [4.8, 8.4], [16, 32], 2048, 256),    # layer-5 chains
    (256, [[256, 512, 512]], [12.8], [64], 2048, 100),                             # ns 64: groups span 2 waves; ragged M
    (256, [[256, 512, 1024]], [6.4], [32], 1024, 77),                              # KITTI layer 5 scale 1
    (1, [[16, 16, 32]], [2.0], [8], 1000, 33),                                     # ns 8, odd sizes
    (256, [[256, 256, 512]], [20.0], [128], 512, 16),                              # ns 128: 4 waves per group
])
@pytest.mark.parametrize("wide_split", [False, True])
def test_fused_matches_unfused(c_in, mlps, radii, nsamples, n, m, wide_split):
    """wide_split = False: every scale on the fused f32-MFMA kernel (csrc/sa_mlp.hip); True (the default): the wide scales
    take the per-point first layer + split-bf16 GEMMs (pointnet2_utils.sa_wide_scale_infer) where the sizes allow it."""
    from pdanet_amd import synth, fused_ops, pointnet2_utils as pu
    keep, pu.SA_WIDE_INFER_SPLIT = pu.SA_WIDE_INFER_SPLIT, wide_split
    try:
        _fused_matches_unfused(c_in, mlps, radii, nsamples, n, m)
    finally:
        pu.SA_WIDE_INFER_SPLIT = keep


def _fused_matches_unfused(c_in, mlps, radii, nsamples, n, m):
    from pdanet_amd import synth, fused_ops
    xyz = torch.from_numpy(synth.batch_xyz(2, n, config_id=n + m)).cuda()
    feats = torch.randn(2, c_in, n, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    layer = make_layer(c_in, mlps, radii, nsamples, m)
    with torch.no_grad():
        ref_xyz, ref, _, ref_idx = layer(xyz, feats, None)
        assert fused_ops.enable_fused(layer) == 1
        new_xyz, out, _, idx = layer(xyz, feats, None)
    assert not layer.fused.unsupported, "fused kernel refused a chain: %s" % layer.fused.unsupported
    assert torch.equal(ref_idx, idx) and torch.equal(ref_xyz, new_xyz)
    assert out.shape == ref.shape
    scale = float(ref.abs().max())
    err = float((out - ref).abs().max())
    assert err <= 2e-4 * max(1.0, scale), (err, scale)


def test_fused_skipped_in_training_and_unsupported_chain():
    from pdanet_amd import synth, fused_ops
    xyz = torch.from_numpy(synth.batch_xyz(2, 1024, config_id=5)).cuda()
    feats = torch.randn(2, 5, 1024, device="cuda")
    layer = make_layer(5, [[64, 64, 64]], [2.0], [16], 256)   # 8->64->64->64: no kernel built for it
    fused_ops.enable_fused(layer)
    with torch.no_grad():
        a = layer(xyz, feats, None)[1]
    assert layer.fused.unsupported == {0}                  # fell back to the unfused HIP ops
    fused_ops.enable_fused(layer, False)
    with torch.no_grad():
        b = layer(xyz, feats, None)[1]
    assert torch.equal(a, b)
    layer2 = make_layer(1, [[16, 16, 32]], [2.0], [16], 256)
    fused_ops.enable_fused(layer2)
    layer2.train()
    layer2(xyz, feats[:, :1].contiguous(), None)           # batch-stat BN: must not touch the fused path
    assert not layer2.fused.cache


def test_fused_weight_update_invalidates_cache():
    from pdanet_amd import synth, fused_ops
    xyz = torch.from_numpy(synth.batch_xyz(1, 1024, config_id=6)).cuda()
    feats = torch.randn(1, 1, 1024, device="cuda")
    layer = make_layer(1, [[16, 16, 32]], [2.0], [16], 256)
    fused_ops.enable_fused(layer)
    with torch.no_grad():
        a = layer(xyz, feats, None)[1].clone()
        layer.mlps[0][0].weight.mul_(0.5)
        b = layer(xyz, feats, None)[1]
        fused_ops.enable_fused(layer, False)
        c = layer(xyz, feats, None)[1]
    assert not torch.allclose(a, b)
    assert float((b - c).abs().max()) <= 2e-4 * max(1.0, float(c.abs().max()))


def test_fused_cache_sees_raw_pointer_updates():
    """eval(fused) -> k training steps of the flat-buffer optimizer -> eval(fused) must equal eval(unfused).
    FlatAdamOneCycle.step and the BN+ReLU kernels' running-statistics updates write through raw pointers: neither
    _version nor data_ptr of the conv / BN tensors moves, only _lib.PARAM_EPOCH does."""
    from pdanet_amd import synth, fused_ops, optimization
    xyz = torch.from_numpy(synth.batch_xyz(2, 1024, config_id=8)).cuda()
    feats = torch.randn(2, 1, 1024, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    layer = make_layer(1, [[16, 16, 32]], [2.0], [16], 256)
    fused_ops.enable_fused(layer)
    with torch.no_grad():
        a = layer(xyz, feats, None)[1].clone()
    assert layer.fused.cache
    opt = optimization.FlatAdamOneCycle(layer, wd=0.01, grad_norm_clip=10, lr=1e-2, mom=0.9)
    layer.train()
    for _ in range(3):
        opt.zero_grad()
        layer(xyz, feats, None)[1].pow(2).mean().backward()
        opt.step()
    # a training forward WITHOUT an optimizer step (BN recalibration) must invalidate the cache as well
    with torch.no_grad():
        layer(xyz, feats * 3.0 + 1.0, None)
    layer.eval()
    with torch.no_grad():
        b = layer(xyz, feats, None)[1].clone()
        assert not layer.fused.unsupported
        fused_ops.enable_fused(layer, False)
        c = layer(xyz, feats, None)[1]
    assert not torch.allclose(a, b)
    assert float((b - c).abs().max()) <= 2e-4 * max(1.0, float(c.abs().max()))

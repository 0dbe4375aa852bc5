#!/usr/bin/env python3
"""One hot kernel (or operator call) of the bench, launched a few times, as the program under `rocprofv3 --pmc`
(tools/pmc_passes.sh).  Targets and the bench keys they feed (benchmarks/workloads.py pmc_record):
  fps          FPS 16384 -> 4096, 2 scenes                  -> traffic["pda::fps_chain_kernel FPS 16384->4096 b2"]
  ball_query   layer-0 ball query 16384 x 16384, 2 radii    -> traffic["pda::ball_query 16384x16384 r2 b2"]
  wgrad        dW(512x512) over 131072 tokens               -> traffic[...], mfma_util["pda::wgrad_split_kernel ..."]
  sa_mlp       fused SA scale 259->256->512->512, ns 64     -> mfma_util["pda::sa_mlp_kernel ..."]
  lin_cols     training-form contraction 512->512, 131072 tokens -> mfma_util["pda::lin_cols_kernel ..."]
  lin_split    the same contraction on the split-bf16 kernel     -> mfma_util["pda::lin_split_kernel ..."]
  gemm_split   the same on the LDS-tiled split-bf16 kernel       -> mfma_util["pda::gemm_split_wide_kernel ..."]
  sa_small     ONCE layer 0, scale 2, training passes             -> mfma_util["pda::ss_fwd_kernel / ss_bwd_kernel ..."]
Inputs are the bench's (synth scene config_id 2, distribution L)."""
import sys

import torch

sys.path.insert(0, '.')
from pdanet_amd import pointnet2_batch_cuda as ext, pointnet2_utils as pu, synth  # noqa: E402

target, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
xyz = torch.from_numpy(synth.batch_xyz(2, 16384, config_id=2, dist="L")).to(dev)
if target == "fps":
    idx = torch.zeros((2, 4096), dtype=torch.int32, device=dev)
    for _ in range(reps):
        temp = torch.full((2, 16384), 1e10, device=dev)
        ext.farthest_point_sampling_wrapper(2, 16384, 4096, xyz, temp, idx)
elif target == "ball_query":
    for _ in range(reps):
        pu.ball_query_multi([0.2, 0.8], [16, 32], xyz, xyz)
elif target == "wgrad":
    t, ni, no = 131072, 512, 512
    x, g = torch.randn(t, ni, device=dev), torch.randn(t, no, device=dev)
    gw, gb = torch.empty(no, ni, device=dev), torch.empty(no, device=dev)
    for _ in range(reps):
        ext.linear_wgrad(x, g, gw, gb, t, ni, no)
elif target == "sa_mlp":
    from pdanet_amd import fused_ops
    from pdanet_amd.pointnet2_modules import PointnetSAModuleMSG_WithSampling
    layer = PointnetSAModuleMSG_WithSampling(npoint_list=[1024], sample_range_list=[-1], sample_type_list=["D-FPS"], radii=[12.8],
                                             nsamples=[64], mlps=[[256, 256, 512, 512]], use_xyz=True, dilated_group=False,
                                             aggregation_mlp=None, confidence_mlp=None, num_class=5).to(dev).eval()
    fused_ops.enable_fused(layer)
    pts = xyz[:, :2048].contiguous()
    feats = torch.randn(2, 256, 2048, device=dev)
    with torch.no_grad():
        for _ in range(reps):
            layer(pts, feats, None, ctr_xyz=pts[:, :1024].contiguous())
elif target == "lin_cols":
    # training form of the group MLP: ONCE layer 5, scale 3, layer 3 forward (131072 tokens, 512 -> 512)
    t, k, n = 131072, 512, 512
    x, w, y = torch.randn(t, k, device=dev), torch.randn(n, k, device=dev), torch.empty(t, n, device=dev)
    wf = ext.linear_cols_pack(w, n, k)
    for _ in range(reps):
        ext.linear_cols(x, wf, y, t, k, n)
elif target == "gemm_split":
    # the LDS-tiled split-bf16 GEMM (256 x 256 tiles): ONCE layer 5, scale 3, layer 3 forward (131072 tokens, 512 -> 512)
    t, k, n = 131072, 512, 512
    x, w, y = torch.randn(t, k, device=dev), torch.randn(n, k, device=dev), torch.empty(t, n, device=dev)
    wf = ext.linear_split_pack(w, n, k)
    for _ in range(reps):
        ext.gemm_split(x, wf, None, y, t, k, n)
elif target == "lin_split":
    t, k, n = 131072, 512, 512
    x, w, y = torch.randn(t, k, device=dev), torch.randn(n, k, device=dev), torch.empty(t, n, device=dev)
    wf = ext.linear_split_pack(w, n, k)
    for _ in range(reps):
        ext.linear_split(x, wf, None, y, t, k, n)
elif target == "sa_small":
    # the narrow SA scale in training form (csrc/sa_train_small.hip): ONCE layer 0, scale 2 (4 -> 32 -> 32 -> 64, 32 neighbours,
    # 2 x 16384 centres = 1 M tokens), forward + backward passes
    import torch.nn as nn
    mlp = nn.Sequential(*[m for k in range(3) for m in (nn.Conv2d((4, 32, 32)[k], (32, 32, 64)[k], 1, bias=False), nn.BatchNorm2d((32, 32, 64)[k]), nn.ReLU())]).to(dev).train()
    feats = torch.rand(2, 16384, 1, device=dev)
    idx = pu.ball_query(0.8, 32, xyz, xyz)
    for _ in range(reps):
        out = pu.sa_small_chain_train(xyz, xyz, feats, idx, mlp)
        out.backward(torch.ones_like(out))
else:
    raise SystemExit("unknown target " + target)
torch.cuda.synchronize()

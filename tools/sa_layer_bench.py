#!/usr/bin/env python3
"""Device time of ONE vanilla SA layer of the ONCE PDA-SSD backbone in training form (forward, backward), in isolation:
layer 0 (4->16->16->32 | 4->32->32->64 over 0.5 / 1 M tokens) or layer 5 (259->256->256->512 x2 | 259->256->512->512).
Usage: python tools/sa_layer_bench.py [layer] [iters]      (MI355X box; add rocprofv3 --kernel-trace --stats in front)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdanet_amd import synth  # noqa: E402
from pdanet_amd.backbone import build_backbone  # noqa: E402

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(1234)
model, _ = build_backbone("once_pda_ssd.yaml")
model = model.cuda().train()
m = model.SA_modules[layer]
B = 2
g = torch.Generator().manual_seed(7)
if layer == 0:
    pts = torch.from_numpy(synth.batch_points(B, 16384, config_id=2, dist="L")).cuda()
    xyz = pts[:, 1:4].reshape(B, -1, 3).contiguous()
    feats = pts[:, 4:].reshape(B, -1, 1).permute(0, 2, 1).contiguous()
    args = dict(xyz=xyz, features=feats)
elif layer in (1, 2):
    # PDA layers: layer 1 (16384 pts, 64 ch, D-FPS -> 4096), layer 2 (4096 pts, 128 ch, ctr-aware top-k -> 2048)
    n_in, c_in = (16384, 64) if layer == 1 else (4096, 128)
    xyz = torch.from_numpy(synth.batch_xyz(B, 16384, config_id=2, dist="L")).cuda()
    if layer == 2:      # the D-FPS picks of layer 1 as input points
        from pdanet_amd import pointnet2_utils as pu
        idx = pu.furthest_point_sample(xyz, 4096)
        xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    feats = torch.randn(B, c_in, n_in, generator=g).cuda().requires_grad_(True)
    cls = torch.randn(B, n_in, 5, generator=g).cuda()
    args = dict(xyz=xyz, features=feats, cls_features=cls)
else:
    xyz = torch.from_numpy(synth.batch_xyz(B, 2048, config_id=2, dist="L")).cuda()
    feats = torch.randn(B, 256, 2048, generator=g).cuda().requires_grad_(True)
    ctr = (xyz[:, :1024] + 0.3 * torch.randn(B, 1024, 3, generator=g).cuda()).contiguous()
    args = dict(xyz=xyz, features=feats, ctr_xyz=ctr)


def fwd():
    out = m(**args)
    return out[1]


y = fwd()
gy = torch.randn_like(y)
for _ in range(3):
    y = fwd()
    y.backward(gy)
torch.cuda.synchronize()
ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(iters):
    for p in m.parameters():
        p.grad = None
    ef[0].record()
    y = fwd()
    ef[1].record()
    y.backward(gy)
    ef[2].record()
    torch.cuda.synchronize()
    tf += ef[0].elapsed_time(ef[1])
    tb += ef[1].elapsed_time(ef[2])
print("SA layer %d, B=%d: forward %.3f ms, backward %.3f ms, total %.3f ms (mean of %d, events around the enqueue: host-bound "
      "parts count as device idle)" % (layer, B, tf / iters, tb / iters, (tf + tb) / iters, iters))

"""Fused inference path for the vanilla SA layers: one HIP kernel per scale
(group -> 3 x [1x1 conv, folded BN, ReLU] -> max-pool, csrc/sa_mlp.hip) instead of the
reference's group_points x2 + cat + 9 torch kernels (pointnet2_modules.py:1657-1670).

`enable_fused(model)` attaches the fused callable to every PointnetSAModuleMSG_WithSampling.
It is used only in eval mode under torch.no_grad(); in training the layers need batch
statistics and run the unfused operator sequence.  Chains the library has no kernel for
(PDA_ERR_UNSUPPORTED) fall back to the unfused sequence as well -- both are HIP paths.
"""
import ctypes

import torch

from . import _lib
from .pointnet2_modules import PointnetSAModuleMSG_WithSampling

PDA_ERR_UNSUPPORTED = 3
# bench.py sets this to a list to collect (event0, event1, flops, dims, ns) per fused launch
PROFILE = None


def _pad32(v):
    n = (v.numel() + 31) // 32 * 32
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v
    return out


class FusedSAMlp:
    """Callable stored on the module as `module.fused`."""

    def __init__(self):
        self.cache = {}
        self.unsupported = set()

    def _prepare(self, i, module):
        seq = module.mlps[i]
        convs = [m for m in seq if isinstance(m, torch.nn.Conv2d)]
        bns = [m for m in seq if isinstance(m, torch.nn.BatchNorm2d)]
        if len(convs) != 3 or len(bns) != 3:
            return None
        # PARAM_EPOCH: the flat-buffer optimizer and the BN kernels' running-statistics updates write through raw
        # pointers, which moves neither _version nor data_ptr (eval -> train k steps -> eval must re-pack)
        key = (_lib.PARAM_EPOCH[0],) + tuple((c.weight._version, c.weight.data_ptr()) for c in convs) + \
            tuple((b.weight._version, b.bias._version, b.running_mean._version, b.running_var._version) for b in bns)
        hit = self.cache.get(i)
        if hit is not None and hit["key"] == key:
            return hit
        lib = _lib.load()
        dev = convs[0].weight.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        wf, scale, shift, dims = [], [], [], [convs[0].weight.shape[1]]
        with torch.no_grad():
            for l, (c, b) in enumerate(zip(convs, bns)):
                w = c.weight.detach().reshape(c.weight.shape[0], -1).contiguous().float()
                rows, cols = w.shape
                n = lib.pda_sa_mlp_packed_size(rows, cols, 1 if l == 0 else 0)
                packed = torch.empty(n, dtype=torch.float32, device=dev)
                _lib.check(lib.pda_sa_mlp_pack_weights(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()),
                                                       rows, cols, 1 if l == 0 else 0, stream), "pda_sa_mlp_pack_weights")
                s = b.weight / torch.sqrt(b.running_var + b.eps)
                t = b.bias - b.running_mean * s
                wf.append(packed); scale.append(_pad32(s.float())); shift.append(_pad32(t.float()))
                dims.append(rows)
        hit = dict(key=key, wf=wf, scale=scale, shift=shift, dims=dims)
        self.cache[i] = hit
        return hit

    def __call__(self, i, module, xyz, new_xyz, features, idx):
        if i in self.unsupported or idx is None:
            return None
        prep = self._prepare(i, module)
        if prep is None:
            self.unsupported.add(i)
            return None
        lib = _lib.load()
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        C = 0 if features is None else features.shape[1]
        out = torch.empty((B, prep["dims"][3], M), dtype=torch.float32, device=xyz.device)
        dims = (ctypes.c_int32 * 4)(*prep["dims"])
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and idx.is_contiguous()
        assert features is None or (features.is_contiguous() and features.dtype == torch.float32)
        prof = None if torch.cuda.is_current_stream_capturing() else PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.cuda.device(xyz.device):
            st = lib.pda_sa_mlp_maxpool(
                ctypes.c_void_p(xyz.data_ptr()), ctypes.c_void_p(new_xyz.data_ptr()),
                ctypes.c_void_p(features.data_ptr()) if features is not None else None,
                ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                B, N, M, C, ns, dims, arr(prep["wf"]), arr(prep["scale"]), arr(prep["shift"]),
                ctypes.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream))
        if st == PDA_ERR_UNSUPPORTED:
            self.unsupported.add(i)
            return None
        _lib.check(st, "pda_sa_mlp_maxpool")
        if prof is not None:
            e1.record()
            d = prep["dims"]
            prof.append((e0, e1, 2.0 * B * M * ns * (d[0] * d[1] + d[1] * d[2] + d[2] * d[3]), tuple(d), ns))
        return out


def enable_fused(model, enabled=True):
    """Attach (or detach) the fused SA-MLP path on every vanilla SA layer of `model`."""
    n = 0
    for m in model.modules():
        if isinstance(m, PointnetSAModuleMSG_WithSampling):
            m.fused = FusedSAMlp() if enabled else None
            n += 1
    return n

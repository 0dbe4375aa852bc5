"""Deterministic, name-keyed weights: the same state-dict key always gets the same values,
whatever order a model creates its parameters in.  Used by make_golden.py (on the REFERENCE
modules) and by the tests (on this repo's modules), so fixtures need not store weights."""
import zlib

import torch


def _gen(key):
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode()) & 0x7fffffff)
    return g


def fill_deterministic(module, salt=""):
    sd = module.state_dict()
    new = {}
    for key, t in sd.items():
        g = _gen(salt + key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            new[key] = t.clone()
        elif leaf == "running_var":
            new[key] = torch.rand(t.shape, generator=g) + 0.5
        elif leaf == "running_mean":
            new[key] = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() <= 1:
            # BN / LayerNorm weight ~ U(0.5,1.5); every 1-D bias ~ N(0, 0.1)
            if leaf == "weight":
                new[key] = torch.rand(t.shape, generator=g) + 0.5
            else:
                new[key] = torch.randn(t.shape, generator=g) * 0.1
        else:
            fan_in = t[0].numel()
            new[key] = torch.randn(t.shape, generator=g) * (1.5 / fan_in ** 0.5)
    module.load_state_dict(new, strict=True)
    return module

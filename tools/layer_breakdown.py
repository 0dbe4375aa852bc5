#!/usr/bin/env python3
"""Device time of the training iteration by backbone layer (forward and backward), head + losses and optimizer: CUDA events at
the layer boundaries of the main stream (forward: module hooks; backward: gradient hooks on the layer inputs).
Usage (MI355X): python tools/layer_breakdown.py [workload] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "detector_train"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
os.environ.setdefault("PDA_GRAPH_TAIL", "0")
os.environ.setdefault("PDA_GRAPH_HEAD", "0")
dev = torch.device("cuda", 0)
wl = workloads.create(name, 4 if name.startswith("kitti") else 2, 16384, dev, 0, 1)
wl.begin()
bb = wl.model.backbone_3d
marks = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((tag, e))


hooks = []
for i, m in enumerate(bb.SA_modules):
    hooks.append(m.register_forward_pre_hook(lambda mod, inp, i=i: mark("fwd L%d begin" % i)))

    def post(mod, inp, out, i=i):
        mark("fwd L%d end" % i)
        feats = out[1] if isinstance(out, tuple) and len(out) > 1 and torch.is_tensor(out[1]) and out[1].requires_grad else None
        if feats is not None:
            feats.register_hook(lambda g, i=i: mark("bwd reaches L%d output" % i))
    hooks.append(m.register_forward_hook(post))
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
acc = {}
for _ in range(iters):
    marks.clear()
    mark("step begin")
    wl.step()
    mark("step end")
    torch.cuda.synchronize()
    for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
        acc.setdefault((t0, t1), []).append(e0.elapsed_time(e1))
print("%-62s %8s" % ("interval (main stream, %s)" % wl.name, "ms"))
tot = 0.0
for (t0, t1), v in acc.items():
    ms = sum(v) / len(v)
    tot += ms
    print("%-62s %8.3f" % ("%s -> %s" % (t0, t1), ms))
print("%-62s %8.3f" % ("sum", tot))

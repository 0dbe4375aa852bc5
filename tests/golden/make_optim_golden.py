"""Generates tests/golden/optim_onecycle.npz by running the REFERENCE's optimizer code
(/root/reference/tools/train_utils/optimization: build_optimizer / build_scheduler with
OPTIMIZER=adam_onecycle, plus clip_grad_norm_ as in train_utils.py:56) on CPU for 12 iterations of a
small model with seeded parameters and gradients.  Only inputs and outputs are stored."""
import os
import sys
import types

import numpy as np
import torch
from torch import nn
from torch.nn.utils import clip_grad_norm_

sys.path.insert(0, "/root/reference/tools/train_utils")
from optimization import build_optimizer, build_scheduler  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv1d(5, 16, 1, bias=False), nn.BatchNorm1d(16), nn.ReLU())
        self.attn = nn.MultiheadAttention(16, 4)
        self.block = nn.Sequential(nn.Conv2d(16, 9, 1), nn.BatchNorm2d(9), nn.ReLU(), nn.Conv2d(9, 3, 1))
        self.norm = nn.LayerNorm(16)
        self.head = nn.Linear(16, 7)


def main():
    torch.manual_seed(20240)
    model = Net()
    names = [n for n, _ in model.named_parameters()]
    init = {n: p.detach().clone().numpy() for n, p in model.named_parameters()}
    cfg = types.SimpleNamespace(OPTIMIZER="adam_onecycle", LR=0.01, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85],
                                PCT_START=0.4, DIV_FACTOR=10, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1,
                                LR_CLIP=1e-7, GRAD_NORM_CLIP=10)
    opt = build_optimizer(model, cfg)
    sched, _ = build_scheduler(opt, total_iters_each_epoch=10, total_epochs=3, last_epoch=-1, optim_cfg=cfg)
    group_names = []
    pid = {id(p): n for n, p in model.named_parameters()}
    for g in opt.opt.param_groups:
        group_names.append([pid[id(p)] for p in g["params"]])
    steps = 12
    gen = torch.Generator().manual_seed(7)
    out = {"names": np.array(names), "group0": np.array(group_names[0]), "group1": np.array(group_names[1])}
    for n in names:
        out["init/" + n] = init[n]
    lrs, moms, norms = [], [], []
    for it in range(steps):
        sched.step(it)
        lrs.append(float(opt.lr)); moms.append(float(opt.mom))
        opt.zero_grad()
        scale = 3.0 if it % 3 == 0 else 0.05           # clipping active on every third step
        for n, p in model.named_parameters():
            g = torch.randn(p.shape, generator=gen) * scale
            out["grad/%d/%s" % (it, n)] = g.numpy().copy()
            p.grad = g
        norms.append(float(clip_grad_norm_(model.parameters(), cfg.GRAD_NORM_CLIP)))
        opt.step()
        for n, p in model.named_parameters():
            out["param/%d/%s" % (it, n)] = p.detach().numpy().copy()
    sd = opt.state_dict()
    out["lr"], out["mom"], out["norm"] = np.array(lrs), np.array(moms), np.array(norms)
    out["sd_groups"] = np.array([len(g["params"]) for g in sd["param_groups"]])
    for k, st in sd["state"].items():
        out["state/%d/exp_avg" % k] = st["exp_avg"].numpy()
        out["state/%d/exp_avg_sq" % k] = st["exp_avg_sq"].numpy()
        out["state/%d/step" % k] = np.array(float(st["step"]))
    np.savez_compressed(os.path.join(HERE, "optim_onecycle.npz"), **out)
    print("groups:", [len(g) for g in group_names], "untrained:", sorted(set(names) - set(sum(group_names, []))))
    print("norms", np.round(norms, 3), "lr", np.round(lrs, 5))


if __name__ == "__main__":
    main()

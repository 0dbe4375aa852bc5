"""adam_onecycle train-step arithmetic, MI355X form.

Mirrors tools/train_utils/optimization/__init__.py:11-63 (build_optimizer / build_scheduler),
fastai_optim.py:104-236 (OptimWrapper with true_wd=True, bn_wd=True around torch.optim.Adam with
betas=(mom, 0.99)), learning_schedules_fastai.py:12-77 (OneCycle) and the clip_grad_norm_ of
train_utils.py:56.  The reference updates ~330 tensors one by one from Python; here every trained
parameter, its gradient and both Adam moments are views into four flat fp32 buffers and one step is
two launches of csrc/optim.hip (gradient norm, fused decay+Adam).

Reference behaviours kept on purpose:
  * only parameters of LEAF modules are trained: build_optimizer flattens the model with
    `flatten_model` (__init__.py:27), which drops parameters owned directly by a module that has
    children -- nn.MultiheadAttention.in_proj_weight / in_proj_bias (its only child is out_proj) are
    never stepped nor decayed, although they receive gradients and count in the clipping norm;
  * two parameter groups [non-BatchNorm leaves, BatchNorm leaves] in flatten order (split_bn_bias,
    fastai_optim.py:15-27): that is the parameter numbering of the checkpoint's optimizer_state;
  * decay `p *= 1 - wd*lr` is applied to BN parameters too (bn_wd=True) and before the Adam update;
  * OneCycle.step(it) is called BEFORE the forward of iteration `it` (train_utils.py:34).
"""
import math

import numpy as np
import torch
from torch import nn

from . import _lib

BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)  # fastai_optim.py:12


def flatten_model(m):
    """optimization/__init__.py:27: the leaf modules of `m` in children order."""
    children = list(m.children())
    return sum((flatten_model(c) for c in children), []) if children else [m]


def trained_parameter_groups(model):
    """[non-BN leaf parameters, BN leaf parameters] as (name, parameter) lists, in the order
    OptimWrapper.create hands them to Adam (fastai_optim.py:15-27,118-123)."""
    names = {id(p): n for n, p in model.named_parameters()}
    groups, seen = ([], []), set()
    for leaf in flatten_model(model):
        g = groups[1] if isinstance(leaf, BN_TYPES) else groups[0]
        for p in leaf.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                g.append((names.get(id(p), "?"), p))
    return groups


def annealing_cos(start, end, pct):
    """learning_schedules_fastai.py:54-58."""
    cos_out = np.cos(np.pi * pct) + 1
    return end + (start - end) / 2 * cos_out


class OneCycle:
    """learning_schedules_fastai.py:12-77: two cosine phases for lr (lr_max/div -> lr_max ->
    lr_max/div/1e4) and momentum (moms[0] -> moms[1] -> moms[0]), split at pct_start."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, total_step
        low_lr = lr_max / div_factor
        lr_phases = ((0, (low_lr, lr_max)), (pct_start, (lr_max, low_lr / 1e4)))
        mom_phases = ((0, tuple(moms)), (pct_start, tuple(moms[::-1])))
        self.lr_phases = self._phases(lr_phases)
        self.mom_phases = self._phases(mom_phases)
        optimizer.lr, optimizer.mom = low_lr, moms[0]

    def _phases(self, phases):
        out = []
        for i, (start, ends) in enumerate(phases):
            stop = int(phases[i + 1][0] * self.total_step) if i < len(phases) - 1 else self.total_step
            out.append((int(start * self.total_step), stop, ends))
        return out

    def values(self, step):
        lr = mom = None
        for start, end, (a, b) in self.lr_phases:
            if step >= start:
                lr = annealing_cos(a, b, (step - start) / (end - start))
        for start, end, (a, b) in self.mom_phases:
            if step >= start:
                mom = annealing_cos(a, b, (step - start) / (end - start))
        return float(lr), float(mom)

    def step(self, step):
        self.optimizer.lr, self.optimizer.mom = self.values(step)


class FlatAdamOneCycle:
    """OptimWrapper(Adam, true_wd=True, bn_wd=True) over flat buffers.

    After construction every trained parameter's `.data` and `.grad` are views into `flat_p` /
    `flat_g` (16-byte aligned slots); parameters that receive gradients but are not trained (see the
    module docstring) follow in the tail of `flat_g` so the clipping norm covers them.  Construct it
    BEFORE wrapping the model in DistributedDataParallel."""

    def __init__(self, model, wd, beta2=0.99, eps=1e-8, grad_norm_clip=None, lr=3e-3, mom=0.9):
        self.groups = trained_parameter_groups(model)
        trained = [p for g in self.groups for _, p in g]
        ids = {id(p) for p in trained}
        self.untrained = [(n, p) for n, p in model.named_parameters() if p.requires_grad and id(p) not in ids]
        everything = trained + [p for _, p in self.untrained]
        if not everything:
            raise ValueError("model has no trainable parameters")
        dev = everything[0].device
        self.offsets, off = [], 0
        for p in everything:
            if p.dtype != torch.float32 or p.device != dev:
                raise _lib.PdaError("FlatAdamOneCycle: fp32 parameters on one device only")
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
            if p is trained[-1]:
                self.n_trained = off
        self.flat_p = torch.zeros(self.n_trained, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(self.n_trained, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_trained, device=dev)
        self._norm = torch.zeros(1, device=dev)
        self._scratch = torch.zeros(1024, device=dev)
        for p, o in zip(everything, self.offsets):
            n = p.numel()
            if o < self.n_trained:
                self.flat_p[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        self._all = everything
        self._grad_views = [p.grad for p in everything]
        self._gather = False
        self.params = trained
        self.wd, self.beta2, self.eps, self.grad_norm_clip = wd, beta2, eps, grad_norm_clip
        self.lr, self.mom = lr, mom
        self.step_count = 0
        self._dp_group, self._dp_world, self._dp_avg = None, 1, False

    # ---- data parallel: the gradient exchange of tools/train.py:153-154 (DDP), outside autograd ----------------
    def data_parallel(self, group=None, sync_parameters=True, model=None):
        """Make step() all-reduce the flat gradient buffer over `group` (default: the world) before the norm.

        The reference wraps the model in DistributedDataParallel, whose reducer hooks fire inside backward.  Here the
        whole gradient already is ONE fp32 buffer, so the exchange is one collective on it between backward and the
        clipping norm: no hooks (nothing of the exchange can end up inside a captured hipGraph region, so N ranks run the
        same graphed step as one rank), no buckets and no bucket -> view copy, .grad stay views of the flat buffer.
        RCCL averages in the collective (ReduceOp.AVG: no extra launch); backends without AVG (gloo) sum, then one scale
        launch.  25.5 MB per rank: latency-bound on xGMI.  sync_parameters: broadcast rank 0's parameters (and, with
        `model`, its buffers and untrained parameters) once, as DDP does at construction.  Not reproduced: DDP's
        per-forward broadcast of BatchNorm running statistics -- training-mode forwards never read them and rank 0
        writes the checkpoint, so rank 0's statistics are what a DDP run would have kept too."""
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return self
        self._dp_group, self._dp_world = group, dist.get_world_size(group)
        self._dp_avg = False
        if dist.get_backend(group) == "nccl":
            # RCCL averages inside the collective (ncclAvg); probed once on a scratch value so that a build without it
            # falls back to SUM + scale on every rank alike instead of failing in the first step
            try:
                probe = torch.ones(4, device=self.flat_g.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                self._dp_avg = bool((probe == 1).all().item())
            except (RuntimeError, ValueError):
                self._dp_avg = False
        if sync_parameters:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(self.flat_p, src, group=group)
            if model is not None:
                mine = {id(p) for p in self.params}
                for t in [p.data for p in model.parameters() if id(p) not in mine] + list(model.buffers()):
                    dist.broadcast(t, src, group=group)
            _lib.PARAM_EPOCH[0] += 1
            _lib.WEIGHT_EPOCH[0] += 1
        return self

    def exchange_gradients(self):
        """Mean of flat_g over the ranks, in place, on the current stream (called by step(); public for loops that
        clip or log between backward and step)."""
        import torch.distributed as dist
        if self._gather:
            self._gather_grads()
        if self._dp_world == 1:
            return
        if self._dp_avg:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.AVG, group=self._dp_group)
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self._dp_group)
            self.flat_g.mul_(1.0 / self._dp_world)

    def zero_grad(self, set_to_none=False):
        """Default: zero the flat gradient buffer; `.grad` stay views of it, so autograd ACCUMULATES into them (one small
        add launch per parameter per backward -- ~300 launches for PDA-SSD).  set_to_none=True: `.grad` = None, autograd
        hands each parameter its gradient tensor as produced, and step() gathers them into the flat buffer with one
        multi-tensor copy (a handful of launches).  Same result; use the default when something else (DDP built with
        `grads_are_views`) writes into the views."""
        if set_to_none:
            for p in self._all:
                p.grad = None
            self._gather = True
        else:
            if self._gather:
                for p, v in zip(self._all, self._grad_views):
                    p.grad = v
                self._gather = False
            self.flat_g.zero_()

    def _gather_grads(self):
        self.flat_g.zero_()                                  # slots of parameters that received no gradient
        dsts, srcs = [], []
        for p, v in zip(self._all, self._grad_views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dsts.append(v); srcs.append(g.detach())
            p.grad = v                                       # .grad is a view of flat_g again (total_norm, checkpoints, hooks)
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        self._gather = False

    def step(self):
        """clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP) + OptimWrapper.step()."""
        if self.flat_p.device.type != "cuda":
            # the flat buffers and the gradient exchange are plumbing and work anywhere (the gloo tests build them on the
            # CPU); the update itself is csrc/optim.hip and nothing else
            raise _lib.PdaError("FlatAdamOneCycle.step needs the parameters on the GPU (no CPU path)")
        lib = _lib.load()
        if self._gather:
            self._gather_grads()
        if self._dp_world > 1:
            self.exchange_gradients()
        stream = torch.cuda.current_stream(self.flat_p.device).cuda_stream
        with torch.cuda.device(self.flat_p.device):
            norm_ptr = None
            if self.grad_norm_clip is not None:
                _lib.check(lib.pda_grad_norm(self.flat_g.data_ptr(), self.flat_g.numel(), self._norm.data_ptr(),
                                             self._scratch.data_ptr(), stream), "pda_grad_norm")
                norm_ptr = self._norm.data_ptr()
            self.step_count += 1
            _lib.PARAM_EPOCH[0] += 1        # parameters change without their version counters moving
            _lib.WEIGHT_EPOCH[0] += 1
            _lib.check(lib.pda_adam_onecycle_step(
                self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.n_trained, self.lr, self.mom, self.beta2, self.eps, self.wd, self.step_count, norm_ptr,
                float(self.grad_norm_clip or 0.0), stream), "pda_adam_onecycle_step")

    @property
    def total_norm(self):
        """Gradient norm of the last step() (device tensor; clip_grad_norm_'s return value)."""
        return self._norm

    # ---- checkpoint format of torch.optim.Adam.state_dict(), which the reference saves as
    # 'optimizer_state' (train_utils.py:158-173 through OptimWrapper.__getattr__) ----------------
    def state_dict(self):
        state, k = {}, 0
        group_params = []
        for g in self.groups:
            idxs = []
            for _, p in g:
                o, n = self.offsets[k], p.numel()
                state[k] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
                idxs.append(k)
                k += 1
            group_params.append(idxs)
        if self.step_count == 0:
            state = {}
        pg = [{"lr": self.lr, "betas": (self.mom, self.beta2), "eps": self.eps, "weight_decay": 0,
               "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
               "differentiable": False, "fused": None, "params": idxs} for idxs in group_params]
        return {"state": state, "param_groups": pg}

    def load_state_dict(self, sd):
        pg = sd["param_groups"]
        n = sum(len(g["params"]) for g in pg)
        if n != len(self.params):
            raise ValueError("optimizer state has %d parameters, model trains %d" % (n, len(self.params)))
        self.lr, (self.mom, self.beta2) = pg[0]["lr"], pg[0]["betas"]
        self.step_count = 0
        for k, p in enumerate(self.params):
            st = sd["state"].get(k)
            o, m = self.offsets[k], p.numel()
            if st is None:
                self.exp_avg[o:o + m].zero_(); self.exp_avg_sq[o:o + m].zero_()
                continue
            self.exp_avg[o:o + m].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + m].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])


def build_optimizer(model, optim_cfg):
    """optimization/__init__.py:11-39 for OPTIMIZER == 'adam_onecycle' (the only one the PDA-SSD yamls use)."""
    if optim_cfg["OPTIMIZER"] != "adam_onecycle":
        raise NotImplementedError(optim_cfg["OPTIMIZER"])
    return FlatAdamOneCycle(model, wd=optim_cfg["WEIGHT_DECAY"], grad_norm_clip=optim_cfg.get("GRAD_NORM_CLIP"))


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, optim_cfg):
    """optimization/__init__.py:42-63 (adam_onecycle branch)."""
    total_steps = total_iters_each_epoch * total_epochs
    return OneCycle(optimizer, total_steps, optim_cfg["LR"], list(optim_cfg["MOMS"]), optim_cfg["DIV_FACTOR"],
                    optim_cfg["PCT_START"])


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    """train_utils.py:151-170: the dict the reference torch.save()s as checkpoint_epoch_N.pth."""
    if isinstance(model, nn.parallel.DistributedDataParallel):
        model_state = type(model.module.state_dict())((k, v.cpu()) for k, v in model.module.state_dict().items())
    else:
        model_state = model.state_dict() if model is not None else None
    return {"epoch": epoch, "it": it, "model_state": model_state,
            "optimizer_state": optimizer.state_dict() if optimizer is not None else None, "version": "none"}


def save_checkpoint(state, filename="checkpoint"):
    """train_utils.py:173-182."""
    torch.save(state, "{}.pth".format(filename))

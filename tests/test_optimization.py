"""adam_onecycle against tests/golden/optim_onecycle.npz, which
tests/golden/make_optim_golden.py produced by running the reference's own optimizer code on CPU."""
import os

import numpy as np
import pytest
import torch
from torch import nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_onecycle.npz")


class Net(nn.Module):  # same topology as make_optim_golden.Net
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv1d(5, 16, 1, bias=False), nn.BatchNorm1d(16), nn.ReLU())
        self.attn = nn.MultiheadAttention(16, 4)
        self.block = nn.Sequential(nn.Conv2d(16, 9, 1), nn.BatchNorm2d(9), nn.ReLU(), nn.Conv2d(9, 3, 1))
        self.norm = nn.LayerNorm(16)
        self.head = nn.Linear(16, 7)


CFG = dict(OPTIMIZER="adam_onecycle", LR=0.01, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85], PCT_START=0.4,
           DIV_FACTOR=10, GRAD_NORM_CLIP=10)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _model(gold, device):
    m = Net()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(gold["init/" + n]))
    return m.to(device)


def test_parameter_groups_follow_the_reference_flattening(gold):
    from pdanet_amd import optimization as opt
    g0, g1 = opt.trained_parameter_groups(_model(gold, "cpu"))
    assert [n for n, _ in g0] == list(gold["group0"])
    assert [n for n, _ in g1] == list(gold["group1"])
    # the reference never trains MultiheadAttention.in_proj_* (flatten_model drops non-leaf owners)
    trained = {n for n, _ in g0 + g1}
    assert set(gold["names"]) - trained == {"attn.in_proj_weight", "attn.in_proj_bias"}


def test_onecycle_schedule_matches_reference(gold):
    from pdanet_amd import optimization as opt

    class Dummy:
        lr = mom = None
    d = Dummy()
    sched = opt.OneCycle(d, 30, CFG["LR"], CFG["MOMS"], CFG["DIV_FACTOR"], CFG["PCT_START"])
    assert d.lr == pytest.approx(0.001) and d.mom == 0.95
    for it in range(len(gold["lr"])):
        sched.step(it)
        assert d.lr == pytest.approx(float(gold["lr"][it]), rel=1e-12)
        assert d.mom == pytest.approx(float(gold["mom"][it]), rel=1e-12)
    # end of the cycle: lr_max / div / 1e4 is approached, momentum returns to moms[0]
    lr, mom = sched.values(29)
    assert lr < 2e-4 and mom > 0.94


def test_cpu_parameters_are_rejected(gold):
    from pdanet_amd import optimization as opt
    from pdanet_amd._lib import PdaError
    # the flat buffers are plumbing and may be built anywhere (the gloo tests exchange gradients on the CPU); the update
    # itself is csrc/optim.hip and nothing else
    o = opt.FlatAdamOneCycle(_model(gold, "cpu"), wd=0.01)
    with pytest.raises(PdaError, match="no CPU path"):
        o.step()


@pytest.mark.gpu
def test_trajectory_matches_reference_optimizer(gold):
    """12 iterations of clip_grad_norm_ + decoupled decay + Adam under OneCycle: parameters within
    2e-6 absolute of the reference's CPU run (fp32 update arithmetic in a different op order)."""
    from pdanet_amd import optimization as opt
    model = _model(gold, "cuda")
    o = opt.build_optimizer(model, CFG)
    sched = opt.build_scheduler(o, 10, 3, CFG)
    names = list(gold["names"])
    for it in range(len(gold["lr"])):
        sched.step(it)
        o.zero_grad()
        for n, p in model.named_parameters():
            p.grad.copy_(torch.from_numpy(gold["grad/%d/%s" % (it, n)]))    # .grad stays a view of flat_g
        o.step()
        assert float(o.total_norm) == pytest.approx(float(gold["norm"][it]), rel=1e-5)
        for n, p in model.named_parameters():
            ref = gold["param/%d/%s" % (it, n)]
            err = np.abs(p.detach().cpu().numpy() - ref).max()
            assert err < 2e-6, (it, n, err)
    # in_proj_* untouched (reference quirk), optimizer state in torch.optim.Adam's checkpoint layout
    assert torch.equal(model.attn.in_proj_weight.cpu(), torch.from_numpy(gold["init/attn.in_proj_weight"]))
    sd = o.state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == list(gold["sd_groups"])
    for k, st in sd["state"].items():
        assert float(st["step"]) == float(gold["state/%d/step" % k])
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), gold["state/%d/exp_avg" % k], atol=1e-6)
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), gold["state/%d/exp_avg_sq" % k], rtol=1e-5, atol=1e-9)
    # round trip
    o2 = opt.build_optimizer(_model(gold, "cuda"), CFG)
    o2.load_state_dict(sd)
    assert o2.step_count == o.step_count and torch.equal(o2.exp_avg, o.exp_avg)


@pytest.mark.gpu
def test_gradients_accumulate_into_the_flat_buffer(gold):
    from pdanet_amd import optimization as opt
    model = _model(gold, "cuda")
    o = opt.build_optimizer(model, CFG)
    x = torch.randn(4, 16, device="cuda")
    model.head(model.norm(x)).sum().backward()
    assert model.head.weight.grad.data_ptr() >= o.flat_g.data_ptr()
    assert model.head.weight.grad.data_ptr() < o.flat_g.data_ptr() + o.flat_g.numel() * 4
    assert float(o.flat_g.abs().sum()) > 0
    o.zero_grad()
    assert float(model.head.weight.grad.abs().sum()) == 0


@pytest.mark.gpu
def test_derived_tensor_caches_follow_the_raw_pointer_optimizer():
    """FlatAdamOneCycle writes parameters through a raw pointer (version counters do not move): the cached bf16 weight
    copies (dense-bf16 mode) and the BatchNorm-folded inference weights must still be refreshed after a step."""
    import torch.nn as nn
    from pdanet_amd import optimization, pointnet2_utils as pu, pointnet2_modules as pm
    torch.manual_seed(0)
    model = nn.Sequential(nn.Conv2d(8, 16, 1, bias=False), nn.BatchNorm2d(16), nn.ReLU()).cuda()
    opt = optimization.FlatAdamOneCycle(model, wd=0.01, lr=1e-2)
    conv, bn = model[0], model[1]
    wb0 = pu._b16p(conv.weight).clone()
    assert pu._b16p(conv.weight) is pu._b16p(conv.weight)          # cached while nothing changes
    model.eval()
    with torch.no_grad():
        w0 = pm._folded_conv_bn(conv, bn)[0].clone()
    model.train()
    x = torch.randn(4, 8, 5, 5, device="cuda")
    opt.zero_grad()
    model(x).square().mean().backward()
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(pu._b16p(conv.weight), wb0)
    assert torch.equal(pu._b16p(conv.weight), conv.weight.detach().bfloat16())
    model.eval()
    with torch.no_grad():
        w1, b1 = pm._folded_conv_bn(conv, bn)
        s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        assert not torch.equal(w1, w0)
        assert torch.allclose(w1, conv.weight.flatten(1) * s[:, None])


@pytest.mark.gpu
def test_set_to_none_gather_path_equals_accumulating_views():
    """zero_grad(set_to_none=True): autograd hands out fresh gradient tensors and step() gathers them with one
    multi-tensor copy.  Same parameters, norm and moments as the default path (gradients accumulated into the views),
    including a parameter that receives no gradient and two backward passes accumulated into one step."""
    from pdanet_amd import optimization
    results = []
    for mode in (False, True):
        torch.manual_seed(11)
        model = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 3)).cuda()
        opt = optimization.FlatAdamOneCycle(model, wd=0.01, lr=5e-3, grad_norm_clip=0.5)
        x = torch.randn(32, 6, device="cuda")
        norms = []
        for it in range(4):
            opt.zero_grad(set_to_none=mode)
            h = model[2](model[1](model[0](x)))              # model[3] unused: no gradient this step
            h.square().mean().backward()
            if it == 2:
                (h.detach() * 0 + model[2](model[1](model[0](x * 0.5)))).abs().mean().backward()   # accumulate a second backward
            opt.step()
            norms.append(float(opt.total_norm))
            for p in model.parameters():                     # .grad are views of the flat buffer after step()
                assert p.grad is not None and opt.flat_g.data_ptr() <= p.grad.data_ptr() < opt.flat_g.data_ptr() + 4 * opt.flat_g.numel()
        results.append((norms, opt.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()))
    (na, pa, ma, va), (nb, pb, mb, vb) = results
    assert na == pytest.approx(nb, rel=1e-6)
    assert torch.allclose(pa, pb, atol=1e-7, rtol=1e-6) and torch.allclose(ma, mb, atol=1e-9, rtol=1e-5) and torch.allclose(va, vb, atol=1e-12, rtol=1e-5)

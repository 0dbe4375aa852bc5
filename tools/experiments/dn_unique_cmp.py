import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from pdanet_amd import pointnet2_utils as pu, pointnet2_modules as pm
import test_detector_train as T
res = {}
for flag in (False, True):
    pu.DENSITYNET_UNIQUE = flag
    model, opt, sched, bd = T._setup("once_pda_ssd.yaml", "once")
    outs = []
    hooks = []
    for i, m in enumerate(model.backbone_3d.SA_modules):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: outs.append((i, [o.detach().clone() if torch.is_tensor(o) else o for o in out]))))
    dn_out = []
    orig = pu.densitynet
    def wrapped(dn, x, part=None):
        y = orig(dn, x, part)
        dn_out.append((x.detach().clone(), y.detach().clone(), None if part is None else [t.clone() if torch.is_tensor(t) else t for t in part]))
        return y
    pu.densitynet = wrapped
    ret, tb = T._iteration(model, opt, sched, bd, 0)
    pu.densitynet = orig
    res[flag] = (float(ret['loss']), outs, dn_out)
    print(flag, "loss", float(ret['loss']))
a, b = res[False], res[True]
for k, ((xa, ya, _), (xb, yb, part)) in enumerate(zip(a[2], b[2])):
    d = (ya - yb).abs()
    print("densitynet call", k, tuple(xa.shape), "x equal", torch.equal(xa, xb), "max |dy|", d.max().item(), "n bad", int((d > 1e-5).sum()))
    if part is not None and d.max().item() > 1e-5:
        cnt, off, rowmap, roww, G, ns = part
        bad = (d.view(-1) > 1e-5).nonzero().view(-1)[:10].tolist()
        print("  bad flat slots", bad, "ns", ns, "U", int(off[-1]))
        for e in bad[:5]:
            g, s = divmod(e, ns)
            print("   group", g, "slot", s, "cnt", int(cnt[g]), "x", xa.view(-1)[g*ns:(g+1)*ns].tolist()[:ns], "y dense", ya.view(-1)[e].item(), "y uniq", yb.view(-1)[e].item())
for (i, oa), (_, ob) in zip(a[1], b[1]):
    for j, (ta, tb_) in enumerate(zip(oa, ob)):
        if torch.is_tensor(ta) and ta.shape == tb_.shape and ta.is_floating_point():
            print("layer", i, "out", j, tuple(ta.shape), "max diff", (ta - tb_).abs().max().item())

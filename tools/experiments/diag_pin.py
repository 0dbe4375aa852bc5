import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from pdanet_amd import pointnet2_utils as pu
import test_detector_train as t
for cfg, ds in t.CASES:
    res = {}
    for fused in (True, False):
        pu.SA_SMALL_TRAIN = fused
        model, opt, sched, bd = t._setup(cfg, ds)
        feats = {}
        hook = model.backbone_3d.register_forward_hook(lambda m, i, o: feats.update(o))
        sched.step(0); opt.zero_grad()
        ret, tb, _ = model(bd())
        hook.remove()
        res[fused] = (float(ret['loss']), [x.clone() if torch.is_tensor(x) else x for x in feats['sample_list_id']],
                      [f.clone() if f is not None else None for f in feats['encoder_features']], {k: float(v) for k, v in tb.items()})
    lf, sf, ff, tf = res[True]; ll, sl, fl, tl = res[False]
    print(ds, "loss fused %.9f layerwise %.9f rel %.2e" % (lf, ll, abs(lf - ll) / ll))
    for i, (a, b) in enumerate(zip(sf, sl)):
        if torch.is_tensor(a) and a.numel():
            same = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(a, b))
            print("  layer", i, "sampled ids: common", same, "of", a.numel())
    for i, (a, b) in enumerate(zip(ff, fl)):
        if a is not None and a.shape == b.shape:
            print("  features", i, "max rel diff %.2e" % ((a - b).abs().max() / b.abs().max()).item())
    for k in tf:
        if abs(tf[k] - tl[k]) > 1e-4 * max(1e-6, abs(tl[k])):
            print("   ", k, tf[k], tl[k])

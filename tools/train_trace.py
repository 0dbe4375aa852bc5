"""Per-iteration trace of the small training setup of tests/test_detector_train.py (B=2, N=4096, 100-iteration
OneCycle): total loss, every loss term of tb_dict, the pre-clip gradient norm and lr.  Written to explain the
loss swings of the first iterations.  Usage (MI355X):
    python tools/train_trace.py [iters] [cfg ...]"""
import sys

import torch

sys.path.insert(0, '.')
from pdanet_amd import detector, optimization, synth  # noqa: E402


def trace(cfg, dataset, iters, B=2, N=4096):
    torch.manual_seed(7)
    model, c = detector.build_detector(cfg)
    model = model.cuda().train()
    pts = synth.batch_points(B, N, config_id=2, dist="L", dataset=dataset)
    gt = synth.gt_boxes(pts, B, config_id=2, dataset=dataset)
    opt = optimization.build_optimizer(model, c.OPTIMIZATION)
    sched = optimization.build_scheduler(opt, 100, 2, c.OPTIMIZATION)
    pts_d, gt_d = torch.from_numpy(pts).cuda(), torch.from_numpy(gt).cuda()
    print("==", cfg, flush=True)
    for it in range(iters):
        sched.step(it)
        opt.zero_grad()
        ret, tb, _ = model({'batch_size': B, 'points': pts_d, 'gt_boxes': gt_d})
        ret['loss'].backward()
        opt.step()
        terms = " ".join("%s=%.4g" % (k, float(v)) for k, v in tb.items() if 'loss' in k)
        print("it %3d loss %.17g norm %.5g lr %.3g | %s" % (it, float(ret['loss']), float(opt.total_norm), opt.lr, terms), flush=True)


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cfgs = sys.argv[2:] or ["kitti", "once"]
    for d in cfgs:
        trace("%s_pda_ssd.yaml" % d, d, iters)

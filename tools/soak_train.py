"""200 training iterations of the whole detector (fp32 ONCE and dense-bf16 KITTI) on synthetic scenes: prints the loss every 20
iterations and whether every value stayed finite.  Usage (MI355X): PYTHONPATH=. python tools/soak_train.py"""
import sys, time, torch
sys.path.insert(0, '.')
from benchmarks import workloads as bw
dev = torch.device('cuda:0')
for name, batch in (('kitti_detector_train_bf16', 4), ('kitti_detector_train', 4), ('detector_train', 2)):
    wl = bw.create(name, batch, 16384, dev, 0, 1); wl.begin()
    losses = []
    t = time.perf_counter()
    for i in range(200):
        l = wl.step()
        if i % 20 == 0 or i == 199:
            losses.append(float(l.detach()))
    torch.cuda.synchronize()
    ok = all(x == x and abs(x) < 1e6 for x in losses)
    print(name, "200 steps in %.1f s" % (time.perf_counter() - t), "finite" if ok else "NON-FINITE", " ".join("%.3f" % x for x in losses),
          "mem GB %.2f" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del wl

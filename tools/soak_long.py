"""Long soak of the whole training iteration on synthetic scenes: N iterations per workload (default 3000), the loss every
N/10 iterations, finiteness, peak memory.  Usage (MI355X, repo root): python tools/soak_long.py [N] > gpurun_out/soak.txt"""
import sys
import time

import torch

sys.path.insert(0, '.')
from benchmarks import workloads as bw  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device('cuda:0')
for name, batch in (('detector_train', 2), ('kitti_detector_train', 4)):
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    wl = bw.create(name, batch, 16384, dev, 0, 1)
    wl.begin()
    t = time.perf_counter()
    losses = []
    for i in range(n):
        loss = wl.step()
        if i % max(1, n // 10) == 0 or i == n - 1:
            losses.append(float(loss))
            print(name, i, "%.1f s" % (time.perf_counter() - t), "%.3f" % losses[-1], flush=True)
    torch.cuda.synchronize()
    ok = all(x == x and abs(x) < 1e6 for x in losses)
    print(name, "%d iterations in %.1f s (%.2f ms each)" % (n, time.perf_counter() - t, 1e3 * (time.perf_counter() - t) / n),
          "finite" if ok else "NON-FINITE", "peak mem GB %.2f" % (torch.cuda.max_memory_allocated() / 2**30),
          "graph_tail", wl.model.graph_tail, flush=True)
    del wl

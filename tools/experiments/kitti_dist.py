import sys, time, torch
sys.path.insert(0, '.')
from benchmarks import workloads
dev = torch.device("cuda:0")
wl = workloads.create("kitti_detector_train", 4, 16384, dev, 0, 1)
for _ in range(10): wl.step()
torch.cuda.synchronize()
ts = []
for blk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): wl.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
print("kitti 12 blocks of 10 steps, ms/step:", " ".join("%.2f" % t for t in ts), "graph_tail", wl.model.graph_tail, flush=True)
import os
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "threads", torch.get_num_threads())

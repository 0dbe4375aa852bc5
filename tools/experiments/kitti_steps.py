import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchmarks import workloads
dev = torch.device("cuda:0")
for name in ("kitti_detector_train_bf16", "kitti_detector_train"):
    torch.cuda.empty_cache()
    wl = workloads.create(name, 4, 16384, dev, 0, 1)
    wl.begin()
    ts = []
    for i in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wl.step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "host_bound", getattr(wl, "host_bound", None), "graph_tail", wl.model.graph_tail, " ".join("%.1f" % t for t in ts), flush=True)
    del wl

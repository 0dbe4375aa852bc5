import sys, time, os, torch
sys.path.insert(0, '.')
from benchmarks import workloads as bw
dev = torch.device('cuda:0')
mode = os.environ.get("MODE", "keep")
wl = bw.create("kitti_detector_train", 4, 16384, dev, 0, 1); wl.begin()
for i in range(12):
    if mode == "keep":
        l = wl.step()
    elif mode == "discard":
        wl.step()
    elif mode == "sync":
        wl.step(); torch.cuda.synchronize()
torch.cuda.synchronize()
print(mode, "ok graph_tail", wl.model.graph_tail, flush=True)

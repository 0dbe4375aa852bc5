import sys, time, torch
sys.path.insert(0, '.')
from benchmarks import workloads
dev = torch.device("cuda:0")
wl = workloads.create("detector_train", 2, 16384, dev, 0, 1)
for _ in range(8): wl.step()
torch.cuda.synchronize()
for trial in range(3):
    hs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter(); wl.step(); hs.append(time.perf_counter() - a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("trial %d: wall/step %.2f ms | host enqueue/step mean %.2f ms (min %.2f max %.2f) | host done at %.1f %% of wall | graph_tail %s graph_head %s" % (
        trial, (t2 - t0) / 20 * 1e3, sum(hs) / 20 * 1e3, min(hs) * 1e3, max(hs) * 1e3, (t1 - t0) / (t2 - t0) * 100, wl.model.graph_tail, getattr(wl.model, 'graph_head', None)), flush=True)
# one step from an idle device: host time alone (no back-pressure)
for trial in range(3):
    torch.cuda.synchronize(); a = time.perf_counter(); wl.step(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print("single step from idle: host %.2f ms, total %.2f ms" % ((b - a) * 1e3, (c - a) * 1e3), flush=True)

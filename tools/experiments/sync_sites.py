import sys, time, warnings, traceback, torch
sys.path.insert(0, '.')
from benchmarks import workloads
dev = torch.device("cuda:0")
wl = workloads.create("detector_train", 2, 16384, dev, 0, 1)
for _ in range(8): wl.step()
torch.cuda.synchronize()
sites = {}
def hook(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()
    key = " <- ".join("%s:%d" % (f.filename.split('/')[-1], f.lineno) for f in st[-8:-1] if 'repo' in f.filename)
    sites[key] = sites.get(key, 0) + 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(2): wl.step()
torch.cuda.set_sync_debug_mode("default")
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(v / 2, "x/step:", k)
print("total synchronizing calls per step:", sum(sites.values()) / 2)

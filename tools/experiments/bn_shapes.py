import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchmarks import workloads
from pdanet_amd import pointnet2_batch_cuda as ext
log = collections.Counter()
on = [False]
def wrap(name, rows_i, c_i):
    f = getattr(ext, name)
    def g(*a, **k):
        if on[0]: log[(name, int(a[rows_i]) if not isinstance(rows_i, tuple) else int(a[rows_i[0]]) * int(a[rows_i[1]]), int(a[c_i]))] += 1
        return f(*a, **k)
    setattr(ext, name, g)
wrap("bn_relu_fwd", 8, 9); wrap("bn_relu_bwd", 9, 10)
wrap("bn_relu_max_pool_fwd", (9, 10), 11); wrap("bn_relu_max_pool_bwd", (10, 11), 12)
wrap("bn_relu_fwd_weighted", 8, 9); wrap("bn_relu_bwd_weighted", 9, 10)
wrap("bn_stats_fwd", 5, 6)
dev = torch.device("cuda:0")
wl = workloads.create("detector_train", 2, 16384, dev, 0, 1)
wl.model.graph_head = False
wl.begin()
for _ in range(3): wl.step()
on[0] = True
wl.step()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: (kv[0][0], kv[0][1] * kv[0][2])): print(k, v)

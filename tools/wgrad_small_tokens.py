import sys, torch
sys.path.insert(0, "/root/repo")
from pdanet_amd import pointnet2_batch_cuda as ext
def tg(fn, reps=20):
    # device time per call: replay a captured graph of `reps` calls
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3
for T,M,N in [(8192,128,256),(8192,256,128),(8192,128,128),(4096,256,512),(4096,512,256),(4096,512,1536),(4096,256,256),(4096,128,128),(4096,512,512),(8192,64,128),(12979,256,256),(12979,128,256)]:
    x=torch.randn(T,M,device="cuda"); g=torch.randn(T,N,device="cuda"); gw=torch.empty(N,M,device="cuda"); gb=torch.empty(N,device="cuda")
    own=tg(lambda: ext.linear_wgrad(x,g,gw,gb,T,M,N))
    lib=tg(lambda: (g.t().mm(x), g.sum(0)))
    lib1=tg(lambda: g.t().mm(x))
    print(f"T={T:6d} M={M:4d} N={N:4d}  own {own:6.1f} us  lib mm+sum {lib:6.1f} us  (mm alone {lib1:6.1f})", flush=True)

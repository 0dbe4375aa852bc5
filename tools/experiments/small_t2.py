import sys, torch
sys.path.insert(0, '.')
from pdanet_amd import pointnet2_batch_cuda as ext
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
import os
for T, K, N in ((4096, 512, 512), (4096, 512, 1536), (4096, 1536, 512), (4096, 512, 256), (4096, 256, 512), (12979, 768, 256), (12979, 256, 768), (8192, 512, 512), (32307, 768, 256)):
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.zeros(T, N, device="cuda")
    wf = ext.linear_split_pack(w, N, K)
    a = t(lambda: ext.gemm_split(x, wf, None, y, T, K, N))
    ext.gemm_split(x, wf, None, y, T, K, N)
    ref = x[:512].double() @ w.double().t()
    err = ((y[:512].double() - ref).abs() / (x[:512].double().abs() @ w.double().abs().t())).max().item()
    print(os.environ.get("PDA_GEMM_SPLIT_DEEP", "1"), (T, K, N), "gemm_split %.1f us  err %.1e | torch f32 mm %.1f us" % (a, err, t(lambda: torch.mm(x, w.t()))), flush=True)

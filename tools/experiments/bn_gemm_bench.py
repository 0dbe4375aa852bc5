import sys, torch
sys.path.insert(0, '.')
from pdanet_amd import pointnet2_batch_cuda as ext
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for T, K, N in ((131072, 256, 512), (131072, 512, 512), (131072, 512, 256), (65536, 256, 256), (65536, 256, 512), (32768, 256, 256), (32768, 512, 256)):
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; y = torch.empty(T, N, device="cuda")
    z = torch.randn(T, N, device="cuda")
    wf = ext.linear_split_pack(w, N, K)
    mi = torch.cat([torch.zeros(K), torch.ones(K)]).cuda(); g = torch.ones(K, device="cuda"); b = torch.zeros(K, device="cuda")
    mo = torch.cat([torch.zeros(N), torch.ones(N)]).cuda(); go = torch.ones(N, device="cuda"); bo = torch.zeros(N, device="cuda")
    tiles = ext.gemm_split_bn_tiles(T)
    part = torch.empty(tiles * 2 * N, dtype=torch.float64, device="cuda")
    r = {}
    r["gemm_split"] = t(lambda: ext.gemm_split(x, wf, None, y, T, K, N))
    if K <= 512 and N % 128 == 0: r["lin_split"] = t(lambda: ext.linear_split(x, wf, None, y, T, K, N))
    r["bn(0,0)"] = t(lambda: ext.gemm_split_bn(x, wf, y, T, K, N))
    r["PRO"] = t(lambda: ext.gemm_split_bn(x, wf, y, T, K, N, in_bn=(mi, g, b)))
    r["EPI1"] = t(lambda: ext.gemm_split_bn(x, wf, y, T, K, N, stats_mode=1, partial=part))
    r["PRO+EPI1"] = t(lambda: ext.gemm_split_bn(x, wf, y, T, K, N, in_bn=(mi, g, b), stats_mode=1, partial=part))
    r["EPI2"] = t(lambda: ext.gemm_split_bn(x, wf, y, T, K, N, stats_mode=2, partial=part, z=z, out_bn=(mo, go, bo)))
    print((T, K, N), " ".join("%s %.0f" % kv for kv in r.items()), flush=True)

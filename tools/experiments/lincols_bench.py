import sys, torch
sys.path.insert(0, '.')
from pdanet_amd import pointnet2_batch_cuda as ext, pointnet2_utils as pu, synth
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for T, K, N in ((32768, 256, 256), (65536, 256, 256), (131072, 256, 512), (131072, 512, 512), (65536, 256, 512), (131072, 512, 256)):
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(T, N, device="cuda")
    wf = ext.linear_cols_pack(w, N, K)
    fl = 2.0 * T * K * N
    a = t(lambda: ext.linear_cols(x, wf, y, T, K, N)); b = t(lambda: torch.nn.functional.linear(x, w)); c = t(lambda: ext.linear_cols_pack(w, N, K))
    print("T=%6d K=%3d N=%4d  own %.3f ms %5.1f TF | lib %.3f ms %5.1f TF | pack %.3f ms" % (T, K, N, a, fl / a / 1e9, b, fl / b / 1e9, c), flush=True)
B, Npts, C = 2, 2048, 256
xyz = torch.from_numpy(synth.batch_xyz(B, Npts, config_id=3)).cuda(); feats = torch.randn(B, Npts, C, device="cuda")
for M, ns in ((1024, 16), (1024, 32), (1024, 64)):
    new_xyz = xyz[:, :M].contiguous(); idx = pu.ball_query(12.8, ns, xyz, new_xyz)
    w = torch.randn(256, 259, device="cuda"); wf = ext.linear_cols_pack(w, 256, 259, gather_order=True)
    y = torch.empty(B, M, ns, 256, device="cuda"); fl = 2.0 * B * M * ns * 259 * 256
    a = t(lambda: ext.sa_gather_linear(xyz, new_xyz, feats, idx, wf, y, B, Npts, M, C, ns, 256))
    def lib():
        x0 = torch.cat([pu.group_rows(xyz, idx) - new_xyz.unsqueeze(2), pu.group_rows(feats, idx)], dim=-1)
        return torch.nn.functional.linear(x0, w)
    b = t(lib)
    print("gather M=%d ns=%d: own %.3f ms %5.1f TF | lib (group+cat+gemm) %.3f ms %5.1f TF" % (M, ns, a, fl / a / 1e9, b, fl / b / 1e9), flush=True)

import sys, ctypes, torch
vp, i64, i = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
libs = {"NOSTORE": ctypes.CDLL("build/libgs_NOSTORE.so"), "full": ctypes.CDLL("pdanet_amd/libpda_pointnet2.so")}
for l in libs.values():
    l.pda_linear_split_packed_bytes.restype = i64
    l.pda_linear_split_pack.argtypes = [vp, vp, i, i, i, vp]
    l.pda_gemm_split.argtypes = [vp, vp, vp, vp, i64, i, i, i, i, vp]
for T, K, N in ((131072, 512, 512), (131072, 256, 512), (65536, 256, 512), (65536, 256, 256), (131072, 256, 256), (196608, 128, 256), (196608, 256, 256), (100000, 256, 512), (100000, 512, 256)):
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(T, N, device="cuda")
    out = []
    for name, l in libs.items():
        wf = torch.empty(l.pda_linear_split_packed_bytes(N, K), dtype=torch.uint8, device="cuda")
        l.pda_linear_split_pack(w.data_ptr(), wf.data_ptr(), N, K, 0, None)
        a = t(lambda: l.pda_gemm_split(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), T, K, N, 0, 0, None))
        out.append("%s %.3f ms %.0f TF(x6 %.0f)" % (name, a, 2.0 * T * K * N / a / 1e9, 12.0 * T * K * N / a / 1e9))
    hbm = (T * K + T * N) * 4 / 1e9
    print((T, K, N), " | ".join(out), "| HBM bytes %.0f MB (%.0f us at 4 TB/s), mfma ideal %.0f us at 1.9 PF" % (hbm * 1e3, hbm / 4e3 * 1e6, 12.0 * T * K * N / 1.9e15 * 1e6), flush=True)

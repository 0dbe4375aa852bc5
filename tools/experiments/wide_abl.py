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
names = ["NOSTORE", "NOBAR", "NOREAD", "NOSPLIT", "NOWRITE", "NOLOAD", "ALL"]
libs = {n: ctypes.CDLL("build/libg%s_%s.so" % ("s" if n == "NOSTORE" else "w", n)) for n in names}
for l in libs.values():
    l.pda_linear_split_packed_bytes.restype = i64
    l.pda_linear_split_pack.argtypes = [vp, vp, i, i, i, vp]
    l.pda_gemm_split.argtypes = [vp, vp, vp, vp, i64, i, i, i, i, vp]
for T, K, N in ((131072, 512, 512), (131072, 256, 256)):
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(T, N, device="cuda")
    ideal = 12.0 * T * K * N / 2.5e15 * 1e3
    for name, l in libs.items():
        wf = torch.empty(l.pda_linear_split_packed_bytes(N, K), dtype=torch.uint8, device="cuda")
        l.pda_linear_split_pack(w.data_ptr(), wf.data_ptr(), N, K, 0, None)
        a = t(lambda: l.pda_gemm_split(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), T, K, N, 0, 0, None))
        print((T, K, N), "%-8s %.3f ms  x6 %.0f TF  (ideal at 2.5 PF %.3f ms)" % (name, a, 12.0 * T * K * N / a / 1e9, ideal), flush=True)

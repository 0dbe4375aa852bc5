#!/usr/bin/env python
"""per-operator timings, CPU oracle (1 thread / all cores) next to the HIP
kernels on the same box, at the config-1 and config-2 layer shapes.  Parity gate first: a shape is
only timed after the HIP result matched the oracle (indices bit-exact).

  python tools/op_baseline.py > profiles/rNN_op_baseline.md      (needs the MI355X)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (CPU baseline leg)
from pdanet_amd import pointnet2_batch_cuda as ext, synth  # noqa: E402

NCORES = oracle.num_threads()


def cpu_time(fn, threads, reps=5):
    oracle.set_num_threads(threads)
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    oracle.set_num_threads(NCORES)
    return float(np.median(ts))


def gpu_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def row(name, alg_bytes, t1, tall, tg):
    print("| %s | %.1f MB | %.2f ms (%.2f GB/s) | %.2f ms (%.1f GB/s) | %.3f ms (%.0f GB/s) | %.0fx / %.0fx |" % (
        name, alg_bytes / 1e6, t1 * 1e3, alg_bytes / t1 / 1e9, tall * 1e3, alg_bytes / tall / 1e9,
        tg * 1e3, alg_bytes / tg / 1e9, t1 / tg, tall / tg))


def fps_case(b, n, m, tag):
    xyz = synth.batch_xyz(b, n, config_id=2)
    idx_o = np.zeros((b, m), np.int32)

    def cpu():
        temp = np.full((b, n), 1e10, np.float32)
        oracle.farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx_o)
    xyz_d = torch.from_numpy(xyz).cuda()
    idx_d = torch.zeros((b, m), dtype=torch.int32, device="cuda")

    def gpu():
        temp = torch.full((b, n), 1e10, device="cuda")
        ext.farthest_point_sampling_wrapper(b, n, m, xyz_d, temp, idx_d)
    t1, tall = cpu_time(cpu, 1, 3), cpu_time(cpu, NCORES, 3)
    tg = gpu_time(gpu)
    assert np.array_equal(idx_o, idx_d.cpu().numpy()), "parity gate failed: FPS"
    row("FPS %d->%d, %d scene%s (%s)" % (n, m, b, "s" * (b > 1), tag), b * ((m - 1) * n * 20 + m * 4), t1, tall, tg)


def bq_case(b, n, m, r, ns, tag):
    xyz = synth.batch_xyz(b, n, config_id=2)
    new_xyz = np.ascontiguousarray(xyz[:, :m])
    idx_o = np.zeros((b, m, ns), np.int32)

    def cpu():
        idx_o[:] = 0
        oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, idx_o)
    xd, nd = torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda()
    idx_d = torch.zeros((b, m, ns), dtype=torch.int32, device="cuda")

    def gpu():
        idx_d.zero_()
        ext.ball_query_wrapper(b, n, m, r, ns, nd, xd, idx_d)
    t1, tall = cpu_time(cpu, 1, 3), cpu_time(cpu, NCORES, 5)
    tg = gpu_time(gpu)
    assert np.array_equal(idx_o, idx_d.cpu().numpy()), "parity gate failed: ball query"
    alg = b * (-(-m // 256) * n * 12 + m * 12 + m * ns * 4)
    row("ball_query %dx%d r=%.1f ns=%d, %d scene%s (%s)" % (m, n, r, ns, b, "s" * (b > 1), tag), alg, t1, tall, tg)
    return idx_o, idx_d


def group_case(b, c, n, m, ns, tag):
    rng = np.random.default_rng(c)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    out_o = np.zeros((b, c, m, ns), np.float32)
    pd, idd = torch.from_numpy(pts).cuda(), torch.from_numpy(idx).cuda()
    out_d = torch.empty((b, c, m, ns), device="cuda")
    t1 = cpu_time(lambda: oracle.group_points_wrapper(b, c, n, m, ns, pts, idx, out_o), 1, 3)
    tall = cpu_time(lambda: oracle.group_points_wrapper(b, c, n, m, ns, pts, idx, out_o), NCORES, 5)
    tg = gpu_time(lambda: ext.group_points_wrapper(b, c, n, m, ns, pd, idd, out_d))
    assert np.array_equal(out_o, out_d.cpu().numpy()), "parity gate failed: group"
    row("group_points C=%d M=%d ns=%d, %d scene%s (%s)" % (c, m, ns, b, "s" * (b > 1), tag),
        b * (m * ns * 4 + c * m * ns * 8), t1, tall, tg)


def three_nn_case(b, n, m, c, tag):
    unknown = synth.batch_xyz(b, n, config_id=2)
    known = np.ascontiguousarray(unknown[:, :m])
    d_o = np.zeros((b, n, 3), np.float32); i_o = np.zeros((b, n, 3), np.int32)
    ud, kd = torch.from_numpy(unknown).cuda(), torch.from_numpy(known).cuda()
    d_d = torch.zeros((b, n, 3), device="cuda"); i_d = torch.zeros((b, n, 3), dtype=torch.int32, device="cuda")
    t1 = cpu_time(lambda: oracle.three_nn_wrapper(b, n, m, unknown, known, d_o, i_o), 1, 3)
    tall = cpu_time(lambda: oracle.three_nn_wrapper(b, n, m, unknown, known, d_o, i_o), NCORES, 5)
    tg = gpu_time(lambda: ext.three_nn_wrapper(b, n, m, ud, kd, d_d, i_d))
    assert np.array_equal(i_o, i_d.cpu().numpy()) and np.array_equal(d_o, d_d.cpu().numpy()), "parity gate failed: three_nn"
    row("three_nn n=%d m=%d, %d scenes (%s)" % (n, m, b, tag), b * (-(-n // 256) * m * 12 + n * 36), t1, tall, tg)
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(b, c, m)).astype(np.float32)
    w = rng.uniform(0.1, 1, size=(b, n, 3)).astype(np.float32); w /= w.sum(-1, keepdims=True)
    out_o = np.zeros((b, c, n), np.float32)
    pd, wd = torch.from_numpy(pts).cuda(), torch.from_numpy(w).cuda()
    out_d = torch.empty((b, c, n), device="cuda")
    t1 = cpu_time(lambda: oracle.three_interpolate_wrapper(b, c, m, n, pts, i_o, w, out_o), 1, 3)
    tall = cpu_time(lambda: oracle.three_interpolate_wrapper(b, c, m, n, pts, i_o, w, out_o), NCORES, 5)
    tg = gpu_time(lambda: ext.three_interpolate_wrapper(b, c, m, n, pd, i_d, wd, out_d))
    assert np.abs(out_o - out_d.cpu().numpy()).max() <= 1e-4, "parity gate failed: three_interpolate (1e-4)"
    row("three_interpolate C=%d n=%d, %d scenes (%s)" % (c, n, b, tag), b * (n * 24 + c * n * 16), t1, tall, tg)


def main():
    print("## Operator baseline: CPU oracle vs HIP kernels on the same box\n")
    print("Host: %d hardware threads (`nproc`), CPU oracle = oracle/libpda_oracle.so (gcc -O3 -mavx2 -mfma, OpenMP); "
          "GPU: %s.  Median of 3-5 CPU runs after one warm-up; GPU = mean of 20 launches between HIP events.  "
          "GB/s are effective rates under the algorithmic-byte formulas of BASELINE.md section 2 (they exceed "
          "real traffic for FPS, which keeps the scene on-chip).  Every row passed the parity gate (bit-exact "
          "indices / copies) before it was timed.\n" % (NCORES, torch.cuda.get_device_name(0)))
    print("| op / shape | algorithmic bytes | CPU 1 thread | CPU %d threads | MI355X | speed-up vs 1 / all |" % NCORES)
    print("|---|---|---|---|---|---|")
    fps_case(1, 4096, 1024, "config 1")
    bq_case(1, 4096, 1024, 0.8, 16, "config 1")
    bq_case(1, 4096, 1024, 1.6, 32, "config 1")
    fps_case(2, 16384, 4096, "config 2, layer 1")
    bq_case(2, 16384, 16384, 0.2, 16, "config 2, layer 0")
    bq_case(2, 16384, 16384, 0.8, 32, "config 2, layer 0")
    bq_case(2, 16384, 4096, 0.8, 16, "config 2, layer 1")
    bq_case(2, 16384, 4096, 1.6, 32, "config 2, layer 1")
    group_case(2, 4, 16384, 16384, 32, "config 2, layer 0")
    group_case(2, 67, 16384, 4096, 32, "config 2, layer 1")
    three_nn_case(2, 16384, 4096, 128, "SURVEY 8a row a5 shape")


if __name__ == "__main__":
    main()

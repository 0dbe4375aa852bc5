"""Workloads of bench.py: the reference's training / inference iterations and the bare operator sequence, on
synthetic scenes, plus the live (HIP-event) roofline legs and the CPU-baseline legs of the bench line."""
import glob
import hashlib
import json
import os
import time

import numpy as np
import torch

from pdanet_amd.tuning import enable_tuned_gemms  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "detector_train"

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA (v_mfma_f32_32x32x2_f32) = the f32 vector peak
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), no sparsity
# What a kernel of nothing but v_mfma_f32_32x32x16_bf16 on register operands holds on random normal data, two waves per SIMD
# (tools/microbench/mfma_sustained.hip, profiles/r03_mfma_sustained.txt: 1813 TFLOP/s = a 1.73 GHz matrix clock; 2350 on
# zeros): the power-managed ceiling under `frac`'s nominal peak.  Reported beside `frac`, never instead of it.
BF16_MFMA_SUSTAINED_TF = 1813.0
SPLIT_PRODUCTS = 6          # csrc/split_bf16.h: an f32 product block = 6 bf16 MFMA blocks (3-term operands, 6 of 9 products)
# plain (non-packed) VALU lane-operations per second: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.  A ball-query
# distance test is 7 VALU instructions (3 sub, 1 mul, 2 fma, 1 compare), none of them packable without losing
# the reference's float expression, so the VALU bound is VALU_LANE_OPS / 7 tests per second.
VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9
BALL_QUERY_VALU_PER_TEST = 7

# ONCE PDA-SSD layer shapes at N_in = 16384:
#   (centres M, points N, [(radius, nsample)], feature channels C)
ONCE16K_LAYERS = [
    dict(name="L0", M=16384, N=16384, scales=[(0.2, 16), (0.8, 32)], C=1, fps=None),
    dict(name="L1", M=4096, N=16384, scales=[(0.8, 16), (1.6, 32)], C=64, fps=(16384, 4096)),
    dict(name="L2", M=2048, N=4096, scales=[(1.6, 16), (4.8, 32)], C=128, fps=None),
    dict(name="L5", M=1024, N=2048, scales=[(4.8, 16), (8.4, 32), (12.8, 64)], C=256, fps=None),
]


def _csrc_hash(*files):
    """sha1 of the kernel sources a committed counter pass belongs to: a PMC record whose hash differs from the
    sources in the tree is STALE (the kernel changed after the pass) and is not reported."""
    h = hashlib.sha1()
    for f in files:
        with open(os.path.join(ROOT, "pdanet_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_record(kind, key, sources, prefix=False, by=None):
    """Entry `key` of the newest profiles/r*_pmc/<kind>.json (kind = 'traffic' | 'mfma_util'), or None.  PMC
    counters cannot be read from inside the process (rocprofv3 passes, tools/pmc_passes.sh), so the bench line
    carries the committed record -- but only while the record's `csrc_sha1` still equals the hash of `sources`."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", kind + ".json"))):
        try:
            doc = json.load(open(f))
            sha = doc.get("csrc_sha1", {})
            if any(sha.get(s) != _csrc_hash(s) for s in sources):
                continue
            for name, k in doc["kernels"].items():
                if name == key or (prefix and name.startswith(key)):
                    if by is None or best is None or k[by] > best[by]:
                        best = k
        except (OSError, ValueError, KeyError):
            pass
    return best


def pmc_traffic(key, sources, prefix=False):
    k = pmc_record("traffic", key, sources, prefix)
    return None if k is None else k["hbm_bytes_per_launch_mean"]


def fps_algorithmic_bytes(n, m):
    return (m - 1) * n * 20 + m * 4  # SURVEY.md 8(d) / BASELINE.md section 2, per scene


def ball_query_algorithmic_bytes(n, m, nsamples):
    """SURVEY.md 8(d), per scene and per radius: ceil(M/256)*N*12 + M*12 + M*ns*4."""
    return sum(-(-m // 256) * n * 12 + m * 12 + m * ns * 4 for ns in nsamples)


class KernelTimers:
    """HIP-event timing of this repo's own kernels at their Python entry points, on the stream they are launched
    on (torch events record on the CURRENT stream, and the ops launch on the current stream -- for the D-FPS
    prefetch that is the side stream of backbone._presample).  Active only while `record` is set."""

    @property
    def record(self):
        return self._record

    @record.setter
    def record(self, on):
        from pdanet_amd import pointnet2_utils as pu
        self._record = bool(on)
        pu.SA_MFMA_EVENTS = self.sa_mfma_events if on else None     # the SA group-MLP launches time themselves

    def _install_timers(self):
        from pdanet_amd import pointnet2_utils as pu
        wl = self
        self.fps_events, self.bq_events, self.wgrad_events, self.sa_mfma_events = [], [], [], []
        self.record = False

        def timed(orig, sink, describe):
            def f(*a):
                if not wl.record:
                    return orig(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig(*a)
                e1.record()
                sink.append((e0, e1) + describe(*a))
                return out
            f._pda_orig = orig
            return f

        def unwrap(fn):
            return getattr(fn, "_pda_orig", fn)
        ext = pu.pointnet2
        ext.farthest_point_sampling_wrapper = timed(unwrap(ext.farthest_point_sampling_wrapper), self.fps_events,
                                                    lambda b, n, m, *r: (b, n, m))
        ext.ball_query_multi = timed(unwrap(ext.ball_query_multi), self.bq_events,
                                     lambda b, n, m, radii, ns, *r: (b, n, m, tuple(radii), tuple(ns), "scalar-stream brute force"))
        ext.ball_query_cells = timed(unwrap(ext.ball_query_cells), self.bq_events,
                                     lambda b, n, m, radii, ns, *r: (b, n, m, tuple(radii), tuple(ns), "cell list"))
        ext.linear_wgrad = timed(unwrap(ext.linear_wgrad), self.wgrad_events,
                                 lambda x, g, gw, gb, tokens, n_in, n_out: (tokens, n_in, n_out))
        ext.linear_wgrad_bn = timed(unwrap(ext.linear_wgrad_bn), self.wgrad_events,      # the same kernel, BatchNorm in its operand load
                                    lambda x, g, gw, tokens, n_in, n_out, *r: (tokens, n_in, n_out))

    @staticmethod
    def _mean_s(events):
        ts = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in events]
        return sum(ts) / len(ts)

    def _ball_query_alone_s(self, b, n, m, radii, nss):
        """Layer 0's query (centres = all points) on the current stream with nothing else on the device; None where the step's
        largest query is not of that form or the workload keeps no point tensor."""
        pts = getattr(self, "points", None)
        if m != n or pts is None or not torch.is_tensor(pts) or pts.shape[0] != b * n:
            return None
        from pdanet_amd import pointnet2_utils as pu
        xyz = pts[:, 1:4].reshape(b, n, 3).contiguous()
        rec, self.record = self.record, False
        try:
            torch.cuda.synchronize()
            evs = []
            for i in range(13):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pu.ball_query_multi(list(radii), list(nss), xyz, xyz)
                e1.record()
                if i >= 3:
                    evs.append((e0, e1))
            torch.cuda.synchronize()
        finally:
            self.record = rec
        return self._mean_s(evs)

    def roofline_fps(self):
        """The D-FPS launch of the step (dominant sampling kernel): SURVEY 8(d) bytes / mean launch duration."""
        if not self.fps_events:
            return None
        b, n, m = max((e[2:] for e in self.fps_events), key=lambda s: s[1] * s[2])
        t = self._mean_s([e[:2] for e in self.fps_events if e[2:] == (b, n, m)])
        alg = fps_algorithmic_bytes(n, m) * b
        ach = alg / t / 1e9
        kern = "fps_chain_kernel" if 2048 <= n <= 16384 else ("fps_chain_coop_kernel" if 24576 < n <= 65536 else "fps_reg_kernel")
        out = {"kernel": "%s (FPS %d->%d, %d scenes/launch)" % (kern, n, m, b), "bound": "hbm", "achieved": ach,
               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": min(1.0, ach / HBM_PEAK_GBS)}
        if ach > HBM_PEAK_GBS:
            # many scenes x several workgroups each: the algorithmic bytes (no on-chip reuse) per second exceed what HBM could
            # deliver -- the kernel never moves them.  `frac` is capped; the ratio is kept beside it.
            out["algorithmic_rate_over_hbm_peak"] = ach / HBM_PEAK_GBS
        out.update({
                "traffic": pmc_traffic("pda::%s FPS %d->%d b%d" % (kern, n, m, b), ["fps.hip"]),
                "avg_launch_ms": t * 1e3, "algorithmic_bytes_per_launch": alg,
                "compulsory_bytes_per_launch": (n * 16 + m * 4) * b,
                "note": "achieved = algorithmic bytes ((m-1)*N*20+m*4 per scene: the specified algorithm with no "
                        "on-chip reuse) / kernel time.  The kernel keeps the scene on chip (exact spatial pruning "
                        "skips no-op updates), so its real HBM traffic is the compulsory N*16+m*4 bytes per scene "
                        "and it is bound by the latency of the dependent arg-max rounds (fps_chain_kernel: 6.65 samples per synchronisation on the bench scene; "
                        "fps_chain_coop_kernel, N > 16384: ~12 per exchange between the K workgroups of a scene), not by HBM.  traffic = "
                        "(2*FETCH_SIZE+WRITE_SIZE) KB per launch from the committed rocprofv3 --pmc passes "
                        "(null when the kernel source changed since the pass)"})
        return out

    def roofline_ball_query(self):
        """The layer-0 ball-query launch (largest M*N of the step; both radii in one pass)."""
        if not self.bq_events:
            return None
        b, n, m, radii, nss, path = max((e[2:] for e in self.bq_events), key=lambda s: s[1] * s[2])
        t_step = self._mean_s([e[:2] for e in self.bq_events if e[2:] == (b, n, m, radii, nss, path)])
        # In the training / inference workloads this call is issued on the sampling side stream (beside the D-FPS chain, under
        # the previous iteration): the events above then time launches that wait for, and share the chip with, the main
        # stream's kernels.  The kernel's own rate is taken from the same call alone on the device, right here.
        t = self._ball_query_alone_s(b, n, m, radii, nss)
        shared = t is not None
        if t is None:
            t = t_step
        alg = ball_query_algorithmic_bytes(n, m, nss) * b
        ach = alg / t / 1e9
        tests = float(b) * n * m * len(radii)
        peak_tests = VALU_LANE_OPS / BALL_QUERY_VALU_PER_TEST
        return {"kernel": "ball query (%s) %dx%d, radii %s, nsample %s, %d scenes/call" % (path, m, n, list(radii), list(nss), b),
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic("pda::ball_query %dx%d r%d b%d" % (m, n, len(radii), b), ["ball_query.hip", "ball_query_cells.hip"]),
                "avg_launch_ms": t * 1e3, "avg_launch_ms_inside_the_step": t_step * 1e3,
                "timed": ("the same call alone on the device after the timed steps (10 calls, HIP events); inside the step it runs on "
                          "the side stream and shares the chip with the main stream's kernels" if shared else "inside the step"),
                "algorithmic_bytes_per_launch": alg,
                "distance_tests_per_launch": tests, "tests_per_s": tests / t,
                "valu_bound_tests_per_s": peak_tests, "valu_frac": tests / t / peak_tests,
                "note": "bytes per SURVEY 8(d) (xyz streamed once per 256 centres, per radius); tests = M*N per "
                        "radius = the brute-force count (the cell-list path performs far fewer, so its valu_frac is an "
                        "equivalent rate and may exceed 1); VALU bound = 256 CUs x 64 lanes x 2.4 GHz / 7 instructions per "
                        "test; the cell-list call is 5 launches (grid, count, scan, scatter, query) timed together"}

    def roofline_sa_mlp_train(self):
        """The vanilla-SA group MLPs in training form (forward and input-gradient contractions of ONCE layers 0 and 5; the
        weight gradients are csrc/wgrad.hip launches and appear in roofline_mfma).  The fraction is taken on the pipe the
        kernels really issue on: the split GEMMs run six v_mfma_f32_32x32x16_bf16 per f32 product block, so their MFMA work
        is 6 x the algorithmic flops, against the dense bf16 peak.  The gather-fused first layers (f32-input MFMA) are
        reported beside it against the f32 MFMA peak."""
        if not self.sa_mfma_events:
            return None
        steps = max(1, len(self.fps_events))

        def part(pipe):
            ev = [e for e in self.sa_mfma_events if e[3] == pipe]
            return (sum(e[0].elapsed_time(e[1]) for e in ev) * 1e-3, sum(e[2] for e in ev), len(ev))
        ts, fs, ns_ = part("bf16x6")
        tf, ff, nf = part("f32")
        f32_part = None if nf == 0 else {
            "kernel": "lin_cols_kernel / sa_gather_linear (v_mfma_f32_32x32x2_f32)", "launches_per_step": nf // steps,
            "achieved": ff / tf / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ff / tf / 1e12 / F32_MFMA_PEAK_TF,
            "ms_per_step": tf / steps * 1e3, "gflop_per_step": ff / steps / 1e9}
        tr, fr, nr = part("f32_recompute")
        small = None if nr == 0 else {
            "kernel": "ss_fwd_kernel / ss_bwd_kernel (csrc/sa_train_small.hip: SA layer 0 as recompute passes, v_mfma_f32_32x32x2_f32)",
            "calls_per_step": nr // steps, "achieved": fr / tr / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": fr / tr / 1e12 / F32_MFMA_PEAK_TF, "ms_per_step": tr / steps * 1e3, "gflop_per_step": fr / steps / 1e9,
            "note": "ALGORITHMIC flops (one forward and one backward evaluation of the chain per token) / time of all passes incl. "
                    "their finalize launches; the passes issue ~2.6x (forward) and ~3x (backward) that on the matrix cores -- "
                    "recomputing a token from its 16-byte input instead of moving 0.4-1 KB of activations per token through HBM"}
        if ns_ == 0:            # PDA_SPLIT_GEMM=0 / dense-bf16 mode: no split-bf16 launch among the SA group MLPs
            if f32_part is None:
                return None if small is None else dict(small, bound="mfma", traffic=None)
            util = pmc_record("mfma_util", "pda::lin_cols_kernel", ["sa_mlp.hip"], prefix=True, by="avg_ns")
            out = dict(f32_part, bound="mfma", traffic=None, mfma_busy_pmc=None if util is None else round(util["mfma_util"], 4),
                       layer0_recompute_part=small,
                       note="v_mfma_f32_32x32x2_f32; launches include the weight-packing kernel in front of each contraction")
            out["kernel"] += " (SA group MLP, training form: forward + input gradient)"
            return out
        util = (pmc_record("mfma_util", "pda::gemm_split_wide_kernel", ["gemm_split.hip"], prefix=True, by="avg_ns")
                or pmc_record("mfma_util", "pda::lin_split_kernel", ["gemm_split.hip"], prefix=True, by="avg_ns"))
        work = SPLIT_PRODUCTS * fs / ts / 1e12
        return {"kernel": "gemm_split_wide_kernel / lin_split_kernel (SA group MLP, training form: %d launches per step, "
                          "forward + input gradient)" % (ns_ // steps),
                "bound": "mfma", "achieved": work, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": work / BF16_MFMA_PEAK_TF,
                "sustained_mfma_only_kernel": BF16_MFMA_SUSTAINED_TF, "frac_of_sustained": work / BF16_MFMA_SUSTAINED_TF,
                "achieved_f32_equiv": fs / ts / 1e12, "peak_f32_input_mfma": F32_MFMA_PEAK_TF,
                "traffic": None, "ms_per_step": ts / steps * 1e3, "gflop_per_step": fs / steps / 1e9,
                "mfma_busy_pmc": None if util is None else round(util["mfma_util"], 4), "f32_mfma_part": f32_part,
                "layer0_recompute_part": small,
                "note": "f32 contractions on v_mfma_f32_32x32x16_bf16 with every operand split into three bf16 terms (x = h + m + l "
                        "exactly; 6 of 9 products kept, error at the f32 fmaf chain's: tests/test_gemm_split.py).  achieved = "
                        "6 x algorithmic flops / time = the bf16 MFMA work really issued, peak = dense bf16 MFMA; "
                        "achieved_f32_equiv = algorithmic flops / time, next to the f32-input MFMA peak the same contraction "
                        "would be priced against without the split.  Launch times include the weight-packing kernel.  "
                        "PDA_SPLIT_GEMM=0 restores lin_cols_kernel (f32 MFMA) everywhere"}

    def roofline_wgrad(self):
        """The weight-gradient kernel shape with the largest total time in the timed region: algorithmic flops
        2*T*in*out per launch (the dW GEMM; the fused bias gradient is not counted) / mean launch duration, on the pipe
        the kernel issues on (split form: 6 x the flops against the dense bf16 peak)."""
        by = {}
        for e in self.wgrad_events:
            by.setdefault(e[2:], []).append(e[0].elapsed_time(e[1]) * 1e-3)
        if not by:
            return None
        (t, ni, no), times = max(by.items(), key=lambda kv: sum(kv[1]))
        avg = sum(times) / len(times)
        flops = 2.0 * t * ni * no
        total = sum(sum(v) for v in by.values())
        steps = max(1, len(self.fps_events))          # one D-FPS launch per step
        from pdanet_amd import _lib
        lib = _lib.load()
        split = int(lib.pda_linear_wgrad_form(t, ni, no)) == 2
        kern = "wgrad_split_kernel" if split else "wgrad_kernel"
        key = "pda::linear_wgrad dW(%dx%d) over %d tokens" % (no, ni, t)
        util = pmc_record("mfma_util", "pda::" + kern, ["wgrad.hip"], prefix=True, by="launches")
        f32_equiv = flops / avg / 1e12
        # algorithmic bytes: both operands once + dW; the split-K partials the kernel writes and its second stage reads
        # again are listed beside it (scratch = S x (out x in + out) floats)
        alg_bytes = 4.0 * t * (ni + no) + 4.0 * ni * no
        part_bytes = 2.0 * int(lib.pda_linear_wgrad_scratch_bytes(t, ni, no))
        traffic = pmc_traffic(key, ["wgrad.hip"])
        out = {"kernel": "%s dW(%dx%d) over %d tokens" % (kern, no, ni, t), "bound": "mfma",
               "mfma_busy_pmc": None if util is None else round(util["mfma_util"], 4)}
        if split:
            work = SPLIT_PRODUCTS * f32_equiv
            out.update(achieved=work, peak=BF16_MFMA_PEAK_TF, unit="TFLOP/s", frac=work / BF16_MFMA_PEAK_TF,
                       sustained_mfma_only_kernel=BF16_MFMA_SUSTAINED_TF, frac_of_sustained=work / BF16_MFMA_SUSTAINED_TF,
                       achieved_f32_equiv=f32_equiv, peak_f32_input_mfma=F32_MFMA_PEAK_TF)
            note = ("three-term bf16 split of both operands, six v_mfma_f32_32x32x16_bf16 per f32 product block (f32 accuracy): "
                    "achieved = 6 x algorithmic flops / time = the bf16 MFMA work really issued, peak = dense bf16 MFMA; "
                    "achieved_f32_equiv = algorithmic flops / time.  On random data the chip holds ~1.8 GHz under this load "
                    "(mfma_util.json clock_ghz)")
        else:
            out.update(achieved=f32_equiv, peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", frac=f32_equiv / F32_MFMA_PEAK_TF)
            note = "v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate)"
        out.update(traffic=traffic, algorithmic_bytes_per_launch=alg_bytes, split_k_partial_bytes_per_launch=part_bytes,
                   traffic_over_algorithmic=None if traffic is None else traffic / alg_bytes, avg_launch_ms=avg * 1e3,
                   note=note + "; event-timed launch = split-K kernel + fixed-order second stage; all %d wgrad launches of a "
                        "step: %.2f ms" % (sum(len(v) for v in by.values()) // steps, total / steps * 1e3))
        return out


class SamplingGroupingWorkload(KernelTimers):
    """All sampling/grouping operator calls of one PDA-SSD backbone forward (ONCE-16k)."""

    name = "once16k_b2_sampling_grouping"

    def __init__(self, batch, n_points, device, rank):
        from pdanet_amd import pointnet2_utils as pu, synth
        self.pu = pu
        self.B, self.N = batch, n_points
        xyz = synth.batch_xyz(batch, n_points, config_id=2 + 10 * rank, dist="L")
        self.xyz_np = xyz
        self.xyz = torch.from_numpy(xyz).to(device)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.feats = {L["name"]: torch.randn(batch, L["C"], L["N"], generator=g).to(device)
                      for L in ONCE16K_LAYERS}
        self.dtype = "f32"
        self._install_timers()

    def begin(self):
        self._install_timers()

    def step(self):
        pu = self.pu
        xyz = self.xyz
        out = 0
        for L in ONCE16K_LAYERS:
            pts = xyz[:, :L["N"]].contiguous() if L["N"] != xyz.shape[1] else xyz
            if L["fps"] is not None:
                idx = pu.furthest_point_sample(pts, L["M"])
                new_xyz = pu.gather_operation(pts.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            else:
                new_xyz = pts[:, :L["M"]].contiguous()
            idxs = pu.ball_query_multi([r for r, _ in L["scales"]], [ns for _, ns in L["scales"]], pts, new_xyz)
            pts_t = pts.transpose(1, 2).contiguous()
            for idx_s in idxs:
                gx = pu.grouping_operation(pts_t, idx_s)
                gf = pu.grouping_operation(self.feats[L["name"]], idx_s)
                out = out + gx.numel() + gf.numel()
            xyz = new_xyz if L["name"] != "L5" else xyz
        return out

    def rooflines(self):
        return {"roofline": self.roofline_fps(), "roofline_ball_query": self.roofline_ball_query()}

    def cpu_baseline(self, budget_s=20.0):
        """The same operator sequence through the CPU oracle (kind 'port'), on scene 0, all cores."""
        return sampling_grouping_cpu(self.xyz_np, None)


def sampling_grouping_cpu(xyz_np, threads):
    """FPS + ball queries + groupings of one ONCE-16k scene (SURVEY 8(d) CPU baseline) through the oracle with
    `threads` OpenMP threads (None = all cores)."""
    import oracle
    ncores = oracle.num_threads()
    nthreads = ncores if threads is None else threads
    oracle.set_num_threads(nthreads)
    xyz = np.ascontiguousarray(xyz_np[:1])
    t0 = time.perf_counter()
    cur = xyz
    for L in ONCE16K_LAYERS:
        pts = np.ascontiguousarray(cur[:, :L["N"]])
        n = pts.shape[1]
        if L["fps"] is not None:
            temp = np.full((1, n), 1e10, np.float32)
            idx = np.zeros((1, L["M"]), np.int32)
            oracle.farthest_point_sampling_wrapper(1, n, L["M"], pts, temp, idx)
            new_xyz = np.ascontiguousarray(pts[0][idx[0]][None])
        else:
            new_xyz = np.ascontiguousarray(pts[:, :L["M"]])
        feats = np.zeros((1, L["C"], n), np.float32)
        pts_t = np.ascontiguousarray(pts.transpose(0, 2, 1))
        for r, ns in L["scales"]:
            bq = np.zeros((1, L["M"], ns), np.int32)
            oracle.ball_query_wrapper(1, n, L["M"], r, ns, new_xyz, pts, bq)
            gx = np.empty((1, 3, L["M"], ns), np.float32)
            oracle.group_points_wrapper(1, 3, n, L["M"], ns, pts_t, bq, gx)
            gf = np.empty((1, L["C"], L["M"], ns), np.float32)
            oracle.group_points_wrapper(1, L["C"], n, L["M"], ns, feats, bq, gf)
        cur = new_xyz if L["name"] != "L5" else cur
    dt = time.perf_counter() - t0
    oracle.set_num_threads(ncores)
    return dict(value=1.0 / dt, unit="scenes/s", cores=nthreads, kind="port",
                sample="sampling/grouping operators only (FPS 16384->4096, 9 ball queries, 18 groupings at the "
                       "ONCE-16k layer shapes) of 1 scene through oracle/libpda_oracle.so, %d OpenMP thread%s, "
                       "%.2f s" % (nthreads, "" if nthreads == 1 else "s", dt))


class BackboneWorkload(KernelTimers):
    """PDA-SSD backbone (IASSD_Backbone, ONCE yaml) forward + backward on synthetic ONCE scenes.

    One step = forward over `batch` scenes of `n_points` points in training mode (batch-stat
    BatchNorm, as the reference trains) + backward of a scalar loss that touches every backbone
    output the detection head consumes (centers_features, ctr_offsets, sa_ins_preds), so every
    parameter receives a gradient.  No optimiser step: the head/loss/optimiser are the "next"
    rows -- the metric here is the backbone fwd+bwd rate.
    With world > 1 the model is wrapped in DDP (RCCL gradient all-reduce over xGMI).
    """

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False, cfg="once_pda_ssd.yaml"):
        from pdanet_amd import synth
        from pdanet_amd.backbone import build_backbone
        self.B, self.N = batch, n_points
        self.name = "once16k_b%d_backbone_fwd_bwd" % batch if n_points == 16384 else \
            "once%d_b%d_backbone_fwd_bwd" % (n_points, batch)
        self.device = device
        self.cfg_name = cfg
        from pdanet_amd import pointnet2_utils as _pu
        self.dense_bf16 = bool(dense_bf16)
        _pu.DENSE_BF16 = self.dense_bf16        # DESIGN.md "Dense-bf16 mode" (not torch.autocast); see begin()
        self.dtype = "bf16 GEMMs (f32 accumulate) and GEMM-adjacent tensors; residual stream, statistics and kernel arithmetic f32" if dense_bf16 else "f32"
        self.tuned = enable_tuned_gemms() if os.environ.get("PDA_NO_TUNED_GEMMS") != "1" else False
        torch.manual_seed(1234)  # same initial weights on every rank
        model, self.cfg = build_backbone(cfg)
        self.model = model.to(device).train()
        from pdanet_amd import parallel
        self.ddp = parallel.wrap_ddp(self.model, device) if world > 1 else None
        self.points_np = synth.batch_points(batch, n_points, config_id=2 + 10 * rank, dist="L")
        self.points = torch.from_numpy(self.points_np).to(device)
        self._install_timers()

    def begin(self):
        """Make this workload's process-wide switches current (several workloads run one after another in one
        bench.py process: the headline and the `extra` timings)."""
        from pdanet_amd import pointnet2_utils as _pu
        _pu.DENSE_BF16 = self.dense_bf16
        self._install_timers()

    def rooflines(self):
        return {"roofline": self.roofline_fps(), "roofline_ball_query": self.roofline_ball_query(),
                "roofline_mfma": self.roofline_wgrad(), "roofline_mfma_sa_mlp": self.roofline_sa_mlp_train()}

    @staticmethod
    def loss_of(bd):
        loss = bd['centers_features'].float().pow(2).mean() + bd['ctr_offsets'][:, 1:].pow(2).mean()
        for p in bd['sa_ins_preds']:
            if not isinstance(p, list):
                loss = loss + p[..., 1:].float().pow(2).mean()
        return loss

    def step(self):
        model = self.ddp if self.ddp is not None else self.model
        for p in self.model.parameters():
            p.grad = None
        bd = model({'batch_size': self.B, 'points': self.points, 'inputs_resident': True})
        self._prefetch_next()
        loss = self.loss_of(bd)
        loss.backward()
        return loss.detach()

    def _prefetch_next(self):
        """The next iteration's batch is resident already (synthetic data; a data loader's prefetch in a real run): start its
        coordinate-only front -- D-FPS, layer-1 ball queries and unique-token plans -- on the side stream, under this
        iteration's backward.  Every iteration still does this work exactly once."""
        bb = self.model.backbone_3d if hasattr(self.model, "backbone_3d") else self.model
        if getattr(self.model, "graph_tail", False) and self.model.tail_start() == 1:
            return                       # every sampling layer is inside the replayed graph
        if self.PREFETCH:
            bb.prefetch(self.points, self.B)

    PREFETCH = os.environ.get("PDA_PREFETCH", "1") != "0"
    PREFETCH_EARLY = os.environ.get("PDA_PREFETCH_EARLY", "1") != "0"
    _primed = False

    def cpu_baseline(self, budget_s=30.0):
        """Same step on the host cores: this repo's model code with the operator extension
        replaced by the CPU oracle (kind 'port'), dense layers on torch CPU.  One scene pair is
        too slow for a default run, so the sample is ONE step over ONE scene."""
        import oracle
        from pdanet_amd import pointnet2_utils as pu
        from pdanet_amd.backbone import build_backbone

        class Stub:
            pass
        stub = Stub()
        for name in ["ball_query_wrapper", "ball_query_dilated_wrapper", "group_points_wrapper",
                     "group_points_grad_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper",
                     "farthest_point_sampling_wrapper", "furthest_point_sampling_with_dist_wrapper",
                     "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"]:
            def mk(fn):
                return lambda *a: fn(*[x.numpy() if isinstance(x, torch.Tensor) else x for x in a])
            setattr(stub, name, mk(getattr(oracle, name)))

        def bq_multi(b, n, m, radii, nsamples, new_xyz, xyz, idxs):
            for r, ns, idx in zip(radii, nsamples, idxs):
                oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz.numpy(), xyz.numpy(), idx.numpy())
            return 1
        stub.ball_query_multi = bq_multi
        saved, saved_cells = pu.pointnet2, pu.BALL_QUERY_CELLS
        pu.pointnet2, pu.BALL_QUERY_CELLS = stub, False      # the oracle has the reference's brute-force query only
        try:
            torch.manual_seed(1234)
            model, _ = build_backbone(self.cfg_name)
            model.train()
            pts = torch.from_numpy(self.points_np[: self.N].copy())
            nthreads = max(oracle.num_threads(), torch.get_num_threads())
            t0 = time.perf_counter()
            bd = model({'batch_size': 1, 'points': pts})
            self.loss_of(bd).backward()
            dt = time.perf_counter() - t0
        finally:
            pu.pointnet2, pu.BALL_QUERY_CELLS = saved, saved_cells
        return dict(value=1.0 / dt, unit="scenes/s", cores=nthreads, kind="port",
                    sample="1 step over 1 scene (scene 0 of the GPU batch, %d pts): this repo's backbone "
                           "with the extension replaced by oracle/libpda_oracle.so and dense layers on "
                           "torch CPU, fwd+bwd, %.1f s" % (self.N, dt))


class BackboneInferWorkload(BackboneWorkload):
    """BASELINE configs[1] literally: PDA-SSD backbone FORWARD (eval BatchNorm, no_grad) with
    the fused SA-scale kernel on layers 0 and 5."""

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False):
        super().__init__(batch, n_points, device, rank, world, dense_bf16=dense_bf16)
        from pdanet_amd import fused_ops
        self.fused_ops = fused_ops
        self.name = self.name.replace("fwd_bwd", "fwd_eval_fused")
        self.model.eval()
        fused_ops.enable_fused(self.model)
        # the layers behind the last token-count read as one hipGraph replay (PDA_GRAPH_TAIL_INFER=0: enqueued launch by launch)
        self.model.graph_tail_infer = os.environ.get("PDA_GRAPH_TAIL_INFER", "1") != "0"
        self.ddp = None
        self.sa_events = []
        self._primed = False

    def step(self):
        """One forward.  The NEXT batch is resident (synthetic data; a serving queue in a real run), so its coordinate-only
        front (D-FPS 16384 -> 4096: 2.6 ms on 2 CUs, layer-1 ball queries, token plans) is started on the side stream BEFORE
        this batch's forward and runs under all of it; every step still does that work exactly once."""
        self.fused_ops.PROFILE = self.sa_events if self.record else None
        with torch.no_grad():
            if not self._primed:
                self._prefetch_next()        # this batch's own front (first step only)
                self._primed = True
            self._prefetch_next()            # the next batch's front
            bd = self.model({'batch_size': self.B, 'points': self.points, 'inputs_resident': True})
        self.fused_ops.PROFILE = None
        return bd['centers_features']

    def rooflines(self):
        if self.model.graph_tail_infer and not any(e[3] == "bf16x6_infer" for e in self.sa_mfma_events):
            # layer 5 ran inside the replayed graph, where nothing times itself: three forwards launch by launch for the rooflines
            keep_rec, self.model.graph_tail_infer = self.record, False
            del self.sa_mfma_events[:]
            self.record = True
            try:
                for _ in range(3):
                    self.step()
                torch.cuda.synchronize()
            finally:
                self.model.graph_tail_infer, self.record = True, keep_rec
            self._roofline_forwards = 3
        return {"roofline": self.roofline_sa_wide_infer() or self.roofline_sa_mlp(), "roofline_fused_sa_kernel": self.roofline_sa_mlp(),
                "roofline_fps": self.roofline_fps(), "roofline_ball_query": self.roofline_ball_query()}

    def roofline_sa_wide_infer(self):
        """The wide SA scales (layer 5) in inference: per-point first layer + split-bf16 GEMMs for layers 2 and 3
        (pointnet2_utils.sa_wide_scale_infer).  achieved = 6 x the flops of layers 2 and 3 / the time of the whole scale
        (gather of the first layer and the max-pool included), against the dense bf16 peak."""
        ev = [e for e in self.sa_mfma_events if e[3] == "bf16x6_infer"]
        if not ev:
            return None
        t = sum(e[0].elapsed_time(e[1]) for e in ev) * 1e-3
        fl = sum(e[2] for e in ev)
        steps = getattr(self, "_roofline_forwards", None) or max(1, len(self.fps_events))
        work = SPLIT_PRODUCTS * fl / t / 1e12
        return {"kernel": "sa_wide_scale_infer: per-point projection, then gemm_split_wide_kernel with the first layer's row gather in its operand "
                          "load, then gemm_split_wide_kernel with the max over nsample in its epilogue; %d scales per step" % (len(ev) // steps),
                "bound": "mfma", "achieved": work, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": work / BF16_MFMA_PEAK_TF,
                "sustained_mfma_only_kernel": BF16_MFMA_SUSTAINED_TF, "frac_of_sustained": work / BF16_MFMA_SUSTAINED_TF,
                "achieved_f32_equiv": fl / t / 1e12, "peak_f32_input_mfma": F32_MFMA_PEAK_TF, "traffic": None,
                "ms_per_step": t / steps * 1e3, "gflop_per_step": fl / steps / 1e9,
                "note": "flops of layers 2 and 3 only (the first layer is a per-point projection of 0.5 GFLOP + a row gather); time of "
                        "the whole scale.  PDA_SA_WIDE_INFER_SPLIT=0 restores the fused f32-MFMA kernel (roofline_fused_sa_kernel)"}

    def roofline_sa_mlp(self):
        """Dominant fused kernel = the launch shape with the largest mean duration."""
        by = {}
        for e0, e1, flops, dims, ns in self.sa_events:
            by.setdefault((dims, ns), []).append((e0.elapsed_time(e1) * 1e-3, flops))
        if not by:
            return None
        stats = {k: (sum(t for t, _ in v) / len(v), v[0][1]) for k, v in by.items()}
        (dims, ns), (t, flops) = max(stats.items(), key=lambda kv: kv[1][0])
        l5 = {k: v for k, v in stats.items() if k[0][0] > 100}
        l5_t, l5_f = sum(v[0] for v in l5.values()), sum(v[1] for v in l5.values())
        peak = F32_MFMA_PEAK_TF
        util = pmc_record("mfma_util", "pda::sa_mlp_kernel", ["sa_mlp.hip"], prefix=True, by="avg_ns")
        return {"kernel": "sa_mlp_kernel %s ns=%d (%d scenes/launch)" % ("->".join(map(str, dims)), ns, self.B),
                "bound": "mfma", "achieved": flops / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / peak, "traffic": None, "avg_launch_ms": t * 1e3,
                "mfma_busy_pmc": None if util is None else round(util["mfma_util"], 4),
                "note": "f32-input MFMA (v_mfma_f32_32x32x2_f32); layer 5, all scales: %.1f GFLOP in %.3f ms = "
                        "%.1f TFLOP/s = %.1f %% of peak" % (l5_f / 1e9, l5_t * 1e3, l5_f / l5_t / 1e12,
                                                            100 * l5_f / l5_t / 1e12 / peak) if l5_t > 0 else ""}

    def cpu_baseline(self, budget_s=30.0):
        base = super().cpu_baseline(budget_s)
        base["sample"] += " (training-mode fwd+bwd step; the inference forward alone is ~1/3 of it)"
        return base


class TrainStepWorkload(BackboneWorkload):
    """Forward + backward + clip_grad_norm_(10) + adam_onecycle step (pdanet_amd/optimization.py,
    csrc/optim.hip) -- the iteration of tools/train_utils/train_utils.py:34-60 around the backbone.
    `kitti_train_bf16`: KITTI yaml, 4 scenes per GPU, dense-bf16 mode (pointnet2_utils.DENSE_BF16;
    operators, activations and statistics stay fp32)."""

    OPTIM = dict(OPTIMIZER="adam_onecycle", LR=0.01, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85], PCT_START=0.4,
                 DIV_FACTOR=10, GRAD_NORM_CLIP=10)   # once/kitti PDA-SSD.yaml OPTIMIZATION

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False, cfg="once_pda_ssd.yaml", dataset="once"):
        from pdanet_amd import optimization, parallel, synth
        super().__init__(batch, n_points, device, rank, 1, dense_bf16=dense_bf16, cfg=cfg)
        if dataset != "once":
            self.points_np = synth.batch_points(batch, n_points, config_id=3 + 10 * rank, dist="L", dataset=dataset)
            self.points = torch.from_numpy(self.points_np).to(device)
        self.name = "%s%dk_b%d_backbone_fwd_bwd_adam%s" % (dataset, n_points // 1024, batch, "_bf16" if dense_bf16 else "")
        self.opt = optimization.build_optimizer(self.model, self.OPTIM)
        self.sched = optimization.build_scheduler(self.opt, 1000, 80, self.OPTIM)
        self._data_parallel(world)
        self.it = 0

    def _data_parallel(self, world):
        """The gradient exchange of tools/train.py:153-154.  Default: one all-reduce of the optimizer's flat gradient buffer
        between backward and the clipping norm (optimization.FlatAdamOneCycle.data_parallel) -- no autograd hooks, so the
        N-rank step is the 1-rank step plus one collective.  PDA_DDP=1: the reference-shaped DistributedDataParallel wrapper
        (reducer hooks, bucket -> view copy per step; eager head, because hooks must not fire inside a captured region)."""
        from pdanet_amd import parallel
        self.ddp, self.exchange = None, None
        if world == 1:
            return
        if os.environ.get("PDA_DDP") == "1":
            self.model.graph_head = self.model.graph_tail = False
            self._tail_auto = False
            self.ddp = parallel.wrap_ddp(self.model, self.device, grads_are_views=True)
            self.exchange = "DistributedDataParallel (reducer hooks in backward, 25 MB buckets)"
        else:
            self.opt.data_parallel(model=self.model)
            self.exchange = "one all-reduce (%s) of the flat fp32 gradient buffer (%.1f MB) between backward and the clipping " \
                            "norm, outside autograd" % ("AVG" if self.opt._dp_avg else "SUM + scale", self.opt.flat_g.numel() * 4 / 1e6)

    def step(self):
        model = self.ddp if self.ddp is not None else self.model
        self.sched.step(self.it)
        self.opt.zero_grad(set_to_none=self.ddp is None)    # DDP copies its reduced buckets into the flat-buffer views
        bd = model({'batch_size': self.B, 'points': self.points, 'inputs_resident': True})
        self._prefetch_next()
        loss = self.loss_of(bd)
        loss.backward()
        self.opt.step()
        self.it += 1
        return loss.detach()


class DetectorTrainWorkload(TrainStepWorkload):
    """The whole training iteration of PDA-SSD on synthetic scenes + ground truth: IASSD detector
    (backbone + IASSD_Head: target assignment, all configured losses) forward, backward, gradient
    clipping and the adam_onecycle step -- tools/train_utils/train_utils.py:34-60 with model_func =
    model_fn_decorator (pcdet/models/__init__.py).  No host synchronisation inside the step."""

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False, cfg="once_pda_ssd.yaml", dataset="once"):
        from pdanet_amd import detector, optimization, parallel, synth
        BackboneWorkload.__init__(self, batch, n_points, device, rank, 1, dense_bf16=dense_bf16, cfg=cfg)
        self.points_np = synth.batch_points(batch, n_points, config_id=(2 if dataset == "once" else 3) + 10 * rank,
                                            dist="L", dataset=dataset)
        self.points = torch.from_numpy(self.points_np).to(device)
        self.gt = torch.from_numpy(synth.gt_boxes(self.points_np, batch, config_id=2 + 10 * rank, dataset=dataset)).to(device)
        torch.manual_seed(1234)
        model, self.cfg = detector.build_detector(cfg)
        self.model = model.to(device).train()
        self.name = "%s%dk_b%d_detector_fwd_bwd_adam%s" % (dataset, n_points // 1024, batch, "_bf16" if dense_bf16 else "")
        # head + losses replayed as hipGraphs (detector.IASSD.graph_head), at every world size: the gradient exchange is
        # outside autograd (_data_parallel), so nothing of it can land inside a captured region
        self.model.graph_head = os.environ.get("PDA_GRAPH_HEAD", "1") != "0"
        # graph_tail: layers 3-5 + head + losses as hipGraphs.  Measured: -3 ms on the host-bound dense-bf16 iteration
        # (18.4 -> 15.4 ms), +0.4 ms on the fp32 one where the device is the limit -- which depends on the box: the fp32
        # iteration takes 20.5 ms of device time and 18-23 ms of host time to enqueue (the pool's hosts differ), so the
        # fp32 workload decides on its fourth iteration: one synchronised measurement of enqueue time against elapsed
        # time, graph_tail on when the host is the limit (21.6 instead of 23.0 ms there).  PDA_GRAPH_TAIL=0|1 fixes it.
        mode = os.environ.get("PDA_GRAPH_TAIL", "1" if dense_bf16 else "auto")
        self.model.graph_tail = mode == "1"
        self._tail_auto = mode == "auto"
        self.opt = optimization.build_optimizer(self.model, self.cfg.OPTIMIZATION)
        self.sched = optimization.build_scheduler(self.opt, 1000, 80, self.cfg.OPTIMIZATION)
        self._data_parallel(world)
        self.it = 0

    def cpu_baseline(self, budget_s=30.0):
        """The WHOLE training iteration on the host cores (kind 'port'): this repo's detector (backbone + IA-SSD head with
        target assignment and all losses) with every extension call replaced by the CPU oracle (point ops, points_in_boxes),
        dense layers and the torch formulation of the losses on torch CPU, backward, clip_grad_norm_ and an Adam step with
        decoupled decay (what the reference's OptimWrapper does, fastai_optim.py:138-156).  One step over ONE scene."""
        import oracle
        from pdanet_amd import detector, pointnet2_utils as pu, roiaware_pool3d_utils as ru, synth

        class Stub:
            pass
        stub = Stub()
        for name in ["ball_query_wrapper", "ball_query_dilated_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
                     "gather_points_wrapper", "gather_points_grad_wrapper", "farthest_point_sampling_wrapper",
                     "furthest_point_sampling_with_dist_wrapper", "three_nn_wrapper", "three_interpolate_wrapper",
                     "three_interpolate_grad_wrapper"]:
            def mk(fn):
                return lambda *a: fn(*[x.numpy() if isinstance(x, torch.Tensor) else x for x in a])
            setattr(stub, name, mk(getattr(oracle, name)))

        def bq_multi(b, n, m, radii, nsamples, new_xyz, xyz, idxs):
            for r, ns, idx in zip(radii, nsamples, idxs):
                oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz.numpy(), xyz.numpy(), idx.numpy())
            return 1
        stub.ball_query_multi = bq_multi

        def pib(points, boxes):
            out = np.full(tuple(points.shape[:2]), -1, np.int32)
            oracle.points_in_boxes_gpu(boxes.contiguous().numpy(), points.contiguous().numpy(), out)
            return torch.from_numpy(out)
        saved = (pu.pointnet2, pu.BALL_QUERY_CELLS, ru.points_in_boxes_gpu)
        pu.pointnet2, pu.BALL_QUERY_CELLS, ru.points_in_boxes_gpu = stub, False, pib
        try:
            torch.manual_seed(1234)
            model, cfg = detector.build_detector(self.cfg_name)
            model.train()
            dataset = "kitti" if "kitti" in self.cfg_name else "once"
            pts = torch.from_numpy(self.points_np[: self.N].copy())
            gt = torch.from_numpy(synth.gt_boxes(self.points_np[: self.N], 1, config_id=2, dataset=dataset))
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.95, 0.99))
            nthreads = max(oracle.num_threads(), torch.get_num_threads())
            t0 = time.perf_counter()
            ret, _, _ = model({'batch_size': 1, 'points': pts, 'gt_boxes': gt})
            ret['loss'].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1 - 0.01 * 1e-3)
            opt.step()
            dt = time.perf_counter() - t0
        finally:
            pu.pointnet2, pu.BALL_QUERY_CELLS, ru.points_in_boxes_gpu = saved
        return dict(value=1.0 / dt, unit="scenes/s", cores=nthreads, kind="port",
                    sample="1 whole training iteration over 1 scene (scene 0 of the GPU batch, %d pts): this repo's detector with "
                           "every extension call replaced by oracle/libpda_oracle.so (point ops, points_in_boxes), dense layers "
                           "and losses on torch CPU, forward + backward + gradient clipping + Adam step, %.1f s" % (self.N, dt))

    def step(self):
        probe = self._tail_auto and self.it == 3
        if probe:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model = self.ddp if self.ddp is not None else self.model
        self.sched.step(self.it)
        self.opt.zero_grad(set_to_none=self.ddp is None)    # DDP copies its reduced buckets into the flat-buffer views
        if self.PREFETCH_EARLY:
            # the NEXT batch's coordinate-only front goes to the side stream before this iteration's forward (two-deep, like the
            # inference workload): a whole iteration to hide under -- the 15 ms of D-FPS on 60 000-point scenes do not fit
            # under a backward pass alone.  Every iteration still does that work exactly once.
            if not self._primed:
                self._prefetch_next()
                self._primed = True
            self._prefetch_next()
        ret, tb, _ = model({'batch_size': self.B, 'points': self.points, 'gt_boxes': self.gt, 'inputs_resident': True})
        if not self.PREFETCH_EARLY:
            self._prefetch_next()
        ret['loss'].backward()
        self.opt.step()
        self.it += 1
        self.tb = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in tb.items()}
        if probe:
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            self.host_bound = (t1 - t0) > 0.93 * (t2 - t0)      # the GPU was done (almost) as soon as the host was
            if os.environ.get("PDA_PROBE_FORCE") in ("0", "1"):   # test hook: the probe's verdict, whatever the box
                self.host_bound = os.environ["PDA_PROBE_FORCE"] == "1"
            self.model.graph_tail = self.host_bound
            self._tail_auto = False
        # DETACHED: a caller that keeps the returned loss (`loss = wl.step()` in a loop) must not keep the iteration's autograd
        # graph alive with it.  With the graph of an earlier iteration alive, the capture of the tail graph on the next
        # iteration (graph_tail chosen by the probe above) dies inside hipStreamEndCapture (ROCm 7.2; torch/cuda/graphs.py
        # capture_end; reproduced with tools/experiments/soak4.py MODE=keep, gone with the reference dropped).
        return ret['loss'].detach()


def create(name, batch, n_points, device, rank, world):
    if name == "sampling_grouping":
        return SamplingGroupingWorkload(batch, n_points, device, rank)
    if name == "detector_train":
        return DetectorTrainWorkload(batch, n_points, device, rank, world)
    if name == "kitti_detector_train_bf16":
        return DetectorTrainWorkload(batch, n_points, device, rank, world, dense_bf16=True, cfg="kitti_pda_ssd.yaml",
                                     dataset="kitti")
    if name == "kitti_detector_train":       # the same iteration in the default f32 mode (unique-token encoder, split-bf16 GEMMs)
        return DetectorTrainWorkload(batch, n_points, device, rank, world, dense_bf16=False, cfg="kitti_pda_ssd.yaml", dataset="kitti")
    if name == "train_step":
        return TrainStepWorkload(batch, n_points, device, rank, world)
    if name == "kitti_train_bf16":
        return TrainStepWorkload(batch, n_points, device, rank, world, dense_bf16=True, cfg="kitti_pda_ssd.yaml",
                                 dataset="kitti")
    if name == "backbone_infer":
        return BackboneInferWorkload(batch, n_points, device, rank, world)
    if name == "backbone_infer_bf16":
        return BackboneInferWorkload(batch, n_points, device, rank, world, dense_bf16=True)
    if name == "backbone":
        return BackboneWorkload(batch, n_points, device, rank, world, dense_bf16=False)
    if name == "backbone_bf16":
        return BackboneWorkload(batch, n_points, device, rank, world, dense_bf16=True)
    raise ValueError("unknown workload %r" % name)

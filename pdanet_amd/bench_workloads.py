"""Workloads for bench.py beyond the bare operator sequence."""
import os
import time

import numpy as np
import torch

DEFAULT = "backbone"

TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_mi355x_once16k_b2.csv")


def enable_tuned_gemms(path=TUNING_FILE):
    """The dense layers run on hipBLASLt through torch; its default heuristics pick poor kernels for
    the backward GEMMs of this model (tall-skinny weight gradients with K = 65k-262k tokens).
    PyTorch's TunableOp benchmarks every GEMM shape once and records the fastest solution; the
    recorded choices for the ONCE-16k / batch-2 step (12 min of tuning on one MI355X,
    `tools/tune_gemms.sh`) are committed and only LOADED here (tuning stays off, unknown shapes use the
    default heuristic).  Measured: 78.2 -> 58.3 ms per training step."""
    try:
        import torch.cuda.tunable as tn
        if not os.path.exists(path):
            return False
        tn.enable(True)
        tn.tuning_enable(False)
        tn.record_untuned_enable(False) if hasattr(tn, "record_untuned_enable") else None
        return bool(tn.read_file(path))
    except Exception:  # noqa: BLE001  (TunableOp is an optimisation, never a requirement)
        return False


class BackboneWorkload:
    """PDA-SSD backbone (IASSD_Backbone, ONCE yaml) forward + backward on synthetic ONCE scenes.

    One step = forward over `batch` scenes of `n_points` points in training mode (batch-stat
    BatchNorm, as the reference trains) + backward of a scalar loss that touches every backbone
    output the detection head consumes (centers_features, ctr_offsets, sa_ins_preds), so every
    parameter receives a gradient.  No optimiser step: the head/loss/optimiser are the "next"
    rows -- the metric here is the backbone fwd+bwd rate.
    With world > 1 the model is wrapped in DDP (RCCL gradient all-reduce over xGMI).
    """

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False, cfg="once_pda_ssd.yaml"):
        from . import synth
        from .backbone import build_backbone
        self.B, self.N = batch, n_points
        self.name = "once16k_b%d_backbone_fwd_bwd" % batch if n_points == 16384 else \
            "once%d_b%d_backbone_fwd_bwd" % (n_points, batch)
        self.device = device
        self.cfg_name = cfg
        from . import pointnet2_utils as _pu
        _pu.DENSE_BF16 = bool(dense_bf16)       # DESIGN.md "Dense-bf16 mode" (not torch.autocast)
        self.dtype = "bf16 GEMMs (f32 accumulate) and GEMM-adjacent tensors; residual stream, statistics and kernel arithmetic f32" if dense_bf16 else "f32"
        self.tuned = enable_tuned_gemms() if os.environ.get("PDA_NO_TUNED_GEMMS") != "1" else False
        torch.manual_seed(1234)  # same initial weights on every rank
        model, self.cfg = build_backbone(cfg)
        self.model = model.to(device).train()
        from . import parallel
        self.ddp = parallel.wrap_ddp(self.model, device) if world > 1 else None
        self.points_np = synth.batch_points(batch, n_points, config_id=2 + 10 * rank, dist="L")
        self.points = torch.from_numpy(self.points_np).to(device)
        self.fps_events = []
        self.record = False
        self._hook_fps()

    def _hook_fps(self):
        # time the D-FPS launches with events on the launch stream (roofline leg of bench.py)
        from . import pointnet2_utils as pu
        orig = pu.FarthestPointSampling.apply
        wl = self

        def timed(xyz, npoint):
            if not wl.record:
                return orig(xyz, npoint)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(xyz, npoint)
            e1.record()
            wl.fps_events.append((e0, e1, xyz.shape[1], npoint))
            return out
        pu.furthest_point_sample = pu.farthest_point_sample = timed
        # ... and the weight-gradient kernel launches (second roofline object: MFMA-bound own kernel)
        self.wgrad_events = []
        orig_wg = pu.pointnet2.linear_wgrad

        def timed_wg(x, grad_out, grad_weight, grad_bias, tokens, n_in, n_out):
            if not wl.record:
                return orig_wg(x, grad_out, grad_weight, grad_bias, tokens, n_in, n_out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_wg(x, grad_out, grad_weight, grad_bias, tokens, n_in, n_out)
            e1.record()
            wl.wgrad_events.append((e0, e1, tokens, n_in, n_out))
            return out
        if not getattr(orig_wg, "_pda_timed", False):
            timed_wg._pda_timed = True
            pu.pointnet2.linear_wgrad = timed_wg

    @staticmethod
    def _pmc_traffic(key):
        import glob
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc", "traffic.json"))):
            try:
                k = json.load(open(f))["kernels"].get(key)
                if k:
                    return k["hbm_bytes_per_launch_mean"]
            except (OSError, ValueError, KeyError):
                pass
        return None

    @staticmethod
    def _pmc_mfma_util(prefix, by="launches"):
        """MFMA-pipe busy fraction of a kernel from the committed counter pass (profiles/r*_pmc/mfma_util.json): the
        launch shape with the most launches (or the longest one, by="avg_ns"); None when no pass is committed."""
        import glob
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        best = None
        for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc", "mfma_util.json"))):
            try:
                for name, k in json.load(open(f))["kernels"].items():
                    if name.startswith(prefix) and (best is None or k[by] > best[by]):
                        best = k
            except (OSError, ValueError, KeyError):
                pass
        return None if best is None else round(best["mfma_util"], 4)

    def roofline_mfma(self):
        """The weight-gradient kernel shape with the largest total time in the timed region: algorithmic flops
        2*T*in*out per launch (the dW GEMM; the fused bias gradient is not counted) / mean launch duration."""
        by = {}
        for e0, e1, t, ni, no in self.wgrad_events:
            by.setdefault((t, ni, no), []).append(e0.elapsed_time(e1) * 1e-3)
        if not by:
            return None
        (t, ni, no), times = max(by.items(), key=lambda kv: sum(kv[1]))
        avg = sum(times) / len(times)
        flops = 2.0 * t * ni * no
        peak = 157.3
        total = sum(sum(v) for v in by.values())
        steps = max(1, len(self.fps_events))          # one D-FPS launch per step
        pmc = self._pmc_mfma_util("pda::wgrad_kernel")
        return {"kernel": "wgrad_kernel dW(%dx%d) over %d tokens" % (no, ni, t), "bound": "mfma", "mfma_busy_pmc": pmc,
                "achieved": flops / avg / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / avg / 1e12 / peak,
                "traffic": self._pmc_traffic("pda::wgrad_kernel dW(%dx%d) over %d tokens" % (no, ni, t)), "avg_launch_ms": avg * 1e3,
                "note": "v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate); traffic = committed PMC passes (profiles/r01_pmc);  event-timed launch = split-K kernel + fixed-order "
                        "second stage; all %d wgrad launches of a step: %.2f ms" % (
                            sum(len(v) for v in by.values()) // steps, total / steps * 1e3)}

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
        loss = self.loss_of(bd)
        loss.backward()
        return loss

    def fps_shape(self):
        if self.fps_events:
            return self.fps_events[0][2], self.fps_events[0][3]
        return self.N, 4096

    def cpu_baseline(self, budget_s=30.0):
        """Same step on the host cores: this repo's model code with the operator extension
        replaced by the CPU oracle (kind 'port'), dense layers on torch CPU.  One scene pair is
        too slow for a default run, so the sample is ONE step over ONE scene."""
        import oracle
        from . import pointnet2_utils as pu
        from .backbone import build_backbone

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
        saved = pu.pointnet2
        pu.pointnet2 = stub
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
            pu.pointnet2 = saved
        return dict(value=1.0 / dt, unit="scenes/s", cores=nthreads, kind="port",
                    sample="1 step over 1 scene (scene 0 of the GPU batch, %d pts): this repo's backbone "
                           "with the extension replaced by oracle/libpda_oracle.so and dense layers on "
                           "torch CPU, fwd+bwd, %.1f s" % (self.N, dt))


class BackboneInferWorkload(BackboneWorkload):
    """BASELINE configs[1] literally: PDA-SSD backbone FORWARD (eval BatchNorm, no_grad) with
    the fused SA-scale kernel on layers 0 and 5."""

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False):
        super().__init__(batch, n_points, device, rank, world, dense_bf16=dense_bf16)
        from . import fused_ops
        self.fused_ops = fused_ops
        self.name = self.name.replace("fwd_bwd", "fwd_eval_fused")
        self.model.eval()
        fused_ops.enable_fused(self.model)
        self.ddp = None
        self.sa_events = []

    def step(self):
        self.fused_ops.PROFILE = self.sa_events if self.record else None
        with torch.no_grad():
            bd = self.model({'batch_size': self.B, 'points': self.points, 'inputs_resident': True})
        self.fused_ops.PROFILE = None
        return bd['centers_features']

    def roofline(self):
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
        peak = 157.3  # TFLOP/s f32 matrix, MI355X_MICROARCH.md
        return {"kernel": "sa_mlp_kernel %s ns=%d (%d scenes/launch)" % ("->".join(map(str, dims)), ns, self.B),
                "bound": "mfma", "achieved": flops / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / peak, "traffic": None, "avg_launch_ms": t * 1e3,
                "mfma_busy_pmc": self._pmc_mfma_util("pda::sa_mlp_kernel", by="avg_ns"),
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
        from . import optimization, parallel, synth
        super().__init__(batch, n_points, device, rank, 1, dense_bf16=dense_bf16, cfg=cfg)
        if dataset != "once":
            self.points_np = synth.batch_points(batch, n_points, config_id=3 + 10 * rank, dist="L", dataset=dataset)
            self.points = torch.from_numpy(self.points_np).to(device)
        self.name = "%s%dk_b%d_backbone_fwd_bwd_adam%s" % (dataset, n_points // 1024, batch, "_bf16" if dense_bf16 else "")
        self.opt = optimization.build_optimizer(self.model, self.OPTIM)     # flat buffers BEFORE DDP
        self.sched = optimization.build_scheduler(self.opt, 1000, 80, self.OPTIM)
        self.ddp = parallel.wrap_ddp(self.model, device, grads_are_views=True) if world > 1 else None
        self.it = 0

    def step(self):
        model = self.ddp if self.ddp is not None else self.model
        self.sched.step(self.it)
        self.opt.zero_grad(set_to_none=self.ddp is None)    # DDP copies its reduced buckets into the flat-buffer views
        bd = model({'batch_size': self.B, 'points': self.points, 'inputs_resident': True})
        loss = self.loss_of(bd)
        loss.backward()
        self.opt.step()
        self.it += 1
        return loss


class DetectorTrainWorkload(TrainStepWorkload):
    """The whole training iteration of PDA-SSD on synthetic scenes + ground truth: IASSD detector
    (backbone + IASSD_Head: target assignment, all configured losses) forward, backward, gradient
    clipping and the adam_onecycle step -- tools/train_utils/train_utils.py:34-60 with model_func =
    model_fn_decorator (pcdet/models/__init__.py).  No host synchronisation inside the step."""

    def __init__(self, batch, n_points, device, rank, world, dense_bf16=False, cfg="once_pda_ssd.yaml", dataset="once"):
        from . import detector, optimization, parallel, synth
        BackboneWorkload.__init__(self, batch, n_points, device, rank, 1, dense_bf16=dense_bf16, cfg=cfg)
        self.points_np = synth.batch_points(batch, n_points, config_id=(2 if dataset == "once" else 3) + 10 * rank,
                                            dist="L", dataset=dataset)
        self.points = torch.from_numpy(self.points_np).to(device)
        self.gt = torch.from_numpy(synth.gt_boxes(self.points_np, batch, config_id=2 + 10 * rank, dataset=dataset)).to(device)
        torch.manual_seed(1234)
        model, self.cfg = detector.build_detector(cfg)
        self.model = model.to(device).train()
        self.name = "%s%dk_b%d_detector_fwd_bwd_adam%s" % (dataset, n_points // 1024, batch, "_bf16" if dense_bf16 else "")
        self.opt = optimization.build_optimizer(self.model, self.cfg.OPTIMIZATION)
        self.sched = optimization.build_scheduler(self.opt, 1000, 80, self.cfg.OPTIMIZATION)
        self.ddp = parallel.wrap_ddp(self.model, device, grads_are_views=True) if world > 1 else None
        self.it = 0

    def step(self):
        model = self.ddp if self.ddp is not None else self.model
        self.sched.step(self.it)
        self.opt.zero_grad(set_to_none=self.ddp is None)    # DDP copies its reduced buckets into the flat-buffer views
        ret, tb, _ = model({'batch_size': self.B, 'points': self.points, 'gt_boxes': self.gt, 'inputs_resident': True})
        ret['loss'].backward()
        self.opt.step()
        self.it += 1
        self.tb = tb
        return ret['loss']


def create(name, batch, n_points, device, rank, world):
    if name == "detector_train":
        return DetectorTrainWorkload(batch, n_points, device, rank, world)
    if name == "kitti_detector_train_bf16":
        return DetectorTrainWorkload(batch, n_points, device, rank, world, dense_bf16=True, cfg="kitti_pda_ssd.yaml",
                                     dataset="kitti")
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

#!/usr/bin/env python
"""bench.py -- scenes/sec of the PDA-SSD hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W           (N > 1: spawns its own N ranks, see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W           (the driver's form: RANK/LOCAL_RANK/WORLD_SIZE from env)

A step = one pass of the hot path over one per-GPU batch of synthetic scenes (ONCE, 16384 points, batch 2 per
GPU as in the ONCE yaml; weak scaling: every rank processes its own scenes, the only exchange step is the
gradient all-reduce over RCCL -- one collective on the optimizer's flat gradient buffer, outside autograd).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Launch contract (reference: tools/scripts/torch_train.sh:17, dist_train.sh:18 start one process per GPU):
`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this file under torch.distributed.run with
N ranks BEFORE anything touches the GPU (a child process, never an exec) and exits with its code; under a
launcher, WORLD_SIZE must equal --gpus or the run stops with exit code 2 -- a single-rank run is never reported
as an N-GPU result.

Workloads (--workload; default detector_train):
  detector_train     backbone + IA-SSD head (target assignment, all losses) + backward + grad clip + adam_onecycle:
                     the reference's whole training iteration (train_utils.py:34-64) on synthetic scenes and boxes.
  backbone           backbone forward+backward with a synthetic loss (round 1's headline; now under `extra`).
  backbone_infer     backbone forward, eval BN, fused SA kernel (BASELINE configs[1]).
  kitti_detector_train_bf16   detector_train on the KITTI yaml in dense-bf16 mode (BASELINE configs[2]; --batch 4).
  kitti_detector_train        the same iteration in the default mode (f32 tensors, split-bf16 contractions, unique tokens).
  train_step, kitti_train_bf16, backbone_bf16, backbone_infer_bf16   see benchmarks/workloads.py.
  sampling_grouping  only the sampling/grouping operators at the ONCE-16k layer shapes.

The default run (N = 1) also times `backbone`, `backbone_infer`, `kitti_detector_train_bf16 --batch 4` and
`kitti_detector_train --batch 4` for a few steps each and reports them under "extra" (--no-extra skips that).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default: the yaml's 2 for ONCE, 4 for KITTI)")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary timings under 'extra'")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N ranks of this file under torch.distributed.run as a CHILD process.
    Nothing in this process has touched the GPU yet (torch is not even imported)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("PYTHONFAULTHANDLER", "1")     # a rank that dies on a signal leaves its Python stack on stderr
    return subprocess.call(cmd, env=env)


def time_workload(wl, steps, warmup, device, parallel):
    """W (+2 priming) untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over
    ranks.  The two priming steps keep first-use costs (code-object loading of every library GEMM, allocator
    growth, TunableOp table lookups) out of a run with a small W."""
    trace = [] if os.environ.get("PDA_BENCH_STEP_TIMES") else None     # host-side enqueue time of every step -> stderr
    if trace is not None:
        import gc
        gcs = []

        def on_gc(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gcs.append((len(trace), info["generation"], 1e3 * (time.perf_counter() - _t[0]), info["collected"]))
        gc.callbacks.append(on_gc)
    wl.begin()
    for _ in range(2 + warmup):
        ts = time.perf_counter()
        wl.step()
        if trace is not None:
            trace.append(time.perf_counter() - ts)
    # What is alive after the warm-up (modules, captured graphs, cached plans and packed weights: ~10^6 objects by the fourth
    # workload of a run) is taken out of the collector's generations, as a long-running training loop does after start-up:
    # a generation-2 pass that walks all of it costs the host 82 ms (measured with gc.callbacks, PDA_BENCH_STEP_TIMES=1), once per
    # ~10 iterations of the host-bound KITTI workload -- 6 ms per step over a 10-step window.  The collector stays on for what
    # the steps themselves allocate.  PDA_GC_FREEZE=0: leave it alone.
    import gc
    frozen = os.environ.get("PDA_GC_FREEZE", "1") != "0"
    if frozen:
        gc.collect()
        gc.freeze()
    parallel.barrier(device)
    wl.record = True
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        wl.step()
        if trace is not None:
            trace.append(time.perf_counter() - ts)
    parallel.barrier(device)
    dt = time.perf_counter() - t0
    wl.record = False
    if frozen:
        gc.unfreeze()
    if trace is not None:
        gc.callbacks.remove(on_gc)
        print("bench.py: %s collections (step, generation, ms, objects): %s" % (wl.name, [g for g in gcs if g[2] > 1.0]), file=sys.stderr)
        print("bench.py: %s host ms per step (warm-up %d): %s" % (wl.name, 2 + warmup, " ".join("%.1f" % (1e3 * t) for t in trace)),
              file=sys.stderr)
    return parallel.max_over_ranks(dt, device)


def safe(fn, default, *a):
    """The timing is the measurement; a failure in a secondary leg (roofline bookkeeping, CPU baseline) must not cost the line.
    The error text is kept in place of the value."""
    try:
        return fn(*a)
    except Exception as e:  # noqa: BLE001
        print("bench.py: %s failed: %r" % (getattr(fn, "__name__", fn), e), file=sys.stderr)
        return dict(default, error=repr(e)) if isinstance(default, dict) else default


def metric_name(wl_name, points):
    ds = "KITTI" if wl_name.startswith("kitti") else "ONCE"
    if "fwd_bwd" in wl_name:
        what = "forward+backward"
    elif "fwd_eval" in wl_name:
        what = "forward"
    else:
        what = "sampling/grouping ops"
    return "scenes/sec (%d-pt %s, PDA-SSD %s)" % (points, ds, what)


STEP_TEXT = {
    "detector": "IASSD detector (backbone + IA-SSD head with target assignment and all losses) forward, backward, "
                "gradient clipping, adam_onecycle step = the reference's training iteration (train_utils.py:34-64)",
    "backbone_fwd_bwd_adam": "backbone forward + synthetic loss + backward + gradient clipping + adam_onecycle step",
    "backbone_fwd_bwd": "backbone forward + synthetic L2 loss on every head input + backward",
    "backbone_fwd_eval": "backbone forward, eval BatchNorm, fused SA kernel, no_grad",
    "sampling_grouping": "sampling/grouping operators only",
}


def step_text(name):
    for k, v in STEP_TEXT.items():
        if k in name:
            return v
    return name


def pin_to_gpu_node(torch, device_index):
    """Keep this rank's threads on the CPUs of the NUMA node its GPU hangs off (what `numactl --cpunodebind` does for a launcher;
    the driver starts `python bench.py` bare).  A host-bound iteration is 10-15 % slower when the launching thread sits on the
    other socket, and the scheduler moves it between the two from run to run (KITTI, B = 4: 12.3-12.8 against 14.1-14.5 ms per
    step, tools/experiments/kitti_dist.py).  PDA_PIN_CPUS=0 leaves the affinity alone.  Returns a description for the result line."""
    if os.environ.get("PDA_PIN_CPUS", "1") == "0":
        return "unchanged (PDA_PIN_CPUS=0)"
    global ALL_CPUS
    ALL_CPUS = os.sched_getaffinity(0)
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        cpus = set()
        for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "unchanged (no local CPU in the allowed set)"
        for tid in os.listdir("/proc/self/task"):          # the threads torch has started already; later ones inherit
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        return "GPU-local NUMA node of %s: %d CPUs" % (bdf, len(cpus))
    except Exception as e:  # noqa: BLE001  (an unreadable sysfs must not cost the run)
        return "unchanged (%r)" % (e,)


ALL_CPUS = None


def unpin():
    """The CPU baselines use every core of the box (OpenMP in the oracle, torch's intra-op pool): give the threads the whole
    machine back for them (pinned to one socket, 256 OpenMP threads on 128 CPUs took 4 x as long)."""
    if ALL_CPUS:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), ALL_CPUS)
            except OSError:
                pass


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    env_world = int(os.environ["WORLD_SIZE"]) if "WORLD_SIZE" in os.environ else None
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args, argv))
    if env_world is not None and env_world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a mislabelled "
              "run" % (args.gpus, env_world), file=sys.stderr)
        sys.exit(2)

    if env_world is not None and env_world > 1:
        # graph capture (head / tail replayed as hipGraphs) next to a live process group: the watchdog's asynchronous error
        # handling polls collective events from another thread (torch's CUDA-graphs notes ask for this with captures)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "0")
    if os.environ.get("PDA_DUMP_STACKS_AFTER"):      # debugging aid: every thread's Python stack after N seconds (a hung rank)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PDA_DUMP_STACKS_AFTER"]), exit=False)
    import torch
    import torch.distributed as dist
    from benchmarks import workloads
    from pdanet_amd import parallel

    rank, local_rank, world = parallel.env_world()
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    rehearse = os.environ.get("PDA_REHEARSE_ONE_GPU") == "1"   # several ranks on one card (with PDA_DIST_BACKEND=gloo)
    if rehearse:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        print("bench.py: %d ranks but only %d GPUs visible" % (world, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cpu_affinity = pin_to_gpu_node(torch, local_rank)
    parallel.init_distributed("nccl", device)  # backend "nccl" is RCCL on ROCm; no-op for 1 GPU
    seen_world = dist.get_world_size() if dist.is_initialized() else 1
    assert seen_world == args.gpus, (seen_world, args.gpus)
    backend = dist.get_backend() if dist.is_initialized() else None

    name = workloads.DEFAULT if args.workload == "auto" else args.workload
    batch = args.batch if args.batch is not None else (4 if name.startswith("kitti") else 2)
    wl = workloads.create(name, batch, args.points, device, rank, world)
    if args.warmup < 5 and getattr(wl, "_tail_auto", False):
        wl._tail_auto = False        # the probe (iteration 4) and the capture behind it (iteration 5) must stay out of the timed steps
    dt = time_workload(wl, args.steps, args.warmup, device, parallel)
    roofs = safe(wl.rooflines, {})

    line = {
        "metric": metric_name(wl.name, args.points),
        "value": batch * world * args.steps / dt,
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": getattr(wl, "dtype", "f32"),
        "data": "synthetic",
        "config": {"workload": wl.name, "step": step_text(wl.name), "scenes_per_gpu": batch,
                   "global_batch": batch * world, "points_per_scene": args.points, "parallelism": "dp%d" % world,
                   "world_size_seen": seen_world, "dist_backend": backend,
                   "gradient_exchange": getattr(wl, "exchange", None) or (
                       "DDP all-reduce over %s" % backend if getattr(wl, "ddp", None) is not None else None)},
    }
    line.update({k: v for k, v in roofs.items() if v is not None})
    line.setdefault("roofline", None)
    if hasattr(wl, "tuned"):
        line["config"]["hipblaslt_tunableop_results_loaded"] = bool(wl.tuned)
    line["config"]["peak_mem_GiB"] = round(torch.cuda.max_memory_allocated(device) / 2**30, 2)
    line["config"]["cpu_affinity"] = cpu_affinity
    if hasattr(getattr(wl, "model", None), "graph_tail"):
        # layers 3-5 + head + losses replayed as hipGraphs: chosen by the workload when the host was the limit (workloads.py)
        line["config"]["graph_tail"] = bool(wl.model.graph_tail)
        line["config"]["graph_head"] = bool(getattr(wl.model, "graph_head", False))
        if hasattr(wl, "host_bound"):
            line["config"]["host_bound_at_probe"] = bool(wl.host_bound)
    if hasattr(getattr(wl, "model", None), "graph_tail_infer"):
        line["config"]["graph_tail_infer"] = bool(wl.model.graph_tail_infer)     # inference: layers 3-5 replayed as one hipGraph

    single = world == 1 and rank == 0
    if single and not args.no_cpu_baseline:
        unpin()          # (the timed GPU part of this workload is over; the extra workloads below pin again)
        line["cpu_baseline"] = safe(wl.cpu_baseline, {"value": None, "unit": "scenes/s", "cores": None, "kind": "port", "sample": "failed"})
        # the operator baseline "at 1 thread and at all cores"
        xyz_np = getattr(wl, "xyz_np", None)
        if xyz_np is None:
            xyz_np = wl.points_np[: args.points, 1:4].reshape(1, args.points, 3).copy()
        line["cpu_baseline_ops"] = {"threads_1": safe(workloads.sampling_grouping_cpu, {}, xyz_np, 1),
                                    "threads_all": safe(workloads.sampling_grouping_cpu, {}, xyz_np, None)}

    if single and not args.no_extra and args.workload == "auto":
        pin_to_gpu_node(torch, local_rank)
        # secondary timings (few steps each): round 1's headline and BASELINE configs[1] / configs[2]
        del wl
        extra = {}
        # configs[2] (KITTI, B = 4, whole iteration) in both of this repo's forms: dense-bf16 mode (bf16 tensors between the
        # GEMMs, dense encoder) and the default mode (f32 tensors, every large contraction as a 3-term bf16 split on the bf16
        # matrix cores, unique-token encoder)
        for xname, xbatch in (("backbone", 2), ("backbone_infer", 2), ("kitti_detector_train_bf16", 4), ("kitti_detector_train", 4)):
            try:
                torch.cuda.empty_cache()
                xw = workloads.create(xname, xbatch, args.points, device, rank, world)
                xsteps = max(5, min(args.steps, 10))
                xdt = time_workload(xw, xsteps, 6, device, parallel)     # two warm-up steps left first-use costs (allocator growth, library heuristics) in the timed ones
                xr = safe(xw.rooflines, {})
                extra[xname] = {"workload": xw.name, "step": step_text(xw.name), "scenes_per_gpu": xbatch,
                                "steps": xsteps, "ms_per_step": xdt / xsteps * 1e3, "scenes_per_s": xbatch * xsteps / xdt,
                                "dtype": xw.dtype}
                if xname == "backbone_infer" and xr.get("roofline") is not None:
                    extra[xname]["roofline_mfma_fused_sa"] = xr["roofline"]
                if xname == "backbone_infer":
                    extra[xname]["graph_tail_infer"] = bool(getattr(xw.model, "graph_tail_infer", False))
                if hasattr(xw, "host_bound"):       # which form the probe of this workload chose (workloads.py)
                    extra[xname].update(host_bound_at_probe=bool(xw.host_bound), graph_tail=bool(xw.model.graph_tail),
                                        graph_head=bool(getattr(xw.model, "graph_head", False)))
                del xw
            except Exception as e:  # noqa: BLE001  (a secondary timing must not cost the headline line)
                print("bench.py: extra workload %s failed: %r" % (xname, e), file=sys.stderr)
                extra[xname] = {"error": repr(e)}
        line["extra"] = extra

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

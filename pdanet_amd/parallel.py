"""Data-parallel plumbing: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference scales PDA-SSD only by plain data parallelism: DistributedSampler shards the
scenes, DDP all-reduces the 25.5 MB of fp32 gradients (tools/train.py:71-73,153-154;
datasets/__init__.py:62-67).  Scenes are independent, so the forward path has no collective
at all; the only exchange step is the gradient all-reduce, which DDP overlaps with backward.
Backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (no-op when
    WORLD_SIZE == 1).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PDA_DIST_BACKEND=gloo: rehearse the multi-process path with several ranks on ONE GPU (RCCL refuses
        # two ranks on a device; gloo stages CUDA tensors through the host)
        backend = os.environ.get("PDA_DIST_BACKEND") or backend
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_scenes(num_scenes, rank, world):
    """Contiguous, balanced shard of scene ids for this rank (DistributedSampler without
    shuffling: every scene is processed by exactly one rank)."""
    base, rem = divmod(num_scenes, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device):
    """MAX-reduce a python float over all ranks (bench timing contract)."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def wrap_ddp(model, device=None, bucket_cap_mb=25, grads_are_views=False):
    """DDP wrap as the reference does (tools/train.py:153-154).  The whole gradient (25.5 MB)
    fits ~1 default bucket: at this size the xGMI all-reduce is latency-bound
    and hides under backward.  grads_are_views: the parameters' .grad are views of an optimizer's flat
    buffer (optimization.FlatAdamOneCycle) and must stay so -- DDP then copies the reduced bucket into
    them instead of re-pointing .grad at its own buckets."""
    if not dist.is_initialized():
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=not grads_are_views)

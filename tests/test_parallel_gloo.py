"""world_size-2 gloo test of the data-parallel path on CPU: scene sharding, DDP gradient
averaging (the path's only exchange step) and the bench timing reduction.  The dense blocks
used here (TransformerEncoderLayerPreNorm, Vote_layer) are the backbone's own pure-torch
modules; the HIP operators have no CPU path and are covered by the -m gpu tests."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from pdanet_amd import parallel
    from pdanet_amd.pointnet2_modules import TransformerEncoderLayerPreNorm, Vote_layer
    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # 4 "scenes", sharded 2 + 2
    g = torch.Generator().manual_seed(1)
    scenes = torch.randn(4, 6, 5, 16, generator=g)           # (scene, ns, points, d)
    xyz = torch.randn(4, 5, 3, generator=g)
    mine = parallel.shard_scenes(4, rank, world)
    assert mine == [2 * rank, 2 * rank + 1]
    # DDP needs forward through the wrapper: use a tiny wrapper module
    class Step(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, feats, pts):                      # (S, ns, points, d), (S, points, 3)
            tot = 0
            for f, p in zip(feats, pts):                      # one DDP forward per step over all local scenes
                y = self.m["tr"](f)                           # (ns, points, d)
                v = self.m["vote"](p.unsqueeze(0), y.max(dim=0)[0].t().unsqueeze(0))
                tot = tot + y.pow(2).mean() + v[0].pow(2).mean()
            return tot / len(feats)
    torch.manual_seed(0)
    step = Step(torch.nn.ModuleDict(dict(tr=TransformerEncoderLayerPreNorm(16, 4, 8, dropout=0.0),
                                         vote=Vote_layer([8], 16, [3.0, 3.0, 2.0]))))
    step.m["vote"].eval()                                     # BN with 1-sample batches: use eval stats
    dstep = parallel.wrap_ddp(step)
    total = dstep(scenes[mine], xyz[mine])
    total.backward()
    grads = {k: p.grad.clone() for k, p in step.named_parameters() if p.grad is not None}
    t = parallel.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert t == float(world)
    parallel.barrier()
    torch.save(grads, os.path.join(out_dir, "grads_%d.pt" % rank))
    if rank == 0:
        # single-process reference over all 4 scenes
        torch.manual_seed(0)
        ref = Step(torch.nn.ModuleDict(dict(tr=TransformerEncoderLayerPreNorm(16, 4, 8, dropout=0.0),
                                            vote=Vote_layer([8], 16, [3.0, 3.0, 2.0]))))
        ref.m["vote"].eval()
        ref(scenes, xyz).backward()
        torch.save({k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None},
                   os.path.join(out_dir, "grads_ref.pt"))
    torch.distributed.destroy_process_group()


def test_ddp_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "grads_0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "grads_1.pt"))
    ref = torch.load(os.path.join(tmp_path, "grads_ref.pt"))
    assert set(g0) == set(ref) and len(ref) > 10
    for k in ref:
        assert torch.allclose(g0[k], g1[k], atol=0, rtol=0), k          # all-reduced: identical on both ranks
        assert torch.allclose(g0[k], ref[k], atol=1e-6, rtol=1e-5), k    # == gradient of the global batch


def _flat_worker(rank, world, port, out_dir):
    """The bench / training path's exchange: no DDP wrapper, one all-reduce of the optimizer's flat gradient buffer."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from pdanet_amd import optimization, parallel
    from pdanet_amd.pointnet2_modules import TransformerEncoderLayerPreNorm
    parallel.init_distributed(backend="gloo")

    def build(seed):
        torch.manual_seed(seed)
        return torch.nn.ModuleDict(dict(tr=TransformerEncoderLayerPreNorm(16, 4, 8, dropout=0.0),
                                        bn=torch.nn.BatchNorm1d(16), fc=torch.nn.Linear(16, 3)))

    def loss_of(m, feats):                                    # mean over the scenes it is given
        tot = 0
        for f in feats:
            y = m["tr"](f).max(dim=0)[0]                      # (points, d)
            tot = tot + m["fc"](m["bn"](y)).pow(2).mean()
        return tot / len(feats)
    g = torch.Generator().manual_seed(3)
    scenes = torch.randn(4, 6, 5, 16, generator=g)
    mine = parallel.shard_scenes(4, rank, world)
    model = build(100 + rank)                                 # ranks start DIFFERENT: data_parallel must broadcast rank 0's state
    model["bn"].running_mean.fill_(float(rank))
    opt = optimization.FlatAdamOneCycle(model, wd=0.01, grad_norm_clip=10.0)
    opt.data_parallel(model=model)
    assert opt._dp_world == 2 and not opt._dp_avg             # gloo: SUM + scale
    state = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    for mode in (True, False):                                # .grad handed over by autograd / accumulated into the views
        opt.zero_grad(set_to_none=mode)
        loss_of(model, scenes[mine]).backward()
        opt.exchange_gradients()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(opt._all, opt._grad_views))
        out[mode] = opt.flat_g.clone()
    with pytest.raises(RuntimeError, match="no CPU path"):    # the update itself is csrc/optim.hip only
        opt.step()
    torch.save(dict(flat=out, state=state, names=[n for grp in opt.groups for n, _ in grp] + [n for n, _ in opt.untrained],
                    offsets=opt.offsets), os.path.join(out_dir, "flat_%d.pt" % rank))
    if rank == 0:
        ref = build(100)
        ref["bn"].running_mean.fill_(0.0)
        loss_of(ref, scenes[:2]).backward()                   # BatchNorm statistics are per rank (no SyncBN): mean of the
        g0 = {n: p.grad.clone() for n, p in ref.named_parameters()}      # two ranks' losses, each on its own scenes
        ref.zero_grad()
        loss_of(ref, scenes[2:]).backward()
        torch.save({n: 0.5 * (g0[n] + p.grad) for n, p in ref.named_parameters()}, os.path.join(out_dir, "flat_ref.pt"))
    torch.distributed.destroy_process_group()


def test_flat_gradient_exchange_gloo_world2(tmp_path):
    """tools/train.py:153-154 through optimization.FlatAdamOneCycle.data_parallel: after the exchange both ranks hold
    the mean of the per-rank gradients in the flat buffer (== DDP's result), parameters and buffers start as rank 0's."""
    port = _free_port()
    mp.spawn(_flat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "flat_0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "flat_1.pt"))
    ref = torch.load(os.path.join(tmp_path, "flat_ref.pt"))
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k             # broadcast from rank 0, buffers included
    for mode in (True, False):
        assert torch.equal(r0["flat"][mode], r1["flat"][mode])           # identical on both ranks
    assert torch.allclose(r0["flat"][True], r0["flat"][False], atol=1e-7)
    assert len(r0["names"]) == len(r0["offsets"]) > 10
    for n, o in zip(r0["names"], r0["offsets"]):
        g = ref[n]
        got = r0["flat"][True][o:o + g.numel()].view(g.shape)
        assert torch.allclose(got, g, atol=1e-6, rtol=1e-5), n


def test_shard_scenes_covers_everything():
    sys.path.insert(0, ROOT)
    from pdanet_amd import parallel
    for total in (0, 1, 7, 16):
        for world in (1, 2, 3, 8):
            got = sum((parallel.shard_scenes(total, r, world) for r in range(world)), [])
            assert got == list(range(total))

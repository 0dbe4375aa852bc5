"""Mirror of the reference's ``pointnet2_batch/pointnet2_utils.py`` operator API.

Same public names, argument order, allocation/initialisation contracts and gradients as
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (cited per class);
the kernels underneath are the hand-written gfx950 ones reached through the C ABI
(``pointnet2_batch_cuda`` mirror).  Differences that are deliberate:
  * outputs are allocated with ``torch.empty(..., device=input.device)`` and kernels run on
    the current stream (the reference uses ``torch.cuda.IntTensor(...)`` + the null stream),
    so the ops work under side streams and hipGraph capture;
  * the grouper modules use one multi-radius ball query when asked for several scales.
"""
import weakref
from typing import List, Tuple

import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import pointnet2_batch_cuda as pointnet2

# Under torch.autocast the dense layers may run in bf16; the point operators always compute in
# fp32 ("ops stay fp32"): inputs are cast back at the operator boundary.
_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


class FarthestPointSampling(Function):
    """pointnet2_utils.py:10-33.  xyz (B,N,3) -> idx (B,npoint) int32; temp pre-filled 1e10."""

    @staticmethod
    @_fwd
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class FurthestPointSamplingWithDist(Function):
    """pointnet2_utils.py:39-62.  dist matrix (B,N,N) -> idx (B,npoint)."""

    @staticmethod
    @_fwd
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_with_dist_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply


class GatherOperation(Function):
    """pointnet2_utils.py:67-98.  features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""

    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out_data, idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """pointnet2_utils.py:104-130.  Returns (sqrt(dist2), idx), both (B,N,3)."""

    @staticmethod
    @_fwd
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:136-178.  features (B,C,M), idx/weight (B,n,3) -> (B,C,n)."""

    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out_data, idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:184-222.  features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""

    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out_data, idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class GroupRows(Function):
    """MI355X extension: grouping in the point-major layout.  rows (B,N,C), idx (B,...) int32 ->
    (B, ..., C) with out[b, e, :] = rows[b, idx[b, e], :].  Same values as
    `grouping_operation(rows^T, idx)` permuted to channel-last; gradient = scatter-add of rows."""

    @staticmethod
    @_fwd
    def forward(ctx, rows: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert rows.is_contiguous() and idx.is_contiguous()
        B, N, C = rows.size()
        E = idx.numel() // B
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=rows.device)
        pointnet2.group_rows(B, N, C, E, rows, idx, out)
        ctx.for_backwards = (idx, N, C, E)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        idx, N, C, E = ctx.for_backwards
        B = idx.shape[0]
        grad_rows = torch.zeros((B, N, C), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_rows_grad(B, N, C, E, grad_out.detach().contiguous(), idx, grad_rows)
        return grad_rows, None


group_rows = GroupRows.apply


class GroupAttention(Function):
    """MI355X extension: multi-head self-attention over the nsample tokens of each group, on the
    in_proj output qkv (G, S, 3*D) laid out [q | k | v] x heads x head_dim (csrc/group_attention.hip).
    Same math as F.scaled_dot_product_attention without mask/dropout."""
    SUPPORTED_SEQ, SUPPORTED_HD = (8, 16, 32), (32, 64, 128)

    @staticmethod
    def supported(qkv, heads):
        G, S, D3 = qkv.shape
        return (qkv.is_cuda and qkv.dtype == torch.float32 and S in GroupAttention.SUPPORTED_SEQ
                and D3 % (3 * heads) == 0 and (D3 // (3 * heads)) in GroupAttention.SUPPORTED_HD)

    @staticmethod
    @_fwd
    def forward(ctx, qkv, heads):
        qkv = qkv.contiguous()
        G, S, D3 = qkv.shape
        hd = D3 // (3 * heads)
        out = torch.empty((G, S, heads * hd), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((G, heads, S), dtype=torch.float32, device=qkv.device)
        pointnet2.group_attention_fwd(qkv, out, lse, G, S, heads, hd)
        ctx.save_for_backward(qkv, lse)
        ctx.heads = heads
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        qkv, lse = ctx.saved_tensors
        G, S, D3 = qkv.shape
        hd = D3 // (3 * ctx.heads)
        grad_qkv = torch.empty_like(qkv)
        pointnet2.group_attention_bwd(qkv, grad_out.contiguous(), lse, grad_qkv, G, S, ctx.heads, hd)
        return grad_qkv, None


group_attention = GroupAttention.apply


class BatchNormReLU(Function):
    """MI355X extension: training-mode BatchNorm over the last dim of x (..., C) followed by ReLU, forward
    and backward in csrc/bn_relu.hip (statistics over all leading dims, like BatchNorm2d over (B, C, H, W)
    of the channel-major tensor; running statistics updated in place)."""

    @staticmethod
    def supported_module(bn, c):
        return bn.training and bn.affine and bn.momentum is not None and 4 <= c <= 1024 and (c & (c - 1)) == 0

    @staticmethod
    def supported(x, bn):
        # x: fp32, or bf16 where it is the output of a dense-bf16 GEMM (`linear(..., out_bf16=True)`)
        return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0
                and BatchNormReLU.supported_module(bn, x.shape[-1]))

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, out_bf16=False):
        x = x.contiguous()
        c = x.shape[-1]
        rows = x.numel() // c
        y = torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_fwd(x, weight, bias, running_mean, running_var, y, stats, scratch, rows, c, eps, momentum)
        ctx.save_for_backward(x, weight, bias, stats)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight, bias, stats = ctx.saved_tensors
        c = x.shape[-1]
        rows = x.numel() // c
        grad_x = torch.empty_like(x)
        gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_bwd(x, grad_y.contiguous(), weight, bias, stats, grad_x, gw, gb, scratch, rows, c)
        return grad_x, gw, gb, None, None, None, None, None


class BatchNormReLUMaxPool(Function):
    """MI355X extension: relu(bn(x)) in training mode followed by the max over dim -2 (the nsample axis of a grouped
    (..., nsample, C) tensor), as one autograd node: the activation and the dense gradient of the max-pool are never
    materialised (csrc/bn_relu.hip, pda_bn_relu_max_pool_{fwd,bwd})."""

    @staticmethod
    def supported(x, bn):
        return BatchNormReLU.supported(x, bn) and x.dim() >= 3 and 1 <= x.shape[-2] <= 255

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        x = x.contiguous()
        ns, c = x.shape[-2], x.shape[-1]
        groups = x.numel() // (ns * c)
        out = torch.empty(x.shape[:-2] + (c,), dtype=torch.float32, device=x.device)
        arg = torch.empty(x.shape[:-2] + (c,), dtype=torch.uint8, device=x.device)
        stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_max_pool_fwd(x, weight, bias, running_mean, running_var, out, arg, stats, scratch, groups, ns, c, eps, momentum)
        ctx.save_for_backward(x, weight, bias, stats, arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, bias, stats, arg = ctx.saved_tensors
        ns, c = x.shape[-2], x.shape[-1]
        groups = x.numel() // (ns * c)
        grad_x = torch.empty_like(x)
        gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_max_pool_bwd(x, grad_out.contiguous().float(), arg, weight, bias, stats, grad_x, gw, gb, scratch, groups, ns, c)
        return grad_x, gw, gb, None, None, None, None


def batch_norm_relu_max_pool(bn, x):
    """max over dim -2 of relu(bn(x)) for an nn.BatchNorm{1,2}d module over the last dim of x (training mode)."""
    bump_bn_counter(bn)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return BatchNormReLUMaxPool.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum)


# num_batches_tracked bookkeeping: 51 one-element `add_` launches per step when every layer bumps its own
# counter.  IASSD_Backbone.forward collects them here and bumps them with one _foreach_add_ at the end.
BN_COUNTERS_PENDING = None


def bump_bn_counter(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if BN_COUNTERS_PENDING is not None:
            BN_COUNTERS_PENDING.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)


def batch_norm_relu(bn, x, out_bf16=False):
    """relu(bn(x)) for an nn.BatchNorm{1,2}d module over the LAST dim of x (training mode; fp32 arithmetic).
    out_bf16: write the result as bf16 (dense-bf16 mode, when it only feeds the next bf16 GEMM)."""
    bump_bn_counter(bn)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return BatchNormReLU.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, out_bf16)


class AssembleTokens(Function):
    """MI355X extension: the encoder input of a PDA scale, x (B,M,ns,4C) = [rppe | f*dscale | f | glob], with f =
    feats_pm (B,N,C) gathered by idx -- one kernel forward, one backward (csrc/assemble.hip) instead of a grouping,
    a multiply, an expand and a concatenation (and their autograd chain)."""

    @staticmethod
    def supported(rppe, feats_pm):
        c = feats_pm.shape[-1]
        return (FUSED_ASSEMBLE and rppe.is_cuda and rppe.dtype == torch.float32 and feats_pm.dtype == torch.float32
                and c in (16, 32, 64, 128, 256) and rppe.shape[-1] == c and not torch.is_autocast_enabled())

    @staticmethod
    def forward(ctx, rppe, dscale, feats_pm, idx, glob):
        B, M, ns, C = rppe.shape
        N = feats_pm.shape[1]
        rppe, dscale, feats_pm, glob = rppe.contiguous(), dscale.contiguous(), feats_pm.contiguous(), glob.contiguous()
        out = torch.empty((B, M, ns, 4 * C), dtype=torch.float32, device=rppe.device)
        pointnet2.assemble_tokens(rppe, dscale, feats_pm, idx, glob, out, B, N, M, ns, C)
        ctx.save_for_backward(dscale, feats_pm, idx)
        ctx.dims = (B, N, M, ns, C)
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dscale, feats_pm, idx = ctx.saved_tensors
        B, N, M, ns, C = ctx.dims
        dev = grad_out.device
        g_rppe = torch.empty((B, M, ns, C), dtype=torch.float32, device=dev)
        g_ds = torch.empty_like(dscale)
        g_feats = torch.zeros((B, N, C), dtype=torch.float32, device=dev)
        g_glob = torch.empty((B, M, C), dtype=torch.float32, device=dev)
        pointnet2.assemble_tokens_grad(grad_out.contiguous(), dscale, feats_pm, idx, g_rppe, g_ds, g_feats, g_glob, B, N, M, ns, C)
        return g_rppe, g_ds, g_feats, None, g_glob


FUSED_ASSEMBLE = True


class DensityNetFused(Function):
    """MI355X extension: DensityNet (1 -> 16 -> 8 -> 1, conv + BatchNorm + ReLU each) in training mode on a scalar
    input per token, 4 launches forward and 5 backward (csrc/densitynet.hip) instead of ~45 small ones.  The
    12 parameter tensors travel as one packed 227-float block; their gradients come back the same way."""
    SPLITS = (16, 16, 16, 16, 128, 8, 8, 8, 8, 1, 1, 1)     # w1 b1 g1 be1 W2 b2 g2 be2 w3 b3 g3 be3

    @staticmethod
    def supported(x, dn):
        convs, bns = dn.mlp_convs, dn.mlp_bns
        return (FUSED_DENSITYNET and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and dn.training
                and len(convs) == 3 and [c.out_channels for c in convs] == [16, 8, 1] and convs[0].in_channels == 1
                and all(c.bias is not None for c in convs) and all(b.affine and b.momentum is not None for b in bns)
                and len({b.eps for b in bns}) == 1 and len({b.momentum for b in bns}) == 1
                and not torch.is_autocast_enabled() and x.numel() > 0)

    @staticmethod
    def forward(ctx, x, eps, momentum, running, part, *params):
        """part: None, or the device half of the scale's unique-token plan (ragged_plan_parts: cnt, off, rowmap, roww, groups,
        ns) -- x (groups, ns) then holds ball_query's repeats of each group's first neighbour behind its distinct slots, and
        only the distinct slots are evaluated (csrc/densitynet.hip, DnRows): same y in every slot, same statistics and
        gradients, 5-10 x fewer tokens walked by the nine passes.  No host read: the count stays on the device."""
        x = x.contiguous()
        n = x.numel()
        packed = torch.cat([p.reshape(-1) for p in params])
        nparam, nscratch = pointnet2.densitynet_sizes()
        assert packed.numel() == nparam
        y = torch.empty_like(x)
        stats = torch.empty((46,), dtype=torch.float32, device=x.device)
        scratch = torch.empty((nscratch,), dtype=torch.uint8, device=x.device)
        if part is None:
            pointnet2.densitynet_fwd(x, packed, y, stats, scratch, running, n, eps, momentum)
            ctx.save_for_backward(x, packed, stats)
        else:
            _, off, rowmap, roww, G, ns = part
            pointnet2.densitynet_fwd_unique(x, packed, y, stats, scratch, running, n, rowmap, roww, off, G, ns, eps, momentum)
            ctx.save_for_backward(x, packed, stats, off, rowmap, roww)
            ctx.groups, ctx.ns = G, ns
        ctx.unique = part is not None
        ctx.eps = eps
        ctx.shapes = [p.shape for p in params]
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, packed, stats = ctx.saved_tensors[:3]
        nparam, nscratch = pointnet2.densitynet_sizes()
        grads = torch.empty((nparam,), dtype=torch.float32, device=x.device)
        scratch = torch.empty((nscratch,), dtype=torch.uint8, device=x.device)
        if ctx.unique:
            off, rowmap, roww = ctx.saved_tensors[3:]
            pointnet2.densitynet_bwd_unique(x, grad_y.contiguous(), packed, stats, grads, scratch, x.numel(), rowmap, roww, off,
                                            ctx.groups, ctx.ns, ctx.eps)
        else:
            pointnet2.densitynet_bwd(x, grad_y.contiguous(), packed, stats, grads, scratch, x.numel(), ctx.eps)
        pieces = torch.split(grads, DensityNetFused.SPLITS)
        return (None, None, None, None, None) + tuple(g.view(s) for g, s in zip(pieces, ctx.shapes))


class DensityNetFusedMulti(Function):
    """DensityNetFused for the k scales of a PDA layer in ONE set of launches (pda_densitynet_{fwd,bwd}_multi): each of the
    nine passes is ~10 us of dependency latency on <= 128 workgroups, so k problems per launch cost what one does.  Per scale
    the same arithmetic as DensityNetFused (bit for bit).  apply(metas, *xs, *params): metas = [(eps, momentum, running,
    part)] per scale, 12 parameter tensors per scale; returns the k outputs."""

    @staticmethod
    def forward(ctx, metas, *flat):
        k = len(metas)
        xs, params = [x.contiguous() for x in flat[:k]], flat[k:]
        nparam, nscratch = pointnet2.densitynet_sizes()
        problems, saved, ys = [], [], []
        for i, (x, (eps, momentum, running, part)) in enumerate(zip(xs, metas)):
            packed = torch.cat([p.reshape(-1) for p in params[12 * i:12 * i + 12]])
            assert packed.numel() == nparam
            y = torch.empty_like(x)
            stats = torch.empty((46,), dtype=torch.float32, device=x.device)
            scratch = torch.empty((nscratch,), dtype=torch.uint8, device=x.device)
            prob = dict(x=x, params=packed, y=y, stats=stats, scratch=scratch, running=running, n=x.numel(), eps=eps, momentum=momentum)
            keep = [x, packed, stats]
            if part is not None:
                _, off, rowmap, roww, G, ns = part
                prob.update(rowmap=rowmap, roww=roww, off=off, groups=G, nsample=ns)
                keep += [off, rowmap, roww]
            problems.append(prob)
            saved.append((keep, None if part is None else (part[4], part[5]), eps))
            ys.append(y)
        pointnet2.densitynet_multi(problems)
        ctx.save_for_backward(*[t for keep, _, _ in saved for t in keep])
        ctx.layout = [(len(keep), uq, eps) for keep, uq, eps in saved]
        ctx.shapes = [p.shape for p in params]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *grad_ys):
        nparam, nscratch = pointnet2.densitynet_sizes()
        tensors = list(ctx.saved_tensors)
        problems, grads = [], []
        for (cnt, uq, eps), gy in zip(ctx.layout, grad_ys):
            keep, tensors = tensors[:cnt], tensors[cnt:]
            x, packed, stats = keep[:3]
            g = torch.empty((nparam,), dtype=torch.float32, device=x.device)
            scratch = torch.empty((nscratch,), dtype=torch.uint8, device=x.device)
            gy = torch.zeros_like(x) if gy is None else gy.contiguous()
            prob = dict(x=x, grad_y=gy, params=packed, stats=stats, grad_params=g, scratch=scratch, n=x.numel(), eps=eps)
            if uq is not None:
                prob.update(off=keep[3], rowmap=keep[4], roww=keep[5], groups=uq[0], nsample=uq[1])
            problems.append(prob)
            grads.append(g)
        pointnet2.densitynet_multi(problems, backward=True)
        out = []
        for i, g in enumerate(grads):
            out += [t.view(s) for t, s in zip(torch.split(g, DensityNetFused.SPLITS), ctx.shapes[12 * i:12 * i + 12])]
        return (None,) + (None,) * len(grads) + tuple(out)


FUSED_DENSITYNET = True
DENSITYNET_MULTI = os.environ.get("PDA_DENSITYNET_MULTI", "1") != "0"       # the scales of a layer share their launches
DENSITYNET_UNIQUE = os.environ.get("PDA_DENSITYNET_UNIQUE", "1") != "0"     # distinct slots only where a plan exists


def densitynet(dn, x, part=None):
    """DensityNet module `dn` (pointnet2_modules.DensityNet) on x (..., 1) in training mode.  part: see DensityNetFused."""
    convs, bns = dn.mlp_convs, dn.mlp_bns
    running = None
    if all(b.track_running_stats for b in bns):
        running = [t for b in bns for t in (b.running_mean, b.running_var)]
        for b in bns:
            bump_bn_counter(b)
    params = []
    for c, b in zip(convs, bns):
        params += [c.weight, c.bias, b.weight, b.bias]
    return DensityNetFused.apply(x, bns[0].eps, bns[0].momentum, running, part if DENSITYNET_UNIQUE else None, *params)


def densitynet_multi(dns, xs, parts):
    """The DensityNets `dns` of the scales of one layer on their inputs `xs` (parts: the scales' plan parts or None each): one
    set of launches for all of them."""
    if not DENSITYNET_MULTI or len(dns) == 1 or len(dns) > 4:
        return [densitynet(dn, x, part) for dn, x, part in zip(dns, xs, parts)]
    metas, params = [], []
    for dn, part in zip(dns, parts):
        convs, bns = dn.mlp_convs, dn.mlp_bns
        running = None
        if all(b.track_running_stats for b in bns):
            running = [t for b in bns for t in (b.running_mean, b.running_var)]
            for b in bns:
                bump_bn_counter(b)
        metas.append((bns[0].eps, bns[0].momentum, running, part if DENSITYNET_UNIQUE else None))
        for c, b in zip(convs, bns):
            params += [c.weight, c.bias, b.weight, b.bias]
    return list(DensityNetFusedMulti.apply(metas, *xs, *params))


class LinearLongTokens(Function):
    """y = x W^T + b over the last dim of x (..., in).  Forward and the input gradient are _gemm_nt / _gemm_nn (the
    split-bf16 kernels, or the library where those do not apply); the weight and bias gradients -- GEMMs with a tiny
    output and a reduction over all tokens, which the libraries run at a fraction of their peak -- come from
    csrc/wgrad.hip in one pass."""
    MIN_TOKENS = 4096
    SPLIT_MIN_TOKENS = 32768      # forward / input gradient below this: the library (measured: the step got 0.6 ms slower with the
                                  # 4096..32768-token layers on the split kernels and their weight-packing launches)

    @staticmethod
    def supported(x, weight):
        """Where the kernel wins on MI355X: both feature dims >= 64 (half of its 128 x 128 output tile empty at 64: 8192 x
        64 -> 128 still 16 us against 71; both <= 64 take the streaming form) from 4096 tokens -- device time per call, library g^T x plus the bias sum
        against pda_linear_wgrad (profiles/r02_wgrad_small_tokens.txt): 8192 x 128 -> 256: 64 / 22 us, 4096 x 256 -> 512:
        43 / 30, 4096 x 512 -> 1536: 80 / 82, 12979 x 256 -> 256: 109 / 31."""
        return torch.is_grad_enabled() and not torch.is_autocast_enabled() and LinearLongTokens.kernel_wins(x, weight)

    @staticmethod
    def kernel_wins(x, weight):
        if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2):
            return False
        n_out, n_in = weight.shape
        tokens = x.numel() // max(1, x.shape[-1])
        if n_out % 4 or n_in % 4:
            return False
        if max(n_out, n_in) <= 64:       # narrow layers: the streaming form (wgrad_skinny_kernel), 3-7x the library
            return tokens >= 4096
        return min(n_out, n_in) >= 64 and tokens >= LinearLongTokens.MIN_TOKENS

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if x.numel() // x.shape[-1] < LinearLongTokens.SPLIT_MIN_TOKENS or min(weight.shape) < 128:
            return torch.nn.functional.linear(x, weight, bias)
        return _gemm_nt(x.reshape(-1, x.shape[-1]), weight, bias).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if grad_out.numel() // grad_out.shape[-1] < LinearLongTokens.SPLIT_MIN_TOKENS or min(weight.shape) < 128:
                gx = grad_out.matmul(weight)
            else:
                gx = _gemm_nn(grad_out.reshape(-1, grad_out.shape[-1]), weight).view(*grad_out.shape[:-1], weight.shape[1])
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            n_out, n_in = weight.shape
            xc = x.contiguous()
            tokens = xc.numel() // n_in
            gw = torch.empty_like(weight)
            gb = torch.empty((n_out,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
            pointnet2.linear_wgrad(xc, grad_out, gw, gb, tokens, n_in, n_out)
        return gx, gw, gb


# ---- "bf16 dense" mode (BASELINE configs[2]: bf16 on the dense layers, operators fp32) -------------------------
# DENSE_BF16 = True: the large GEMMs (forward, input gradient and weight gradient of the 1x1 convolutions /
# projections) take bf16 operands with fp32 accumulation and fp32 OUTPUT (torch.mm(..., out_dtype=float32)); every
# activation that a kernel of this repo reads, every statistic, normalisation and the attention stay fp32, so all of
# those kernels run unchanged.  This is what torch.autocast does to the GEMM operands; the only bf16 tensors stored
# are the GEMM input copies kept for the weight gradients.
DENSE_BF16 = False
BF16_MIN_TOKENS = 32768     # below this (and for narrow layers) the operand casts cost more than the GEMM saves
BF16_MIN_FEATURES = 128


def _dense_bf16(x, weight=None):
    if not (DENSE_BF16 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not torch.is_autocast_enabled()):
        return False
    if x.numel() // max(1, x.shape[-1]) < BF16_MIN_TOKENS:
        return False
    return weight is None or min(weight.shape[0], weight.shape[1]) >= BF16_MIN_FEATURES


def _b16(t):
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


_B16_PARAMS = {}


def _b16p(p):
    """bf16 copy of a parameter (weight / bias), cached until the parameter is written again (its version counter moves:
    optimizer step, load_state_dict) -- a step uses each weight in 2-3 GEMMs, and each cast is a launch."""
    if p.dtype == torch.bfloat16:
        return p
    if p.is_cuda and torch.cuda.is_current_stream_capturing():
        return p.detach().to(torch.bfloat16)        # the cast belongs INSIDE the graph: a replay must see the new weights
    key = id(p)
    stamp = (p._version, p.data_ptr(), _lib.PARAM_EPOCH[0])
    hit = _B16_PARAMS.get(key)
    if hit is not None and hit[0] == stamp and hit[1]() is p:
        return hit[2]
    pb = p.detach().to(torch.bfloat16)
    _B16_PARAMS[key] = (stamp, weakref.ref(p, lambda _r, k=key: _B16_PARAMS.pop(k, None)), pb)
    return pb


# SPLIT_GEMM: f32 contractions run on the bf16 matrix cores with three-term bf16 operands (csrc/gemm_split.hip: f32-grade
# results; LinearColsMFMA: lin_split_kernel, 1.3-1.5x the f32 MFMA kernel; the encoder projections and input gradients:
# gemm_split_kernel, 1.1-1.35x the library's f32 GEMMs).  False (PDA_SPLIT_GEMM=0): lin_cols_kernel / the library.
SPLIT_GEMM = os.environ.get("PDA_SPLIT_GEMM", "1") != "0"

# ---- f32 GEMMs of the step on csrc/gemm_split.hip (bf16 matrix cores, three-term operands, f32-grade error) ----------
SPLIT_GEMM_MIN_TOKENS = 4096      # below: the library (one small launch) is as fast and saves the packing launch


def _split_gemm_ok(x2d, k):
    return (SPLIT_GEMM and x2d.is_cuda and x2d.dtype == torch.float32 and k % 32 == 0 and x2d.shape[0] >= SPLIT_GEMM_MIN_TOKENS
            and not DENSE_BF16 and not torch.is_autocast_enabled())


ROWS_IN_REGISTERS_MAX_TOKENS = int(os.environ.get("PDA_ROWS_MAX_TOKENS", "98304"))


def _rows_in_registers(tokens, k, n_out):
    """Which split-bf16 kernel a (tokens, k) x (n_out, k)^T product takes: lin_split_kernel (a wave's rows stay in registers:
    short K only) is ahead of the 256 x 256 tiles of gemm_split_wide_kernel while those leave a ragged last round of
    workgroups on the 256 CUs; from ~100k tokens the tiles win (131072 x 256 -> 256: 0.109 against 0.131 ms, -> 512: 0.217
    against 0.233; 65536 x 256 -> 512: level; 32307 x 256 -> 768: 0.077 against 0.101 for the rows form)."""
    return k <= 256 and n_out % 128 == 0 and n_out <= 2048 and tokens < ROWS_IN_REGISTERS_MAX_TOKENS


# Packed bf16 planes of the PARAMETER weights, keyed on the storage address: {ptr: (tag, planes of W, planes of W^T, weak
# reference to the owning nn.Parameter, byte offset of the weight inside it)} with tag = (version counter,
# _lib.WEIGHT_EPOCH, shape).  Training packs both forms in one launch in the forward pass; the backward pass (which sees
# the weight as an unpacked saved tensor, not as the Parameter) finds the transposed planes by address.  An entry is valid
# only while its Parameter is alive AND still sits at that address -- then no other tensor can occupy it; a freed model or a
# re-pointed `.data` (optimization.FlatAdamOneCycle moves the parameters into its flat buffer) invalidates it.  Inference
# packs once and hits until the weights change.
_PACKED_PLANES = {}


def _owning_parameter(w):
    if isinstance(w, nn.Parameter):
        return w
    base = getattr(w, "_base", None)
    return base if isinstance(base, nn.Parameter) else None


def _split_planes(w, transposed=False):
    """Planes of W (n_out, k) for Y = X W^T (transposed=False) or of W^T for dX = dY W (True); w is the (n_out, k) source."""
    n_out, k = w.shape
    wd = w.detach()

    def pack(t):
        src = wd if wd.is_contiguous() else wd.contiguous()
        return pointnet2.linear_split_pack(src, k, n_out, transposed_source=True) if t else pointnet2.linear_split_pack(src, n_out, k)
    if not wd.is_contiguous() or torch.cuda.is_current_stream_capturing():
        return pack(transposed)      # under capture the packing belongs INSIDE the graph: a replay must see the new weights
    key, tag = wd.data_ptr(), (wd._version, _lib.WEIGHT_EPOCH[0], n_out, k)
    ent = _PACKED_PLANES.get(key)
    if ent is not None:
        owner = ent[3]()
        if owner is None or owner.data_ptr() + ent[4] != key:
            del _PACKED_PLANES[key]                       # the parameter is gone or lives elsewhere now
            ent = None
        elif ent[0] != tag:
            ent = None                                    # the weights changed: repack below (if this is the parameter)
        elif ent[2 if transposed else 1] is not None:
            return ent[2 if transposed else 1]
    param = _owning_parameter(w)
    if param is None:
        return pack(transposed)
    if w.requires_grad and n_out % 32 == 0 and k >= 128:               # (grad mode is off inside Function.forward: ask the tensor)
        wf, wft = pointnet2.linear_split_pack_both(wd, n_out, k)       # the backward pass of this iteration wants W^T
    elif transposed:
        wf, wft = (ent[1] if ent is not None else None), pack(True)
    else:
        wf, wft = pack(False), (ent[2] if ent is not None else None)
    # (the entry goes with its parameter: a callback on the weak reference, as for the bf16 copies above)
    _PACKED_PLANES[key] = (tag, wf, wft, weakref.ref(param, lambda _r, k=key: _PACKED_PLANES.pop(k, None)), key - param.data_ptr())
    return wft if transposed else wf


def _gemm_nt(x2d, w, bias=None, relu=False):
    """relu?(x2d (T, K) @ w (N, K)^T + bias) in f32."""
    n_out, k = w.shape
    if _split_gemm_ok(x2d, k) and w.dtype == torch.float32 and n_out >= 128:
        x2d = x2d.contiguous()
        T = x2d.shape[0]
        y = torch.empty((T, n_out), dtype=torch.float32, device=x2d.device)
        wf = _split_planes(w)
        bias = None if bias is None else bias.detach().contiguous()
        if _rows_in_registers(T, k, n_out):
            pointnet2.linear_split(x2d, wf, bias, y, T, k, n_out, relu=relu)
        else:
            pointnet2.gemm_split(x2d, wf, bias, y, T, k, n_out, relu=relu)
        return y
    if relu and bias is not None:
        return torch._addmm_activation(bias, x2d, w.t())
    y = torch.nn.functional.linear(x2d, w, bias)
    return torch.relu_(y) if relu else y


def _gemm_nn(g2d, w, acc=None):
    """g2d (T, N) @ w (N, K) [+ acc, in place: acc is owned by the caller] in f32: the input gradient of y = x w^T."""
    n, k_out = w.shape
    if _split_gemm_ok(g2d, n) and w.dtype == torch.float32 and k_out >= 128:
        g2d = g2d.contiguous()
        T = g2d.shape[0]
        y = acc if acc is not None else torch.empty((T, k_out), dtype=torch.float32, device=g2d.device)
        wf = _split_planes(w, transposed=True)
        if acc is None and _rows_in_registers(T, n, k_out):
            pointnet2.linear_split(g2d, wf, None, y, T, n, k_out)
        else:
            pointnet2.gemm_split(g2d, wf, None, y, T, n, k_out, accumulate=acc is not None)
        return y
    if acc is not None:
        return acc.addmm_(g2d, w)
    return g2d.mm(w)


def _mm_nt(x2d, w, bf16):
    """x2d (T, in) @ w (out, in)^T -> (T, out) fp32."""
    if bf16:
        return torch.mm(_b16(x2d), _b16p(w).t(), out_dtype=torch.float32)
    return x2d.mm(w.t())


def _mm_nn(g2d, w, bf16, acc=None):
    """g2d (T, out) @ w (out, in) -> (T, in) fp32; acc: a (T, in) fp32 tensor OWNED by the caller to accumulate into."""
    if bf16:
        gb, wb = _b16(g2d), _b16p(w)
        if acc is not None:
            return torch.addmm(acc, gb, wb, out_dtype=torch.float32)
        return torch.mm(gb, wb, out_dtype=torch.float32)
    if acc is not None:
        return acc.addmm_(g2d, w)
    return g2d.mm(w)


def _lin(x, w, b, bf16, keep=False):
    """linear over the last dim with the dense-mode GEMM.  keep=True also returns the GEMM's (T, in) input operand as
    it was fed (the bf16 copy in dense-bf16 mode): what a backward pass needs for the weight gradient."""
    if not bf16:
        y = torch.nn.functional.linear(x, w, b)
        return (y, x.reshape(-1, x.shape[-1])) if keep else y
    xb = _b16(x.reshape(-1, x.shape[-1]))
    y = _mm_nt(xb, w, True)
    if b is not None:
        y.add_(b)
    y = y.view(*x.shape[:-1], w.shape[0])
    return (y, xb) if keep else y


class LinearBF16(Function):
    """y = x W^T + b with bf16 GEMM operands (fp32 accumulation) in the forward, the input gradient and the weight
    gradient (`_wgrad_bf16`); keeps only the bf16 copy of x for backward.  x may already be bf16 (the output of a kernel
    that emits the operand copy); out_bf16: y leaves the GEMM as bf16 (its consumer reads bf16, e.g. `batch_norm_relu`).
    The input gradient has the dtype of x."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_bf16=False):
        xb = _b16(x.reshape(-1, x.shape[-1]))
        if out_bf16:
            wb = _b16p(weight)
            y = torch.mm(xb, wb.t()) if bias is None else torch.addmm(_b16p(bias), xb, wb.t())
        else:
            y = _mm_nt(xb, weight, True)
            if bias is not None:
                y.add_(bias)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias, ctx.x_shape, ctx.x_bf16 = bias is not None, x.shape, x.dtype == torch.bfloat16
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_out):
        xb, weight = ctx.saved_tensors
        n_out, n_in = weight.shape
        g2 = _b16(grad_out.reshape(-1, n_out))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (torch.mm(g2, _b16p(weight)) if ctx.x_bf16 else _mm_nn(g2, weight, True)).view(ctx.x_shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _wgrad(xb, g2, weight, ctx.has_bias, True)
        return gx, gw, gb, None


def linear(x, weight, bias=None, out_bf16=False):
    """F.linear with the long-token weight-gradient kernel where it applies (and bf16 GEMMs in DENSE_BF16 mode;
    out_bf16 is honoured there only -- callers check `_dense_bf16(x, weight)` first)."""
    if weight.dim() == 2 and _dense_bf16(x, weight):
        if torch.is_grad_enabled():
            return LinearBF16.apply(x, weight, bias, out_bf16)
        return _lin(x, weight, bias, True)
    assert not out_bf16 and x.dtype == torch.float32, "bf16 boundary tensors exist only between dense-bf16 operators"
    if LINEAR_WGRAD_KERNEL and LinearLongTokens.supported(x, weight):
        return LinearLongTokens.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


LINEAR_WGRAD_KERNEL = True


def _wgrad_bf16(xb, gb, want_bias):
    """Dense-bf16 weight gradient g^T x over T tokens: a GEMM with a tiny output and a huge reduction, which the
    library runs at a fraction of its rate as one problem.  Split the tokens into S slabs -- one batched GEMM with S
    well-shaped (out x T/S) @ (T/S x in) problems, fp32 partials -- and add the S partials (a few MB)."""
    T, n_in = xb.shape
    n_out = gb.shape[1]
    tiles = max(1, (n_in // 128) * (n_out // 128))
    S = 64
    while S > 16 and S * tiles > 768:
        S //= 2
    while S > 1 and (T % S or T // S < 512):
        S //= 2
    if S > 1:
        gw = torch.bmm(gb.view(S, T // S, n_out).transpose(1, 2), xb.view(S, T // S, n_in), out_dtype=torch.float32).sum(0)
    else:
        gw = torch.mm(gb.t(), xb, out_dtype=torch.float32)
    if not want_bias:
        return gw, None
    if n_out % 8 == 0 and n_out <= 2048 and gb.is_contiguous():
        gbias = torch.empty((n_out,), dtype=torch.float32, device=gb.device)
        pointnet2.colsum_bf16(gb, gbias, T, n_out)
        return gw, gbias
    return gw, gb.sum(0, dtype=torch.float32)


def _wgrad(x2d, g2d, weight, want_bias, bf16=False):
    """grad_weight (out, in) and grad_bias of y = x W^T + b from x (T, in), g (T, out).  fp32: csrc/wgrad.hip where it
    wins (LinearLongTokens.kernel_wins), the library otherwise.  Dense-bf16 mode: bf16 operands (cast here unless the
    caller already holds the bf16 copies), fp32 accumulation, split over the tokens."""
    if bf16:
        return _wgrad_bf16(_b16(x2d), _b16(g2d), want_bias)
    if LINEAR_WGRAD_KERNEL and LinearLongTokens.kernel_wins(x2d, weight):
        gw = torch.empty_like(weight)
        gb = torch.empty((weight.shape[0],), dtype=torch.float32, device=x2d.device) if want_bias else None
        pointnet2.linear_wgrad(x2d, g2d, gw, gb, x2d.shape[0], weight.shape[1], weight.shape[0])
        return gw, gb
    return g2d.t().mm(x2d), (g2d.sum(0) if want_bias else None)


class TransformerBlock(Function):
    """TransformerEncoderLayerPreNorm.forward (PointFormer.py:28-38; dropout 0) on x (groups, seq, D) as ONE
    autograd node with a hand-scheduled backward: LayerNorm and attention on this repo's kernels, the
    projections on the library GEMMs, weight/bias gradients through `_wgrad`, and the two places where a
    tensor's gradient is the sum of a residual branch and a projection's input gradient handled without a
    separate T x D addition pass (accumulated by the GEMM in place, or summed inside the LayerNorm backward)."""

    @staticmethod
    def supported(x, heads):
        g, s_, d = x.shape
        return (LayerNormResidual.supported(x, d) and d % heads == 0 and s_ in GroupAttention.SUPPORTED_SEQ
                and d // heads in GroupAttention.SUPPORTED_HD and not torch.is_autocast_enabled())

    @staticmethod
    def forward(ctx, x, n1w, n1b, in_w, in_b, out_w, out_b, n2w, n2b, w1, b1, w2, b2, heads, eps1, eps2, pool):
        bf16 = _dense_bf16(x)
        x = x.contiguous()
        G, S, D = x.shape
        T, hd = G * S, D // heads
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        src1, ssum, src2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        st1, st2 = torch.empty((T, 2), **f32), torch.empty((T, 2), **f32)
        lse = torch.empty((G, heads, S), **f32)
        if bf16:
            # Dense-bf16 mode.  Every tensor that sits between a GEMM and one of this repo's kernels crosses HBM as
            # bf16: the GEMMs write bf16 with bias (+ ReLU) in their epilogue, the kernels read bf16 / emit the bf16
            # operand copy next to their fp32 output.  The residual stream (x, src1, ssum, src2), the statistics and
            # all kernel arithmetic stay fp32.  No cast or bias pass remains.
            b16 = dict(dtype=torch.bfloat16, device=dev)
            src1_s, src2_s = torch.empty((T, D), **b16), torch.empty((T, D), **b16)
            pointnet2.layer_norm_fwd(x, None, n1w, n1b, None, src1, st1, T, D, eps1, y_bf16=src1_s)
            qkv = torch.addmm(_b16p(in_b), src1_s, _b16p(in_w).t())
            a_s = torch.empty((T, D), **b16)
            pointnet2.group_attention_fwd(qkv, a_s, lse, G, S, heads, hd)
            proj = torch.addmm(_b16p(out_b), a_s, _b16p(out_w).t())
            pointnet2.layer_norm_fwd(proj, src1, n2w, n2b, ssum, src2, st2, T, D, eps2, y_bf16=src2_s)
            del proj
            h = h_s = torch._addmm_activation(_b16p(b1), src2_s, _b16p(w1).t())      # relu(src2 W1^T + b1)
            ffn = torch.addmm(_b16p(b2), h, _b16p(w2).t()).view(G, S, D)
        else:
            pointnet2.layer_norm_fwd(x, None, n1w, n1b, None, src1, st1, T, D, eps1)
            qkv = _gemm_nt(src1.view(T, D), in_w, in_b).view(G, S, 3 * D)
            a = torch.empty((G, S, D), **f32)
            pointnet2.group_attention_fwd(qkv, a, lse, G, S, heads, hd)
            proj = _gemm_nt(a.view(T, D), out_w, out_b).view(G, S, D)
            pointnet2.layer_norm_fwd(proj, src1, n2w, n2b, ssum, src2, st2, T, D, eps2)
            del proj
            h = _gemm_nt(src2.view(T, D), w1, b1, relu=True).view(G, S, -1)     # relu(src2 W1^T + b1), epilogue ReLU
            ffn = _gemm_nt(h.view(T, -1), w2, b2).view(G, S, D)
            src1_s, a_s, src2_s, h_s = src1.view(T, D), a.view(T, D), src2.view(T, D), h.view(T, -1)
        if pool:   # max over the tokens of a group of src2 + ffn, without materialising the sum
            y = torch.empty((G, D), **f32)
            arg = torch.empty((G, D), dtype=torch.uint8, device=dev)
            pointnet2.add_max_pool(src2, ffn, y, arg, G, S, D)
        else:
            y = src2 + ffn
            arg = torch.empty((0,), dtype=torch.uint8, device=dev)
        # backward needs src1 / a / src2 only as weight-gradient operands: in dense-bf16 mode their bf16 copies suffice
        ctx.save_for_backward(x, st1, src1_s, qkv, lse, a_s, ssum, st2, src2_s, h, h_s, n1w, in_w, out_w, n2w, w1, w2, arg)
        ctx.heads, ctx.pool, ctx.bf16 = heads, pool, bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st1, src1_s, qkv, lse, a_s, ssum, st2, src2_s, h, h_s, n1w, in_w, out_w, n2w, w1, w2, arg = ctx.saved_tensors
        G, S, D = x.shape
        T, heads, bf16 = G * S, ctx.heads, ctx.bf16
        hd = D // heads
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        b16 = dict(dtype=torch.bfloat16, device=dev)
        # Each gradient below feeds two GEMMs (an input gradient and a weight gradient).  In dense-bf16 mode `*_g` is
        # its bf16 copy, written by the kernel that produces the gradient (or the GEMM itself); otherwise the tensor.
        if ctx.pool:   # dy is (G, D): route it to the arg-max tokens (dense (T, D) gradient, written once)
            dy2 = torch.empty((T, D), **f32)
            dy_g = torch.empty((T, D), **b16) if bf16 else dy2
            pointnet2.max_pool_scatter(dy.contiguous(), arg, dy2, G, S, D, grad_x_bf16=dy_g if bf16 else None)
        else:
            dy2 = dy.contiguous().view(T, D)
            dy_g = _b16(dy2) if bf16 else dy2
        # y = src2 + h W2^T + b2
        d_h = torch.mm(dy_g, _b16p(w2)) if bf16 else _gemm_nn(dy_g, w2)
        if bf16 and ctx.pool:   # the scattered gradient's column sums are those of the (G, D) tensor it was scattered from
            gw2, gb2 = _wgrad(h_s, dy_g, w2, False, True)[0], dy.sum(0)
        else:
            gw2, gb2 = _wgrad(h_s, dy_g, w2, True, bf16)
        del dy_g
        d_h = torch.ops.aten.threshold_backward(d_h, h.view(T, -1), 0)
        # h = relu(src2 W1^T + b1); the gradient of src2 is dy (residual branch) + d_h W1: the LayerNorm backward
        # kernel adds its two incoming gradients on the fly (torch.addmm would first copy dy into its output)
        gw1, gb1 = _wgrad(src2_s, d_h, w1, True, bf16)
        d_lin1 = torch.mm(d_h, _b16p(w1)) if bf16 else _gemm_nn(d_h, w1)
        del d_h
        # src2 = LayerNorm2(ssum), ssum = src1 + a Wo^T + bo
        d_s = torch.empty((T, D), **f32)
        ds_g = torch.empty((T, D), **b16) if bf16 else d_s
        gn2w, gn2b = torch.empty_like(n2w), torch.empty_like(n2w)
        scratch = torch.empty((pointnet2.layer_norm_scratch_bytes(D),), dtype=torch.uint8, device=dev)
        pointnet2.layer_norm_bwd(ssum, dy2, n2w, st2, d_s, gn2w, gn2b, scratch, T, D, grad_y2=d_lin1, grad_x_bf16=ds_g if bf16 else None)
        del d_lin1
        d_a = torch.mm(ds_g, _b16p(out_w)) if bf16 else _gemm_nn(d_s, out_w)
        gwo, gbo = _wgrad(a_s, ds_g, out_w, True, bf16)
        del ds_g
        dqkv = torch.empty_like(qkv)
        pointnet2.group_attention_bwd(qkv, d_a.view(G, S, D), lse, dqkv, G, S, heads, hd)
        del d_a
        dqkv2 = dqkv.view(T, 3 * D)
        gwi, gbi = _wgrad(src1_s, dqkv2, in_w, True, bf16)
        d_src1 = _mm_nn(dqkv2, in_w, True, acc=d_s) if bf16 else _gemm_nn(dqkv2, in_w, acc=d_s)   # d_s + dqkv Win, accumulated (d_s is ours)
        del dqkv, dqkv2
        d_x = torch.empty_like(x)
        gn1w, gn1b = torch.empty_like(n1w), torch.empty_like(n1w)
        pointnet2.layer_norm_bwd(x, d_src1, n1w, st1, d_x, gn1w, gn1b, scratch, T, D)
        return d_x, gn1w, gn1b, gwi, gbi, gwo, gbo, gn2w, gn2b, gw1, gb1, gw2, gb2, None, None, None, None


def transformer_block(tr, x, pool=False):
    """TransformerBlock on the parameters of a TransformerEncoderLayerPreNorm module; pool=True returns the max over
    the sequence dimension, (groups, D), instead of (groups, seq, D)."""
    at = tr.self_attn
    return TransformerBlock.apply(x, tr.norm1.weight, tr.norm1.bias, at.in_proj_weight, at.in_proj_bias, at.out_proj.weight,
                                  at.out_proj.bias, tr.norm2.weight, tr.norm2.bias, tr.linear1.weight, tr.linear1.bias,
                                  tr.linear2.weight, tr.linear2.bias, at.num_heads, tr.norm1.eps, tr.norm2.eps, pool)


# ---- training form of the vanilla-SA group MLP on this repo's MFMA code (csrc/sa_mlp.hip, lin_cols_kernel) -----------
# SA_MFMA_TRAIN: the contractions of the set-abstraction MLPs -- forward and input gradient -- run on the hand-written
# f32 MFMA kernels instead of the library GEMMs (the weight gradient is csrc/wgrad.hip already); layer 1 gathers its
# grouped input in the kernel, so the (B, M, ns, 3 + C) tensor only exists in the backward pass.
SA_MFMA_TRAIN = True
SA_MFMA_EVENTS = None      # bench: a list collecting (event0, event1, flops, pipe) per MFMA launch of the SA group MLPs


def _sa_timed(flops, fn, pipe="f32"):
    """pipe: "f32" = v_mfma_f32_32x32x2_f32, "bf16x6" = six v_mfma_f32_32x32x16_bf16 per f32 product block (split GEMMs)."""
    ev = SA_MFMA_EVENTS
    if ev is None or torch.cuda.is_current_stream_capturing():     # (a timing event inside a capture is a node, not a clock)
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    ev.append((e0, e1, flops, pipe))




def _lin_cols(x2, weight, y, T, k, n_out, transposed):
    """y (T, n_out) = x2 (T, k) W'^T with W' = weight (n_out, k), or weight^T when `transposed` (weight is (k, n_out))."""
    if SPLIT_GEMM:
        wf = _split_planes(weight, transposed)       # `weight` is the parameter's (out, in) source either way
        if _rows_in_registers(T, k, n_out):
            _sa_timed(2.0 * T * k * n_out, lambda: pointnet2.linear_split(x2, wf, None, y, T, k, n_out), "bf16x6")
        else:
            _sa_timed(2.0 * T * k * n_out, lambda: pointnet2.gemm_split(x2, wf, None, y, T, k, n_out), "bf16x6")
    else:
        wf = pointnet2.linear_cols_pack(weight, n_out, k, transposed_source=transposed)
        _sa_timed(2.0 * T * k * n_out, lambda: pointnet2.linear_cols(x2, wf, y, T, k, n_out))


class LinearColsMFMA(Function):
    """y = x W^T (no bias) over the last dim of x (..., K): forward and input gradient on lin_cols_kernel, weight
    gradient on csrc/wgrad.hip (or the library below its break-even)."""

    @staticmethod
    def supported(x, weight):
        n_out, k = weight.shape
        return (SA_MFMA_TRAIN and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and k in (256, 512)
                and n_out % 128 == 0 and n_out <= 1024 and x.shape[-1] == k and not torch.is_autocast_enabled() and not DENSE_BF16)

    @staticmethod
    def forward(ctx, x, weight):
        n_out, k = weight.shape
        x2 = x.contiguous().view(-1, k)
        T = x2.shape[0]
        y = torch.empty((T, n_out), dtype=torch.float32, device=x.device)
        _lin_cols(x2, weight.contiguous(), y, T, k, n_out, False)
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, grad_out):
        x2, weight = ctx.saved_tensors
        n_out, k = weight.shape
        g2 = grad_out.contiguous().view(-1, n_out)
        T = g2.shape[0]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if n_out in (256, 512) and k % 128 == 0:
                gx = torch.empty((T, k), dtype=torch.float32, device=g2.device)
                # dX = dY W: the same kernel with the weights packed from the transposed source
                _lin_cols(g2, weight.contiguous(), gx, T, n_out, k, True)
            else:
                gx = _gemm_nn(g2, weight)
            gx = gx.view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            gw = _wgrad(x2, g2, weight, False)[0]
        return gx, gw


class SaGatherLinear(Function):
    """Layer 1 of a vanilla SA scale with the grouping fused in: z1 (B, M, ns, C1) = [xyz[idx] - new_xyz | feats[idx]] W1^T
    (QueryAndGroup + the first 1x1 convolution, pointnet2_utils.py:671-704 + pointnet2_modules.py:1657).  The grouped
    input is rebuilt only in the backward pass (weight gradient), where the input gradient goes through the library
    (3 + C = 259 columns) and on to the features (scatter-add) and the centres."""

    @staticmethod
    def supported(xyz, feats_pm, weight):
        return (SA_MFMA_TRAIN and xyz.is_cuda and feats_pm is not None and feats_pm.dtype == torch.float32 and feats_pm.shape[-1] == 256
                and weight.shape[1] == 259 and weight.shape[0] % 128 == 0 and weight.shape[0] <= 1024
                and not torch.is_autocast_enabled() and not DENSE_BF16)

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_pm, idx, weight):
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        C, n_out = feats_pm.shape[-1], weight.shape[0]
        xyz, new_xyz, feats_pm = xyz.contiguous(), new_xyz.contiguous(), feats_pm.contiguous()
        y = torch.empty((B, M, ns, n_out), dtype=torch.float32, device=xyz.device)
        wf = pointnet2.linear_cols_pack(weight.contiguous(), n_out, 3 + C, gather_order=True)
        _sa_timed(2.0 * B * M * ns * (3 + C) * n_out,
                  lambda: pointnet2.sa_gather_linear(xyz, new_xyz, feats_pm, idx, wf, y, B, N, M, C, ns, n_out))
        ctx.save_for_backward(xyz, new_xyz, feats_pm, idx, weight)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        """By column block of W1 = [W_xyz (n_out, 3) | W_f (n_out, C)]: the feature block is a plain (tokens x C) problem --
        weight gradient on csrc/wgrad.hip from the re-gathered neighbour rows, input gradient on csrc/gemm_split.hip, then the
        scatter-add onto the points; the coordinate block (weight gradient and the centres' gradient) is one streaming
        pass over grad_out (csrc/sa_xyz_grad.hip).  No 259-column tensor is built and nothing goes to the library."""
        xyz, new_xyz, feats_pm, idx, weight = ctx.saved_tensors
        B, M, ns = idx.shape
        n_out, C, N = weight.shape[0], feats_pm.shape[-1], xyz.shape[1]
        g2 = grad_out.contiguous().view(-1, n_out)
        T = g2.shape[0]
        g_new = g_feats = gw = None
        with torch.no_grad():
            if ctx.needs_input_grad[4] or ctx.needs_input_grad[1]:
                gw = torch.empty_like(weight)
                g_new = torch.empty_like(new_xyz) if ctx.needs_input_grad[1] else None
                pointnet2.sa_xyz_grad(g2, xyz, new_xyz, idx, weight.contiguous(), gw, g_new, B, N, M, ns, n_out)
            if ctx.needs_input_grad[4]:
                gf_in = group_rows(feats_pm, idx).view(T, C)            # the grouped features exist for this product only
                gw[:, 3:] = _wgrad(gf_in, g2, torch.empty((n_out, C), dtype=weight.dtype, device=weight.device), False)[0]
                del gf_in
            else:
                gw = None
            if ctx.needs_input_grad[2]:
                gxf = _gemm_nn(g2, weight[:, 3:].contiguous())           # (T, C)
                g_feats = torch.zeros_like(feats_pm)
                pointnet2.group_rows_grad(B, feats_pm.shape[1], C, M * ns, gxf, idx, g_feats)
        return None, g_new, g_feats, None, gw


# SA_POINT_LINEAR: the first layer of a wide SA scale as a per-POINT projection + a gather (a linear layer commutes with the
# gather): z1[token] = (F W_f^T)[idx] + W_xyz (xyz[idx] - centre).  Forward: one (B N, C) x (C, c1) product instead of a
# (B M ns, 3 + C) one -- 4096 rows instead of 229 376 at ONCE layer 5 -- and a row gather; backward: the token gradients are
# scatter-added onto the points FIRST (the scatter the grouping gradient needs anyway), then weight and feature gradients
# are (B N)-row products.  The per-token contraction, its weight gradient and its input gradient (3 x 30 GFLOP per step) are gone.
SA_POINT_LINEAR = os.environ.get("PDA_SA_POINT_LINEAR", "1") != "0"


class SaPointLinear(Function):
    """SaGatherLinear's contract -- z1 (B, M, ns, C1) = [xyz[idx] - new_xyz | feats[idx]] W1^T -- computed per point."""

    @staticmethod
    def supported(xyz, feats_pm, weight):
        return (SA_POINT_LINEAR and SPLIT_GEMM and SaGatherLinear.supported(xyz, feats_pm, weight) and weight.shape[0] in (128, 256, 512, 1024)
                and feats_pm.shape[0] * feats_pm.shape[1] >= SPLIT_GEMM_MIN_TOKENS)

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_pm, idx, weight):
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        C, n_out = feats_pm.shape[-1], weight.shape[0]
        xyz, new_xyz, feats_pm, idx = xyz.contiguous(), new_xyz.contiguous(), feats_pm.contiguous(), idx.contiguous()
        w = weight.detach().contiguous()
        w_f = w[:, 3:].contiguous()
        rows = _gemm_nt(feats_pm.view(B * N, C), w_f)                   # (B N, c1): the feature part, once per point
        z = torch.empty((B, M, ns, n_out), dtype=torch.float32, device=xyz.device)
        pointnet2.sa_point_gather(rows, xyz, new_xyz, idx, w, z, B, N, M, ns, n_out)
        ctx.save_for_backward(xyz, new_xyz, feats_pm, idx, weight)
        return z

    @staticmethod
    def backward(ctx, grad_out):
        xyz, new_xyz, feats_pm, idx, weight = ctx.saved_tensors
        B, M, ns = idx.shape
        n_out, C, N = weight.shape[0], feats_pm.shape[-1], xyz.shape[1]
        g2 = grad_out.contiguous().view(-1, n_out)
        g_new = g_feats = gw = None
        with torch.no_grad():
            w = weight.contiguous()
            if ctx.needs_input_grad[4] or ctx.needs_input_grad[1]:
                gw = torch.empty_like(weight)
                g_new = torch.empty_like(new_xyz) if ctx.needs_input_grad[1] else None
                pointnet2.sa_xyz_grad(g2, xyz, new_xyz, idx, w, gw, g_new, B, N, M, ns, n_out)     # coordinate columns, centres
            if ctx.needs_input_grad[4] or ctx.needs_input_grad[2]:
                g_pts = torch.zeros((B, N, n_out), dtype=torch.float32, device=g2.device)
                pointnet2.group_rows_grad(B, N, n_out, M * ns, g2, idx, g_pts)                    # token gradients onto their points
                g_pts = g_pts.view(B * N, n_out)
                if ctx.needs_input_grad[4]:
                    gw[:, 3:] = _wgrad(feats_pm.view(B * N, C), g_pts, torch.empty((n_out, C), dtype=weight.dtype, device=weight.device), False)[0]
                if ctx.needs_input_grad[2]:
                    g_feats = _gemm_nn(g_pts, w[:, 3:].contiguous()).view(B, N, C)
            if not ctx.needs_input_grad[4]:
                gw = None
        return None, g_new, g_feats, None, gw


SA_WIDE_INFER_SPLIT = os.environ.get("PDA_SA_WIDE_INFER_SPLIT", "1") != "0"
SA_WIDE_INFER_GATHER_IN_GEMM = os.environ.get("PDA_SA_WIDE_INFER_GATHER_IN_GEMM", "1") != "0"


def sa_wide_scale_infer(xyz, new_xyz, feats_pm, idx, folded):
    """One WIDE vanilla SA scale in inference (BatchNorm folded into the convolutions: `folded` = [(W, b)] x 3): the first
    layer as per-point projection + row gather with bias and ReLU (sa_point_gather), layers 2 and 3 on the split-bf16 GEMMs
    with bias + ReLU in their epilogue, then the max over nsample.  172 GFLOP of ONCE layer 5 become 142 on the bf16 matrix
    cores (f32-grade) instead of 172 on the f32-input MFMA of the fused kernel.  Returns (B, M, c3) or None (shape not covered)."""
    (w1, b1), (w2, b2), (w3, b3) = folded
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = feats_pm.shape[-1]
    T = B * M * ns
    c1, c2, c3 = w1.shape[0], w2.shape[0], w3.shape[0]
    if not (SA_WIDE_INFER_SPLIT and SPLIT_GEMM and not DENSE_BF16 and w1.shape[1] == 3 + C and C % 32 == 0 and c1 in (128, 256, 512, 1024)
            and c1 % 32 == 0 and c2 % 128 == 0 and c3 % 128 == 0 and B * N >= SPLIT_GEMM_MIN_TOKENS and T >= SPLIT_GEMM_MIN_TOKENS):
        return None
    out = []

    def run():
        rows = _gemm_nt(feats_pm.reshape(B * N, C), w1[:, 3:].contiguous())
        if SA_WIDE_INFER_GATHER_IN_GEMM and c1 % 32 == 0 and c1 <= 1024:
            # the first layer's output is formed in the operand load of the second contraction: (T, c1) is never written
            y2 = torch.empty((T, c2), dtype=torch.float32, device=xyz.device)
            pointnet2.gemm_split_gather(rows, xyz.contiguous(), new_xyz.contiguous(), idx.contiguous(), w1.contiguous(), b1.detach().contiguous(),
                                        _split_planes(w2), b2.detach().contiguous(), y2, B, N, M, ns, c1, c2, relu=True)
        else:
            y1 = torch.empty((T, c1), dtype=torch.float32, device=xyz.device)
            pointnet2.sa_point_gather(rows, xyz.contiguous(), new_xyz.contiguous(), idx.contiguous(), w1, y1, B, N, M, ns, c1, bias=b1, relu=True)
            y2 = _gemm_nt(y1, w2, b2, relu=True)
            del y1
        if ns in (16, 32, 64) and c2 % 32 == 0:
            # the max over nsample in the epilogue of the last contraction: (T, c3) is never written
            pooled = torch.empty((B, M, c3), dtype=torch.float32, device=xyz.device)
            pointnet2.gemm_split_maxpool(y2, _split_planes(w3), b3.detach().contiguous(), pooled, T, c2, c3, ns, relu=True)
            out.append(pooled)
            return
        y3 = _gemm_nt(y2, w3, b3, relu=True)
        del y2
        out.append(y3.view(B, M, ns, c3).amax(dim=2))
    # bench: (layers 2 + 3 flops, the part that runs on the bf16 pipe), timed over the whole scale incl. gather and max-pool
    _sa_timed(2.0 * T * (c1 * c2 + c2 * c3), run, "bf16x6_infer")
    return out[0]


# ---- the narrow vanilla SA scale (layer 0) in training form as recompute passes (csrc/sa_train_small.hip) ------------
# SA_SMALL_TRAIN: group -> [conv1x1 -> BN(batch statistics) -> ReLU] x 3 -> max of a scale whose widths are <= 64 runs as
# four forward and four backward passes over the neighbour lists; no (B, M, ns, C) tensor exists in the forward pass.
SA_SMALL_TRAIN = True
FUSED_WIDE_CHAIN_OK = True      # (tests switch the wide chain off to compare it with the unfused path)


class SaSmallChainTrain(Function):
    """One scale of PointnetSAModuleMSG_WithSampling (pointnet2_modules.py:1657-1670) for the chains of layer 0
    (4 -> 16 -> 16 -> 32 over 16 neighbours, 4 -> 32 -> 32 -> 64 over 32): out (B, M, c3) = max over nsample of the MLP of
    [xyz[idx] - new_xyz | feats[idx]].  Gradients: the three convolution weights and the three BatchNorms' affine
    parameters; the gathered inputs are the raw points (no gradient)."""

    @staticmethod
    def supported(xyz, new_xyz, feats_pm, idx, mlp):
        layers = list(mlp)
        if not (SA_SMALL_TRAIN and len(layers) == 9 and xyz.is_cuda and xyz.dtype == torch.float32 and not DENSE_BF16
                and not torch.is_autocast_enabled() and not xyz.requires_grad and not new_xyz.requires_grad
                and (feats_pm is None or (feats_pm.dtype == torch.float32 and not feats_pm.requires_grad))):
            return False
        for k in range(3):
            conv, bn, act = layers[3 * k:3 * k + 3]
            if not (isinstance(conv, nn.Conv2d) and conv.bias is None and conv.kernel_size == (1, 1) and isinstance(bn, nn.BatchNorm2d)
                    and isinstance(act, nn.ReLU) and bn.training and bn.affine and bn.momentum is not None
                    and conv.weight.dtype == torch.float32):
                return False
        c = 0 if feats_pm is None else feats_pm.shape[-1]
        c1, c2, c3 = (layers[3 * k].out_channels for k in range(3))
        return layers[0].in_channels == 3 + c and pointnet2.sa_small_train_supported(c, idx.shape[2], c1, c2, c3, idx.numel())

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_pm, idx, w1, w2, w3, g1, b1, g2, b2, g3, b3, bns):
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        c = 0 if feats_pm is None else feats_pm.shape[-1]
        c3 = w3.shape[0]
        dev = xyz.device
        xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
        feats_pm = None if feats_pm is None else feats_pm.contiguous()
        ws = torch.empty((pointnet2.sa_small_train_workspace_bytes(),), dtype=torch.uint8, device=dev)
        out = torch.empty((B, M, c3), dtype=torch.float32, device=dev)
        zmax = torch.empty((B, M, c3), dtype=torch.float32, device=dev)
        arg = torch.empty((B, M, c3), dtype=torch.uint8, device=dev)
        track = [bn.track_running_stats for bn in bns]
        macs = w1.numel() + w2.numel() + w3.numel()            # algorithmic: one pass over the chain per token
        _sa_timed(2.0 * B * M * ns * macs, lambda: pointnet2.sa_small_train_fwd(
            xyz, new_xyz, feats_pm, idx, (w1.contiguous(), w2.contiguous(), w3.contiguous()), (g1, g2, g3), (b1, b2, b3),
            [bn.running_mean if t else None for bn, t in zip(bns, track)],
            [bn.running_var if t else None for bn, t in zip(bns, track)],
            [bn.eps for bn in bns], [bn.momentum for bn in bns], ws, out, zmax, arg, B, N, M, c, ns), "f32_recompute")
        ctx.save_for_backward(xyz, new_xyz, feats_pm, idx, zmax, arg, ws, w1, w2, w3)
        ctx.dims = (B, N, M, c, ns)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xyz, new_xyz, feats_pm, idx, zmax, arg, ws, w1, w2, w3 = ctx.saved_tensors
        B, N, M, c, ns = ctx.dims
        tokens = B * M * ns
        dev = xyz.device
        dz2 = torch.empty((tokens, w2.shape[0]), dtype=torch.float32, device=dev)
        dz1 = torch.empty((tokens, w1.shape[0]), dtype=torch.float32, device=dev)
        dws = [torch.empty(w.shape, dtype=torch.float32, device=dev) for w in (w1, w2, w3)]
        dgs = [torch.empty((w.shape[0],), dtype=torch.float32, device=dev) for w in (w1, w2, w3)]
        dbs = [torch.empty((w.shape[0],), dtype=torch.float32, device=dev) for w in (w1, w2, w3)]
        gout = grad_out.contiguous().float()
        macs = w1.numel() + w2.numel() + w3.numel()            # algorithmic backward: weight gradients + input gradients of layers 2, 3
        _sa_timed(2.0 * tokens * (2 * macs - w1.numel()), lambda: pointnet2.sa_small_train_bwd(
            xyz, new_xyz, feats_pm, idx, gout, zmax, arg, ws, dz2, dz1, dws, dgs, dbs, B, N, M, c, ns), "f32_recompute")
        return (None, None, None, None, dws[0], dws[1], dws[2], dgs[0], dbs[0], dgs[1], dbs[1], dgs[2], dbs[2], None)


def sa_small_chain_train(xyz, new_xyz, feats_pm, idx, mlp):
    """The scale's [conv -> BN -> ReLU] x 3 nn.Sequential `mlp` applied through SaSmallChainTrain: (B, M, c3)."""
    layers = list(mlp)
    convs, bns = [layers[0], layers[3], layers[6]], [layers[1], layers[4], layers[7]]
    for bn in bns:
        bump_bn_counter(bn)
    return SaSmallChainTrain.apply(xyz, new_xyz, feats_pm, idx, convs[0].weight.flatten(1), convs[1].weight.flatten(1),
                                   convs[2].weight.flatten(1), bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias,
                                   bns[2].weight, bns[2].bias, bns)


# ---- the WIDE vanilla SA scale (layer 5) in training form with the BatchNorm passes folded into the contractions ------------
# SA_WIDE_CHAIN: group -> [conv1x1 -> BN(batch statistics) -> ReLU] x 3 -> max of a scale whose first layer is the per-point
# projection of SaPointLinear and whose widths are multiples of 256.  Between the three contractions only the PRE-BatchNorm
# tensors z1, z2, z3 reach HBM: a layer's statistics come out of the epilogue of the GEMM that produces it and its normalised
# activation is formed in the operand load of the GEMM (and of the weight gradient) that consumes it
# (pda_gemm_split_bn, pda_linear_wgrad_bn): per mid layer the forward statistics pass, the normalise pass and the activation's
# write + two reads go.  (The reduction pass of the BatchNorm BACKWARD in the epilogue of the input-gradient GEMM was built and
# measured too: reading z tile by tile behind a 256 x 256 tile's stores costs more than the standalone pass, +0.23 ms on
# 131072 x 512 x 512 against 0.10 ms, and was removed.)
SA_WIDE_CHAIN = os.environ.get("PDA_SA_WIDE_CHAIN", "1") != "0"
WIDE_CHAIN_DGRAD_FIRST = os.environ.get("PDA_WIDE_CHAIN_DGRAD_FIRST", "1") != "0"
SA_WIDE_CHAIN_MIN_TOKENS = int(os.environ.get("PDA_SA_WIDE_CHAIN_MIN_TOKENS", "32768"))   # (at 32768 tokens the 256 x 256 tiles take 50 us against lin_split's 35 and the passes saved are still worth more; at 16384 they are not)


class SaWideChainTrain(Function):
    """One scale of PointnetSAModuleMSG_WithSampling (pointnet2_modules.py:1657-1670): out (B, M, c3) = max over nsample of the
    three-layer MLP of [xyz[idx] - new_xyz | feats[idx]], training-mode BatchNorm.  Same results as SaPointLinear +
    BatchNormReLU + the split GEMMs + BatchNormReLUMaxPool up to the order of the statistics' sums."""

    @staticmethod
    def supported(xyz, new_xyz, feats_pm, idx, mlp):
        layers = list(mlp)
        if not (SA_WIDE_CHAIN and FUSED_WIDE_CHAIN_OK and len(layers) == 9 and feats_pm is not None and idx is not None):
            return False
        convs, bns = [layers[0], layers[3], layers[6]], [layers[1], layers[4], layers[7]]
        if not (all(isinstance(c, nn.Conv2d) and c.bias is None and c.weight.dtype == torch.float32 for c in convs)
                and all(isinstance(b, nn.BatchNorm2d) and b.affine and b.training and b.momentum is not None for b in bns)
                and all(isinstance(layers[k], nn.ReLU) for k in (2, 5, 8))):
            return False
        w1 = convs[0].weight.flatten(1)
        if not SaPointLinear.supported(xyz, feats_pm, w1):
            return False
        c1, c2, c3 = (c.weight.shape[0] for c in convs)
        T = idx.shape[0] * idx.shape[1] * idx.shape[2]
        if not (c1 % 256 == 0 and c2 % 256 == 0 and c3 % 256 == 0 and max(c1, c2, c3) <= 1024 and 1 <= idx.shape[2] <= 255
                and T >= SA_WIDE_CHAIN_MIN_TOKENS and not DENSE_BF16 and not torch.is_autocast_enabled()):
            return False
        lib = _lib.load()
        return int(lib.pda_linear_wgrad_form(T, c1, c2)) == 2 and int(lib.pda_linear_wgrad_form(T, c2, c3)) == 2

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_pm, idx, w1, w2, w3, g1, b1, g2, b2, g3, b3, bns):
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        C, c1, c2, c3 = feats_pm.shape[-1], w1.shape[0], w2.shape[0], w3.shape[0]
        T = B * M * ns
        dev = xyz.device
        xyz, new_xyz, feats_pm, idx = xyz.contiguous(), new_xyz.contiguous(), feats_pm.contiguous(), idx.contiguous()
        gs = [t.detach().contiguous() for t in (g1, g2, g3)]
        bs = [t.detach().contiguous() for t in (b1, b2, b3)]
        stats = [(bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None) for bn in bns]
        w1d = w1.detach().contiguous()
        # layer 1: per-point projection + row gather (SaPointLinear), then its statistics
        rows = _gemm_nt(feats_pm.view(B * N, C), w1d[:, 3:].contiguous())
        z1 = torch.empty((T, c1), dtype=torch.float32, device=dev)
        pointnet2.sa_point_gather(rows, xyz, new_xyz, idx, w1d, z1, B, N, M, ns, c1)
        mi1 = torch.empty((2 * c1,), dtype=torch.float32, device=dev)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c1),), dtype=torch.uint8, device=dev)
        pointnet2.bn_stats_fwd(z1, stats[0][0], stats[0][1], mi1, scratch, T, c1, bns[0].eps, bns[0].momentum)
        # layers 2 and 3: z = relu(bn(z_prev)) W^T with the statistics of z in the epilogue
        tiles = pointnet2.gemm_split_bn_tiles(T)

        def layer(z_in, mi_in, k, w, n_out, bn, st, gi, bi):
            z = torch.empty((T, n_out), dtype=torch.float32, device=dev)
            part = torch.empty((tiles * 2 * n_out,), dtype=torch.float64, device=dev)
            wf = _split_planes(w)
            _sa_timed(2.0 * T * k * n_out, lambda: pointnet2.gemm_split_bn(z_in, wf, z, T, k, n_out, in_bn=(mi_in, gi, bi), stats_mode=1,
                                                                           partial=part), "bf16x6")
            mi = torch.empty((2 * n_out,), dtype=torch.float32, device=dev)
            pointnet2.bn_finalize_fwd(part, tiles, n_out, T, bn.eps, bn.momentum, mi, st[0], st[1])
            return z, mi
        z2, mi2 = layer(z1, mi1, c1, w2, c2, bns[1], stats[1], gs[0], bs[0])
        z3, mi3 = layer(z2, mi2, c2, w3, c3, bns[2], stats[2], gs[1], bs[1])
        out = torch.empty((B, M, c3), dtype=torch.float32, device=dev)
        arg = torch.empty((B, M, c3), dtype=torch.uint8, device=dev)
        pointnet2.bn_relu_max_pool_apply(z3, gs[2], bs[2], mi3, out, arg, B * M, ns, c3)
        ctx.save_for_backward(xyz, new_xyz, feats_pm, idx, w1, w2, w3, g1, b1, g2, b2, g3, b3, z1, z2, z3, mi1, mi2, mi3, arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xyz, new_xyz, feats_pm, idx, w1, w2, w3, g1, b1, g2, b2, g3, b3, z1, z2, z3, mi1, mi2, mi3, arg = ctx.saved_tensors
        B, M, ns = idx.shape
        N, C = xyz.shape[1], feats_pm.shape[-1]
        c1, c2, c3 = w1.shape[0], w2.shape[0], w3.shape[0]
        T = B * M * ns
        dev = xyz.device
        tiles = pointnet2.gemm_split_bn_tiles(T)
        with torch.no_grad():
            g1c, b1c, g2c, b2c, g3c, b3c = (t.contiguous() for t in (g1, b1, g2, b2, g3, b3))
            # BatchNorm 3 + max-pool: the gradient reaches one row per (group, channel)
            gz3 = torch.empty((T, c3), dtype=torch.float32, device=dev)
            dg3, db3 = torch.empty_like(g3c), torch.empty_like(b3c)
            scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c3),), dtype=torch.uint8, device=dev)
            pointnet2.bn_relu_max_pool_bwd(z3, grad_out.contiguous().float(), arg, g3c, b3c, mi3, gz3, dg3, db3, scratch, B * M, ns, c3)

            def layer(gz, w, z_in, mi_in, gi, bi, k_in, n_out):
                """gz (T, n_out) = gradient at this layer's output z; returns (dW, gradient at z_in, dgamma_in, dbeta_in)."""
                dw = torch.empty((n_out, k_in), dtype=torch.float32, device=dev)
                ga = torch.empty((T, k_in), dtype=torch.float32, device=dev)
                if WIDE_CHAIN_DGRAD_FIRST:
                    _lin_cols(gz, w, ga, T, n_out, k_in, True)                                # ga = gz W (timed as a bf16x6 launch)
                    pointnet2.linear_wgrad_bn(z_in, gz, dw, T, k_in, n_out, mi_in, gi, bi)
                else:
                    pointnet2.linear_wgrad_bn(z_in, gz, dw, T, k_in, n_out, mi_in, gi, bi)
                    _lin_cols(gz, w, ga, T, n_out, k_in, True)
                dg, db = torch.empty_like(gi), torch.empty_like(bi)
                scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(k_in),), dtype=torch.uint8, device=dev)
                pointnet2.bn_relu_bwd(z_in, ga, gi, bi, mi_in, ga, dg, db, scratch, T, k_in)   # in place: a thread reads what it writes
                return dw, ga, dg, db
            dw3, gz2, dg2, db2 = layer(gz3, w3, z2, mi2, g2c, b2c, c2, c3)
            del gz3
            dw2, gz1, dg1, db1 = layer(gz2, w2, z1, mi1, g1c, b1c, c1, c2)
            del gz2
            # layer 1 (SaPointLinear.backward): coordinate columns and centres, then the feature columns through the points
            w1c = w1.contiguous()
            dw1 = torch.empty_like(w1c)
            g_new = torch.empty_like(new_xyz) if ctx.needs_input_grad[1] else None
            pointnet2.sa_xyz_grad(gz1, xyz, new_xyz, idx, w1c, dw1, g_new, B, N, M, ns, c1)
            g_pts = torch.zeros((B, N, c1), dtype=torch.float32, device=dev)
            pointnet2.group_rows_grad(B, N, c1, M * ns, gz1, idx, g_pts)
            g_pts = g_pts.view(B * N, c1)
            dw1[:, 3:] = _wgrad(feats_pm.view(B * N, C), g_pts, torch.empty((c1, C), dtype=torch.float32, device=dev), False)[0]
            g_feats = _gemm_nn(g_pts, w1c[:, 3:].contiguous()).view(B, N, C) if ctx.needs_input_grad[2] else None
        return None, g_new, g_feats, None, dw1, dw2, dw3, dg1, db1, dg2, db2, dg3, db3, None


def sa_wide_chain_train(xyz, new_xyz, feats_pm, idx, mlp):
    """The scale's [conv -> BN -> ReLU] x 3 nn.Sequential `mlp` applied through SaWideChainTrain: (B, M, c3)."""
    layers = list(mlp)
    convs, bns = [layers[0], layers[3], layers[6]], [layers[1], layers[4], layers[7]]
    for bn in bns:
        bump_bn_counter(bn)
    return SaWideChainTrain.apply(xyz, new_xyz, feats_pm, idx, convs[0].weight.flatten(1), convs[1].weight.flatten(1),
                                  convs[2].weight.flatten(1), bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias,
                                  bns[2].weight, bns[2].bias, bns)


# ---- unique-token ("ragged") execution of a PDA scale (csrc/ragged.hip) --------------------------------------------
# RAGGED_TOKENS: run the encoder of a PDA scale on the distinct (centre, neighbour) tokens only.  RAGGED_MAX_FRACTION:
# use it when the distinct tokens are at most this share of B*M*nsample (above it the bookkeeping costs more than it saves).
RAGGED_TOKENS = True
RAGGED_MAX_FRACTION = 0.85


class RaggedPlan:
    """cnt (G) / off (G+1) / rowmap (U) of one scale (device tensors) and U, the number of distinct tokens (host int)."""
    __slots__ = ("cnt", "off", "rowmap", "roww", "tokens", "groups", "nsample")

    def __init__(self, cnt, off, rowmap, roww, tokens, groups, nsample):
        self.cnt, self.off, self.rowmap, self.roww = cnt, off, rowmap, roww      # roww: multiplicity of each compact token
        self.tokens, self.groups, self.nsample = tokens, groups, nsample

    @property
    def fraction(self):
        return self.tokens / float(self.groups * self.nsample)


def ragged_plan_parts(idxs):
    """Device half of ragged_plans: per index tensor (cnt, off, rowmap, groups, nsample) and the stacked token counts
    (a device tensor).  No synchronisation."""
    parts = []
    for idx in idxs:
        G, ns = idx.shape[0] * idx.shape[1], idx.shape[2]
        cnt = torch.empty((G,), dtype=torch.int32, device=idx.device)
        off = torch.empty((G + 1,), dtype=torch.int32, device=idx.device)
        rowmap = torch.empty((G * ns,), dtype=torch.int32, device=idx.device)
        roww = torch.empty((G * ns,), dtype=torch.float32, device=idx.device)
        pointnet2.ragged_plan(idx, cnt, off, rowmap, G, ns, roww)
        parts.append((cnt, off, rowmap, roww, G, ns))
    return parts, torch.stack([p[1][-1] for p in parts])


def ragged_plans_from(parts, totals):
    return [RaggedPlan(c, o, r[:u], w[:u], int(u), G, ns) for (c, o, r, w, G, ns), u in zip(parts, totals)]


def ragged_plans(idxs):
    """One RaggedPlan per neighbour-index tensor (B, M, ns) of a layer.  ONE host synchronisation for all of them: the
    token counts size the GEMMs of the encoder (the reference's backbone synchronises once per forward as well,
    IASSD_backbone.py:134-137).  A layer whose centres depend on coordinates only gets its plans ahead of time on the
    sampling side stream instead (backbone._presample), with the counts copied to pinned memory."""
    parts, totals = ragged_plan_parts(idxs)
    return ragged_plans_from(parts, totals.tolist())


ASSEMBLE_BWD_TOKEN_PARALLEL = os.environ.get("PDA_ASSEMBLE_BWD_TOKEN_PARALLEL", "1") != "0"


class AssembleTokensRagged(Function):
    """AssembleTokens writing only the distinct tokens: x (U, 4C).  rppe is either dense (B,M,ns,C) -- its gradient then
    comes back dense with zeros at the repeat slots -- or compact (U, C) when the position MLP ran on the distinct tokens
    (RAGGED_POSITION_MLP); dscale is dense (B,M,ns,1) either way (DensityNet runs on all slots)."""

    @staticmethod
    def forward(ctx, rppe, dscale, feats_pm, idx, glob, plan):
        B, M, ns = idx.shape
        C, N = feats_pm.shape[-1], feats_pm.shape[1]
        compact = rppe.dim() == 2
        rppe, dscale, feats_pm, glob = rppe.contiguous(), dscale.contiguous(), feats_pm.contiguous(), glob.contiguous()
        out = torch.empty((plan.tokens, 4 * C), dtype=torch.float32, device=rppe.device)
        pointnet2.assemble_tokens_ragged(rppe, dscale, feats_pm, idx, glob, plan.rowmap, plan.off, out, plan.tokens, B, N, M, ns, C,
                                         rppe_compact=compact)
        ctx.save_for_backward(dscale, feats_pm, idx, plan.cnt, plan.off, plan.rowmap)
        ctx.dims = (B, N, M, ns, C, plan.tokens, compact)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dscale, feats_pm, idx, cnt, off, rowmap = ctx.saved_tensors
        B, N, M, ns, C, U, compact = ctx.dims
        dev = grad_out.device
        g_rppe = torch.empty((U, C) if compact else (B, M, ns, C), dtype=torch.float32, device=dev)
        g_ds = torch.empty_like(dscale)
        g_feats = torch.zeros((B, N, C), dtype=torch.float32, device=dev)
        g_glob = torch.empty((B, M, C), dtype=torch.float32, device=dev)
        pointnet2.assemble_tokens_ragged_grad(grad_out.contiguous(), dscale, feats_pm, idx, cnt, off, g_rppe, g_ds, g_feats, g_glob,
                                              U, B, N, M, ns, C, rppe_compact=compact,
                                              rowmap=rowmap if ASSEMBLE_BWD_TOKEN_PARALLEL else None)
        return g_rppe, g_ds, g_feats, None, g_glob, None


# The position MLP of a PDA scale (pointnet2_modules.py:907-915: Conv 12 -> C/2 -> C with BatchNorm + ReLU) on the distinct
# tokens: the rows are gathered, the convolutions run on (U, .) and the BatchNorms use multiplicity-weighted statistics, i.e.
# exactly the statistics of the dense (B, M, ns, .) tensor (csrc/bn_relu.hip, pda_bn_relu_{fwd,bwd}_weighted).
RAGGED_POSITION_MLP = True


class BatchNormReLUWeighted(Function):
    """relu(batch_norm(x)) in training mode on rows that stand for roww[r] copies each (dense row count = `count`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, roww, count):
        x = x.contiguous()
        rows, c = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_fwd_weighted(x, weight, bias, running_mean, running_var, y, stats, scratch, rows, c, eps, momentum, roww, count)
        ctx.save_for_backward(x, weight, bias, stats, roww)
        ctx.count = count
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight, bias, stats, roww = ctx.saved_tensors
        rows, c = x.shape
        gx = torch.empty_like(x)
        gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        scratch = torch.empty((pointnet2.bn_relu_scratch_bytes(c),), dtype=torch.uint8, device=x.device)
        pointnet2.bn_relu_bwd_weighted(x, grad_y.contiguous(), weight, bias, stats, gx, gw, gb, scratch, rows, c, roww, ctx.count)
        return gx, gw, gb, None, None, None, None, None, None


def position_mlp_ragged_supported(layers, rppe):
    layers = list(layers)
    ok = (RAGGED_POSITION_MLP and rppe.is_cuda and rppe.dtype == torch.float32 and len(layers) == 6 and torch.is_grad_enabled()
          and not DENSE_BF16 and not torch.is_autocast_enabled())
    if not ok:
        return False
    for conv, bn in ((layers[0], layers[1]), (layers[3], layers[4])):
        c = conv.weight.shape[0]
        if (conv.bias is not None or not bn.training or not bn.affine or bn.momentum is None or c < 4 or c > 1024
                or (c & (c - 1)) != 0):
            return False
    return True


def position_mlp_ragged(layers, rppe, plan):
    """[Conv -> BN -> ReLU] x 2 of `layers` on the distinct tokens: rppe (B,M,ns,12) dense in, (U, C) compact out."""
    layers = list(layers)
    x = rppe.reshape(-1, rppe.shape[-1]).index_select(0, plan.rowmap)        # (U, 12)
    count = plan.groups * plan.nsample
    for conv, bn in ((layers[0], layers[1]), (layers[3], layers[4])):
        x = linear(x, conv.weight.flatten(1), None)
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        bump_bn_counter(bn)
        x = BatchNormReLUWeighted.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, plan.roww, count)
    return x


def _wgrad_ragged(x2d, g2d, weight, want_bias):
    """Weight / bias gradient at the token counts of the ragged path.  They change from step to step, so no library
    selection was tuned for them, and the default heuristic runs these K = tokens reductions at 20-70 TFLOP/s where
    csrc/wgrad.hip reaches 60-115 (profiles/r02_gemm_probe.txt), 150-200 in its split-bf16 form; below ~4k tokens the library is level."""
    n_out, n_in = weight.shape
    tokens = x2d.shape[0]
    if (LINEAR_WGRAD_KERNEL and tokens >= 4096 and (min(n_out, n_in) >= 64 or max(n_out, n_in) <= 64) and n_out % 4 == 0
            and n_in % 4 == 0 and x2d.dtype == torch.float32 and g2d.dtype == torch.float32):
        gw = torch.empty_like(weight)
        gb = torch.empty((n_out,), dtype=torch.float32, device=x2d.device) if want_bias else None
        pointnet2.linear_wgrad(x2d.contiguous(), g2d.contiguous(), gw, gb, tokens, n_in, n_out)
        return gw, gb
    return g2d.t().mm(x2d), (g2d.sum(0) if want_bias else None)


class RaggedTransformerBlock(Function):
    """TransformerBlock (pool=True) on the compact token matrix x (U, D) of a scale: the same operator sequence on U
    instead of groups * nsample rows, attention and the max-pool tail on the ragged groups (fp32 path)."""

    @staticmethod
    def supported(d, heads, nsample, x):
        return (RAGGED_TOKENS and x.is_cuda and x.dtype == torch.float32 and d in (256, 512, 1024) and d % heads == 0
                and nsample in GroupAttention.SUPPORTED_SEQ and d // heads in GroupAttention.SUPPORTED_HD
                and not torch.is_autocast_enabled() and not DENSE_BF16)

    @staticmethod
    def forward(ctx, x, plan, n1w, n1b, in_w, in_b, out_w, out_b, n2w, n2b, w1, b1, w2, b2, heads, eps1, eps2):
        x = x.contiguous()
        U, D = x.shape
        G, S, hd = plan.groups, plan.nsample, D // heads
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        src1, ssum, src2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        st1, st2 = torch.empty((U, 2), **f32), torch.empty((U, 2), **f32)
        lse = torch.empty((G, heads, S), **f32)
        pointnet2.layer_norm_fwd(x, None, n1w, n1b, None, src1, st1, U, D, eps1)
        qkv = _gemm_nt(src1, in_w, in_b)
        a = torch.empty((U, D), **f32)
        pointnet2.group_attention_ragged_fwd(qkv, plan.cnt, plan.off, a, lse, U, G, S, heads, hd)
        proj = _gemm_nt(a, out_w, out_b)
        pointnet2.layer_norm_fwd(proj, src1, n2w, n2b, ssum, src2, st2, U, D, eps2)
        del proj
        h = _gemm_nt(src2, w1, b1, relu=True)                   # relu(src2 W1^T + b1), ReLU in the GEMM epilogue
        ffn = _gemm_nt(h, w2, b2)
        y = torch.empty((G, D), **f32)
        arg = torch.empty((G, D), dtype=torch.uint8, device=dev)
        pointnet2.add_max_pool_ragged(src2, ffn, plan.cnt, plan.off, y, arg, U, G, D)
        ctx.save_for_backward(x, st1, src1, qkv, lse, a, ssum, st2, src2, h, n1w, in_w, out_w, n2w, w1, w2, arg,
                              plan.cnt, plan.off, plan.rowmap)
        ctx.heads, ctx.dims = heads, (G, S)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st1, src1, qkv, lse, a, ssum, st2, src2, h, n1w, in_w, out_w, n2w, w1, w2, arg, cnt, off, rowmap = ctx.saved_tensors
        U, D = x.shape
        G, S = ctx.dims
        heads = ctx.heads
        hd = D // heads
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dy2 = torch.empty((U, D), **f32)
        pointnet2.max_pool_scatter_ragged(dy.contiguous(), arg, rowmap, off, dy2, U, G, S, D)
        d_h = _gemm_nn(dy2, w2)
        gw2, gb2 = _wgrad_ragged(h, dy2, w2, True)
        d_h = torch.ops.aten.threshold_backward(d_h, h, 0)
        gw1, gb1 = _wgrad_ragged(src2, d_h, w1, True)
        d_lin1 = _gemm_nn(d_h, w1)
        del d_h
        d_s = torch.empty((U, D), **f32)
        gn2w, gn2b = torch.empty_like(n2w), torch.empty_like(n2w)
        scratch = torch.empty((pointnet2.layer_norm_scratch_bytes(D),), dtype=torch.uint8, device=dev)
        pointnet2.layer_norm_bwd(ssum, dy2, n2w, st2, d_s, gn2w, gn2b, scratch, U, D, grad_y2=d_lin1)
        del d_lin1
        d_a = _gemm_nn(d_s, out_w)
        gwo, gbo = _wgrad_ragged(a, d_s, out_w, True)
        dqkv = torch.empty_like(qkv)
        pointnet2.group_attention_ragged_bwd(qkv, d_a, lse, cnt, off, dqkv, U, G, S, heads, hd)
        del d_a
        gwi, gbi = _wgrad_ragged(src1, dqkv, in_w, True)
        d_src1 = _gemm_nn(dqkv, in_w, acc=d_s)             # residual gradient d_s + dqkv Win, accumulated (d_s is ours)
        del dqkv
        d_x = torch.empty_like(x)
        gn1w, gn1b = torch.empty_like(n1w), torch.empty_like(n1w)
        pointnet2.layer_norm_bwd(x, d_src1, n1w, st1, d_x, gn1w, gn1b, scratch, U, D)
        return d_x, None, gn1w, gn1b, gwi, gbi, gwo, gbo, gn2w, gn2b, gw1, gb1, gw2, gb2, None, None, None


def ragged_transformer_block(tr, x, plan):
    at = tr.self_attn
    return RaggedTransformerBlock.apply(x, plan, tr.norm1.weight, tr.norm1.bias, at.in_proj_weight, at.in_proj_bias,
                                        at.out_proj.weight, at.out_proj.bias, tr.norm2.weight, tr.norm2.bias,
                                        tr.linear1.weight, tr.linear1.bias, tr.linear2.weight, tr.linear2.bias,
                                        at.num_heads, tr.norm1.eps, tr.norm2.eps)


class LayerNormResidual(Function):
    """MI355X extension: y = LayerNorm(x [+ residual]) over the last dim (csrc/layer_norm.hip); one forward
    kernel, one single-pass backward kernel (+ a tiny fixed-order reduction of the gamma/beta partials)."""

    @staticmethod
    def supported(x, d):
        return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == d and d in (256, 512, 1024) and x.numel() > 0

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        x = x.contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        if residual is not None:
            s = torch.empty_like(x)
            pointnet2.layer_norm_fwd(x, residual.contiguous(), weight, bias, s, y, stats, rows, d, eps)
        else:
            s = x
            pointnet2.layer_norm_fwd(x, None, weight, bias, None, y, stats, rows, d, eps)
        ctx.save_for_backward(s, weight, stats)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        s, weight, stats = ctx.saved_tensors
        d = s.shape[-1]
        rows = s.numel() // d
        gx = torch.empty_like(s)
        gw, gb = torch.empty_like(weight), torch.empty_like(weight)
        scratch = torch.empty((pointnet2.layer_norm_scratch_bytes(d),), dtype=torch.uint8, device=s.device)
        pointnet2.layer_norm_bwd(s, grad_y.contiguous(), weight, stats, gx, gw, gb, scratch, rows, d)
        return gx, (gx if ctx.has_residual else None), gw, gb, None


def layer_norm(x, ln, residual=None):
    """nn.LayerNorm module `ln` applied to x (+ residual) over the last dim."""
    return LayerNormResidual.apply(x, residual, ln.weight, ln.bias, ln.eps)


class BallQuery(Function):
    """pointnet2_utils.py:228-253.  idx (B,npoint,nsample) int32, zero-initialised."""

    @staticmethod
    @_fwd
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class BallQueryDilated(Function):
    """pointnet2_utils.py:258-284."""

    @staticmethod
    @_fwd
    def forward(ctx, max_radius: float, min_radius: float, nsample: int, xyz: torch.Tensor,
                new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_dilated_wrapper(B, N, npoint, max_radius, min_radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


ball_query_dilated = BallQueryDilated.apply


# cell-list ball query (csrc/ball_query_cells.hip) for the multi-scale groupers when the cloud is large enough to pay
# for the binning passes; below that the scalar-stream brute force of csrc/ball_query.hip is faster
BALL_QUERY_CELLS = True
BALL_QUERY_CELLS_MIN_N = int(os.environ.get('PDA_BQ_CELLS_MIN_N', '16384'))      # profiles/r02_ball_query_cells.txt: break-even at 8192, 1.7x at 16384, 3.5-5.4x at 65536


def ball_query_multi(radii: List[float], nsamples: List[int], xyz: torch.Tensor,
                     new_xyz: torch.Tensor) -> List[torch.Tensor]:
    """MI355X extension: the multi-scale groupers' ball queries (one per scale over the same
    centres, pointnet2_modules.py:1657) in one pass over the points.  Results are identical to
    ``[ball_query(r, ns, xyz, new_xyz) for r, ns in zip(radii, nsamples)]``."""
    assert new_xyz.is_contiguous() and xyz.is_contiguous()
    B, N, _ = xyz.size()
    npoint = new_xyz.size(1)
    out: List[torch.Tensor] = []
    with torch.no_grad():
        for start in range(0, len(radii), 3):
            rs, nss = radii[start:start + 3], nsamples[start:start + 3]
            idxs = [torch.zeros((B, npoint, ns), dtype=torch.int32, device=xyz.device) for ns in nss]
            if BALL_QUERY_CELLS and N >= BALL_QUERY_CELLS_MIN_N and max(nss) <= 128 and min(rs) > 0:
                scratch = torch.empty((pointnet2.ball_query_cells_scratch_bytes(B, N),), dtype=torch.uint8, device=xyz.device)
                pointnet2.ball_query_cells(B, N, npoint, rs, nss, new_xyz, xyz, idxs, scratch)
            else:
                pointnet2.ball_query_multi(B, N, npoint, rs, nss, new_xyz, xyz, idxs)
            out.extend(idxs)
    return out


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:671-704: ball query, group xyz (centre-subtracted, :692) and
    features, concatenate -> (B, 3 + C, npoint, nsample)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None,
                idx: torch.Tensor = None) -> torch.Tensor:
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class QueryDilatedAndGroup(nn.Module):
    """pointnet2_utils.py:706-741.  NB the reference passes (radius_in, radius_out) into
    ball_query_dilated's (max_radius, min_radius) slots (:726); kept as is."""

    def __init__(self, radius_in: float, radius_out: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius_in, self.radius_out, self.nsample, self.use_xyz = radius_in, radius_out, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        idx = ball_query_dilated(self.radius_in, self.radius_out, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class QueryAndGroup_alone_grouped_density(nn.Module):
    """pointnet2_utils.py:618-668: like the directional grouper without the direction channels:
    cat [xyz(3, absolute), density(1), features(C)]."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None,
                idx: torch.Tensor = None) -> torch.Tensor:
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        distances = torch.norm(grouped_xyz.permute(0, 2, 3, 1).contiguous() - new_xyz.unsqueeze(2), dim=-1)
        density = torch.exp(-distances ** 2 / (2 * self.radius ** 2)) / (2.5 * self.radius)
        density = density.unsqueeze(1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, density, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class QueryAndGroup_alone_grouped_density_directional(nn.Module):
    """The PDA-layer grouper, pointnet2_utils.py:557-614.  Output channel order (:607):
    [xyz(3, ABSOLUTE coordinates: the centre subtraction is commented out at :602),
     gaussian density exp(-|d|^2 / (2 r^2)) / (2.5 r) (:594-597),
     direction (grouped - centre) / r (:599-600),
     features(C)]  -> (B, 3 + 1 + 3 + C, npoint, nsample)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None,
                idx: torch.Tensor = None) -> torch.Tensor:
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
        distances = torch.norm(grouped_xyz.permute(0, 2, 3, 1).contiguous() - new_xyz.unsqueeze(2), dim=-1)
        density = torch.exp(-distances ** 2 / (2 * self.radius ** 2)) / (2.5 * self.radius)
        density = density.unsqueeze(1)  # (B, 1, npoint, nsample)
        directional_vectors = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        directional_vectors = directional_vectors / self.radius
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, density, directional_vectors, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class GroupAll(nn.Module):
    """pointnet2_utils.py:743-766."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            new_features = grouped_xyz
        return new_features

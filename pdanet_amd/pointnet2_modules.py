"""Set-abstraction layer modules of PDA-SSD, restated on top of the gfx950 operators.

Mirrors, by name, constructor signature, forward signature, return values and state-dict key
layout, the classes of /root/reference/pcdet/ops/pointnet2/pointnet2_batch/:
  TransformerEncoderLayerPreNorm                  PointFormer.py:7-38
  PointnetSAModuleMSG_WithSampling                pointnet2_modules.py:1417-1686  (vanilla SA)
  PointnetSAModuleMSG_WithSampling_Ellipsoid      pointnet2_modules.py:541-954    (the PDA layer)
  DensityNet / PointConvDensitySetAbstraction     pointnet2_modules.py:958-1006
  Vote_layer                                      pointnet2_modules.py:1689-1753
  PointnetFPModule                                pointnet2_modules.py:1776-1824
so a reference checkpoint's ``backbone_3d.*`` keys load with strict=True.

What is done differently (results unchanged):
  * all scales of a layer share ONE multi-radius ball query pass (pda_ball_query_multi);
  * xyz^T is transposed once per layer, not once per grouper;
  * the sampling stage skips the (B,N,C) feature transpose the reference materialises and
    never uses for D-FPS / ctr-aware sampling (pointnet2_modules.py:1554);
  * optional fused group->MLP->max-pool path for the vanilla SA layers (see fused_ops.py).
"""
from typing import List

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import pointnet2_utils


class TransformerEncoderLayerPreNorm(nn.Module):
    """PointFormer.py:7-38.  NB despite the name the residual is taken from the NORMALISED
    input (src = norm1(src); src = src + attn(src)), which is what the reference does."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout, inplace=True)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout, inplace=True)
        self.dropout2 = nn.Dropout(dropout, inplace=True)
        self.activation = nn.ReLU(inplace=True)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        src = self.norm1(src)  # (K, B*N, C)
        src2, _ = self.self_attn(src, src, src, attn_mask=src_mask,
                                 key_padding_mask=src_key_padding_mask, need_weights=False)
        src = src + self.dropout1(src2)
        src = self.norm2(src)
        src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
        src = src + self.dropout2(src2)
        return src



# ---------------------------------------------------------------------------------------------
# Channel-last ("point-major") execution of the per-group networks.
#
# The reference builds every grouped tensor channel-major, (B, C, npoint, nsample), which on the
# gather side means C scattered 4-byte reads per neighbour and on the dense side a cat + permute
# + contiguous of the largest tensors of the model before the transformer
# (pointnet2_modules.py:920-929).  Here grouped tensors are (B, npoint, nsample, C): a neighbour
# is one contiguous row (pda_group_rows), 1x1 convs are F.linear over the last dim, BatchNorm runs
# on the SAME element sets through a channels_last NCHW view (identical statistics), and the
# transformer consumes the tensor batch-first without any relayout.  Same parameters, same
# state-dict, same math; only summation orders differ.
CHANNELS_LAST = True
GROUP_ATTENTION_KERNEL = True   # csrc/group_attention.hip instead of scaled_dot_product_attention


def _bn_lastdim(bn, x):
    """BatchNorm{1,2}d over the last dim of x (B, ..., C): statistics over all other dims, as
    BatchNorm2d over (B, C, H, W) of the channel-major tensor."""
    shp = x.shape
    x4 = x.reshape(shp[0], -1, 1, shp[-1]).permute(0, 3, 1, 2)  # logical (B, C, S, 1), channels_last memory
    pointnet2_utils.bump_bn_counter(bn)
    y = F.batch_norm(x4, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training or bn.running_mean is None, bn.momentum, bn.eps)
    return y.permute(0, 2, 3, 1).reshape(shp)


PLAN_OVERLAP = os.environ.get("PDA_PLAN_OVERLAP", "1") != "0"   # PDA layer: work that needs no token count goes ahead of the host's wait for it
FUSED_BN_RELU = True   # csrc/bn_relu.hip instead of F.batch_norm + F.relu in training mode
FUSED_TRANSFORMER_BLOCK = True   # the whole encoder layer as one autograd node (pointnet2_utils.TransformerBlock)
FUSED_GEOMETRY = True   # one kernel for the PDA grouper's density / direction / position-encoding input
FUSED_LAYER_NORM = True   # csrc/layer_norm.hip (with the residual add fused in) instead of F.layer_norm


def _bn_relu_lastdim(bn, x, out_bf16=False):
    """relu(bn(x)) over the last dim: one fused kernel pair in training mode, else torch."""
    if FUSED_BN_RELU and pointnet2_utils.BatchNormReLU.supported(x, bn):
        return pointnet2_utils.batch_norm_relu(bn, x, out_bf16)
    return F.relu(_bn_lastdim(bn, x))


def _bf16_boundary(x, conv, bn):
    """Dense-bf16 training: may the output of `conv` over x stay bf16 on its way into the fused BN+ReLU kernel?"""
    w = conv.weight.flatten(1)
    return (pointnet2_utils.DENSE_BF16 and FUSED_BN_RELU and torch.is_grad_enabled() and pointnet2_utils._dense_bf16(x, w)
            and pointnet2_utils.BatchNormReLU.supported_module(bn, w.shape[0]))


FOLD_EVAL_BN = True   # inference: BatchNorm (running statistics) folded into the preceding 1x1 convolution


def _folded_conv_bn(conv, bn):
    """(W', b') with W' x + b' == BN_eval(conv(x)): W' = W * s, b' = (conv.bias - running_mean) * s + beta,
    s = gamma / sqrt(running_var + eps).  Cached on the BN module, keyed on the version counters of every tensor
    involved (an optimizer step, a load_state_dict or a training-mode forward invalidates it)."""
    tensors = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = (_lib.PARAM_EPOCH[0],) + tuple(None if t is None else (t._version, t.data_ptr()) for t in tensors)
    cache = bn.__dict__.get("_pda_folded")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            w = (conv.weight.flatten(1) * s[:, None]).contiguous()
            b = bn.bias - bn.running_mean * s
            if conv.bias is not None:
                b = b + conv.bias * s
        cache = (key, w, b.contiguous())
        bn.__dict__["_pda_folded"] = cache
    return cache[1], cache[2]


def _can_fold(conv, bn):
    return (FOLD_EVAL_BN and not bn.training and not torch.is_grad_enabled() and bn.affine and bn.running_mean is not None
            and conv.weight.is_cuda)


def _linear_relu(x, w, b):
    """relu(x W^T + b) over the last dim; fp32 outside the dense-bf16 sizes: bias and ReLU in the GEMM epilogue."""
    if b is not None and x.dtype == torch.float32 and not pointnet2_utils._dense_bf16(x, w) and not torch.is_autocast_enabled():
        y = torch._addmm_activation(b, x.reshape(-1, x.shape[-1]), w.t())
        return y.view(*x.shape[:-1], w.shape[0])
    return torch.relu_(pointnet2_utils.linear(x, w, b))


FUSED_BN_RELU_MAX_POOL = True   # the last BN+ReLU of a grouped MLP and the max over nsample as one autograd node


def _mlp_lastdim(layers, x, pool=False, mfma=False):
    """[Conv 1x1 -> BN -> ReLU]* of an nn.Sequential applied over the last dim of x; pool=True: followed by the max over
    dim -2 (the nsample axis of a grouped tensor), fused into the last BN+ReLU in training mode.  mfma=True (the
    vanilla SA group MLPs): the contractions run on this repo's f32 MFMA kernels where a kernel is built for the shape."""
    layers = list(layers)
    skip = 0
    pooled = False
    for k, m in enumerate(layers):
        if skip:
            skip -= 1
            continue
        if isinstance(m, (nn.Conv2d, nn.Conv1d)):
            nxt = layers[k + 1] if k + 1 < len(layers) else None
            if isinstance(nxt, (nn.BatchNorm1d, nn.BatchNorm2d)) and _can_fold(m, nxt):
                w, b = _folded_conv_bn(m, nxt)
                if k + 2 < len(layers) and isinstance(layers[k + 2], nn.ReLU):
                    x = _linear_relu(x, w, b)
                    skip = 2
                else:
                    x = pointnet2_utils.linear(x, w, b)
                    skip = 1
                continue
            relu_next = k + 2 < len(layers) and isinstance(layers[k + 2], nn.ReLU)
            boundary = isinstance(nxt, (nn.BatchNorm1d, nn.BatchNorm2d)) and relu_next and _bf16_boundary(x, m, nxt)
            w2d = m.weight.flatten(1)
            if mfma and m.bias is None and torch.is_grad_enabled() and pointnet2_utils.LinearColsMFMA.supported(x, w2d):
                x = pointnet2_utils.LinearColsMFMA.apply(x, w2d)     # csrc/sa_mlp.hip, lin_cols_kernel
            else:
                x = pointnet2_utils.linear(x, w2d, m.bias, out_bf16=boundary)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            if k + 1 < len(layers) and isinstance(layers[k + 1], nn.ReLU):
                # dense-bf16 mode: y goes out as bf16 when its only consumer is the next bf16 GEMM of the chain
                after = layers[k + 2] if k + 2 < len(layers) else None
                out_b = (x.dtype == torch.bfloat16 and isinstance(after, (nn.Conv2d, nn.Conv1d))
                         and pointnet2_utils._dense_bf16(x, after.weight.flatten(1)))
                if (pool and k + 2 == len(layers) and FUSED_BN_RELU and FUSED_BN_RELU_MAX_POOL
                        and pointnet2_utils.BatchNormReLUMaxPool.supported(x, m)):
                    x = pointnet2_utils.batch_norm_relu_max_pool(m, x)
                    pooled = True
                else:
                    x = _bn_relu_lastdim(m, x, out_b)
                skip = 1
            else:
                x = _bn_lastdim(m, x)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        else:
            raise NotImplementedError(type(m))
    if pool and not pooled:
        x = x.max(dim=-2)[0]
    return x


FUSED_DENSITYNET_EVAL = os.environ.get("PDA_DENSITYNET_EVAL", "1") != "0"


def _densitynet_eval_ok(dn, x):
    convs, bns = list(dn.mlp_convs), list(dn.mlp_bns)
    return (FUSED_DENSITYNET_EVAL and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 1 and not torch.is_grad_enabled()
            and len(convs) == 3 and [c.out_channels for c in convs] == [16, 8, 1] and convs[0].in_channels == 1
            and all(_can_fold(c, b) for c, b in zip(convs, bns)) and not torch.is_autocast_enabled())


def _densitynet_eval(dn, x):
    """DensityNet (pointnet2_modules.py:958-981) in inference on x (..., 1): relu(W3' relu(W2' relu(w1' x + b1') + b2') + b3') with
    the BatchNorms folded in (_folded_conv_bn), one kernel instead of three tiny GEMMs with bias + ReLU epilogues.  The packed
    parameter block is cached on the module, keyed on the folded tensors (which are themselves re-made when a weight, a BatchNorm
    parameter or a running statistic changes)."""
    folded = [_folded_conv_bn(c, b) for c, b in zip(dn.mlp_convs, dn.mlp_bns)]
    key = tuple((t.data_ptr(), t._version) for wb in folded for t in wb)
    cache = dn.__dict__.get("_pda_eval_block")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            block = torch.cat([t.reshape(-1).float() for wb in folded for t in wb]).contiguous()
        cache = (key, block, folded)       # `folded` kept alive: its addresses are the key
        dn.__dict__["_pda_eval_block"] = cache
    x = x.contiguous()
    y = torch.empty_like(x)
    pointnet2_utils.pointnet2.densitynet_eval(x, cache[1], y, x.numel())
    return y


def _transformer_batch_first(tr, x, pool=False):
    """TransformerEncoderLayerPreNorm.forward (PointFormer.py:28-38) on x (batch, seq, D) instead of
    (seq, batch, D), with the module's own parameters; dropout is 0 in PDA-SSD (:632).  pool=True also takes
    the max over seq (pointnet2_modules.py:931) and returns (batch, D)."""
    attn = tr.self_attn
    assert attn.dropout == 0.0 or not tr.training
    D, H = attn.embed_dim, attn.num_heads
    if (FUSED_TRANSFORMER_BLOCK and FUSED_LAYER_NORM and GROUP_ATTENTION_KERNEL
            and pointnet2_utils.TransformerBlock.supported(x, H)):
        return pointnet2_utils.transformer_block(tr, x, pool)
    fused_ln = FUSED_LAYER_NORM and pointnet2_utils.LayerNormResidual.supported(x, D) and not torch.is_autocast_enabled()
    if fused_ln:
        src = pointnet2_utils.layer_norm(x, tr.norm1)
    else:
        src = F.layer_norm(x, (D,), tr.norm1.weight, tr.norm1.bias, tr.norm1.eps)
    lin = pointnet2_utils.linear
    qkv = lin(src, attn.in_proj_weight, attn.in_proj_bias)
    Bn, S, _ = qkv.shape
    if GROUP_ATTENTION_KERNEL and pointnet2_utils.GroupAttention.supported(qkv, H):
        a = pointnet2_utils.group_attention(qkv, H)              # one wave per (group, head), fp32 MFMA
    else:
        q, k, v = qkv.view(Bn, S, 3, H, D // H).permute(2, 0, 3, 1, 4)  # each (Bn, H, S, hd)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Bn, S, D)
    if fused_ln:   # LayerNorm(src + out_proj(a)) in one kernel
        src = pointnet2_utils.layer_norm(lin(a, attn.out_proj.weight, attn.out_proj.bias), tr.norm2, residual=src)
    else:
        src = src + lin(a, attn.out_proj.weight, attn.out_proj.bias)
        src = F.layer_norm(src, (D,), tr.norm2.weight, tr.norm2.bias, tr.norm2.eps)
    src2 = lin(F.relu(lin(src, tr.linear1.weight, tr.linear1.bias)), tr.linear2.weight, tr.linear2.bias)
    return (src + src2).max(dim=1)[0] if pool else src + src2


def calc_square_dist(a, b):
    """pointnet2_modules.py:19-43: |a|^2 + |b|^2 - 2 a.b  -> (bs, n, m)."""
    a_sq = torch.sum(a * a, dim=-1).unsqueeze(2)
    b_sq = torch.sum(b * b, dim=-1).unsqueeze(1)
    return a_sq + b_sq - 2.0 * torch.matmul(a, b.transpose(1, 2))


def _partition_fps(xyz_tmp, npoint, key_fn, batch):
    """ds-FPS / ry-FPS (pointnet2_modules.py:1595-1642): sort each scene by a scalar key, cut
    into 4 equal parts, D-FPS npoint/4 in every part, map back to scene indices."""
    part_num = 4
    xyz_div, idx_div = [], []
    for per_xyz in xyz_tmp:
        _, order = key_fn(per_xyz).sort(dim=0, descending=False)
        xyz_div.append(per_xyz[order].view(part_num, -1, 3))
        idx_div.append(order.view(part_num, -1))
    xyz_div = torch.cat(xyz_div, dim=0).contiguous()
    idx_div = torch.cat(idx_div, dim=0)
    idx_sampled = pointnet2_utils.furthest_point_sample(xyz_div, npoint // part_num)
    picked = [idx_per[s.long()] for s, idx_per in zip(idx_sampled, idx_div)]
    return torch.cat(picked, dim=-1).reshape(batch, npoint).int()


def sample_points(xyz, features, cls_features, sample_type_list, sample_range_list, npoint_list):
    """The sampling stage shared by both SA layer classes
    (pointnet2_modules.py:1542-1646 == :738-841).  Returns sampled_idx (B, sum npoint) int32."""
    sampled_idx_list = []
    last_sample_end_index = 0
    for i in range(len(sample_type_list)):
        sample_type = sample_type_list[i]
        sample_range = sample_range_list[i]
        npoint = npoint_list[i]
        if npoint <= 0:
            continue
        if sample_range == -1:
            sl = slice(last_sample_end_index, None)
        else:
            sl = slice(last_sample_end_index, sample_range)
            last_sample_end_index += sample_range
        xyz_tmp = xyz[:, sl, :]
        cls_features_tmp = cls_features[:, sl, :] if cls_features is not None else None

        def feature_tmp():  # only the feature-distance samplers need it
            return features.transpose(1, 2)[:, sl, :].contiguous()

        n_tmp = xyz_tmp.shape[1]
        if n_tmp <= npoint:  # no downsampling
            sample_idx = torch.arange(n_tmp, device=xyz.device, dtype=torch.int32).unsqueeze(0) \
                .expand(xyz_tmp.shape[0], n_tmp).contiguous()
        elif ('cls' in sample_type) or ('ctr' in sample_type):
            cls_features_max, _ = cls_features_tmp.max(dim=-1)
            score_pred = torch.sigmoid(cls_features_max)  # (B, N)
            _, sample_idx = torch.topk(score_pred, npoint, dim=-1)
            sample_idx = sample_idx.int()
        elif 'D-FPS' in sample_type or 'DFS' in sample_type:
            sample_idx = pointnet2_utils.furthest_point_sample(xyz_tmp.contiguous(), npoint)
        elif 'F-FPS' in sample_type or 'FFS' in sample_type:
            f = torch.cat([xyz_tmp, feature_tmp()], dim=-1)
            sample_idx = pointnet2_utils.furthest_point_sample_with_dist(calc_square_dist(f, f).contiguous(), npoint)
        elif sample_type == 'FS':
            f = torch.cat([xyz_tmp, feature_tmp()], dim=-1)
            idx1 = pointnet2_utils.furthest_point_sample_with_dist(calc_square_dist(f, f).contiguous(), npoint)
            idx2 = pointnet2_utils.furthest_point_sample(xyz_tmp.contiguous(), npoint)
            sample_idx = torch.cat([idx1, idx2], dim=-1)
        elif 'Rand' in sample_type:
            sample_idx = torch.randperm(n_tmp, device=xyz.device)[None, :npoint].int().repeat(xyz_tmp.shape[0], 1)
        elif sample_type in ('ds_FPS', 'ds-FPS'):
            sample_idx = _partition_fps(xyz_tmp, npoint, lambda p: p.norm(dim=-1) - 5, xyz.shape[0])
        elif sample_type in ('ry_FPS', 'ry-FPS'):
            sample_idx = _partition_fps(xyz_tmp, npoint, lambda p: torch.atan(p[:, 0] / p[:, 1]), xyz.shape[0])
        else:
            raise NotImplementedError("sample type %r" % (sample_type,))
        sampled_idx_list.append(sample_idx)
    return torch.cat(sampled_idx_list, dim=-1).contiguous()


def coordinate_only_sampling(sample_type_list, sample_range_list, npoint_list):
    """True when a layer's sampling needs nothing but xyz (identity or D-FPS over the whole set)."""
    kinds = [(t, r, n) for t, r, n in zip(sample_type_list, sample_range_list, npoint_list) if n > 0]
    return len(kinds) == 1 and kinds[0][1] == -1 and ('D-FPS' in kinds[0][0] or 'DFS' in kinds[0][0]) \
        and 'cls' not in kinds[0][0] and 'ctr' not in kinds[0][0]


def _conv_bn_relu_1d(channels_in, spec):
    layers = []
    for c in spec:
        layers.extend([nn.Conv1d(channels_in, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()])
        channels_in = c
    return layers, channels_in


def _build_heads(module, out_channels, aggregation_mlp, confidence_mlp, num_class, have_branches):
    """aggregation_layer / confidence_layers exactly as pointnet2_modules.py:1495-1524."""
    if (aggregation_mlp is not None) and (len(aggregation_mlp) != 0) and have_branches:
        layers, out_channels = _conv_bn_relu_1d(out_channels, aggregation_mlp)
        module.aggregation_layer = nn.Sequential(*layers)
    else:
        module.aggregation_layer = None
    if (confidence_mlp is not None) and (len(confidence_mlp) != 0):
        layers, out_channels = _conv_bn_relu_1d(out_channels, confidence_mlp)
        layers.append(nn.Conv1d(out_channels, num_class, kernel_size=1, bias=True))
        module.confidence_layers = nn.Sequential(*layers)
    else:
        module.confidence_layers = None


class _SAModuleBase(nn.Module):
    def _ball_queries(self, xyz, new_xyz):
        """One pass over the points for every scale of this layer (None for non-ball groupers)."""
        if len(self.groupers) == 0 or not hasattr(self.groupers[0], "radius"):
            return [None] * len(self.groupers)
        return pointnet2_utils.ball_query_multi([g.radius for g in self.groupers],
                                                [g.nsample for g in self.groupers], xyz, new_xyz)


class PointnetSAModuleMSG_WithSampling(_SAModuleBase):
    """Vanilla SA layer with sampling + multi-scale grouping (pointnet2_modules.py:1417-1686):
    sample -> per scale [QueryAndGroup -> (Conv2d 1x1 no bias, BN2d, ReLU)*k -> max over nsample]
    -> concat -> aggregation Conv1d+BN+ReLU -> optional confidence head."""

    def __init__(self, *, npoint_list: List[int], sample_range_list: List[int], sample_type_list: List[str],
                 radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 dilated_group=False, pool_method='max_pool', aggregation_mlp: List[int],
                 confidence_mlp: List[int], num_class):
        super().__init__()
        self.sample_type_list = sample_type_list
        self.sample_range_list = sample_range_list
        self.dilated_group = dilated_group
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint_list = npoint_list
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        out_channels = 0
        for i in range(len(radii)):
            radius, nsample = radii[i], nsamples[i]
            if npoint_list is None:
                self.groupers.append(pointnet2_utils.GroupAll(use_xyz))
            elif self.dilated_group:
                min_radius = 0. if i == 0 else radii[i - 1]
                self.groupers.append(pointnet2_utils.QueryDilatedAndGroup(radius, min_radius, nsample, use_xyz=use_xyz))
            else:
                self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            mlp_spec = list(mlps[i])
            if use_xyz:
                mlp_spec[0] += 3
            shared_mlps = []
            for k in range(len(mlp_spec) - 1):
                shared_mlps.extend([nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1, bias=False),
                                    nn.BatchNorm2d(mlp_spec[k + 1]), nn.ReLU()])
            self.mlps.append(nn.Sequential(*shared_mlps))
            out_channels += mlp_spec[-1]
        self.pool_method = pool_method
        _build_heads(self, out_channels, aggregation_mlp, confidence_mlp, num_class, len(self.mlps) > 0)
        self.fused = None  # set by fused_ops.enable_fused(): group->MLP->max-pool in one HIP kernel

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, cls_features: torch.Tensor = None,
                new_xyz=None, ctr_xyz=None, presampled=None):
        """xyz (B,N,3), features (B,C,N), cls_features (B,N,num_class) ->
        new_xyz (B,M,3), new_features (B,C',M), cls_features (B,M,num_class)|None, sampled_idx.
        `presampled` = (sampled_idx, new_xyz[, prequery]) computed ahead of time (backbone side stream)."""
        sampled_idx_list = []
        pre_idxs = None
        if ctr_xyz is None:
            if presampled is not None:
                sampled_idx_list, new_xyz = presampled[0], presampled[1]
                if len(presampled) > 2 and presampled[2] is not None:
                    pre_idxs = presampled[2]['idxs']
            else:
                sampled_idx_list = sample_points(xyz, features, cls_features, self.sample_type_list,
                                                 self.sample_range_list, self.npoint_list)
                xyz_flipped = xyz.transpose(1, 2).contiguous()
                new_xyz = pointnet2_utils.gather_operation(xyz_flipped, sampled_idx_list).transpose(1, 2).contiguous()
        else:
            new_xyz = ctr_xyz

        if len(self.groupers) > 0:
            new_features_list = []
            plain_ball = (not self.dilated_group) and isinstance(self.groupers[0], pointnet2_utils.QueryAndGroup)
            idxs = pre_idxs if (pre_idxs is not None and plain_ball) else \
                (self._ball_queries(xyz, new_xyz) if plain_ball else [None] * len(self.groupers))
            use_cl = CHANNELS_LAST and plain_ball and self.pool_method == 'max_pool' and \
                getattr(self, "channels_last", True) and xyz.is_cuda
            feats_pm = features.transpose(1, 2).contiguous() if (use_cl and features is not None) else None
            for i in range(len(self.groupers)):
                if self.fused is not None and plain_ball and self.pool_method == 'max_pool' \
                        and not self.training and not torch.is_grad_enabled():
                    seq = list(self.mlps[i])
                    if (use_cl and feats_pm is not None and feats_pm.shape[-1] >= 128 and len(seq) == 9
                            and all(_can_fold(seq[3 * k], seq[3 * k + 1]) for k in range(3))):
                        # wide scale (layer 5): per-point first layer + split-bf16 GEMMs beat the fused f32-MFMA kernel
                        g = pointnet2_utils.sa_wide_scale_infer(xyz, new_xyz, feats_pm, idxs[i],
                                                                [_folded_conv_bn(seq[3 * k], seq[3 * k + 1]) for k in range(3)])
                        if g is not None:
                            new_features_list.append(g.transpose(1, 2))
                            continue
                    pooled = self.fused(i, self, xyz, new_xyz, features, idxs[i])  # (B, mlp[-1], npoint) | None
                    if pooled is not None:
                        new_features_list.append(pooled)
                        continue
                if use_cl and self.groupers[i].use_xyz:
                    if (self.training and torch.is_grad_enabled()
                            and pointnet2_utils.SaSmallChainTrain.supported(xyz, new_xyz, feats_pm, idxs[i], self.mlps[i])):
                        # narrow chains (layer 0): recompute passes over the neighbour lists, nothing grouped reaches HBM
                        g = pointnet2_utils.sa_small_chain_train(xyz, new_xyz, feats_pm, idxs[i], self.mlps[i])
                        new_features_list.append(g.transpose(1, 2))             # (B, mlp[-1], M) view
                        continue
                    if (self.training and torch.is_grad_enabled()
                            and pointnet2_utils.SaWideChainTrain.supported(xyz, new_xyz, feats_pm, idxs[i], self.mlps[i])):
                        # wide chains (layer 5): BatchNorm statistics in the GEMM epilogues, normalisation in the operand loads
                        g = pointnet2_utils.sa_wide_chain_train(xyz, new_xyz, feats_pm, idxs[i], self.mlps[i])
                        new_features_list.append(g.transpose(1, 2))             # (B, mlp[-1], M) view
                        continue
                    conv1 = self.mlps[i][0]
                    w1 = conv1.weight.flatten(1)
                    if (torch.is_grad_enabled() and isinstance(conv1, nn.Conv2d) and conv1.bias is None
                            and pointnet2_utils.SaGatherLinear.supported(xyz, feats_pm, w1)):
                        # the first contraction never sees a grouped (B, M, ns, 3 + C) tensor: per-point projection + row gather
                        # (SaPointLinear), or the grouping fused into the contraction (SaGatherLinear)
                        first = (pointnet2_utils.SaPointLinear if pointnet2_utils.SaPointLinear.supported(xyz, feats_pm, w1)
                                 else pointnet2_utils.SaGatherLinear)
                        g = first.apply(xyz, new_xyz, feats_pm, idxs[i], w1)
                        g = _mlp_lastdim(list(self.mlps[i])[1:], g, pool=True, mfma=True)
                        new_features_list.append(g.transpose(1, 2))
                        continue
                    # (B, M, ns, 3 + C): [xyz - centre | features], neighbours gathered as rows
                    g = pointnet2_utils.group_rows(xyz, idxs[i]) - new_xyz.unsqueeze(2)
                    if feats_pm is not None:
                        g = torch.cat([g, pointnet2_utils.group_rows(feats_pm, idxs[i])], dim=-1)
                    g = _mlp_lastdim(self.mlps[i], g, pool=True, mfma=True)   # (B, M, mlp[-1]): MLP + max over nsample
                    new_features_list.append(g.transpose(1, 2))             # (B, mlp[-1], M) view
                    continue
                if plain_ball:
                    new_features = self.groupers[i](xyz, new_xyz, features, idx=idxs[i])
                else:
                    new_features = self.groupers[i](xyz, new_xyz, features)  # (B, C, npoint, nsample)
                new_features = self.mlps[i](new_features)  # (B, mlp[-1], npoint, nsample)
                if self.pool_method == 'max_pool':
                    new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
                elif self.pool_method == 'avg_pool':
                    new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
                else:
                    raise NotImplementedError
                new_features_list.append(new_features.squeeze(-1))  # (B, mlp[-1], npoint)
            new_features = torch.cat(new_features_list, dim=1)
            if self.aggregation_layer is not None:
                new_features = self.aggregation_layer(new_features)
        else:
            new_features = pointnet2_utils.gather_operation(features, sampled_idx_list).contiguous()

        if self.confidence_layers is not None:
            cls_features = self.confidence_layers(new_features).transpose(1, 2)
        else:
            cls_features = None
        return new_xyz, new_features, cls_features, sampled_idx_list

    def prequery(self, xyz, new_xyz):
        """The neighbour lists of all scales (coordinates only), for the sampling side stream of the backbone: layer 0's
        16384 x 16384 query (0.13 ms with its cell lists) then runs beside the D-FPS chain of layer 1, which keeps 2 of
        256 CUs busy, instead of in front of the layer on the main stream."""
        if self.dilated_group or len(self.groupers) == 0 or not isinstance(self.groupers[0], pointnet2_utils.QueryAndGroup):
            return None
        return {'idxs': self._ball_queries(xyz, new_xyz), 'parts': None, 'totals': None}


class DensityNet(nn.Module):
    """pointnet2_modules.py:958-981: 1x1 convs 1->16->8->1 WITH bias, BN after each, and ReLU
    after EVERY layer including the last (the sigmoid branch `i == len(mlp_convs)` at :976 can
    never be taken)."""

    def __init__(self, hidden_unit=[16, 8]):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.mlp_convs.append(nn.Conv2d(1, hidden_unit[0], 1))
        self.mlp_bns.append(nn.BatchNorm2d(hidden_unit[0]))
        for i in range(1, len(hidden_unit)):
            self.mlp_convs.append(nn.Conv2d(hidden_unit[i - 1], hidden_unit[i], 1))
            self.mlp_bns.append(nn.BatchNorm2d(hidden_unit[i]))
        self.mlp_convs.append(nn.Conv2d(hidden_unit[-1], 1, 1))
        self.mlp_bns.append(nn.BatchNorm2d(1))

    def forward(self, density_scale):
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            density_scale = F.relu(bn(conv(density_scale)))
        return density_scale


class PointConvDensitySetAbstraction(nn.Module):
    """pointnet2_modules.py:983-1006: density / per-group max -> DensityNet."""

    def __init__(self, bandwidth):
        super().__init__()
        self.densitynet = DensityNet()
        self.bandwidth = bandwidth

    def forward(self, grouped_density):  # (B, 1, npoint, nsample)
        inverse_max_density = grouped_density.max(dim=3, keepdim=True)[0]
        return self.densitynet(grouped_density / inverse_max_density)


class PointnetSAModuleMSG_WithSampling_Ellipsoid(_SAModuleBase):
    """The PDA layer (pointnet2_modules.py:541-954).  Per scale with C = mlp_spec[0]:
    grouper -> [xyz(3) | density(1) | direction(3) | features(C)];
    position_mlp(12->C/2->C) on [centre, nbr, centre-nbr, direction];
    global_mlps(C+3->C->C) on the centre's own [xyz, feature], repeated over nsample;
    DensityNet on density/max; concat [rppe, f*density, f, global] (D = 4C);
    TransformerEncoderLayerPreNorm(D, 4 heads, ff = 2C) over sequences of nsample;
    max over nsample; fin_conv(4C->2C->mlp_spec[-1]).
    Only mlp_spec[0] and mlp_spec[-1] are used (:628-671); use_xyz adds nothing (:674-675)."""

    def __init__(self, *, npoint_list: List[int], sample_range_list: List[int], sample_type_list: List[str],
                 radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 dilated_group=False, pool_method='max_pool', aggregation_mlp: List[int],
                 confidence_mlp: List[int], num_class):
        super().__init__()
        self.sample_type_list = sample_type_list
        self.sample_range_list = sample_range_list
        self.dilated_group = dilated_group
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint_list = npoint_list
        self.groupers = nn.ModuleList()
        self.groupers_global = nn.ModuleList()  # present (and empty) in the reference too (:581)
        self.nsamples = nsamples
        self.point_density = nn.ModuleList()
        self.position_mlp = nn.ModuleList()
        self.Local_pointformer = nn.ModuleList()
        self.fin_conv = nn.ModuleList()
        self.global_mlps = nn.ModuleList()

        def conv_bn_relu2d(cin, cout):
            return [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]

        out_channels = 0
        for i in range(len(radii)):
            radius, nsample = radii[i], nsamples[i]
            if npoint_list is None:
                self.groupers.append(pointnet2_utils.GroupAll(use_xyz))
            elif self.dilated_group:
                min_radius = 0. if i == 0 else radii[i - 1]
                self.groupers.append(pointnet2_utils.QueryDilatedAndGroup(radius, min_radius, nsample, use_xyz=use_xyz))
            else:
                self.groupers.append(pointnet2_utils.QueryAndGroup_alone_grouped_density_directional(
                    radius, nsample, use_xyz=use_xyz))
            c = mlps[i][0]
            self.Local_pointformer.append(TransformerEncoderLayerPreNorm(
                d_model=c * 4, dim_feedforward=2 * c, dropout=0.0, nhead=4))
            self.position_mlp.append(nn.Sequential(*(conv_bn_relu2d(9 + 3, c // 2) + conv_bn_relu2d(c // 2, c))))
            self.global_mlps.append(nn.Sequential(*(conv_bn_relu2d(c + 3, c) + conv_bn_relu2d(c, c))))
            self.point_density.append(PointConvDensitySetAbstraction(radius))
            self.fin_conv.append(nn.Sequential(*(conv_bn_relu2d(4 * c, 2 * c) + conv_bn_relu2d(2 * c, mlps[i][-1]))))
            out_channels += mlps[i][-1]
        self.pool_method = pool_method
        _build_heads(self, out_channels, aggregation_mlp, confidence_mlp, num_class, len(self.fin_conv) > 0)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, cls_features: torch.Tensor = None,
                new_xyz=None, ctr_xyz=None, presampled=None):
        sampled_idx_list = []
        if ctr_xyz is None:
            prequery = None
            if presampled is not None:
                sampled_idx_list, new_xyz = presampled[0], presampled[1]
                prequery = presampled[2] if len(presampled) > 2 else None
            else:
                sampled_idx_list = sample_points(xyz, features, cls_features, self.sample_type_list,
                                                 self.sample_range_list, self.npoint_list)
                xyz_flipped = xyz.transpose(1, 2).contiguous()
                new_xyz = pointnet2_utils.gather_operation(xyz_flipped, sampled_idx_list).transpose(1, 2).contiguous()
            new_xyz_feature = pointnet2_utils.gather_operation(features, sampled_idx_list).transpose(1, 2).contiguous()
        else:
            new_xyz = ctr_xyz  # the reference has no centre features on this branch either (:850-851)
            prequery = None

        directional = len(self.groupers) > 0 and isinstance(
            self.groupers[0], pointnet2_utils.QueryAndGroup_alone_grouped_density_directional)
        if len(self.groupers) > 0 and directional and CHANNELS_LAST and getattr(self, "channels_last", True) \
                and xyz.is_cuda and self.groupers[0].use_xyz and features is not None:
            return self._forward_channels_last(xyz, features, new_xyz, new_xyz_feature, sampled_idx_list, prequery)

        if len(self.groupers) > 0:
            new_features_list = []
            # (B, 3 + C, npoint, 1): the sampled centre's own coordinates and feature (:856)
            global_feature = torch.cat([new_xyz, new_xyz_feature], dim=-1).transpose(1, 2).unsqueeze(dim=-1)
            idxs = self._ball_queries(xyz, new_xyz) if directional else [None] * len(self.groupers)
            B, npoint = new_xyz.shape[0], new_xyz.shape[1]
            for i in range(len(self.groupers)):
                if directional:
                    new_features = self.groupers[i](xyz, new_xyz, features, idx=idxs[i])
                else:
                    new_features = self.groupers[i](xyz, new_xyz, features)
                K_sample = self.nsamples[i]
                new_xyz_k = new_features[:, :3]                    # (B, 3, npoint, ns) absolute neighbour xyz
                grouped_density_feature = new_features[:, 3:4]     # (B, 1, npoint, ns)
                directional_vectors = new_features[:, 4:7]         # (B, 3, npoint, ns)
                new_xyz_K_feature = new_features[:, 7:]            # (B, C, npoint, ns)

                global_feature_k = self.global_mlps[i](global_feature).expand(-1, -1, -1, K_sample)
                density_scale_score = self.point_density[i](grouped_density_feature.contiguous())
                new_density_score_feature = new_xyz_K_feature * density_scale_score

                # relative point position encoding, channel order of :907-913, built channel-major
                # directly (the reference builds it (B,np,ns,12) and permutes)
                extended_coords = new_xyz.transpose(1, 2).unsqueeze(-1).expand(B, 3, npoint, K_sample)
                rppe = torch.cat([extended_coords, new_xyz_k, extended_coords - new_xyz_k, directional_vectors], dim=1)
                rppe = self.position_mlp[i](rppe)

                input_features = torch.cat([rppe, new_density_score_feature, new_xyz_K_feature, global_feature_k], dim=1)
                Bq, D, np_, ns = input_features.shape
                # (B, D, np, ns) -> (ns, B*np, D)
                input_features = input_features.permute(3, 0, 2, 1).reshape(ns, Bq * np_, D)
                transformed = self.Local_pointformer[i](input_features)  # (ns, B*np, D)
                # max over nsample == F.max_pool2d(kernel=[1, ns]) of the reference (:931)
                output_features = transformed.max(dim=0)[0].view(Bq, np_, D).permute(0, 2, 1).unsqueeze(-1)
                output_features = self.fin_conv[i](output_features.contiguous()).squeeze(-1)
                new_features_list.append(output_features)
            new_features = torch.cat(new_features_list, dim=1)
            if self.aggregation_layer is not None:
                new_features = self.aggregation_layer(new_features)
        else:
            new_features = pointnet2_utils.gather_operation(features, sampled_idx_list).contiguous()

        if self.confidence_layers is not None:
            cls_features = self.confidence_layers(new_features).transpose(1, 2)
        else:
            cls_features = None
        return new_xyz, new_features, cls_features, sampled_idx_list


    def ragged_capable(self, features=None):
        """The unique-token encoder applies to this layer (shapes the kernels are built for, switches on)."""
        if len(self.groupers) == 0 or not isinstance(self.groupers[0], pointnet2_utils.QueryAndGroup_alone_grouped_density_directional):
            return False
        probe = features if features is not None else next(self.parameters())
        # repeated tokens are identical only without dropout (PDA-SSD: 0.0, :632); a config that trains with dropout keeps
        # the dense encoder, whose own guard (transformer_block) then applies
        tr0 = self.Local_pointformer[0]
        if self.training and any(float(p) != 0.0 for tr in self.Local_pointformer
                                 for p in (tr.self_attn.dropout, tr.dropout.p, tr.dropout1.p, tr.dropout2.p)):
            return False
        return (pointnet2_utils.RaggedTransformerBlock.supported(self.Local_pointformer[0].self_attn.embed_dim,
                                                                 self.Local_pointformer[0].self_attn.num_heads, max(self.nsamples), probe)
                and all(ns in pointnet2_utils.GroupAttention.SUPPORTED_SEQ for ns in self.nsamples)
                and FUSED_TRANSFORMER_BLOCK and FUSED_LAYER_NORM and GROUP_ATTENTION_KERNEL
                and not torch.cuda.is_current_stream_capturing())    # the plan's token count is read on the host

    def prequery(self, xyz, new_xyz):
        """Everything of this layer that depends on coordinates only, for the sampling side stream of the backbone: the
        neighbour lists of all scales and, when the unique-token encoder applies, their plans with the token counts on
        their way to pinned host memory (no synchronisation here)."""
        idxs = self._ball_queries(xyz, new_xyz)
        out = {'idxs': idxs, 'parts': None, 'totals': None}
        if self.ragged_capable():
            parts, totals = pointnet2_utils.ragged_plan_parts(idxs)
            pinned = torch.empty((len(parts),), dtype=torch.int32, pin_memory=True)
            pinned.copy_(totals, non_blocking=True)
            out.update(parts=parts, totals=pinned)
        return out

    def _forward_channels_last(self, xyz, features, new_xyz, new_xyz_feature, sampled_idx_list, prequery=None):
        """The PDA scale loop of forward() in the point-major layout (see CHANNELS_LAST note):
        grouped tensors are (B, npoint, nsample, C); same parameters and math as the
        channel-major branch below / the reference (:854-945)."""
        B, npoint = new_xyz.shape[0], new_xyz.shape[1]
        # neighbour lists and distinct-token plans of all scales; None: run that scale dense.  Computed ahead of time on the
        # sampling side stream when the centres depend on coordinates only (`prequery`, whose event the caller has waited
        # for on this stream and on the host), otherwise here with ONE host synchronisation for the layer
        plans = [None] * len(self.groupers)
        pending = None
        plan_parts = None            # the plans' device halves (cnt, off, rowmap, roww): usable before any count is read
        if prequery is not None:
            idxs = prequery['idxs']
            if prequery['parts'] is not None and self.ragged_capable(features):
                plan_parts = prequery['parts']
                plans = pointnet2_utils.ragged_plans_from(prequery['parts'], prequery['totals'].tolist())
        else:
            idxs = self._ball_queries(xyz, new_xyz)
            if self.ragged_capable(features):
                # The token counts size the encoder's tensors, so the host has to read them -- but not by draining the
                # stream: the counts go to pinned memory behind the plan kernels, an event marks that copy, and everything
                # of the layer that does not need them (geometry, DensityNet, the centre MLPs of every scale) is enqueued
                # BEFORE the host waits for the event.  The host is several ms ahead of the device when it gets here; a
                # blocking read left the device with an empty queue behind the plan kernel and ~100 small launches to
                # wait for (1.3 ms idle per step, tools/step_gaps.py).
                parts, totals = pointnet2_utils.ragged_plan_parts(idxs)
                plan_parts = parts
                pinned = torch.empty((len(parts),), dtype=torch.int32, pin_memory=True)
                pinned.copy_(totals, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                if not PLAN_OVERLAP:
                    ev.synchronize()
                pending = (parts, pinned, ev)
        feats_pm = features.transpose(1, 2).contiguous()                      # (B, N, C)
        global_in = torch.cat([new_xyz, new_xyz_feature], dim=-1)             # (B, M, 3 + C)   (:856)
        centre = new_xyz.unsqueeze(2)                                         # (B, M, 1, 3)
        pre, geo = [], []
        for i in range(len(self.groupers)):
            r, ns = self.groupers[i].radius, self.nsamples[i]
            fused_geo = (FUSED_GEOMETRY and xyz.is_cuda and xyz.dtype == torch.float32 and ns <= 64 and ns & (ns - 1) == 0
                         and not xyz.requires_grad and not new_xyz.requires_grad)
            if fused_geo:
                # [centre, nbr, centre - nbr, direction] (:907-913) and density / per-group max (:594-595,:1000-1001)
                rppe = torch.empty((B, npoint, ns, 12), dtype=torch.float32, device=xyz.device)
                dscale = torch.empty((B, npoint, ns, 1), dtype=torch.float32, device=xyz.device)
                pointnet2_utils.pointnet2.pda_geometry(xyz.contiguous(), new_xyz.contiguous(), idxs[i], rppe, dscale,
                                                       B, xyz.shape[1], npoint, ns, r)
            else:
                nbr = pointnet2_utils.group_rows(xyz, idxs[i])                # (B, M, ns, 3) absolute xyz
                diff = nbr - centre
                # gaussian density exp(-|d|^2 / (2 r^2)) / (2.5 r) and direction d / r (pointnet2_utils.py:594-600)
                dist = torch.norm(diff, dim=-1, keepdim=True)
                density = torch.exp(-dist ** 2 / (2 * r ** 2)) / (2.5 * r)    # (B, M, ns, 1)
                direction = diff / r
                # DensityNet on density / per-group max (:1000-1003); ReLU after every BN (:973-979)
                dscale = density / density.max(dim=2, keepdim=True)[0]
                # relative position encoding [centre, nbr, centre - nbr, direction] (:907-913)
                rppe = torch.cat([centre.expand(B, npoint, ns, 3), nbr, -diff, direction], dim=-1)
            geo.append((rppe, dscale, fused_geo))
        # DensityNet of every scale: the scales that take the fused passes share ONE set of launches
        dns = [self.point_density[i].densitynet for i in range(len(self.groupers))]
        fused = [i for i in range(len(self.groupers)) if pointnet2_utils.DensityNetFused.supported(geo[i][1], dns[i])]
        dsc = [g[1] for g in geo]
        if fused:
            # 4 + 5 launches instead of ~45; with a plan on the device (its token count not read yet), over the distinct
            # slots only -- dscale of a repeat slot equals its group's slot 0 (the same neighbour)
            parts_f = [plan_parts[i] if (plan_parts is not None and geo[i][2]) else None for i in fused]
            for i, y in zip(fused, pointnet2_utils.densitynet_multi([dns[i] for i in fused], [dsc[i] for i in fused], parts_f)):
                dsc[i] = y
        for i in range(len(self.groupers)):
            rppe, dscale = geo[i][0], dsc[i]
            if i not in fused and _densitynet_eval_ok(dns[i], dscale):
                dscale = _densitynet_eval(dns[i], dscale)          # inference: the three folded layers in one launch
            elif i not in fused:
                for conv, bn in zip(dns[i].mlp_convs, dns[i].mlp_bns):
                    if _can_fold(conv, bn):
                        dscale = _linear_relu(dscale, *_folded_conv_bn(conv, bn))
                    else:
                        dscale = _bn_relu_lastdim(bn, pointnet2_utils.linear(dscale, conv.weight.flatten(1), conv.bias))
            glob = _mlp_lastdim(self.global_mlps[i], global_in)               # (B, M, C)
            pre.append((rppe, dscale, glob))
        if pending is not None:
            parts, pinned, ev = pending
            ev.synchronize()
            plans = pointnet2_utils.ragged_plans_from(parts, pinned.tolist())
        plans = [p if p is not None and p.fraction <= pointnet2_utils.RAGGED_MAX_FRACTION else None for p in plans]
        # the scale with the most tokens first: its long kernels give the host the time to enqueue the other scales' short ones
        outs = [None] * len(self.groupers)
        order = sorted(range(len(self.groupers)), key=lambda i: -(plans[i].tokens if plans[i] is not None else npoint * self.nsamples[i] * B))
        if not PLAN_OVERLAP:
            order = list(range(len(self.groupers)))
        for i in order:
            ns = self.nsamples[i]
            rppe, dscale, glob = pre[i]
            pre[i] = None
            if (plans[i] is not None and feats_pm.shape[-1] in (16, 32, 64, 128, 256) and FUSED_BN_RELU
                    and pointnet2_utils.FUSED_ASSEMBLE and pointnet2_utils.position_mlp_ragged_supported(self.position_mlp[i], rppe)):
                # position MLP, token assembly and encoder all on the distinct tokens
                rppe_c = pointnet2_utils.position_mlp_ragged(self.position_mlp[i], rppe, plans[i])        # (U, C)
                x = pointnet2_utils.AssembleTokensRagged.apply(rppe_c, dscale, feats_pm, idxs[i], glob, plans[i])
                x = pointnet2_utils.ragged_transformer_block(self.Local_pointformer[i], x, plans[i]).view(B, npoint, -1)
                outs[i] = _mlp_lastdim(self.fin_conv[i], x)
                continue
            rppe = _mlp_lastdim(self.position_mlp[i], rppe)                   # (B, M, ns, C)
            if plans[i] is not None and pointnet2_utils.AssembleTokens.supported(rppe, feats_pm):
                # the encoder on the distinct tokens only (csrc/ragged.hip): x (U, 4C) -> (B, M, 4C)
                x = pointnet2_utils.AssembleTokensRagged.apply(rppe, dscale, feats_pm, idxs[i], glob, plans[i])
                x = pointnet2_utils.ragged_transformer_block(self.Local_pointformer[i], x, plans[i]).view(B, npoint, -1)
                outs[i] = _mlp_lastdim(self.fin_conv[i], x)
                continue
            if pointnet2_utils.AssembleTokens.supported(rppe, feats_pm):
                x = pointnet2_utils.AssembleTokens.apply(rppe, dscale, feats_pm, idxs[i], glob)   # (B, M, ns, 4C)
            else:
                g = pointnet2_utils.group_rows(feats_pm, idxs[i])             # (B, M, ns, C)
                x = torch.cat([rppe, g * dscale, g, glob.unsqueeze(2).expand(-1, -1, ns, -1)], dim=-1)
            D = x.shape[-1]
            # encoder layer + max over nsample (:931)
            x = _transformer_batch_first(self.Local_pointformer[i], x.view(B * npoint, ns, D), pool=True).view(B, npoint, D)
            outs[i] = _mlp_lastdim(self.fin_conv[i], x)                       # (B, M, mlp[-1])
        new_features = torch.cat(outs, dim=-1)                                # (B, M, sum)
        if self.aggregation_layer is not None:
            new_features = _mlp_lastdim(self.aggregation_layer, new_features)
        cls_features = _mlp_lastdim(self.confidence_layers, new_features) if self.confidence_layers is not None else None
        return new_xyz, new_features.transpose(1, 2).contiguous(), cls_features, sampled_idx_list


class Vote_layer(nn.Module):
    """Light voting module with limitation (pointnet2_modules.py:1689-1753)."""

    def __init__(self, mlp_list, pre_channel, max_translate_range):
        super().__init__()
        self.mlp_list = mlp_list
        if len(mlp_list) > 0:
            # the reference re-creates `shared_mlps` inside its loop (:1695-1704), so only the LAST
            # entry of mlp_list survives into mlp_modules, fed by the channel count of the one before
            shared_mlps = []
            for i in range(len(mlp_list)):
                shared_mlps = [nn.Conv1d(pre_channel, mlp_list[i], kernel_size=1, bias=False),
                               nn.BatchNorm1d(mlp_list[i]), nn.ReLU()]
                pre_channel = mlp_list[i]
            self.mlp_modules = nn.Sequential(*shared_mlps)
        else:
            self.mlp_modules = None
        self.ctr_reg = nn.Conv1d(pre_channel, 3, kernel_size=1)
        self.max_offset_limit = torch.tensor(max_translate_range).float() if max_translate_range is not None else None
        # device-resident copy (not in the state dict): the reference does a host->device copy of
        # this 3-vector every forward (:1730), which also breaks hipGraph capture
        self.register_buffer("_offset_limit", self.max_offset_limit.clone().view(1, 1, 3)
                             if self.max_offset_limit is not None else None, persistent=False)

    def forward(self, xyz, features):
        xyz_select = xyz
        new_features = self.mlp_modules(features) if self.mlp_modules is not None else features
        ctr_offsets = self.ctr_reg(new_features).transpose(1, 2)  # (B, N, 3)
        new_features = ctr_offsets[..., 3:]
        ctr_offsets = ctr_offsets[..., :3]
        if self.max_offset_limit is not None:
            lim = self._offset_limit
            limited = torch.where(ctr_offsets > lim, lim, ctr_offsets)
            limited = torch.where(limited < -lim, -lim, limited)
            vote_xyz = xyz_select + limited
        else:
            vote_xyz = xyz_select + ctr_offsets
        return vote_xyz, new_features, xyz_select, ctr_offsets


class PointnetFPModule(nn.Module):
    """Feature propagation: three_nn + inverse-distance three_interpolate + shared MLP
    (pointnet2_modules.py:1776-1824).  Not used by IASSD_Backbone; north_star names the ops."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        shared_mlps = []
        for k in range(len(mlp) - 1):
            shared_mlps.extend([nn.Conv2d(mlp[k], mlp[k + 1], kernel_size=1, bias=False),
                                nn.BatchNorm2d(mlp[k + 1]), nn.ReLU()])
        self.mlp = nn.Sequential(*shared_mlps)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)
        else:
            new_features = interpolated_feats
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)

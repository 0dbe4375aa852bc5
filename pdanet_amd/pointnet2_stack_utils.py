"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:1-263 (BallQuery, GroupingOperation,
QueryAndGroup, FarthestPointSampling, StackFarthestPointSampling, ThreeNN, ThreeInterpolate) over
pdanet_amd.pointnet2_stack_cuda.  Allocation contracts as in the reference (idx zero-filled, temp = 1e10,
grads zero-filled); `torch.cuda.IntTensor(...)` constructors replaced by device-aware factories."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        """xyz (N1+N2..., 3), new_xyz (M1+M2..., 3) -> idx (M, nsample) local indices, empty_ball_mask (M)."""
        assert new_xyz.is_contiguous() and new_xyz_batch_cnt.is_contiguous() and xyz.is_contiguous() and xyz_batch_cnt.is_contiguous()
        B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        empty_ball_mask = (idx[:, 0] == -1)
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        """features (N, C), idx (M, nsample) -> (M, C, nsample)."""
        assert features.is_contiguous() and features_batch_cnt.is_contiguous() and idx.is_contiguous() and idx_batch_cnt.is_contiguous()
        M, nsample = idx.size()
        N, C = features.size()
        B = idx_batch_cnt.shape[0]
        output = torch.empty((M, C, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, output)
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_features = torch.zeros((N, C), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, M, C, N, nsample, grad_out.contiguous(), idx, idx_batch_cnt,
                                            features_batch_cnt, grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """-> new_features (M, 3 + C, nsample), idx (pointnet2_utils.py:113-158)."""
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        grouped_xyz = grouped_xyz.masked_fill(empty_ball_mask[:, None, None], 0)
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
            grouped_features = grouped_features.masked_fill(empty_ball_mask[:, None, None], 0)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features, idx


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
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


class StackFarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        """xyz (N1+N2..., 3); npoint int, list or int tensor -> (sum npoint) global indices (:186-216)."""
        assert xyz.is_contiguous() and xyz.shape[1] == 3
        batch_size = len(xyz_batch_cnt)
        if not isinstance(npoint, torch.Tensor):
            if not isinstance(npoint, list):
                npoint = [npoint for _ in range(batch_size)]
            total = int(sum(npoint))
            npoint = torch.tensor(npoint, device=xyz.device).int()
        else:
            total = int(npoint.sum().item())
        temp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
        output = torch.empty((total,), dtype=torch.int32, device=xyz.device)
        pointnet2.stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, output, npoint)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


stack_farthest_point_sample = StackFarthestPointSampling.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        """-> dist (N, 3) l2 distances, idx (N, 3) global indices of the 3 nearest known points."""
        assert unknown.dim() == 2 and unknown.shape[1] == 3 and known.dim() == 2 and known.shape[1] == 3
        assert len(unknown_batch_cnt) == len(known_batch_cnt)
        dist2 = unknown.new_zeros(unknown.shape)
        idx = unknown_batch_cnt.new_zeros(unknown.shape).int()
        pointnet2.three_nn_wrapper(unknown.contiguous(), unknown_batch_cnt.contiguous(), known.contiguous(),
                                   known_batch_cnt.contiguous(), dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (M, C), idx / weight (N, 3) -> (N, C)."""
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        output = features.new_zeros((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), idx.contiguous(), weight.contiguous(), output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, M = ctx.three_interpolate_for_backward
        grad_features = grad_out.new_zeros((M, grad_out.shape[1]))
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), idx.contiguous(), weight.contiguous(), grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply

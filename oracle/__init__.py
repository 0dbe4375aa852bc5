"""CPU oracle for the pointnet2_batch hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``pdanet_amd`` (the product) never does: it fails loudly when the
HIP library is missing instead of falling back to anything here.

Parity status: "parity unpinned" at the CUDA boundary (see pointnet2_oracle.c header and
DESIGN.md): the reference ships no tests/golden vectors for this path and its CUDA sources
cannot be built in this image.

The functions mirror the reference extension entry points on this path
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:12-33) with the
same positional signatures, operating in place on C-contiguous numpy arrays
(float32 / int32) instead of CUDA tensors.
"""
from .binding import (  # noqa: F401
    build,
    lib_path,
    opt_n_threads,
    contract_mode,
    num_threads,
    set_num_threads,
    ball_query_wrapper,
    ball_query_dilated_wrapper,
    ellipsoid_query,
    group_points_wrapper,
    group_points_grad_wrapper,
    gather_points_wrapper,
    gather_points_grad_wrapper,
    farthest_point_sampling_wrapper,
    furthest_point_sampling_with_dist_wrapper,
    three_nn_wrapper,
    three_interpolate_wrapper,
    three_interpolate_grad_wrapper,
    chamfer_forward,
    chamfer_backward,
    points_in_boxes_gpu,
    boxes_overlap_bev_gpu,
    boxes_iou_bev_gpu,
    nms_gpu,
    nms_normal_gpu,
    stack_ball_query_wrapper,
    stack_group_points_wrapper,
    stack_group_points_grad_wrapper,
    stack_farthest_point_sampling_wrapper,
    stack_three_nn_wrapper,
    stack_three_interpolate_wrapper,
    stack_three_interpolate_grad_wrapper,
)

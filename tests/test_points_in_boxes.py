"""points_in_boxes: oracle known answers derived by hand from
roiaware_pool3d_kernel.cu:16-36,313-336 (CPU) and HIP == oracle (GPU)."""
import math

import numpy as np
import pytest


def _run_oracle(oracle, boxes, pts):
    out = np.full(pts.shape[:2], -1, np.int32)
    assert oracle.points_in_boxes_gpu(np.ascontiguousarray(boxes, np.float32), np.ascontiguousarray(pts, np.float32), out) == 1
    return out


def test_oracle_known_answers(oracle):
    # box 0: axis aligned 4 x 2 x 2 at the origin; box 1: same, rotated 90 deg (long side along y), overlapping box 0
    boxes = np.array([[[0, 0, 0, 4, 2, 2, 0.0], [0, 0, 0, 4, 2, 2, math.pi / 2], [10, 0, 0, 0, 0, 0, 0]]], np.float32)
    pts = np.array([[[0, 0, 0],          # in both -> lowest index 0
                     [1.9, 0.9, 0.9],    # inside box 0 only
                     [0.5, 1.9, 0],      # outside box 0 (|y| > 1), inside rotated box 1
                     [1.9, 1.9, 0],      # in neither
                     [0, 0, 1.0],        # |z - cz| == dz/2 is NOT excluded (test is `>`)
                     [0, 0, 1.0001],     # above
                     [2.000005, 0, 0],   # |x| < dx/2 + 1e-5: inside thanks to the margin
                     [2.00002, 0, 0],    # beyond the margin (and outside box 1: |local y| = 2.00002 > 1)
                     [10, 0, 0],         # degenerate zero-size box 2: |0| < 0 + 1e-5 -> inside
                     ]], np.float32)
    got = _run_oracle(oracle, boxes, pts)[0]
    assert got.tolist() == [0, 0, 1, -1, 0, -1, 0, -1, 2]


def test_oracle_against_float64_geometry(oracle):
    rng = np.random.default_rng(5)
    B, T, M = 2, 23, 5000
    boxes = np.concatenate([rng.uniform(-20, 20, (B, T, 2)), rng.uniform(-1, 1, (B, T, 1)),
                            rng.uniform(1, 6, (B, T, 3)), rng.uniform(-4, 4, (B, T, 1))], -1).astype(np.float32)
    pts = np.concatenate([rng.uniform(-22, 22, (B, M, 2)), rng.uniform(-3, 3, (B, M, 1))], -1).astype(np.float32)
    got = _run_oracle(oracle, boxes, pts)
    # independent float64 evaluation; only points farther than 1e-4 from every face are compared
    b64, p64 = boxes.astype(np.float64), pts.astype(np.float64)
    d = p64[:, :, None, :] - b64[:, None, :, :3]
    c, s = np.cos(-b64[..., 6])[:, None], np.sin(-b64[..., 6])[:, None]
    lx, ly = d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c
    mx, my, mz = np.abs(lx) - b64[:, None, :, 3] / 2, np.abs(ly) - b64[:, None, :, 4] / 2, np.abs(d[..., 2]) - b64[:, None, :, 5] / 2
    inside = (mx < 0) & (my < 0) & (mz < 0)
    clear = (np.minimum(np.minimum(np.abs(mx), np.abs(my)), np.abs(mz)) > 1e-4).all(-1)
    want = np.where(inside.any(-1), inside.argmax(-1), -1)
    assert clear.mean() > 0.99
    assert (got[clear] == want[clear]).all()
    assert (got >= 0).mean() > 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,M", [(1, 1, 1), (2, 23, 4099), (3, 300, 1000), (2, 0, 64), (4, 64, 16384)])
def test_hip_matches_oracle(oracle, B, T, M):
    import torch
    from pdanet_amd import roiaware_pool3d_utils as ru
    rng = np.random.default_rng(B * 1000 + T)
    boxes = np.concatenate([rng.uniform(-20, 20, (B, T, 2)), rng.uniform(-1, 1, (B, T, 1)),
                            rng.uniform(0.5, 6, (B, T, 3)), rng.uniform(-7, 7, (B, T, 1))], -1).astype(np.float32)
    if T > 3:
        boxes[:, -2:] = 0                                    # zero padding rows, as collate_batch produces
    pts = np.concatenate([rng.uniform(-22, 22, (B, M, 2)), rng.uniform(-3, 3, (B, M, 1))], -1).astype(np.float32)
    if T > 0:
        pts[:, : min(M, T)] = boxes[:, : min(M, T), :3]      # box centres (and the origin for the padding rows)
    want = _run_oracle(oracle, boxes, pts) if T > 0 else np.full((B, M), -1, np.int32)
    got = ru.points_in_boxes_gpu(torch.from_numpy(pts).cuda(), torch.from_numpy(boxes).cuda())
    assert got.dtype == torch.int32 and tuple(got.shape) == (B, M)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_rejects_cpu_tensors():
    import torch
    from pdanet_amd import roiaware_pool3d_utils as ru
    with pytest.raises(RuntimeError):
        ru.roiaware_pool3d_cuda.points_in_boxes_gpu(torch.zeros(1, 1, 7), torch.zeros(1, 4, 3), torch.zeros(1, 4, dtype=torch.int32))

"""Rotated BEV overlap / IoU / NMS: oracle known answers derived from the
geometry (CPU) and HIP == oracle bit for bit (GPU)."""
import math

import numpy as np
import pytest


def _boxes(rng, n, spread=6.0):
    return np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(-1, 1, (n, 1)), rng.uniform(0.5, 5, (n, 3)),
                           rng.uniform(-4, 4, (n, 1))], -1).astype(np.float32)


def test_oracle_overlap_known_answers(oracle):
    a = np.array([[0, 0, 0, 4, 2, 1, 0.0]], np.float32)
    b = np.array([[0, 0, 0, 4, 2, 1, 0.0],                 # identical: 8
                  [1, 0.5, 0, 4, 2, 1, 0.0],               # shifted axis-aligned: 3 x 1.5 = 4.5
                  [10, 0, 0, 4, 2, 1, 0.3],                # disjoint: 0
                  [0, 0, 0, 2, 2, 1, math.pi / 4],         # 2x2 square rotated 45 deg inside the 4x2 strip: 4 - 2*(sqrt2-1)^2
                  [0, 0, 0, 2, 4, 1, math.pi / 2],         # 2x4 rotated 90 deg == the same 4x2 box: 8
                  ], np.float32)
    out = np.zeros((1, 5), np.float32)
    oracle.boxes_overlap_bev_gpu(a, b, out)
    want = [8.0, 4.5, 0.0, 4 - 2 * (math.sqrt(2) - 1) ** 2, 8.0]
    np.testing.assert_allclose(out[0], want, rtol=2e-5, atol=1e-5)
    iou = np.zeros((1, 5), np.float32)
    oracle.boxes_iou_bev_gpu(a, b, iou)
    np.testing.assert_allclose(iou[0, :3], [1.0, 4.5 / (16 - 4.5), 0.0], rtol=2e-5, atol=1e-6)


def test_oracle_overlap_against_polygon_clipping(oracle):
    """Independent float64 Sutherland-Hodgman clipping; generic boxes only (the reference adds a 1e-2 margin
    to its corner-inside tests, which matters for touching configurations)."""
    rng = np.random.default_rng(3)
    a, b = _boxes(rng, 40), _boxes(rng, 50)
    got = np.zeros((40, 50), np.float32)
    oracle.boxes_overlap_bev_gpu(a, b, got)

    def corners(bx):
        hx, hy, c, s = bx[3] / 2, bx[4] / 2, math.cos(bx[6]), math.sin(bx[6])
        return [(bx[0] + x * c - y * s, bx[1] + x * s + y * c) for x, y in ((-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy))]

    def clip(poly, p, q):
        out = []
        for k in range(len(poly)):
            cur, prv = poly[k], poly[k - 1]
            side = lambda r: (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])  # noqa: E731
            sc, sp = side(cur), side(prv)
            if sc >= 0:
                if sp < 0:
                    t = sp / (sp - sc); out.append((prv[0] + t * (cur[0] - prv[0]), prv[1] + t * (cur[1] - prv[1])))
                out.append(cur)
            elif sp >= 0:
                t = sp / (sp - sc); out.append((prv[0] + t * (cur[0] - prv[0]), prv[1] + t * (cur[1] - prv[1])))
        return out

    def area(pa, pb):
        poly = pa
        for k in range(4):
            if not poly:
                return 0.0
            poly = clip(poly, pb[k], pb[(k + 1) % 4])
        return abs(sum(poly[k - 1][0] * poly[k][1] - poly[k][0] * poly[k - 1][1] for k in range(len(poly)))) / 2 if poly else 0.0
    want = np.array([[area(corners(x.astype(np.float64)), corners(y.astype(np.float64))) for y in b] for x in a])
    assert (want > 0.05).mean() > 0.1
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-2)      # the reference's 1e-2 corner margin
    assert np.abs(got - want).mean() < 2e-3


def test_oracle_nms_known_answers(oracle):
    boxes = np.array([[0, 0, 0, 4, 2, 1, 0.0], [0.2, 0, 0, 4, 2, 1, 0.05], [10, 0, 0, 4, 2, 1, 1.0], [10, 0.1, 0, 4, 2, 1, 1.0],
                      [0, 5, 0, 1, 1, 1, 0.0]], np.float32)
    keep = np.zeros(5, np.int64)
    assert oracle.nms_gpu(boxes, keep, 0.1) == 3 and keep[:3].tolist() == [0, 2, 4]
    assert oracle.nms_gpu(boxes, keep, 0.99) == 5
    assert oracle.nms_normal_gpu(boxes, keep, 0.1) == 3 and keep[:3].tolist() == [0, 2, 4]
    # suppression is by KEPT boxes only: 1 suppresses 2 only if 1 itself survives
    chain = np.array([[0, 0, 0, 4, 2, 1, 0], [2.2, 0, 0, 4, 2, 1, 0], [4.4, 0, 0, 4, 2, 1, 0]], np.float32)
    assert oracle.nms_gpu(chain, keep, 0.2) == 2 and keep[:2].tolist() == [0, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("na,nb", [(1, 1), (37, 130), (256, 64), (5, 1000)])
def test_hip_overlap_iou_match_oracle(oracle, na, nb):
    import torch
    from pdanet_amd import iou3d_nms_utils as iu
    rng = np.random.default_rng(na * 31 + nb)
    a, b = _boxes(rng, na), _boxes(rng, nb)
    b[: min(na, nb)] = a[: min(na, nb)]                       # identical pairs
    if nb > 3:
        b[3, 6] = a[0, 6] + np.float32(math.pi / 2)           # parallel edges
        b[3, :2] = a[0, :2]
    want_o, want_i = np.zeros((na, nb), np.float32), np.zeros((na, nb), np.float32)
    oracle.boxes_overlap_bev_gpu(a, b, want_o); oracle.boxes_iou_bev_gpu(a, b, want_i)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    got_o = torch.zeros(na, nb, device="cuda")
    iu.iou3d_nms_cuda.boxes_overlap_bev_gpu(ta, tb, got_o)
    assert np.array_equal(got_o.cpu().numpy(), want_o)
    assert np.array_equal(iu.boxes_iou_bev(ta, tb).cpu().numpy(), want_i)
    # 3-D IoU composition against the same formula on the oracle's overlaps
    i3 = iu.boxes_iou3d_gpu(ta, tb).cpu().numpy()
    zmax_a, zmin_a = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    zmax_b, zmin_b = (b[:, 2] + b[:, 5] / 2)[None], (b[:, 2] - b[:, 5] / 2)[None]
    o3 = want_o * np.clip(np.minimum(zmax_a, zmax_b) - np.maximum(zmin_a, zmin_b), 0, None)
    vol = (a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None] - o3
    np.testing.assert_allclose(i3, o3 / np.clip(vol, 1e-6, None), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("n,thresh,normal", [(1, 0.1, False), (63, 0.1, False), (64, 0.3, True), (65, 0.01, False),
                                             (1000, 0.1, False), (1024, 0.01, False), (4096, 0.1, True), (4100, 0.25, False)])
def test_hip_nms_matches_oracle(oracle, n, thresh, normal):
    import torch
    from pdanet_amd import iou3d_nms_utils as iu
    rng = np.random.default_rng(n)
    B = 3
    boxes = np.stack([_boxes(rng, n, spread=4.0 + 0.4 * math.sqrt(n)) for _ in range(B)])
    nv = np.array([n, max(0, n - 7), n // 2], np.int32)
    keep, num = iu.nms_batched(torch.from_numpy(boxes).cuda(), thresh, num_valid=torch.from_numpy(nv).cuda(), normal=normal)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for s in range(B):
        want = np.zeros(n, np.int64)
        fn = oracle.nms_normal_gpu if normal else oracle.nms_gpu
        k = fn(np.ascontiguousarray(boxes[s, : nv[s]]), want, thresh) if nv[s] > 0 else 0
        assert num[s] == k
        assert np.array_equal(keep[s, :k], want[:k]) and (keep[s, k:] == -1).all()


@pytest.mark.gpu
def test_reference_named_nms_wrappers(oracle):
    import torch
    from pdanet_amd import iou3d_nms_utils as iu
    rng = np.random.default_rng(9)
    boxes, scores = _boxes(rng, 300, spread=8.0), rng.random(300).astype(np.float32)
    sel, _ = iu.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.1, pre_maxsize=200)
    order = np.argsort(-scores, kind="stable")[:200]
    want = np.zeros(200, np.int64)
    k = oracle.nms_gpu(np.ascontiguousarray(boxes[order]), want, 0.1)
    assert np.array_equal(sel.cpu().numpy(), order[want[:k]])

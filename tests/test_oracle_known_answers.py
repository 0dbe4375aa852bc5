"""Pins the CPU oracle with hand-derivable known answers.

The reference has no tests of its own; every expected value below follows from reading
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/*.cu (cited per test).
"""
import numpy as np
import pytest


def _fps(oracle, xyz, m):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, dtype=np.float32)
    idx = np.full((b, m), -7, dtype=np.int32)
    assert oracle.farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx) == 1
    return idx, temp


def _bitrev(v, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def _fps_bruteforce_key(xyz, m, bs):
    """Independent statement of sampling_gpu.cu:93-209: arg-max of min-dist with the
    tie-break key (bitreverse(k mod bs), k) -- SURVEY.md Appendix A.2."""
    n = xyz.shape[0]
    bits = int(np.log2(bs))
    temp = np.full(n, np.float32(1e10), dtype=np.float32)
    out = [0]
    old = 0
    for _ in range(1, m):
        d = (xyz - xyz[old]).astype(np.float32)
        dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
        # fma(dz,dz,fma(dy,dy,dx*dx)) evaluated exactly in float64 then rounded per step
        t = (dx.astype(np.float64) * dx).astype(np.float32)
        t = (dy.astype(np.float64) * dy + t.astype(np.float64)).astype(np.float32)
        t = (dz.astype(np.float64) * dz + t.astype(np.float64)).astype(np.float32)
        temp = np.minimum(t, temp)
        best = None
        for k in range(n):
            key = (-float(temp[k]), _bitrev(k % bs, bits), k)
            if best is None or key < best:
                best = key
        old = best[2]
        out.append(old)
    return np.array(out, dtype=np.int32)


def test_opt_n_threads(oracle):
    # cuda_utils.h:10-14
    for n, e in [(1, 1), (2, 2), (3, 2), (7, 4), (8, 8), (13, 8), (100, 64), (1023, 512),
                 (1024, 1024), (4096, 1024), (16384, 1024), (65536, 1024), (60000, 1024)]:
        assert oracle.opt_n_threads(n) == e


def test_fps_collinear_tiebreak(oracle):
    # x = 0..7, m = 3: after picking 0 then 7, points 3 and 4 tie (min-dist 9); the
    # shared-memory tree (sampling_gpu.cu:143-203) keeps the lane with the smaller
    # bit-reversed id: bitrev3(3)=6, bitrev3(4)=1 -> 4 wins, NOT the lower index.
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = np.arange(8)
    idx, _ = _fps(oracle, xyz, 3)
    assert idx.tolist() == [[0, 7, 4]]


def test_fps_identical_points(oracle):
    xyz = np.ones((2, 37, 3), np.float32)
    idx, temp = _fps(oracle, xyz, 5)
    assert (idx == 0).all()
    assert (temp == 0).all()


def test_fps_m_edge_cases(oracle):
    xyz = np.random.default_rng(0).normal(size=(1, 10, 3)).astype(np.float32)
    idx, temp = _fps(oracle, xyz, 1)
    assert idx.tolist() == [[0]] and (temp == np.float32(1e10)).all()  # m=1: no iteration
    # m <= 0 returns immediately (sampling_gpu.cu:101)
    temp = np.full((1, 10), 1e10, np.float32)
    idx0 = np.zeros((1, 0), np.int32)
    oracle.farthest_point_sampling_wrapper(1, 10, 0, xyz, temp, idx0)


@pytest.mark.parametrize("n,m", [(8, 8), (13, 6), (37, 12), (100, 30), (300, 40)])
def test_fps_lattice_ties_vs_bruteforce_key(oracle, n, m):
    # integer lattice => many exact distance ties; tree emulation must equal the closed-form key
    rng = np.random.default_rng(n)
    xyz = rng.integers(0, 4, size=(1, n, 3)).astype(np.float32)
    idx, _ = _fps(oracle, xyz, m)
    exp = _fps_bruteforce_key(xyz[0], m, oracle.opt_n_threads(n))
    assert idx[0].tolist() == exp.tolist()


def test_fps_random_vs_bruteforce_key(oracle):
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-10, 10, size=(1, 1500, 3)).astype(np.float32)
    idx, _ = _fps(oracle, xyz, 64)
    exp = _fps_bruteforce_key(xyz[0], 64, 1024)
    assert idx[0].tolist() == exp.tolist()
    assert len(set(idx[0].tolist())) == 64


def test_fps_with_dist_matches_xyz_fps(oracle):
    # sampling_gpu.cu:294: d = dataset[old*n + k]; feeding the exact squared distances
    # (same fma expression, point - sample order) must reproduce D-FPS.
    rng = np.random.default_rng(11)
    xyz = rng.uniform(-3, 3, size=(2, 50, 3)).astype(np.float32)
    idx, _ = _fps(oracle, xyz, 20)
    d = (xyz[:, None, :, :] - xyz[:, :, None, :]).astype(np.float32)  # [b, old, k] = p_k - p_old
    t = (d[..., 0].astype(np.float64) * d[..., 0]).astype(np.float32)
    t = (d[..., 1].astype(np.float64) * d[..., 1] + t).astype(np.float32)
    t = (d[..., 2].astype(np.float64) * d[..., 2] + t).astype(np.float32)
    temp = np.full((2, 50), 1e10, np.float32)
    idx2 = np.zeros((2, 20), np.int32)
    assert oracle.furthest_point_sampling_with_dist_wrapper(2, 50, 20, np.ascontiguousarray(t), temp, idx2) == 2
    assert (idx == idx2).all()


def _bq(oracle, new_xyz, xyz, r, ns, fill=0):
    new_xyz = np.ascontiguousarray(new_xyz, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float32)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.full((b, m, ns), fill, np.int32)
    assert oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, idx) == 1
    return idx


def test_ball_query_semantics(oracle):
    # ball_query_gpu.cu:29-44
    xyz = np.zeros((1, 10, 3), np.float32)
    xyz[0, :, 0] = np.arange(10)            # points at x = 0..9
    ctr = np.array([[[100, 0, 0],           # nothing in range -> untouched
                     [4, 0, 0],             # r=1.5: hits 3,4,5
                     [4, 0, 0]]], np.float32)
    idx = _bq(oracle, ctr, xyz, 1.5, 5, fill=-9)
    assert idx[0, 0].tolist() == [-9] * 5              # row left untouched (:34 never true)
    assert idx[0, 1].tolist() == [3, 4, 5, 3, 3]       # first hit pre-fills all slots (:35-39)
    # strict '<': d2 == r^2 excluded (:34); r = 1 -> only the point itself
    idx = _bq(oracle, ctr, xyz, 1.0, 3)
    assert idx[0, 1].tolist() == [4, 4, 4]
    # more hits than nsample: first nsample in ascending index, then break (:42)
    idx = _bq(oracle, ctr, xyz, 100.0, 4)
    assert idx[0, 1].tolist() == [0, 1, 2, 3]
    # zero-hit row with the caller's zero fill reads as neighbour 0 repeated
    idx = _bq(oracle, ctr, xyz, 1.5, 2, fill=0)
    assert idx[0, 0].tolist() == [0, 0]


def test_ball_query_dilated_double_append(oracle):
    # ball_query_gpu.cu:96-115: d2 == 0 is appended by the first `if` and again by the
    # shell test when min_radius == 0.
    xyz = np.zeros((1, 4, 3), np.float32)
    xyz[0, :, 0] = [5, 0, 1, 2]
    ctr = np.zeros((1, 1, 3), np.float32)
    idx = np.zeros((1, 1, 6), np.int32)
    oracle.ball_query_dilated_wrapper(1, 4, 1, 1.5, 0.0, 6, ctr, xyz, idx)
    assert idx[0, 0].tolist() == [1, 1, 2, 1, 1, 1]
    # shell [1, 2.5): excludes point at 0 by the shell test but d2==0 still accepts it once
    idx = np.zeros((1, 1, 4), np.int32)
    oracle.ball_query_dilated_wrapper(1, 4, 1, 2.5, 1.0, 4, ctr, xyz, idx)
    assert idx[0, 0].tolist() == [1, 2, 3, 1]
    # d2 == 0 fills the last slot -> break before the second append
    idx = np.zeros((1, 1, 1), np.int32)
    oracle.ball_query_dilated_wrapper(1, 4, 1, 1.5, 0.0, 1, ctr, xyz, idx)
    assert idx[0, 0].tolist() == [1]


def test_group_and_gather_arange(oracle):
    # group/gather of arange features reproduces idx (group_points_gpu.cu:53-72, sampling_gpu.cu:8-24)
    rng = np.random.default_rng(3)
    b, c, n, m, ns = 2, 3, 17, 5, 4
    feats = np.tile(np.arange(n, dtype=np.float32), (b, c, 1)) + \
        100 * np.arange(c, dtype=np.float32)[None, :, None]
    feats = np.ascontiguousarray(feats)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    out = np.zeros((b, c, m, ns), np.float32)
    oracle.group_points_wrapper(b, c, n, m, ns, feats, idx, out)
    for ci in range(c):
        assert (out[:, ci] == idx + 100 * ci).all()
    gidx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    gout = np.zeros((b, c, m), np.float32)
    oracle.gather_points_wrapper(b, c, n, m, feats, gidx, gout)
    for ci in range(c):
        assert (gout[:, ci] == gidx + 100 * ci).all()


def test_group_and_gather_grad_scatter_add(oracle):
    b, c, n, m, ns = 1, 2, 6, 3, 2
    idx = np.array([[[0, 0], [5, 0], [2, 2]]], np.int32)
    go = np.arange(b * c * m * ns, dtype=np.float32).reshape(b, c, m, ns)
    gp = np.zeros((b, c, n), np.float32)
    oracle.group_points_grad_wrapper(b, c, n, m, ns, go, idx, gp)
    assert gp[0, 0].tolist() == [0 + 1 + 3, 0, 4 + 5, 0, 0, 2]
    assert gp[0, 1].tolist() == [6 + 7 + 9, 0, 10 + 11, 0, 0, 8]
    gidx = np.array([[4, 4, 1]], np.int32)
    ggo = np.array([[[1, 2, 3], [10, 20, 30]]], np.float32)
    ggp = np.zeros((1, 2, n), np.float32)
    oracle.gather_points_grad_wrapper(1, 2, n, 3, ggo, gidx, ggp)
    assert ggp[0, 0].tolist() == [0, 3, 0, 0, 3, 0]
    assert ggp[0, 1].tolist() == [0, 30, 0, 0, 30, 0]


def test_three_nn_semantics(oracle):
    # interpolate_gpu.cu:37-58: strict '<' keeps the lower index first among equal distances
    known = np.array([[[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, 0, 3], [0, -1, 0]]], np.float32)
    unknown = np.zeros((1, 1, 3), np.float32)
    d2 = np.zeros((1, 1, 3), np.float32)
    idx = np.zeros((1, 1, 3), np.int32)
    oracle.three_nn_wrapper(1, 1, 5, unknown, known, d2, idx)
    assert idx[0, 0].tolist() == [0, 1, 2] and d2[0, 0].tolist() == [1, 1, 1]
    # m < 3: untouched bests stay idx 0 / 1e40 -> +inf in float (:37-38, :57)
    d2 = np.zeros((1, 1, 3), np.float32)
    idx = np.full((1, 1, 3), -5, np.int32)
    oracle.three_nn_wrapper(1, 1, 2, unknown, np.ascontiguousarray(known[:, 3:5]), d2, idx)
    assert idx[0, 0].tolist() == [1, 0, 0]
    assert d2[0, 0, 0] == 1 and d2[0, 0, 1] == 9 and np.isinf(d2[0, 0, 2])


def test_three_interpolate_and_grad(oracle):
    b, c, m, n = 1, 2, 4, 2
    pts = np.array([[[1, 2, 3, 4], [10, 20, 30, 40]]], np.float32)
    idx = np.array([[[0, 1, 2], [3, 3, 0]]], np.int32)
    w = np.array([[[0.5, 0.25, 0.25], [0.5, 0.25, 0.25]]], np.float32)
    out = np.zeros((b, c, n), np.float32)
    oracle.three_interpolate_wrapper(b, c, m, n, pts, idx, w, out)
    assert out[0, 0].tolist() == [0.5 + 0.5 + 0.75, 2 + 1 + 0.25]
    assert out[0, 1].tolist() == [5 + 5 + 7.5, 20 + 10 + 2.5]
    go = np.array([[[1, 2], [4, 8]]], np.float32)
    gp = np.zeros((b, c, m), np.float32)
    oracle.three_interpolate_grad_wrapper(b, c, n, m, go, idx, w, gp)
    assert gp[0, 0].tolist() == [0.5 + 0.5, 0.25, 0.25, 1.0 + 0.5]
    assert gp[0, 1].tolist() == [2 + 2, 1, 1, 4 + 2]


def test_chamfer_semantics(oracle):
    # chamferthreed.cu:12-134: nearest point, LOWEST index among equal distances, also across the
    # 512-point tiles (:124 keeps the earlier tile unless strictly smaller)
    xyz1 = np.zeros((1, 2, 3), np.float32); xyz1[0, 1, 0] = 10
    xyz2 = np.zeros((1, 1100, 3), np.float32); xyz2[0, :, 0] = 5
    xyz2[0, 3, 0] = 1; xyz2[0, 700, 0] = 1; xyz2[0, 1050, 0] = -1   # three equidistant (d2 = 1) from the origin
    xyz2[0, 600, 0] = 9.5
    d1 = np.zeros((1, 2), np.float32); d2 = np.zeros((1, 1100), np.float32)
    i1 = np.zeros((1, 2), np.int32); i2 = np.zeros((1, 1100), np.int32)
    assert oracle.chamfer_forward(xyz1, xyz2, d1, d2, i1, i2) == 1
    assert i1[0].tolist() == [3, 600] and d1[0].tolist() == [1.0, 0.25]
    assert i2[0, 3] == 0 and i2[0, 600] == 1 and d2[0, 0] == 25.0
    # gradient: d(dist1_j)/d(p1_j) = 2 (p1_j - p2_idx), and the opposite sign on the target
    g1 = np.zeros_like(xyz1); g2 = np.zeros_like(xyz2)
    gd1 = np.ones((1, 2), np.float32); gd2 = np.zeros((1, 1100), np.float32)
    assert oracle.chamfer_backward(xyz1, xyz2, g1, g2, gd1, gd2, i1, i2) == 1
    assert g1[0, 0].tolist() == [-2.0, 0.0, 0.0] and g1[0, 1].tolist() == [1.0, 0.0, 0.0]
    assert g2[0, 3].tolist() == [2.0, 0.0, 0.0] and g2[0, 600].tolist() == [-1.0, 0.0, 0.0]


def test_ball_query_non_finite_point_is_never_a_hit(oracle):
    """ball_query_gpu.cu:33-34: d2 = inf (or NaN) fails `d2 < radius2`, so a point with an inf coordinate is skipped."""
    xyz = np.array([[[0, 0, 0], [np.inf, 0, 0], [0.5, 0, 0], [0, -np.inf, 0]]], np.float32)
    new_xyz = np.array([[[0, 0, 0]]], np.float32)
    idx = np.zeros((1, 1, 4), np.int32)
    oracle.ball_query_wrapper(1, 4, 1, 1.0, 4, new_xyz, xyz, idx)
    assert idx.tolist() == [[[0, 2, 0, 0]]]


def test_ellipsoid_query_extends_the_ball_along_the_principal_axis(oracle):
    """ellipsoid_query_gpu.cu:311-498 by hand: hits on a line along x -> the covariance's largest eigenvector is x, which
    the second pass scales by e1 (rspoint[0] uses the LAST eigenvector column, :455-457).  With (e1, e2, e3) = (3, 0.5, 1):
    the first query (radius e3 = 1) finds points 0..3; the point at x = 2.5 lies inside the re-oriented ellipsoid and is
    appended, the points at y = 0.8 and z = 0.8 (inside the ball of radius 1? no: listed already if so) are not new, the
    point at y = 2.5 stays outside.  A centre with fewer than 3 hits keeps the plain ball-query row."""
    xyz = np.array([[[0.1, 0, 0], [0.4, 0, 0], [-0.3, 0, 0], [0.7, 0, 0],      # 0..3: on the x axis, inside radius 1
                     [2.5, 0, 0],                                             # 4: outside the ball, inside the ellipsoid along x
                     [0, 2.5, 0],                                             # 5: same distance along y: outside (e2 = 0.5)
                     [50, 50, 50], [50.2, 50, 50]]], np.float32)             # 6, 7: a far pair for the second centre
    new_xyz = np.array([[[0.05, 0.0, 0.0], [50.1, 50, 50]]], np.float32)
    idx = oracle.ellipsoid_query(new_xyz, xyz, 3.0, 0.5, 1.0, 8)
    assert idx[0, 0].tolist() == [0, 1, 2, 3, 4, 0, 0, 0]
    assert idx[0, 1].tolist() == [6, 7, 6, 6, 6, 6, 6, 6]                    # 2 hits: no second pass

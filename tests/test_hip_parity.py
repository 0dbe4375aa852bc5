"""-m gpu: the HIP kernels, called through the C ABI (ctypes mirror of the reference's
extension module), against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact for sampled indices and neighbour lists; 1e-4 for
interpolated features; gradients (float atomics in the reference too) 1e-4 relative.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

# 1: the build with nvcc's default contraction of the distance expression (the product); 0: tests/test_contract0.py
# re-runs the index-exact cases of this file with the UNcontracted builds of library and oracle
EXPECT_CONTRACT = int(os.environ.get("PDA_EXPECT_CONTRACT", "1"))


@pytest.fixture(scope="module")
def ext(oracle):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pdanet_amd import pointnet2_batch_cuda
    from pdanet_amd import _lib
    lib = _lib.load()  # raises if libpda_pointnet2.so is missing: no fallback
    assert lib.pda_fp_contract_mode() == EXPECT_CONTRACT and oracle.contract_mode() == EXPECT_CONTRACT
    return pointnet2_batch_cuda


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cloud(b, n, seed, dist="L"):
    from pdanet_amd import synth
    return synth.batch_xyz(b, n, config_id=seed, dist=dist)


# ---------------------------------------------------------------- FPS
def fps_both(ext, oracle, xyz, m):
    b, n, _ = xyz.shape
    temp_o = np.full((b, n), 1e10, np.float32)
    idx_o = np.zeros((b, m), np.int32)
    oracle.farthest_point_sampling_wrapper(b, n, m, xyz, temp_o, idx_o)
    xyz_d = dev(xyz)
    temp_d = torch.full((b, n), 1e10, dtype=torch.float32, device="cuda")
    idx_d = torch.full((b, m), -1, dtype=torch.int32, device="cuda")
    assert ext.farthest_point_sampling_wrapper(b, n, m, xyz_d, temp_d, idx_d) == 1
    torch.cuda.synchronize()
    return idx_o, temp_o, idx_d.cpu().numpy(), temp_d.cpu().numpy()


@pytest.mark.parametrize("b,n,m,dist", [
    (1, 4096, 1024, "L"),      # BASELINE config 1
    (2, 16384, 4096, "L"),     # ONCE layer 1 (the headline FPS)
    (2, 16384, 4096, "U"),
    (3, 1000, 300, "L"),       # n < 1024: reference block size 512, bit-reversed tie-break on 9 bits
    (2, 1024, 1024, "U"),      # m == n
    (1, 1500, 7, "L"),         # n not a power of two, P = 2 with empty slots
    (2, 5000, 512, "L"),       # P = 8 (5 slots used)
    (1, 20000, 256, "L"),      # 16384 < n <= 65536: the cooperative chain kernel (K = 2)
    (1, 30000, 128, "L"),      # the same, K = 2
    (1, 70000, 300, "L"),      # streaming kernel (n > 65536)
    (1, 65, 65, "U"), (1, 64, 10, "U"), (1, 3, 3, "U"), (1, 1, 1, "U"),
])
def test_fps_index_exact(ext, oracle, b, n, m, dist):
    xyz = cloud(b, n, seed=n + m, dist=dist)
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d)
    assert np.array_equal(temp_o, temp_d)   # final min-distances, bit for bit


def test_fps_exact_ties_lattice(ext, oracle):
    # integer lattice: massive exact ties exercise the (bitreverse(k mod bs), k) order
    rng = np.random.default_rng(7)
    for n, m in [(2048, 512), (777, 200), (4096, 900), (100, 60)]:
        xyz = rng.integers(0, 6, size=(2, n, 3)).astype(np.float32)
        idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
        assert np.array_equal(idx_o, idx_d), (n, m)
        assert np.array_equal(temp_o, temp_d)


@pytest.mark.parametrize("case", ["select_all", "duplicates", "lattice_16k", "two_clusters", "line", "m3"])
def test_fps_chain_kernel_edge_cases(ext, oracle, case):
    """fps_chain_kernel (2048 <= n <= 16384) decides several samples per synchronisation; the cases where its proofs are
    tight: every point selected (values fall to exact zeros), duplicated points (d = 0 reaches records of equal
    coordinates), dense integer lattices (ties in value between waves, ties between best and second-best), two far
    clusters (long chains alternate between them), collinear points (every sample reaches its neighbours' records)."""
    rng = np.random.default_rng(11)
    if case == "select_all":
        xyz, m = cloud(2, 2048, seed=1), 2048
    elif case == "duplicates":
        base = rng.normal(size=(1, 300, 3)).astype(np.float32)
        xyz, m = np.ascontiguousarray(np.tile(base, (1, 10, 1))[:, rng.permutation(3000)]), 1200   # every point ten times
    elif case == "lattice_16k":
        xyz, m = rng.integers(0, 12, size=(1, 16384, 3)).astype(np.float32), 3000                   # 1728 distinct sites
    elif case == "two_clusters":
        xyz = rng.normal(size=(1, 8192, 3)).astype(np.float32)
        xyz[0, ::2] += np.float32(1000.0)
        m = 2500
    elif case == "line":
        xyz = np.zeros((1, 4096, 3), np.float32); xyz[0, :, 0] = rng.permutation(4096).astype(np.float32) * 0.25
        m = 1500
    else:
        xyz, m = cloud(1, 16384, seed=3), 3
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d)
    assert np.array_equal(temp_o, temp_d)


@pytest.mark.parametrize("case", ["k2_full", "k2_small_tail", "k4_long", "duplicates", "lattice", "two_clusters", "line", "select_many"])
def test_fps_cooperative_chain_kernel_cases(ext, oracle, case):
    """fps_chain_coop_kernel (16384 < n <= 65536: K = 2...4 workgroups per scene, each publishing its 64 row records, every
    workgroup walking the scene's 64 K): whole runs at K = 2 and K = 4 (ragged last workgroup), and the cases where the walk's
    proofs are tight ACROSS workgroups -- duplicated points owned by different workgroups, dense lattices (equal values in
    records of different workgroups: the tie-break value decides), two far clusters, collinear points, a large share of the
    cloud selected."""
    rng = np.random.default_rng(23)
    if case == "k2_full":
        xyz, m = cloud(2, 24576, seed=5), 6144
    elif case == "k2_small_tail":
        xyz, m = cloud(2, 16385, seed=6), 4096                      # the second workgroup owns 8192 points, one more than... a ragged split
    elif case == "k4_long":
        xyz, m = cloud(1, 50000, seed=7), 8000
    elif case == "duplicates":
        base = rng.normal(size=(1, 2000, 3)).astype(np.float32)
        xyz, m = np.ascontiguousarray(np.tile(base, (1, 10, 1))[:, rng.permutation(20000)]), 5000    # every point ten times, K = 2
    elif case == "lattice":
        xyz, m = rng.integers(0, 14, size=(1, 40000, 3)).astype(np.float32), 4000                    # 2744 distinct sites, K = 3
    elif case == "two_clusters":
        xyz = rng.normal(size=(1, 30000, 3)).astype(np.float32)
        xyz[0, ::2] += np.float32(1000.0)
        m = 6000
    elif case == "line":
        xyz = np.zeros((1, 20000, 3), np.float32); xyz[0, :, 0] = rng.permutation(20000).astype(np.float32) * 0.25
        m = 5000
    else:
        xyz, m = cloud(1, 17000, seed=8), 16000
    ext.fps_coop_timeouts(reset=True)
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d)
    assert np.array_equal(temp_o, temp_d)
    assert ext.fps_coop_timeouts() == 0


def test_fps_known_answers_on_gpu(ext):
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = np.arange(8)
    temp = torch.full((1, 8), 1e10, device="cuda")
    idx = torch.zeros((1, 3), dtype=torch.int32, device="cuda")
    ext.farthest_point_sampling_wrapper(1, 8, 3, dev(xyz), temp, idx)
    assert idx.cpu().tolist() == [[0, 7, 4]]
    ones = torch.ones((2, 37, 3), device="cuda")
    temp = torch.full((2, 37), 1e10, device="cuda")
    idx = torch.full((2, 5), -1, dtype=torch.int32, device="cuda")
    ext.farthest_point_sampling_wrapper(2, 37, 5, ones, temp, idx)
    assert (idx == 0).all()
    # m <= 0 returns at once, nothing written
    idx = torch.full((1, 4), -3, dtype=torch.int32, device="cuda")
    ext.farthest_point_sampling_wrapper(1, 37, 0, ones[:1].contiguous(), temp[:1].contiguous(), idx)
    assert (idx == -3).all()


def test_fps_with_dist(ext, oracle):
    rng = np.random.default_rng(3)
    for b, n, m in [(2, 300, 100), (1, 1500, 64), (1, 64, 64)]:
        dist = rng.uniform(0, 10, size=(b, n, n)).astype(np.float32)
        temp_o = np.full((b, n), 1e10, np.float32)
        idx_o = np.zeros((b, m), np.int32)
        assert oracle.furthest_point_sampling_with_dist_wrapper(b, n, m, dist, temp_o, idx_o) == 2
        temp_d = torch.full((b, n), 1e10, device="cuda")
        idx_d = torch.zeros((b, m), dtype=torch.int32, device="cuda")
        assert ext.furthest_point_sampling_with_dist_wrapper(b, n, m, dev(dist), temp_d, idx_d) == 2
        assert np.array_equal(idx_o, idx_d.cpu().numpy())
        assert np.array_equal(temp_o, temp_d.cpu().numpy())


# ---------------------------------------------------------------- ball query
def bq_both(ext, oracle, new_xyz, xyz, r, ns, fill=0):
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx_o = np.full((b, m, ns), fill, np.int32)
    oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, idx_o)
    idx_d = torch.full((b, m, ns), fill, dtype=torch.int32, device="cuda")
    assert ext.ball_query_wrapper(b, n, m, r, ns, dev(new_xyz), dev(xyz), idx_d) == 1
    return idx_o, idx_d.cpu().numpy()


@pytest.mark.parametrize("b,n,m,r,ns,dist", [
    (1, 4096, 1024, 0.8, 16, "L"), (1, 4096, 1024, 1.6, 32, "L"),       # config 1
    (2, 16384, 16384, 0.2, 16, "L"), (2, 16384, 16384, 0.8, 32, "L"),   # ONCE layer 0
    (2, 16384, 4096, 1.6, 32, "L"), (2, 16384, 4096, 0.8, 16, "U"),     # ONCE layer 1
    (2, 4096, 2048, 4.8, 32, "L"),                                      # layer 2
    (2, 2048, 1024, 12.8, 64, "L"),                                     # layer 5, ns 64
    (3, 1000, 77, 2.0, 5, "L"),       # ragged: M not a multiple of 64, N not of 8, odd ns
    (1, 9, 1, 100.0, 1, "U"), (1, 7, 130, 3.0, 3, "L"), (2, 513, 64, 5.0, 128, "L"),
])
def test_ball_query_index_exact(ext, oracle, b, n, m, r, ns, dist):
    xyz = cloud(b, n, seed=n + ns, dist=dist)
    if m == n:
        new_xyz = xyz.copy()
    else:
        rng = np.random.default_rng(m)
        new_xyz = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m] if m <= n else rng.integers(0, n, m)])
        new_xyz = new_xyz + rng.normal(scale=0.05, size=new_xyz.shape).astype(np.float32)  # vote-like centres
    # fill = -9 proves rows without neighbours stay untouched (ball_query_gpu.cu:34)
    idx_o, idx_d = bq_both(ext, oracle, new_xyz, xyz, r, ns, fill=-9)
    assert np.array_equal(idx_o, idx_d)


def test_ball_query_semantics_on_gpu(ext, oracle):
    xyz = np.zeros((1, 10, 3), np.float32)
    xyz[0, :, 0] = np.arange(10)
    ctr = np.array([[[100, 0, 0], [4, 0, 0], [4, 0, 0]]], np.float32)
    _, idx = bq_both(ext, oracle, ctr, xyz, 1.5, 5, fill=-9)
    assert idx[0, 0].tolist() == [-9] * 5
    assert idx[0, 1].tolist() == [3, 4, 5, 3, 3]
    _, idx = bq_both(ext, oracle, ctr, xyz, 1.0, 3)
    assert idx[0, 1].tolist() == [4, 4, 4]          # strict '<'
    _, idx = bq_both(ext, oracle, ctr, xyz, 100.0, 4)
    assert idx[0, 1].tolist() == [0, 1, 2, 3]


def test_ball_query_boundary_distances(ext, oracle):
    # lattice points + r^2 exactly representable: d2 == r^2 occurs many times and must be excluded
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 12, size=(2, 3000, 3)).astype(np.float32)
    new_xyz = rng.integers(0, 12, size=(2, 500, 3)).astype(np.float32)
    for r, ns in [(3.0, 16), (5.0, 32), (1.0, 4)]:
        idx_o, idx_d = bq_both(ext, oracle, new_xyz, xyz, r, ns, fill=-1)
        assert np.array_equal(idx_o, idx_d)


def test_ball_query_dilated(ext, oracle):
    xyz = cloud(2, 3000, seed=17)
    new_xyz = np.ascontiguousarray(xyz[:, :400])   # centres ARE points: d2 == 0 double append
    for rmax, rmin, ns in [(1.6, 0.0, 16), (1.6, 0.8, 16), (4.8, 1.6, 32), (0.5, 0.0, 1)]:
        b, m, n = 2, 400, 3000
        idx_o = np.full((b, m, ns), -2, np.int32)
        oracle.ball_query_dilated_wrapper(b, n, m, rmax, rmin, ns, new_xyz, xyz, idx_o)
        idx_d = torch.full((b, m, ns), -2, dtype=torch.int32, device="cuda")
        ext.ball_query_dilated_wrapper(b, n, m, rmax, rmin, ns, dev(new_xyz), dev(xyz), idx_d)
        assert np.array_equal(idx_o, idx_d.cpu().numpy()), (rmax, rmin, ns)


def test_ball_query_multi_equals_separate(ext, oracle):
    xyz = cloud(2, 4096, seed=23)
    new_xyz = np.ascontiguousarray(xyz[:, :1000])
    radii, nss = [4.8, 8.4, 12.8], [16, 32, 64]
    idxs = [torch.zeros((2, 1000, ns), dtype=torch.int32, device="cuda") for ns in nss]
    ext.ball_query_multi(2, 4096, 1000, radii, nss, dev(new_xyz), dev(xyz), idxs)
    for r, ns, got in zip(radii, nss, idxs):
        exp = np.zeros((2, 1000, ns), np.int32)
        oracle.ball_query_wrapper(2, 4096, 1000, r, ns, new_xyz, xyz, exp)
        assert np.array_equal(exp, got.cpu().numpy())
    # unsorted radii
    radii, nss = [1.6, 0.8], [32, 16]
    idxs = [torch.zeros((2, 1000, ns), dtype=torch.int32, device="cuda") for ns in nss]
    ext.ball_query_multi(2, 4096, 1000, radii, nss, dev(new_xyz), dev(xyz), idxs)
    for r, ns, got in zip(radii, nss, idxs):
        exp = np.zeros((2, 1000, ns), np.int32)
        oracle.ball_query_wrapper(2, 4096, 1000, r, ns, new_xyz, xyz, exp)
        assert np.array_equal(exp, got.cpu().numpy())


# ---------------------------------------------------------------- gather / group
@pytest.mark.parametrize("b,c,n,m,ns", [(2, 3, 16384, 4096, 32), (2, 67, 4096, 512, 16),
                                        (1, 259, 2048, 256, 64), (3, 5, 100, 7, 3), (1, 1, 5, 1, 1)])
def test_group_points_and_grad(ext, oracle, b, c, n, m, ns):
    rng = np.random.default_rng(b * c + n)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    exp = np.zeros((b, c, m, ns), np.float32)
    oracle.group_points_wrapper(b, c, n, m, ns, pts, idx, exp)
    out = torch.empty((b, c, m, ns), device="cuda")
    assert ext.group_points_wrapper(b, c, n, m, ns, dev(pts), dev(idx), out) == 1
    assert np.array_equal(exp, out.cpu().numpy())          # pure copy: bit exact
    go = rng.normal(size=(b, c, m, ns)).astype(np.float32)
    gexp = np.zeros((b, c, n), np.float32)
    oracle.group_points_grad_wrapper(b, c, n, m, ns, go, idx, gexp)
    gp = torch.zeros((b, c, n), device="cuda")
    assert ext.group_points_grad_wrapper(b, c, n, m, ns, dev(go), dev(idx), gp) == 1
    # float atomics: order differs from the oracle's sequential sum (as in the reference)
    np.testing.assert_allclose(gp.cpu().numpy(), gexp, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("b,c,n,m", [(2, 3, 16384, 4096), (2, 128, 4096, 2048), (3, 5, 100, 7), (1, 1, 3, 1)])
def test_gather_points_and_grad(ext, oracle, b, c, n, m):
    rng = np.random.default_rng(b * c + n)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    exp = np.zeros((b, c, m), np.float32)
    oracle.gather_points_wrapper(b, c, n, m, pts, idx, exp)
    out = torch.empty((b, c, m), device="cuda")
    assert ext.gather_points_wrapper(b, c, n, m, dev(pts), dev(idx), out) == 1
    assert np.array_equal(exp, out.cpu().numpy())
    go = rng.normal(size=(b, c, m)).astype(np.float32)
    gexp = np.zeros((b, c, n), np.float32)
    oracle.gather_points_grad_wrapper(b, c, n, m, go, idx, gexp)
    gp = torch.zeros((b, c, n), device="cuda")
    assert ext.gather_points_grad_wrapper(b, c, n, m, dev(go), dev(idx), gp) == 1
    np.testing.assert_allclose(gp.cpu().numpy(), gexp, rtol=1e-4, atol=1e-4)


def test_group_misaligned_views(ext, oracle):
    # storage offsets that break 16-byte alignment take the scalar path
    rng = np.random.default_rng(1)
    b, c, n, m, ns = 1, 4, 50, 9, 4
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    exp = np.zeros((b, c, m, ns), np.float32)
    oracle.group_points_wrapper(b, c, n, m, ns, pts, idx, exp)
    big_idx = torch.zeros(b * m * ns + 1, dtype=torch.int32, device="cuda")
    big_idx[1:] = dev(idx).flatten()
    big_out = torch.zeros(b * c * m * ns + 3, device="cuda")
    idx_v = big_idx[1:].view(b, m, ns)
    out_v = big_out[3:].view(b, c, m, ns)
    assert idx_v.is_contiguous() and out_v.is_contiguous()
    ext.group_points_wrapper(b, c, n, m, ns, dev(pts), idx_v, out_v)
    assert np.array_equal(exp, out_v.cpu().numpy())


# ---------------------------------------------------------------- three_nn / interpolate
@pytest.mark.parametrize("b,n,m", [(2, 16384, 4096), (2, 1000, 300), (1, 70, 2), (1, 5, 1), (1, 64, 3)])
def test_three_nn_exact(ext, oracle, b, n, m):
    unknown = cloud(b, n, seed=n)
    known = cloud(b, max(m, 8), seed=m + 1)[:, :m].copy()
    d_o = np.zeros((b, n, 3), np.float32); i_o = np.zeros((b, n, 3), np.int32)
    oracle.three_nn_wrapper(b, n, m, unknown, known, d_o, i_o)
    d_d = torch.zeros((b, n, 3), device="cuda"); i_d = torch.zeros((b, n, 3), dtype=torch.int32, device="cuda")
    ext.three_nn_wrapper(b, n, m, dev(unknown), dev(known), d_d, i_d)
    assert np.array_equal(i_o, i_d.cpu().numpy())
    assert np.array_equal(d_o, d_d.cpu().numpy())     # includes +inf for m < 3


def test_three_nn_ties(ext, oracle):
    rng = np.random.default_rng(2)
    unknown = rng.integers(0, 5, size=(2, 500, 3)).astype(np.float32)
    known = rng.integers(0, 5, size=(2, 333, 3)).astype(np.float32)
    d_o = np.zeros((2, 500, 3), np.float32); i_o = np.zeros((2, 500, 3), np.int32)
    oracle.three_nn_wrapper(2, 500, 333, unknown, known, d_o, i_o)
    d_d = torch.zeros((2, 500, 3), device="cuda"); i_d = torch.zeros((2, 500, 3), dtype=torch.int32, device="cuda")
    ext.three_nn_wrapper(2, 500, 333, dev(unknown), dev(known), d_d, i_d)
    assert np.array_equal(i_o, i_d.cpu().numpy()) and np.array_equal(d_o, d_d.cpu().numpy())


@pytest.mark.parametrize("b,c,m,n", [(2, 128, 4096, 16384), (1, 7, 50, 33), (2, 1, 3, 1)])
def test_three_interpolate_and_grad(ext, oracle, b, c, m, n):
    rng = np.random.default_rng(c + n)
    pts = rng.normal(size=(b, c, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, size=(b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    exp = np.zeros((b, c, n), np.float32)
    oracle.three_interpolate_wrapper(b, c, m, n, pts, idx, w, exp)
    out = torch.empty((b, c, n), device="cuda")
    ext.three_interpolate_wrapper(b, c, m, n, dev(pts), dev(idx), dev(w), out)
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=0, atol=1e-4)   # north_star tolerance
    go = rng.normal(size=(b, c, n)).astype(np.float32)
    gexp = np.zeros((b, c, m), np.float32)
    oracle.three_interpolate_grad_wrapper(b, c, n, m, go, idx, w, gexp)
    gp = torch.zeros((b, c, m), device="cuda")
    ext.three_interpolate_grad_wrapper(b, c, n, m, dev(go), dev(idx), dev(w), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), gexp, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- error behaviour / streams
def test_wrappers_reject_bad_input(ext):
    x = torch.zeros((1, 8, 3), device="cuda")
    idx = torch.zeros((1, 8, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        ext.ball_query_wrapper(1, 8, 8, 1.0, 4, x.cpu(), x, idx)            # CPU tensor
    with pytest.raises(RuntimeError):
        ext.ball_query_wrapper(1, 8, 8, 1.0, 4, x.transpose(1, 2), x, idx)  # non-contiguous
    with pytest.raises(RuntimeError):
        ext.ball_query_wrapper(1, 8, 8, 1.0, 4, x, x, idx.float())          # wrong dtype
    with pytest.raises(RuntimeError):
        ext.ball_query_wrapper(1, 8, 8, 1.0, 0, x, x, idx)                  # nsample < 1 -> status 1


def test_side_stream_and_graph_capture(ext, oracle):
    xyz = cloud(2, 2048, seed=4)
    idx_o = np.zeros((2, 2048, 16), np.int32)
    oracle.ball_query_wrapper(2, 2048, 2048, 1.0, 16, xyz, xyz, idx_o)
    xyz_d = dev(xyz)
    idx_d = torch.zeros((2, 2048, 16), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ext.ball_query_wrapper(2, 2048, 2048, 1.0, 16, xyz_d, xyz_d, idx_d)
    s.synchronize()
    assert np.array_equal(idx_o, idx_d.cpu().numpy())
    # hipGraph capture + replay: the C ABI neither allocates nor synchronises
    idx_g = torch.zeros_like(idx_d)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ext.ball_query_wrapper(2, 2048, 2048, 1.0, 16, xyz_d, xyz_d, idx_g)
    idx_g.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(idx_o, idx_g.cpu().numpy())


# ---------------------------------------------------------------- point-major gather (MI355X extension)
@pytest.mark.parametrize("b,c,n,m,ns", [(2, 64, 16384, 4096, 32), (2, 3, 4096, 512, 16), (1, 259, 2048, 256, 64),
                                        (3, 5, 100, 7, 3), (1, 128, 50, 9, 4)])
def test_group_rows_matches_channel_major_oracle(ext, oracle, b, c, n, m, ns):
    rng = np.random.default_rng(c + n)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)                 # channel-major reference layout
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    exp = np.zeros((b, c, m, ns), np.float32)
    oracle.group_points_wrapper(b, c, n, m, ns, pts, idx, exp)
    rows = dev(np.ascontiguousarray(pts.transpose(0, 2, 1)))            # (b, n, c)
    out = torch.empty((b, m, ns, c), device="cuda")
    assert ext.group_rows(b, n, c, m * ns, rows, dev(idx), out) == 1
    assert np.array_equal(exp.transpose(0, 2, 3, 1), out.cpu().numpy())
    go = rng.normal(size=(b, c, m, ns)).astype(np.float32)
    gexp = np.zeros((b, c, n), np.float32)
    oracle.group_points_grad_wrapper(b, c, n, m, ns, go, idx, gexp)
    gr = torch.zeros((b, n, c), device="cuda")
    ext.group_rows_grad(b, n, c, m * ns, dev(np.ascontiguousarray(go.transpose(0, 2, 3, 1))), dev(idx), gr)
    np.testing.assert_allclose(gr.cpu().numpy(), gexp.transpose(0, 2, 1), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- BASELINE config 5 sizes (65536-pt scenes)
def test_config5_sizes_fps_and_ball_query(ext, oracle):
    """Dense 65536-pt ONCE scene: FPS 65536 -> 16384 (streaming kernel, n > 24576) and the
    16384 x 65536 ball queries of layer 0, against the oracle directly plus the size-independent
    properties (distinct indices, non-increasing selection distance, ascending neighbour lists)."""
    b, n, m = 1, 65536, 16384
    xyz = cloud(b, n, seed=5)
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    sel = idx_d[0]
    assert len(set(sel.tolist())) == m                     # a point is never picked twice
    # distance of pick j to the set of earlier picks never increases (FPS invariant), checked on a prefix
    p = xyz[0][sel[:512]].astype(np.float64)
    d = ((p[:, None] - p[None]) ** 2).sum(-1)
    mind = np.array([d[j, :j].min() for j in range(1, 512)])
    assert (np.diff(mind) <= 1e-9).all()
    new_xyz = np.ascontiguousarray(xyz[:, sel])
    for r, ns in [(0.2, 16), (0.8, 32)]:
        idx_bo, idx_bd = bq_both(ext, oracle, new_xyz, xyz, r, ns, fill=0)
        assert np.array_equal(idx_bo, idx_bd)
        rows = idx_bd[0]
        first = rows[:, :1]
        # rows are ascending up to the padding, which repeats the first hit
        body_ok = (np.diff(rows, axis=1) > 0) | (rows[:, 1:] == first)
        assert body_ok.all()
        dd = ((xyz[0][rows] - new_xyz[0][:, None]) ** 2).sum(-1)
        assert (dd < r * r * (1 + 1e-5)).all()              # every listed neighbour is inside the ball


def test_shipped_once_yaml_input_size_60000(ext, oracle):
    """The ONCE yaml as shipped feeds 60 000 points (tools/cfgs/once_models/PDA-SSD.yaml:14-17): layer 0
    then runs D-FPS 60000 -> 16384 (cooperative form, K = 4 workgroups per scene with a ragged last one) and two
    16384 x 60000 ball queries (cell list).  Indices, final temp and neighbour lists against the oracle."""
    b, n, m = 2, 60000, 16384
    xyz = cloud(b, n, seed=60)
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    new_xyz = np.ascontiguousarray(np.stack([xyz[i][idx_d[i]] for i in range(b)]))
    from pdanet_amd import pointnet2_utils as pu
    got = pu.ball_query_multi([0.2, 0.8], [16, 32], dev(xyz), dev(new_xyz))     # the product's dispatch: cell list at this size
    for (r, ns), g in zip([(0.2, 16), (0.8, 32)], got):
        exp = np.zeros((b, m, ns), np.int32)
        oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, exp)
        assert np.array_equal(exp, g.cpu().numpy()), (r, ns)


# ---------------------------------------------------------------- Chamfer
@pytest.mark.parametrize("b,n,m", [(2, 1024, 1024), (2, 4096, 300), (1, 7, 1500), (3, 1, 1), (1, 513, 512)])
def test_chamfer_forward_backward(ext, oracle, b, n, m):
    xyz1 = cloud(b, max(n, 8), seed=n)[:, :n].copy()
    xyz2 = cloud(b, max(m, 8), seed=m + 3)[:, :m].copy()
    d1 = np.zeros((b, n), np.float32); d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32); i2 = np.zeros((b, m), np.int32)
    oracle.chamfer_forward(xyz1, xyz2, d1, d2, i1, i2)
    D1 = torch.zeros((b, n), device="cuda"); D2 = torch.zeros((b, m), device="cuda")
    I1 = torch.zeros((b, n), dtype=torch.int32, device="cuda"); I2 = torch.zeros((b, m), dtype=torch.int32, device="cuda")
    assert ext.chamfer_forward(dev(xyz1), dev(xyz2), D1, D2, I1, I2) == 1
    assert np.array_equal(i1, I1.cpu().numpy()) and np.array_equal(i2, I2.cpu().numpy())
    assert np.array_equal(d1, D1.cpu().numpy()) and np.array_equal(d2, D2.cpu().numpy())
    rng = np.random.default_rng(0)
    gd1 = rng.normal(size=(b, n)).astype(np.float32); gd2 = rng.normal(size=(b, m)).astype(np.float32)
    g1 = np.zeros_like(xyz1); g2 = np.zeros_like(xyz2)
    oracle.chamfer_backward(xyz1, xyz2, g1, g2, gd1, gd2, i1, i2)
    G1 = torch.zeros((b, n, 3), device="cuda"); G2 = torch.zeros((b, m, 3), device="cuda")
    assert ext.chamfer_backward(dev(xyz1), dev(xyz2), G1, G2, dev(gd1), dev(gd2), I1, I2) == 1
    np.testing.assert_allclose(G1.cpu().numpy(), g1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(G2.cpu().numpy(), g2, rtol=1e-4, atol=1e-3)


def test_chamfer_ties_and_loss_module(ext, oracle):
    rng = np.random.default_rng(9)
    xyz1 = rng.integers(0, 4, size=(2, 700, 3)).astype(np.float32)   # lattice: many equidistant targets
    xyz2 = rng.integers(0, 4, size=(2, 1300, 3)).astype(np.float32)
    d1 = np.zeros((2, 700), np.float32); d2 = np.zeros((2, 1300), np.float32)
    i1 = np.zeros((2, 700), np.int32); i2 = np.zeros((2, 1300), np.int32)
    oracle.chamfer_forward(xyz1, xyz2, d1, d2, i1, i2)
    from pdanet_amd import chamfer_distance as cdm
    a = dev(xyz1).requires_grad_(True)
    D1, D2, I1, I2 = cdm.chamfer_3DFunction.apply(a, dev(xyz2))
    assert np.array_equal(i1, I1.cpu().numpy()) and np.array_equal(i2, I2.cpu().numpy())
    loss = cdm.cd_loss_L2(a, dev(xyz2))
    loss.backward()
    assert abs(float(loss.detach()) - (d1.mean() + d2.mean())) < 1e-4 and torch.isfinite(a.grad).all()


# ---------------------------------------------------------------- randomized shapes
def test_random_shapes_indices_exact(ext, oracle):
    """Seeded sweep over odd shapes: N not a power of two (block-size selection / tie-break width),
    N < 1024, M not a multiple of 64, ns in {1..64}, clustered + lattice + uniform clouds."""
    rng = np.random.default_rng(20260101)
    for case in range(40):
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 127, 500, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000, 9999]))
        kind = case % 3
        if kind == 0:
            xyz = rng.uniform(-20, 20, size=(b, n, 3)).astype(np.float32)
        elif kind == 1:
            xyz = rng.integers(0, 5, size=(b, n, 3)).astype(np.float32)          # exact ties everywhere
        else:
            xyz = (rng.normal(size=(b, n, 3)) * np.array([8, 8, 0.3])).astype(np.float32)
        m = int(rng.integers(1, n + 1))
        idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
        assert np.array_equal(idx_o, idx_d), ("fps", case, b, n, m)
        assert np.array_equal(temp_o, temp_d), ("fps temp", case, b, n, m)
        mc = int(rng.integers(1, 200))
        new_xyz = (xyz[:, rng.integers(0, n, mc)] + rng.normal(scale=0.2, size=(b, mc, 3))).astype(np.float32)
        new_xyz = np.ascontiguousarray(new_xyz)
        ns = int(rng.choice([1, 2, 3, 7, 16, 32, 33, 64]))
        r = float(rng.choice([0.5, 1.0, 2.5, 6.0]))
        io, idd = bq_both(ext, oracle, new_xyz, xyz, r, ns, fill=-4)
        assert np.array_equal(io, idd), ("ball_query", case, b, n, mc, r, ns)
        d_o = np.zeros((b, mc, 3), np.float32); i_o = np.zeros((b, mc, 3), np.int32)
        oracle.three_nn_wrapper(b, mc, n, new_xyz, xyz, d_o, i_o)
        d_d = torch.zeros((b, mc, 3), device="cuda"); i_d = torch.zeros((b, mc, 3), dtype=torch.int32, device="cuda")
        ext.three_nn_wrapper(b, mc, n, dev(new_xyz), dev(xyz), d_d, i_d)
        assert np.array_equal(i_o, i_d.cpu().numpy()) and np.array_equal(d_o, d_d.cpu().numpy()), ("three_nn", case)


# ---------------------------------------------------------------- ellipsoid_query (pointnet2_api.cpp:16)
@pytest.mark.parametrize("b,n,m,e,ns,dist", [
    (2, 2048, 256, (3.0, 0.8, 1.5), 16, "L"),
    (2, 1024, 128, (1.5, 0.8, 2.0), 16, "L"),         # the ellipsoid lies inside the first ball: rows stay the ball query's
    (1, 4096, 512, (2.0, 1.0, 1.0), 32, "L"),
    (2, 700, 90, (6.0, 3.0, 4.0), 64, "U"),          # wide balls: many centres are full after the first query
    (1, 300, 40, (0.5, 0.5, 0.5), 8, "U"),           # small balls: most centres have < 3 hits and keep their row
])
def test_ellipsoid_query_rows_equal_oracle(ext, oracle, b, n, m, e, ns, dist):
    """csrc/ellipsoid_query.hip against the CPU statement of ellipsoid_query_gpu.cu:311-498 (same float expressions on
    both sides): rows bit-exact.  A hit at the exact origin (the reference's `flag`) is planted in scene 0."""
    xyz = cloud(b, n, seed=31 + ns, dist=dist)
    rng = np.random.default_rng(ns)
    new_xyz = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    xyz[0, 5] = 0.0
    new_xyz[0, 0] = (0.1, 0.0, 0.05)                 # a centre next to the origin point
    exp = oracle.ellipsoid_query(new_xyz, xyz, e[0], e[1], e[2], ns)
    got = ext.ellipsoid_query(dev(new_xyz), dev(xyz), e[0], e[1], e[2], ns)
    torch.cuda.synchronize()
    assert np.array_equal(exp, got.cpu().numpy())
    # the second pass did something: some row differs from the plain ball query of radius e3
    plain = np.zeros((b, m, ns), np.int32)
    oracle.ball_query_wrapper(b, n, m, e[2], ns, new_xyz, xyz, plain)
    if max(e[0], e[1]) > e[2] and ns >= 16:
        assert (plain != exp).any()
    if max(e[0], e[1]) <= e[2]:
        assert np.array_equal(plain, exp)

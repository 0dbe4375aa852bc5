"""-m gpu: the cell-list ball query (csrc/ball_query_cells.hip, pda_ball_query_cells) against the CPU oracle of the
reference kernel (ball_query_gpu.cu:9-45).  Bar: bit-exact rows, including rows that must stay untouched, d2 == r^2
lattices (strict '<'), dense balls (more hits than the in-LDS list holds: ascending-scan fallback), centres outside the
points' bounding box, several radii in one pass, ragged sizes."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_hip_parity import cloud, dev, ext  # noqa: E402,F401


def cells_vs_oracle(ext, oracle, new_xyz, xyz, radii, nss, fill=-9):
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idxs = [torch.full((b, m, ns), fill, dtype=torch.int32, device="cuda") for ns in nss]
    scratch = torch.empty((ext.ball_query_cells_scratch_bytes(b, n),), dtype=torch.uint8, device="cuda")
    assert ext.ball_query_cells(b, n, m, radii, nss, dev(new_xyz), dev(xyz), idxs, scratch) == 1
    torch.cuda.synchronize()
    for r, ns, got in zip(radii, nss, idxs):
        exp = np.full((b, m, ns), fill, np.int32)
        oracle.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, exp)
        assert np.array_equal(exp, got.cpu().numpy()), (r, ns)


@pytest.mark.parametrize("b,n,m,radii,nss,dist", [
    (2, 16384, 16384, [0.2, 0.8], [16, 32], "L"),      # ONCE layer 0 (dense near range: the list overflows there)
    (2, 16384, 4096, [0.8, 1.6], [16, 32], "L"),       # ONCE layer 1
    (2, 16384, 4096, [0.8, 1.6], [16, 32], "U"),
    (1, 65536, 16384, [0.2, 0.8], [16, 32], "L"),      # BASELINE config 5, layer 0
    (2, 4096, 2048, [1.6, 4.8], [16, 32], "L"),
    (2, 2048, 1024, [4.8, 8.4, 12.8], [16, 32, 64], "L"),   # three radii, cells wider than the scene in z
    (3, 1000, 77, [2.0], [5], "L"), (1, 9, 1, [100.0], [1], "U"), (1, 7, 130, [3.0], [3], "L"), (2, 513, 64, [5.0], [128], "L"),
])
def test_cells_rows_equal_reference_rows(ext, oracle, b, n, m, radii, nss, dist):
    xyz = cloud(b, n, seed=n + nss[0], dist=dist)
    if m == n:
        new_xyz = xyz.copy()
    else:
        rng = np.random.default_rng(m)
        new_xyz = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m] if m <= n else rng.integers(0, n, m)])
        new_xyz = new_xyz + rng.normal(scale=0.05, size=new_xyz.shape).astype(np.float32)   # vote-like centres
    cells_vs_oracle(ext, oracle, new_xyz, xyz, radii, nss)


def test_cells_boundary_distances_and_untouched_rows(ext, oracle):
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 12, size=(2, 3000, 3)).astype(np.float32)        # lattice: d2 == r^2 many times
    new_xyz = rng.integers(-3, 15, size=(2, 500, 3)).astype(np.float32)    # some centres outside the bounding box
    new_xyz[:, :5] = 1000.0                                                # far away: rows stay untouched
    cells_vs_oracle(ext, oracle, new_xyz, xyz, [3.0, 5.0, 1.0], [16, 32, 4], fill=-1)


def test_cells_dense_ball_falls_back_to_the_ascending_scan(ext, oracle):
    """3000 points inside one small ball (more than the 512-entry list): the first nsample indices must still come out."""
    rng = np.random.default_rng(9)
    xyz = rng.uniform(-40, 40, size=(1, 9000, 3)).astype(np.float32)
    blob = rng.permutation(9000)[:3000]
    xyz[0, blob] = (rng.normal(scale=0.05, size=(3000, 3)) + np.array([3.0, -2.0, 1.0])).astype(np.float32)
    new_xyz = np.concatenate([xyz[:, blob[:40]], xyz[:, :200]], axis=1).copy()
    cells_vs_oracle(ext, oracle, new_xyz, xyz, [0.5, 2.0], [32, 64])


def test_cells_duplicate_points_and_identical_scene(ext, oracle):
    xyz = np.zeros((1, 700, 3), np.float32)            # every point identical: one cell, every centre sees all of them
    new_xyz = np.zeros((1, 3, 3), np.float32)
    cells_vs_oracle(ext, oracle, new_xyz, xyz, [0.1], [16])
    xyz = np.repeat(cloud(1, 300, seed=4), 3, axis=1).copy()     # exact duplicates (sample_points pads with repeats)
    cells_vs_oracle(ext, oracle, xyz[:, :100].copy(), xyz, [1.0, 3.0], [8, 32])


def test_operator_uses_cells_for_large_clouds_and_matches_brute_force(ext, oracle):
    from pdanet_amd import pointnet2_utils as pu
    xyz = dev(cloud(2, 16384, seed=11)); new_xyz = xyz[:, :4096].contiguous()
    a = pu.ball_query_multi([0.8, 1.6], [16, 32], xyz, new_xyz)
    pu.BALL_QUERY_CELLS = False
    try:
        b = pu.ball_query_multi([0.8, 1.6], [16, 32], xyz, new_xyz)
    finally:
        pu.BALL_QUERY_CELLS = True
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("bad", [np.inf, -np.inf])
def test_cells_scene_with_a_non_finite_point_terminates_and_matches(ext, oracle, bad):
    """A scene holding one +/-inf coordinate has no finite bounding box: the grid sizing must not spin (it did: one
    thread looped for ever at N = 16384) and the rows must still be the reference's, in which such a point is never a
    hit (inf < r^2 is false).  Only scene 1 is affected; scene 0 keeps its normal grid."""
    xyz = cloud(2, 16384, seed=77, dist="L")
    xyz[1, 1234, 0] = bad
    xyz[1, 9, 2] = bad
    rng = np.random.default_rng(3)
    new_xyz = np.ascontiguousarray(xyz[:, rng.permutation(16384)[:512]])
    new_xyz[1, new_xyz[1, :, 0] == bad] = 0.0
    new_xyz[~np.isfinite(new_xyz)] = 0.0
    cells_vs_oracle(ext, oracle, new_xyz, xyz, [0.8, 1.6], [16, 32])

"""pointnet2_stack operator set: oracle known answers derived by hand from
pointnet2_stack/src/*.cu (CPU) and HIP == oracle (GPU: indices bit-exact, features 1e-6)."""
import numpy as np
import pytest

I32, F32 = np.int32, np.float32


def test_oracle_stack_known_answers(oracle):
    # two scenes: 4 points on the x axis, then 3 points; centres: 2 in scene 0, 1 in scene 1
    xyz = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0], [11, 0, 0], [12, 0, 0]], F32)
    xyz_cnt = np.array([4, 3], I32)
    new_xyz = np.array([[0.9, 0, 0], [50, 0, 0], [11.2, 0, 0]], F32)
    new_cnt = np.array([2, 1], I32)
    idx = np.zeros((3, 3), I32)
    oracle.stack_ball_query_wrapper(2, 3, 1.0, 3, new_xyz, new_cnt, xyz, xyz_cnt, idx)
    # centre 0: points 0 (d .9) and 1 (d .1) of scene 0 -> [0, 1, 0]; centre 1: empty -> idx[0] = -1, rest untouched;
    # centre 2 (scene 1): LOCAL indices 1 (d .2) and 2 (d .8) -> [1, 2, 1]   (local 0 at distance 1.2 is out)
    assert idx.tolist() == [[0, 1, 0], [-1, 0, 0], [1, 2, 1]]
    # grouping uses the scene start: features = global row id
    feats = np.arange(7, dtype=F32)[:, None].repeat(2, 1) + np.array([0, 100], F32)
    idx[1] = 0
    out = np.zeros((3, 2, 3), F32)
    oracle.stack_group_points_wrapper(2, 3, 2, 3, feats, xyz_cnt, idx, new_cnt, out)
    assert out[:, 0].tolist() == [[0, 1, 0], [0, 0, 0], [5, 6, 5]] and out[2, 1].tolist() == [105, 106, 105]
    g = np.zeros((7, 2), F32)
    oracle.stack_group_points_grad_wrapper(2, 3, 2, 7, 3, np.ones((3, 2, 3), F32), idx, new_cnt, xyz_cnt, g)
    assert g[:, 0].tolist() == [5, 1, 0, 0, 0, 2, 1]
    # three_nn: global indices, strict '<' keeps the lower index first, fewer than 3 known points -> 1e40 -> inf, idx = start
    known = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [10, 0, 0], [12, 0, 0]], F32)
    kcnt = np.array([3, 2], I32)
    unknown = np.array([[1, 0, 0], [11, 0, 0]], F32)
    d2, i3 = np.zeros((2, 3), F32), np.zeros((2, 3), I32)
    oracle.stack_three_nn_wrapper(unknown, np.array([1, 1], I32), known, kcnt, d2, i3)
    assert i3.tolist() == [[1, 0, 2], [3, 4, 3]] and d2[0].tolist() == [0, 1, 1] and d2[1, :2].tolist() == [1, 1] and np.isinf(d2[1, 2])
    # stack FPS: collinear x = 0..7 -> [0, 7, then 3 vs 4 tie]: block size is ALWAYS 1024 here, so lanes 3 and 4 first
    # differ at bit 0 of... the bit-reversed order over 10 bits: bitrev10(3) = 768 > bitrev10(4) = 128 -> 4 wins
    pts = np.stack([np.arange(8, dtype=F32), np.zeros(8, F32), np.zeros(8, F32)], 1)
    both = np.concatenate([pts, pts + np.array([100, 0, 0], F32)])
    temp = np.full(16, 1e10, F32)
    out_idx = np.zeros(5, I32)
    oracle.stack_farthest_point_sampling_wrapper(both, temp, np.array([8, 8], I32), out_idx, np.array([3, 2], I32))
    assert out_idx.tolist() == [0, 7, 4, 8, 15]


def _scenes(rng, b, lo, hi):
    cnt = rng.integers(lo, hi, b).astype(I32)
    return cnt, rng.uniform(-3, 3, (int(cnt.sum()), 3)).astype(F32)


@pytest.mark.gpu
@pytest.mark.parametrize("b,seed", [(1, 0), (3, 1), (5, 2)])
def test_hip_stack_ops_match_oracle(oracle, b, seed):
    import torch
    from pdanet_amd import pointnet2_stack_utils as su
    rng = np.random.default_rng(seed)
    cnt, xyz = _scenes(rng, b, 200, 1500)
    if b >= 3:
        cnt[1] = 0                                               # an empty scene in the middle
        xyz = xyz[: int(cnt.sum())]
    ncnt = np.maximum(cnt // 3, 0).astype(I32)
    new_xyz = np.concatenate([xyz[int(cnt[:k].sum()): int(cnt[:k].sum()) + int(ncnt[k])] for k in range(b)] + [np.zeros((0, 3), F32)]) \
        + rng.normal(0, 0.05, (int(ncnt.sum()), 3)).astype(F32)
    new_xyz[::7] += 40                                           # some empty balls
    M, N, C, ns = new_xyz.shape[0], xyz.shape[0], 9, 8
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    # ball query + mask
    want = np.zeros((M, ns), I32)
    oracle.stack_ball_query_wrapper(b, M, 0.6, ns, new_xyz, ncnt, xyz, cnt, want)
    idx, empty = su.ball_query(0.6, ns, t(xyz), t(cnt), t(new_xyz), t(ncnt))
    assert np.array_equal(empty.cpu().numpy(), want[:, 0] == -1) and empty.any() and not empty.all()
    want[want[:, 0] == -1] = 0
    assert np.array_equal(idx.cpu().numpy(), want)
    # grouping fwd / grad
    feats = rng.normal(size=(N, C)).astype(F32)
    wout = np.zeros((M, C, ns), F32)
    oracle.stack_group_points_wrapper(b, M, C, ns, feats, cnt, want, ncnt, wout)
    tf = t(feats).requires_grad_(True)
    out = su.grouping_operation(tf, t(cnt), idx, t(ncnt))
    assert np.array_equal(out.detach().cpu().numpy(), wout)
    go = rng.normal(size=(M, C, ns)).astype(F32)
    out.backward(t(go))
    wg = np.zeros((N, C), F32)
    oracle.stack_group_points_grad_wrapper(b, M, C, N, ns, go, want, ncnt, cnt, wg)
    np.testing.assert_allclose(tf.grad.cpu().numpy(), wg, rtol=1e-5, atol=1e-5)
    # QueryAndGroup composition
    nf, _ = su.QueryAndGroup(0.6, ns)(t(xyz), t(cnt), t(new_xyz), t(ncnt), t(feats))
    assert tuple(nf.shape) == (M, 3 + C, ns) and float(nf[empty].abs().sum()) == 0
    # three_nn / interpolate
    d2, i3 = np.zeros((M, 3), F32), np.zeros((M, 3), I32)
    oracle.stack_three_nn_wrapper(new_xyz, ncnt, xyz, cnt, d2, i3)
    dist, gi = su.three_nn(t(new_xyz), t(ncnt), t(xyz), t(cnt))
    assert np.array_equal(gi.cpu().numpy(), i3)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(d2))
    w = rng.random((M, 3)).astype(F32)
    wi = np.zeros((M, C), F32)
    oracle.stack_three_interpolate_wrapper(feats, i3, w, wi)
    tf2 = t(feats).requires_grad_(True)
    io = su.three_interpolate(tf2, gi, t(w))
    np.testing.assert_allclose(io.detach().cpu().numpy(), wi, rtol=1e-6, atol=1e-6)
    gio = rng.normal(size=(M, C)).astype(F32)
    io.backward(t(gio))
    wgi = np.zeros((N, C), F32)
    oracle.stack_three_interpolate_grad_wrapper(gio, i3, w, wgi)
    np.testing.assert_allclose(tf2.grad.cpu().numpy(), wgi, rtol=1e-5, atol=1e-5)
    # stack FPS (scenes with n < m are invalid input for the reference as well: keep m <= n, allow m = 0)
    nsamp = np.minimum(cnt, rng.integers(0, 300, b)).astype(I32)
    wfi = np.zeros(int(nsamp.sum()), I32)
    oracle.stack_farthest_point_sampling_wrapper(xyz, np.full(N, 1e10, F32), cnt, wfi, nsamp)
    fi = su.stack_farthest_point_sample(t(xyz), t(cnt), t(nsamp))
    assert np.array_equal(fi.cpu().numpy(), wfi)
    fi2 = su.stack_farthest_point_sample(t(xyz), t(cnt), [int(v) for v in nsamp])
    assert np.array_equal(fi2.cpu().numpy(), wfi)


@pytest.mark.gpu
def test_batch_layout_fps_of_the_stack_module(oracle):
    import torch
    from pdanet_amd import pointnet2_stack_utils as su
    rng = np.random.default_rng(4)
    xyz = rng.uniform(-5, 5, (2, 700, 3)).astype(F32)
    want = np.zeros((2, 64), I32)
    oracle.farthest_point_sampling_wrapper(2, 700, 64, xyz, np.full((2, 700), 1e10, F32), want)
    assert np.array_equal(su.farthest_point_sample(torch.from_numpy(xyz).cuda(), 64).cpu().numpy(), want)

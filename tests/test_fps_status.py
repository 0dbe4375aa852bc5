"""-m gpu: the multi-workgroup FPS form (16384 < n <= 65536, csrc/fps.hip fps_chain_coop_kernel<16>) can no longer
fail silently: a timed-out winner exchange is counted, reported through the C ABI
(pda_fps_coop_timeouts) and the scene is recomputed on the device, so indices and `temp` stay the exact result."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_hip_parity import cloud, ext, fps_both  # noqa: E402,F401


@pytest.mark.parametrize("b,n,m", [(2, 30000, 200), (1, 50000, 150)])    # K = 2 and K = 4 workgroups per scene
def test_cooperative_fps_reports_no_timeouts_when_healthy(ext, oracle, b, n, m):
    ext.fps_coop_timeouts(reset=True)
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, cloud(b, n, seed=n), m)
    assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    assert ext.fps_coop_timeouts() == 0


def test_forced_exchange_timeout_is_counted_and_recovered(ext, oracle):
    """Poll budget 0: a workgroup gives up at the first exchange whose partner records are not there yet.  The status
    reaches the host (non-zero count) and the follow-up kernel recomputes the scene: still bit-exact."""
    b, n, m = 3, 30000, 120
    xyz = cloud(b, n, seed=77)
    ext.fps_coop_timeouts(reset=True)
    ext.debug_fps_spin_limit(0)
    try:
        idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
        failures = ext.fps_coop_timeouts(reset=True)
    finally:
        ext.debug_fps_spin_limit(-1)
    assert failures > 0, "the forced timeout was not reported"
    assert (idx_d >= 0).all() and (idx_d < n).all()          # nothing unwritten / out of range reaches a gather
    assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    # and the next healthy launch is clean again
    idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
    assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    assert ext.fps_coop_timeouts() == 0


def test_cooperative_fps_on_the_side_stream_gives_the_same_backbone_output(monkeypatch):
    """backbone._presample runs the multi-workgroup FPS (65536-point scenes, config 5) on the sampling side stream, next to
    the main stream's kernels: same centres and features as in program order, and no exchange timed out."""
    from pdanet_amd import backbone as bbm, pointnet2_batch_cuda as ext, synth
    torch.manual_seed(3)
    model = bbm.build_backbone("once_pda_ssd.yaml")[0].cuda().eval()
    pts = torch.from_numpy(synth.batch_points(2, 65536, config_id=5, dist="L", dataset="once")).cuda()
    torch.cuda.synchronize()
    outs = {}
    ext.fps_coop_timeouts(reset=True)
    with torch.no_grad():
        for flag in ("0", "1"):
            monkeypatch.setenv("PDA_COOP_FPS_SIDE_STREAM", flag)
            for _ in range(2):      # the second call runs with a warm allocator / streams
                bd = model({'batch_size': 2, 'points': pts, 'inputs_resident': True})     # pts IS resident (no copy in flight)
            outs[flag] = (bd['centers'].clone(), bd['centers_features'].clone(), bd['encoder_xyz'][1].clone())
    assert ext.fps_coop_timeouts() == 0
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)


def test_exchange_granules_are_wiped_behind_every_launch(ext, oracle):
    """The tag of an exchange granule is (15-bit launch epoch, round): a launch 32767 epochs later would read a leftover
    granule of the same scene, region and round as its own.  fps_recover_kernel, which runs behind every cooperative
    launch, wipes the granules of its scene; so the state after a launch is the state before the first one, and repeated
    launches (here more than the four regions the epochs rotate through) keep giving the exact result."""
    b, n, m = 2, 30000, 300
    xyz = cloud(b, n, seed=91)
    ext.fps_coop_timeouts(reset=True)
    for _ in range(9):
        idx_o, temp_o, idx_d, temp_d = fps_both(ext, oracle, xyz, m)
        assert np.array_equal(idx_o, idx_d) and np.array_equal(temp_o, temp_d)
    assert ext.fps_coop_timeouts() == 0
    from pdanet_amd import _lib
    assert _lib.load().pda_debug_fps_exchange_nonzero() == 0

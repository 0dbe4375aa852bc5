#!/usr/bin/env python
"""BASELINE configs[0]: one synthetic 4096-pt scene through FPS + ball_query on the CPU restatement.
Writes tests/golden/config1.npz (inputs are regenerated from the seed; outputs are stored).  The CPU
test re-runs the oracle against it (guards the oracle), the GPU test runs the HIP kernels against it
(needs no oracle at all on the GPU box)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from pdanet_amd import synth  # noqa: E402


def run():
    xyz = synth.batch_xyz(1, 4096, config_id=1, dist="L")
    temp = np.full((1, 4096), 1e10, np.float32)
    fps = np.zeros((1, 1024), np.int32)
    oracle.farthest_point_sampling_wrapper(1, 4096, 1024, xyz, temp, fps)
    new_xyz = np.ascontiguousarray(xyz[:, fps[0]])
    out = {"fps_idx": fps, "fps_temp": temp}
    for r, ns in [(0.8, 16), (1.6, 32)]:
        idx = np.zeros((1, 1024, ns), np.int32)
        oracle.ball_query_wrapper(1, 4096, 1024, r, ns, new_xyz, xyz, idx)
        out["bq_%g_%d" % (r, ns)] = idx
    return xyz, out


if __name__ == "__main__":
    _, out = run()
    np.savez_compressed(os.path.join(HERE, "config1.npz"), **out)
    print({k: v.shape for k, v in out.items()})

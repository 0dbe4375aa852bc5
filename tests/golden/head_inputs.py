"""Seeded synthetic backbone outputs for the head fixtures: shared by make_head_golden.py (which
feeds them to the REFERENCE head) and tests/test_iassd_head.py (which feeds them to this repo's)."""
import numpy as np


def synth_inputs(num_class, seed, B=2, T=7, sizes=(2048, 2048, 512, 256, 128), C=512):
    """Backbone outputs of the right structure (IASSD_backbone.py:188-203) around random GT boxes."""
    g = np.random.default_rng(seed)
    gt = np.zeros((B, T, 8), np.float32)
    for b in range(B):
        n_real = T - 2 - b                                         # zero-padded rows at the end, as collate_batch makes
        gt[b, :n_real, 0:2] = g.uniform(-30, 30, (n_real, 2))
        gt[b, :n_real, 2] = g.uniform(-1, 0.5, n_real)
        gt[b, :n_real, 3:6] = g.uniform([1.5, 0.6, 1.0], [6.0, 2.5, 2.5], (n_real, 3))
        gt[b, :n_real, 6] = g.uniform(-3.5, 3.5, n_real)
        gt[b, :n_real, 7] = g.integers(1, num_class + 1, n_real)

    def cloud(n):
        pts = np.zeros((B, n, 4), np.float32)
        for b in range(B):
            n_real = T - 2 - b
            k = g.integers(0, n_real, n)
            near = g.random(n) < 0.55                              # about half of the points around a box
            local = g.uniform(-0.75, 0.75, (n, 3)) * gt[b, k, 3:6]
            c, s = np.cos(gt[b, k, 6]), np.sin(gt[b, k, 6])
            rot = np.stack([local[:, 0] * c - local[:, 1] * s, local[:, 0] * s + local[:, 1] * c, local[:, 2]], -1)
            far = np.concatenate([g.uniform(-35, 35, (n, 2)), g.uniform(-2, 1, (n, 1))], -1)
            pts[b, :, 1:4] = np.where(near[:, None], gt[b, k, 0:3] + rot, far)
            pts[b, :, 0] = b
        return pts
    coords = [cloud(n) for n in sizes]
    origin = cloud(sizes[-1])
    offsets = np.clip(g.normal(0, 0.6, (B, sizes[-1], 3)), -2, 2).astype(np.float32)
    centers = origin.copy()
    centers[..., 1:4] += offsets
    coords += [origin, centers]                                    # vote layer appends centers_origin, then centers
    sa_preds = [None, g.normal(0, 1, (B, sizes[2], num_class)).astype(np.float32),
                g.normal(0, 1, (B, sizes[3], num_class)).astype(np.float32), None, None, None]
    feats = g.normal(0, 1, (B * sizes[-1], C)).astype(np.float32)
    return dict(gt_boxes=gt, coords=coords, origin=origin, centers=centers, offsets=offsets, sa_preds=sa_preds, feats=feats)

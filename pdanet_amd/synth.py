"""Deterministic synthetic LiDAR-like scenes; there is no dataset access.

``rng = numpy.random.default_rng(1000 * config_id + scene_id)``.  Two distributions, clipped
to the dataset range (ONCE [-75.2,-75.2,-5, 75.2,75.2,3], once_dataset.yaml:5; KITTI
[0,-40,-3, 70.4,40,1], kitti_dataset.yaml:4):
  "U"  uniform in the box (worst case for ball query: most small balls are empty);
  "L"  LiDAR-like: rho = R*u^2, theta uniform, 70 % ground z = z_ground + 0.1*N(0,1),
       30 % z uniform; intensity ~ U[0,1].
Points are randomly permuted (mirrors shuffle_points, data_processor.py:93-103), fp32.
"""
import numpy as np

RANGES = {
    "once": np.array([-75.2, -75.2, -5.0, 75.2, 75.2, 3.0], np.float32),
    "kitti": np.array([0.0, -40.0, -3.0, 70.4, 40.0, 1.0], np.float32),
}


def scene(n, config_id=2, scene_id=0, dist="L", dataset="once"):
    """Returns (n, 4) float32 [x, y, z, intensity]."""
    rng = np.random.default_rng(1000 * config_id + scene_id)
    lo, hi = RANGES[dataset][:3], RANGES[dataset][3:]
    if dist == "U":
        xyz = rng.uniform(lo, hi, size=(n, 3))
    else:
        R = float(max(abs(lo[0]), abs(hi[0]), abs(lo[1]), abs(hi[1])))
        rho = R * rng.uniform(0, 1, n) ** 2
        if dataset == "kitti":
            theta = rng.uniform(-np.pi / 4, np.pi / 4, n)
        else:
            theta = rng.uniform(0, 2 * np.pi, n)
        x, y = rho * np.cos(theta), rho * np.sin(theta)
        ground = rng.uniform(0, 1, n) < 0.7
        z_ground = lo[2] + 0.25 * (hi[2] - lo[2])
        z = np.where(ground, z_ground + 0.1 * rng.normal(size=n), rng.uniform(lo[2], hi[2], n))
        xyz = np.stack([x, y, z], axis=1)
        xyz = np.clip(xyz, lo, hi)
    inten = rng.uniform(0, 1, size=(n, 1))
    pts = np.concatenate([xyz, inten], axis=1).astype(np.float32)
    return pts[rng.permutation(n)]


def batch_points(b, n, config_id=2, dist="L", dataset="once"):
    """collate_batch layout (datasets/dataset.py:173-178): (b*n, 5) [bs_idx, x, y, z, intensity]."""
    rows = []
    for s in range(b):
        p = scene(n, config_id, s, dist, dataset)
        rows.append(np.concatenate([np.full((n, 1), s, np.float32), p], axis=1))
    return np.concatenate(rows, axis=0)


def batch_xyz(b, n, config_id=2, dist="L", dataset="once"):
    return np.stack([scene(n, config_id, s, dist, dataset)[:, :3] for s in range(b)], axis=0).copy()


MEAN_SIZES = {  # BOX_CODER_CONFIG.mean_size of the two PDA-SSD yamls (once :77-83, kitti :73-77)
    "once": np.array([[4.38, 1.87, 1.59], [11.11, 2.88, 3.41], [7.52, 2.5, 2.62], [0.7, 0.66, 1.69], [2.18, 0.79, 1.43]], np.float32),
    "kitti": np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32),
}


def gt_boxes(points, b, config_id=2, dataset="once", n_boxes=20, pad_to=24):
    """(b, pad_to, 8) [x, y, z, dx, dy, dz, heading, class] zero-padded like collate_batch
    (datasets/dataset.py:179-185): `n_boxes` boxes per scene of the yaml's mean sizes (+-10 %), centred on
    random points of the scene so that they hold points, random heading and class."""
    n = points.shape[0] // b
    out = np.zeros((b, pad_to, 8), np.float32)
    for s in range(b):
        rng = np.random.default_rng(1000 * config_id + s + 500)
        pts = points[s * n:(s + 1) * n, 1:4]
        cls = rng.integers(0, len(MEAN_SIZES[dataset]), n_boxes)
        size = MEAN_SIZES[dataset][cls] * rng.uniform(0.9, 1.1, (n_boxes, 3))
        ctr = pts[rng.integers(0, n, n_boxes)].copy()
        ctr[:, 2] += 0.5 * size[:, 2] - 0.3                      # the chosen point lies near the box floor
        out[s, :n_boxes, 0:3], out[s, :n_boxes, 3:6] = ctr, size
        out[s, :n_boxes, 6] = rng.uniform(-np.pi, np.pi, n_boxes)
        out[s, :n_boxes, 7] = cls + 1
    return out

"""Chain length with DEEP records: every group of n/G Morton-consecutive points lists its top-D points (value + coordinates,
updated exactly during the walk); the rest of the group is bounded by its (D+1)-th value.  The walk is an exact FPS over
the G*D listed points, valid while the running maximum exceeds the largest bound."""
import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from pdanet_amd import synth
n, m = 16384, 4096
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
xyz = synth.batch_xyz(1, n, config_id=cfg)
temp = np.full((1, n), 1e10, np.float32); idx = np.zeros((1, m), np.int32)
oracle.farthest_point_sampling_wrapper(1, n, m, xyz, temp, idx)
p = xyz[0].astype(np.float64); ref = idx[0]
lo, hi = p.min(0), p.max(0); ext = hi - lo
bits = [0, 0, 0]; cell = ext.copy(); seq = []
for s in range(18):
    a = int(np.argmax(cell)); seq.append(a); bits[a] += 1; cell[a] *= 0.5
q = [np.clip(((p[:, a] - lo[a]) / (ext[a] * 1.0001) * (1 << bits[a])).astype(np.int64), 0, (1 << bits[a]) - 1) for a in range(3)]
rem = bits.copy(); code = np.zeros(n, np.int64)
for a in seq:
    rem[a] -= 1; code = (code << 1) | ((q[a] >> rem[a]) & 1)
order = np.argsort(code * n + np.arange(n), kind='stable')
ps = p[order]; orig = order
for G, D, CMAX in ((64, 1, 16), (64, 2, 16), (64, 2, 32), (64, 4, 32), (64, 8, 32), (256, 1, 16), (256, 2, 32), (128, 4, 32)):
    per = n // G
    t = np.minimum(np.full(n, 1e10), ((ps - p[0]) ** 2).sum(1))
    out = [0]; lengths = []
    while len(out) < m:
        tg = t.reshape(G, per)
        srt = np.argsort(-tg, axis=1, kind='stable')[:, :D + 1]          # top D+1 per group (stable: lower index wins ties)
        li = (np.arange(G)[:, None] * per + srt[:, :D]).reshape(-1)      # listed point indices (sorted order)
        lv = t[li].copy()
        B = tg[np.arange(G), srt[:, D]].max()                             # largest unlisted value
        chain = []
        while len(chain) < CMAX and len(out) + len(chain) < m:
            k = int(np.lexsort((li, -lv))[0])                             # max value, then lowest sorted index
            cv = lv[k]
            if not (cv > B): break
            s = ps[li[k]]; chain.append(int(li[k]))
            lv = np.minimum(lv, ((ps[li] - s) ** 2).sum(1))
        if not chain:     # cannot happen for D >= 1? the global max is always listed; equality with B blocks it
            gi = int(np.lexsort((np.arange(n), -t))[0]); chain = [gi]
        for gi in chain:
            t = np.minimum(t, ((ps - ps[gi]) ** 2).sum(1)); out.append(int(orig[gi]))
        lengths.append(len(chain))
    L = np.array(lengths)
    print("G=%3d D=%d cap %2d: exact=%s super-rounds %d, mean chain %.2f" % (G, D, CMAX, np.array_equal(np.array(out[:m]), ref), len(L), L.mean()), flush=True)

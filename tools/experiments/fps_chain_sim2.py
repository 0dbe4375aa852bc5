"""Chain-length potential vs record granularity: G groups of n/G Morton-consecutive points, each with (best, second value);
a group's best stays valid unless a chain sample reaches the best point itself."""
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
lane_lo = ps.reshape(n // 16, 16, 3).min(1); lane_hi = ps.reshape(n // 16, 16, 3).max(1)
for G in (16, 64, 256):
    for CMAX in (8, 16):
        per = n // G
        t = np.minimum(np.full(n, 1e10), ((ps - p[0]) ** 2).sum(1))
        out = [0]; lengths = []; wavescan = []; maxs = []
        while len(out) < m:
            tg = t.reshape(G, per)
            best = tg.max(1); arg = tg.argmax(1)
            sec = np.sort(tg, axis=1)[:, -2]
            bp = ps[np.arange(G) * per + arg]
            invalid = np.zeros(G, bool); bound = np.zeros(G)
            chain = []; hc = np.zeros(16, int)
            while len(chain) < CMAX and len(out) + len(chain) < m:
                cand = np.where(~invalid, best, -1.0)
                gc = int(cand.argmax()); cv = cand[gc]
                if cv < 0: break
                if invalid.any() and bound[invalid].max() >= cv: break
                s = bp[gc]; chain.append(gc * per + arg[gc])
                d = ((bp - s) ** 2).sum(1)
                hit = (d < best); hit[gc] = False
                nb = np.maximum(sec, d)
                bound = np.where(hit, np.where(invalid, np.minimum(bound, nb), nb), bound)
                invalid |= hit
                invalid[gc] = True; bound[gc] = sec[gc]
                e = np.maximum(np.maximum(lane_lo - s, s - lane_hi), 0.0)
                lb = t.reshape(n // 16, 16).max(1)
                hc += ((e * e).sum(1) * 0.9999 < lb).reshape(16, 64).any(1)
            for gi in chain:
                t = np.minimum(t, ((ps - ps[gi]) ** 2).sum(1)); out.append(int(orig[gi]))
            lengths.append(len(chain)); wavescan.append((hc > 0).sum()); maxs.append(hc.max())
        L = np.array(lengths)
        print("G=%3d cap %2d: exact=%s super-rounds %d, mean chain %.2f, waves scanning %.2f, max samples/wave %.2f, hist %s" % (
            G, CMAX, np.array_equal(np.array(out[:m]), ref), len(L), L.mean(), np.mean(wavescan), np.mean(maxs), np.bincount(L, minlength=CMAX + 1)[1:].tolist()), flush=True)

"""How many 64-point slots does the conservative box test flag per sample, for different thresholds?"""
import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from pdanet_amd import synth
n, m = 16384, 4096
xyz = synth.batch_xyz(1, n, config_id=2)
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
ps = p[order]; inv = np.empty(n, int); inv[order] = np.arange(n)
for gran, name in ((64, "slot of 64"), (16, "lane cluster of 16")):
    G = n // gran
    blo = ps.reshape(G, gran, 3).min(1); bhi = ps.reshape(G, gran, 3).max(1)
    t = np.full(n, 1e10)
    cnt = {"slot/cluster max": 0, "wave max": 0, "true": 0}; wav = {"slot/cluster max": 0, "wave max": 0, "true": 0}
    for k in range(m - 1):
        s = ps[inv[ref[k]]]
        e = np.maximum(np.maximum(blo - s, s - bhi), 0.0); db = (e * e).sum(1)
        gmax = t.reshape(G, gran).max(1); wmax = np.repeat(t.reshape(16, n // 16).max(1), G // 16)
        dd = ((ps - s) ** 2).sum(1); upd = dd < t
        for nm, flag in (("slot/cluster max", db * 0.9999 < gmax), ("wave max", db * 0.9999 < wmax), ("true", upd.reshape(G, gran).any(1))):
            cnt[nm] += flag.sum(); wav[nm] += flag.reshape(16, G // 16).any(1).sum()
        t = np.minimum(t, dd)
    print(name, {k: "%.2f units, %.2f waves per sample" % (v / (m - 1), wav[k] / (m - 1)) for k, v in cnt.items()})

#!/usr/bin/env python3
"""Category view of a profiles/*_kernel_stats.csv (tools/rocpd_kernel_stats.py output): ms per step by kernel family."""
import collections, csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
steps = int(re.search(r"\((\d+) steps\)", rows[0][1]).group(1))
cat = collections.OrderedDict()
def family(n):
    if n.startswith("Cijk_"): return "library GEMM (hipBLASLt/rocBLAS)"
    m = re.search(r"pda::(\w+)", n)
    if m:
        k = m.group(1)
        for pre in ("bn_", "layer_norm", "group_attention", "wgrad", "fps_", "ball_query", "densitynet", "assemble", "ragged", "sa_mlp",
                    "group_rows", "gather", "add_max_pool", "max_pool_scatter", "adam", "grad_norm", "points_in_boxes", "assign", "nms"):
            if k.startswith(pre): return "pda::" + pre.rstrip("_") + "*"
        return "pda::" + k
    if "elementwise" in n or "Fill" in n: return "torch elementwise/fill/copy"
    if "reduce_kernel" in n: return "torch reduce"
    if "CatArray" in n: return "torch cat"
    return "torch other"
for r in rows[1:]:
    f = family(r[0]); d = cat.setdefault(f, [0, 0.0]); d[0] += int(r[1]); d[1] += float(r[2])
tot = sum(v[1] for v in cat.values())
print("%-40s %10s %10s %7s" % ("family", "launches/step", "ms/step", "%"))
for f, (n, t) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print("%-40s %10.1f %10.3f %6.1f%%" % (f, n / steps, t / steps / 1e6, 100 * t / tot))
print("%-40s %10.1f %10.3f" % ("total", sum(v[0] for v in cat.values()) / steps, tot / steps / 1e6))

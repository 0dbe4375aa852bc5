#!/usr/bin/env python3
"""Per-kernel statistics of the last N steps of a `rocprofv3 --kernel-trace` run (rocpd sqlite output).

    rocprofv3 --kernel-trace --stats -d /tmp/prof -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline
    python tools/rocpd_kernel_stats.py $(find /tmp/prof -name '*.db' | head -1) 10 > profiles/<round>_kernel_stats.csv

A step is delimited by the launches of `--marker` (default: the D-FPS kernel, launched once per step); warm-up steps,
module loading and TunableOp look-ups before the last N+1 markers are left out."""
import collections
import csv
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
    marker = sys.argv[3] if len(sys.argv) > 3 else "fps_chain"
    rows = list(sqlite3.connect(db).execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < steps + 1:
        sys.exit("only %d launches of %r in the trace" % (len(marks), marker))
    sel = rows[marks[-steps - 1]:marks[-1]]
    agg = collections.OrderedDict()
    for name, a, b in sel:
        d = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d[0] += 1; d[1] += b - a; d[2] = min(d[2], b - a); d[3] = max(d[3], b - a)
    total = sum(v[1] for v in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls(%d steps)" % steps, "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, n, t, t / n, 100.0 * t / total, lo, hi])
    sys.stderr.write("window: %d steps, kernel time %.3f ms/step, wall %.3f ms/step\n"
                     % (steps, total / steps / 1e6, (sel[-1][2] - sel[0][1]) / steps / 1e6))


if __name__ == "__main__":
    main()

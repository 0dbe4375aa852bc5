#!/usr/bin/env python3
"""Where the device idles inside a training step: from a `rocprofv3 --kernel-trace` rocpd database, the intervals of the
last N steps in which NO kernel runs (any stream), largest first, with the kernels on either side.

    python tools/step_gaps.py <trace.db> [steps] [marker] [top]"""
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
    marker = sys.argv[3] if len(sys.argv) > 3 else "fps_chain"
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    rows = list(sqlite3.connect(db).execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    sel = rows[marks[-steps - 1]:marks[-1]]
    t0, t1 = sel[0][1], max(r[2] for r in sel)
    busy_end, idle, gaps, prev = sel[0][1], 0, [], None
    for name, a, b in sel:
        if a > busy_end:
            idle += a - busy_end
            gaps.append((a - busy_end, prev, name, busy_end - t0))
        if b > busy_end:
            busy_end, prev = b, name
    wall = (t1 - t0) / steps / 1e6
    print("window: %d steps, wall %.3f ms/step, device idle (no kernel on any stream) %.3f ms/step in %d gaps/step"
          % (steps, wall, idle / steps / 1e6, len(gaps) // steps))
    hist = [0] * 6
    for g in gaps:
        us = g[0] / 1e3
        hist[0 if us < 2 else 1 if us < 5 else 2 if us < 10 else 3 if us < 20 else 4 if us < 50 else 5] += g[0]
    print("idle by gap length, ms/step: <2us %.3f | 2-5 %.3f | 5-10 %.3f | 10-20 %.3f | 20-50 %.3f | >50 %.3f"
          % tuple(h / steps / 1e6 for h in hist))

    def short(n):
        n = n.split("(")[0].replace("void ", "").replace("pda::", "")
        return n[-70:] if n.startswith("at::") else n[:70]
    agg = {}
    for g in gaps:
        k = (short(g[1]), short(g[2]))
        a = agg.setdefault(k, [0, 0])
        a[0] += g[0]; a[1] += 1
    print("largest idle by (kernel before -> kernel after), us/step:")
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print("  %8.1f us/step  %5.1f x/step  %s  ->  %s" % (t / steps / 1e3, n / steps, k[0], k[1]))


if __name__ == "__main__":
    main()

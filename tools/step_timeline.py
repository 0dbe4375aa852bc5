#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: start (us from the step's first kernel), duration, gap to the previous
kernel's end on the same queue, grid, name."""
import sqlite3, sys, re
db = sys.argv[1]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
sys.stderr.write("columns: %s\n" % cols)
want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "grid_size", "workgroup_size") if c in cols]
rows = list(con.execute("select %s from kernels order by start" % ",".join(want)))
ni = want.index("name")
marks = [i for i, r in enumerate(rows) if "fps_chain" in r[ni]]
sel = rows[marks[-3]:marks[-2]]
t0 = sel[0][want.index("start")]
last_end = {}
print("\t".join(["t_us", "dur_us", "gap_us"] + [w for w in want if w not in ("name", "start", "end")] + ["name"]))
for r in sel:
    d = dict(zip(want, r))
    q = d.get("queue_id", d.get("stream_id", 0))
    gap = (d["start"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = d["end"]
    nm = re.sub(r"\(.*", "", d["name"])[:90]
    print("\t".join(["%.1f" % ((d["start"] - t0) / 1e3), "%.1f" % ((d["end"] - d["start"]) / 1e3), "%.1f" % gap] +
                    [str(d[w]) for w in want if w not in ("name", "start", "end")] + [nm]))

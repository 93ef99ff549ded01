#!/usr/bin/env python3
"""dispatch timeline from a rocprofv3 rocpd database: the launches between the LAST two launches of <anchor> (one step) in time order, those of at least min_us microseconds
   python tools/rocpd_timeline.py results.db <anchor substring> [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute("select start, end, name, %s from kernels order by start" % q))
marks = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(marks) < 2:
    sys.exit("anchor seen %d times" % len(marks))
lo, hi = marks[-2], marks[-1]
t0 = rows[lo][0]
print("# %d launches in the step, %.3f ms from anchor to anchor" % (hi - lo, (rows[hi][0] - t0) / 1e6))
print("start_ms,dur_ms,queue,name")
for s, e, n, qq in rows[lo:hi]:
    if (e - s) / 1e3 < floor:
        continue
    print("%.3f,%.3f,%s,%s" % ((s - t0) / 1e6, (e - s) / 1e6, qq, n.split("(")[0][:60]))

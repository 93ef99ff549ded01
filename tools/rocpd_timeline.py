#!/usr/bin/env python3
"""the dispatches of one step in time order from a rocprofv3 rocpd database: start (us from the first listed), duration, kernel, stream/queue
   python tools/rocpd_timeline.py results.db <first kernel of a step, e.g. k_pack_reads> [step index from the end, default 1 = last]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_pack_reads"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute("select start, end, name, %s from kernels order by start" % q))
starts = [i for i, r in enumerate(rows) if r[2].startswith(first)]
if not starts:
    sys.exit("no dispatch of " + first)
a = starts[-back]
b = starts[-back + 1] if back > 1 else len(rows)
t0 = rows[a][0]
for s, e, name, qid in rows[a:b]:
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, name[:70]))

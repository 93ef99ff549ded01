#!/usr/bin/env python3
"""per-kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2):
   python tools/rocpd_stats.py results.db [min_total_ms] > profiles/<name>.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                       "max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,AccumVGPRs,SGPRs,LDS,Scratch")
for r in rows:
    if r[2] / 1e6 < floor:
        continue
    name = r[0].replace('"', "'")
    if len(name) > 160:   # rocPRIM's template names run to kilobytes
        name = name[:157] + "..."
    print('"%s",%d,%d,%.0f,%d,%d,%.2f,%s,%s,%s,%s,%s' % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))

#!/bin/bash
# development aid: SQ counters per kernel of tools/config4_probe.py (one step) -> gpurun_out/<tag>/sq.txt
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p -- python3 $REPO/tools/config4_probe.py --steps 1 "$@" > $OUT/p.log 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"],)
        if key not in seen: seen.add(key); calls[k] += 1
with open(out + "/sq.txt", "w") as fo:
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:80]:
        w = max(1.0, c.get("SQ_WAVES", 1))
        fo.write("%-40s calls %3d waves %9d  valu/wave %9.0f salu/wave %8.0f lds/wave %8.0f  wave_cycles/wave %10.0f wait_any/wave %10.0f active_lanes %5.1f busy_cycles %12.0f\n" % (
            k, calls[k], w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_WAVE_CYCLES", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w,
            64.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_INSTS_VALU", 1)) if False else c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_INSTS_VALU", 1)), c.get("SQ_BUSY_CYCLES", 0)))
PY
rm -rf $OUT/p
cat $OUT/sq.txt

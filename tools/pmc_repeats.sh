#!/bin/bash
# SQ counters of the configs[4] repeats leg, per kernel: bash tools/pmc_repeats.sh <tag> -> gpurun_out/<tag>/pmc_repeats.json (instructions issued, wave cycles, waits)
TAG=$1
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py --repeats --legs repeats > $OUT/p3.log 2>&1
cd $REPO
python3 tools/pmc_summary.py --out $OUT/pmc_repeats.json --commit "$2" --command "rocprofv3 --pmc SQ_* --output-format csv -- python3 bench.py --repeats --legs repeats (1 warm-up + 3 steps of 400 k pairs)" $(find $OUT/p3 -name "*counter_collection.csv")
rm -rf $OUT/p3
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_repeats.json"))["kernels"]
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:24]
print("%-28s %6s %12s %10s %8s %8s" % ("kernel", "calls", "VALU/call M", "SALU M", "LDS M", "lanes"))
for k, e in rows:
    c = e["calls"]
    v = e.get("SQ_INSTS_VALU", 0) / c / 1e6
    lanes = e.get("SQ_THREAD_CYCLES_VALU", 0) / max(e.get("SQ_ACTIVE_INST_VALU", 1), 1) / 4
    print("%-28s %6d %12.1f %10.1f %8.1f %8.1f" % (k[:28], c, v, e.get("SQ_INSTS_SALU", 0) / c / 1e6, e.get("SQ_INSTS_LDS", 0) / c / 1e6, lanes))
PY

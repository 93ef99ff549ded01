#!/bin/bash
# per-launch timeline of one step: bash tools/prof_timeline.sh <tag> <anchor kernel> [bench.py args...] -> gpurun_out/<tag>/timeline.csv
TAG=$1; shift
ANCHOR=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $REPO/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $REPO
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/rocpd_timeline.py $DB $ANCHOR 200 > $OUT/timeline.csv
python3 tools/rocpd_stats.py $DB 0.05 > $OUT/kernel_stats.csv
rm -rf $OUT/kt

#!/bin/bash
# per-launch timeline of any python tool: bash tools/prof_timeline_cmd.sh <tag> <anchor kernel> <script.py> [args...] -> gpurun_out/<tag>/timeline.csv (the launches between the last two launches of the anchor)
TAG=$1; shift
ANCHOR=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $SCRIPT "$@" > $OUT/out.log 2> $OUT/err.log
cd $REPO
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/rocpd_timeline.py $DB $ANCHOR 200 > $OUT/timeline.csv
rm -rf $OUT/kt

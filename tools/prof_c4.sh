#!/bin/bash
# development profile of tools/config4_probe.py on the GPU box: kernel-trace statistics (k_rfa's per-phase timers live on the branch prof-instrumentation)
# usage: bash tools/prof_c4.sh <tag> [config4_probe args...]   -> gpurun_out/<tag>/{kernel_stats.csv,kt.log}
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $REPO/tools/config4_probe.py "$@" > $OUT/kt.log 2>&1
cd $REPO
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 0.05 > $OUT/kernel_stats.csv
rm -rf $OUT/kt

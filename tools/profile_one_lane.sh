#!/bin/bash
# kernel-trace statistics of bench.py --lanes 1 (every kernel with the device to itself): bash tools/profile_one_lane.sh; writes gpurun_out/prof_l1/
REPO=$PWD
OUT=$REPO/gpurun_out/prof_l1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --lanes 1 > $OUT/kt.log 2>&1
cd $REPO
python3 tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) 0.05 > $OUT/kernel_stats.csv
grep -v "^W2026\|^E2026" $OUT/kt.log | tail -n 1 > $OUT/bench.json
rm -rf $OUT/kt

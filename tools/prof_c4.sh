#!/bin/bash
# development profile of tools/config4_probe.py on the GPU box: kernel-trace statistics (+ k_rfa's phases from the LH_RFA_PROF build when it exists)
# usage: bash tools/prof_c4.sh <tag> [config4_probe args...]   -> gpurun_out/<tag>/{kernel_stats.csv,kt.log,rfa_prof.log}
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
if [ -f lariat_amd/_build/liblariat_hip_prof.so ]; then
  LARIAT_HIP_LIB=$REPO/lariat_amd/_build/liblariat_hip_prof.so python3 tools/config4_probe.py "$@" > $OUT/rfa_prof.log 2>&1
fi

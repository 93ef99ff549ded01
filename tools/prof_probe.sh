#!/bin/bash
# development profile of tools/scale_probe.py on the GPU box: kernel-trace statistics + the SQ counter pass (lane utilisation, LDS, waits)
# usage: bash tools/prof_probe.sh <tag> [scale_probe args...]   -> gpurun_out/<tag>/{kernel_stats.csv,pmc.json,kt.log}
set -x
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $REPO/tools/scale_probe.py "$@" > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p3 -- python3 $REPO/tools/scale_probe.py "$@" > $OUT/p3.log 2>&1
cd $REPO
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 0.05 > $OUT/kernel_stats.csv
python3 tools/pmc_summary.py --out $OUT/pmc.json $(find $OUT/p3 -name "*counter_collection.csv")
rm -rf $OUT/kt $OUT/p3

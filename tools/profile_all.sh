#!/bin/bash
# kernel-trace statistics and the three PMC passes of bench.py on the GPU box (usage: bash tools/profile_all.sh <commit>); writes gpurun_out/prof_r06/
ulimit -c 0
set -x
REPO=$PWD
OUT=$REPO/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
cd $REPO
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 0.05 > $OUT/kernel_stats.csv
python3 tools/pmc_summary.py --out $OUT/pmc.json --commit $1 --command "rocprofv3 --pmc {FETCH_SIZE TCC_HIT_sum | WRITE_SIZE TCC_MISS_sum | SQ_*} --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras (one pipeline, 2 M-pair batches: 5 dispatches of each hot-path kernel = 1 warm-up + 4 steps; one pass per counter group)" --ceiling 50.5 --note "ceiling = independent random 32-B reads from a 3 GiB table (hg38 occurrence table); 36.8 G/s from a 48 GiB table (dense SA / ISA): profiles/r02_random_read_ceiling.log" $(find $OUT/p1 $OUT/p2 $OUT/p3 -name "*counter_collection.csv")
grep "^{" $OUT/kt.log | tail -n 1 > $OUT/kt_bench.json
rm -rf $OUT/kt $OUT/p1 $OUT/p2 $OUT/p3
ls -la $OUT
# the configs[4] legs alone (repeats: every read on the copies of repeat families; mixed: 5 % of every barcode's pairs): kernel-trace statistics of `bench.py --repeats`
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt4 -- python3 $REPO/bench.py --repeats > $OUT/kt_repeats.log 2>&1
cd $REPO
DB=$(find $OUT/kt4 -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 0.05 > $OUT/kernel_stats_repeats.csv
grep "^{" $OUT/kt_repeats.log | tail -n 1 > $OUT/kt_repeats.json
rm -rf $OUT/kt4
# one step of each configs[4] leg launch by launch, and the SQ counters of the repeats leg
bash tools/prof_timeline.sh prof_r06_tl k_pack_reads --repeats --legs repeats
cp $REPO/gpurun_out/prof_r06_tl/timeline.csv $OUT/timeline_repeats.csv
bash tools/pmc_repeats.sh prof_r06_pmc4 $1
cp $REPO/gpurun_out/prof_r06_pmc4/pmc_repeats.json $OUT/pmc_repeats.json
# K1's request stream (needs the -DLH_K1_TRACE build: bash tools/prof_rfa.sh beforehand, here, where hipcc is)
if [ -f lariat_amd/_build/liblariat_hip_prof.so ]; then python3 tools/k1_trace.py --commit $1 --kernel-stats $OUT/kernel_stats.csv > $OUT/k1_request_floor.json 2> $OUT/k1_trace.err; fi

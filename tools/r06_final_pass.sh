#!/bin/bash
# the round's closing pass on the GPU box, in one call (usage: bash tools/r06_final_pass.sh <commit>; beforehand, where hipcc is: bash tools/prof_rfa.sh):
# GPU suite, profiles (kernel statistics, PMC passes, timelines, K1 request trace), the K8 barcode-size table and phase clocks, K7's adversarial batches,
# then the default bench against the PMC file just written.  Everything lands in gpurun_out/; the summaries to keep are copied to profiles/r06_* afterwards.
ulimit -c 0
set -x
C=$1
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_gputest_final.log
cat gpurun_out/r06_gputest_final.log
python tools/valu_rate.py > gpurun_out/r06_valu_rate.log 2>&1
bash tools/profile_all.sh $C > gpurun_out/r06_profile_all.log 2>&1
tail -5 gpurun_out/r06_profile_all.log
bash tools/prof_timeline.sh prof_r06_tl_mixed k_pack_reads --repeats --legs mixed > /dev/null 2>&1
cp gpurun_out/prof_r06_tl_mixed/timeline.csv gpurun_out/prof_r06/timeline_mixed.csv
bash tools/prof_timeline.sh prof_r06_tl_sweep k_pack_reads --repeats --legs mix_sweep > /dev/null 2>&1
cp gpurun_out/prof_r06_tl_sweep/timeline.csv gpurun_out/prof_r06/timeline_mix_sweep.csv
for cfg in "4000 100" "2000 200" "1000 400"; do set -- $cfg; echo "== $1 barcodes x $2 pairs, every pair on repeat copies"; python tools/c4_stats.py --frac 1.0 --pairs-per-barcode $2 --barcodes $1 --steps 2 --dump-barcodes 8 2>&1 | grep -E "^step|k_rfa  |k_rescue|k_aln |k_extend\(rounds" ; done > gpurun_out/r06_k8_barcode_size.log 2>&1
cat gpurun_out/r06_k8_barcode_size.log
python tools/c4_stats.py --lib lariat_amd/_build/liblariat_hip_prof.so --frac 1.0 --pairs-per-barcode 400 --barcodes 1000 --steps 1 --dump-barcodes 8 2>&1 | grep -A26 "k_rfa phases" | tail -27 > gpurun_out/r06_k_rfa_phases_400pair.log
python tools/k7_deep_search.py --gpu --batches 400 --seed 5000 > gpurun_out/r06_k7_deep_gpu.log 2>&1
tail -2 gpurun_out/r06_k7_deep_gpu.log
python tools/config4_probe.py --steps 2 --oracle 30 2>&1 | grep -E "^step|HIP == oracle|counters" > gpurun_out/r06_config4_probe.log
cp gpurun_out/prof_r06/pmc.json profiles/r06_pmc_bench.json
cp gpurun_out/prof_r06/k1_request_floor.json profiles/r06_k1_request_floor.json
cp gpurun_out/r06_valu_rate.log profiles/r06_valu_rate.log
S=$(date +%s); python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

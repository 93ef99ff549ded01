ulimit -c 0
set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_gputest_final.log
cat gpurun_out/r06_gputest_final.log
python tools/valu_rate.py > gpurun_out/r06_valu_rate.log 2>&1
bash tools/profile_all.sh 79e8807 > gpurun_out/r06_profile_all.log 2>&1
tail -5 gpurun_out/r06_profile_all.log
for cfg in "4000 100" "2000 200" "1000 400"; do set -- $cfg; echo "== $1 barcodes x $2 pairs, every pair on repeat copies"; python tools/c4_stats.py --frac 1.0 --pairs-per-barcode $2 --barcodes $1 --steps 2 --dump-barcodes 8 2>&1 | grep -E "^step|k_rfa  |k_rescue|k_aln |k_extend\(rounds" ; done > gpurun_out/r06_k8_barcode_size.log 2>&1
cat gpurun_out/r06_k8_barcode_size.log
S=$(date +%s); python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

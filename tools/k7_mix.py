"""which candidates of the configs[4] input need K7's DP: by mismatches and indels of the final alignment (development probe)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from lariat_amd import capi, workload
lib = capi.load_library()
g = workload.config4_genome(lib, 3060e6)
idx = lib.index_build_device(g["pac"], g["l_pac"], g["contigs"]); idx.set_alt(g["alt_flags"])
r = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 400, n_barcodes=400, pairs_per_barcode=100)
ctx = idx.context(r["n_pairs"])
res = ctx.align_barcodes(capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"]))
mm, ind, sc = res.mismatches, res.indels, res.soft_clipped
n = len(mm)
print("candidates", n)
for lo, hi in ((0, 3), (4, 4), (5, 7), (8, 12), (13, 999)):
    sel = (ind == 0) & (mm >= lo) & (mm <= hi)
    print("  no indel, %d..%d mismatches: %.3f" % (lo, hi, sel.mean()))
print("  with indels: %.3f" % (ind > 0).mean())
print("  glob_cells per candidate %.0f" % (res.counters["glob_cells"] / n))

"""development aid: run one synthetic case through the HIP path with per-kernel syncs, compare with the oracle"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers, oracle_py
from lariat_amd import capi
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
junk = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
inf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib = capi.load_library(os.environ.get('LH_LIB'))
o = oracle_py.load()
names, contigs = helpers.small_genome()
oidx = o.index_build_naive(names, contigs)
idx = lib.index_from_arrays(oidx.arrays())
rs = helpers.small_reads(names, contigs, n_barcodes=nb, pairs=pairs, seed=5, junk=junk)
b = helpers.batch_of(rs)
print("ctx...", flush=True)
ctx = idx.context(rs.n_pairs)
print("align...", flush=True)
t = time.time()
res = ctx.align_barcodes(b, lib.opts(run_inference=inf))
print("gpu done %.2fs" % (time.time() - t), ctx.timings(), flush=True)
ores = oidx.align_barcodes(b, o.opts(run_inference=inf), threads=8)
helpers.assert_same_result(res, ores, inference=bool(inf))
print("PARITY OK n_cand=%d" % res.n_cand, flush=True)

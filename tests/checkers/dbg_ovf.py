"""development aid: the k_rfa slab-overflow path on the GPU with per-launch logging (LH_DEBUG_SYNC=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["LH_RFA_SLAB_KB"] = os.environ.get("LH_RFA_SLAB_KB", "8")
import helpers, oracle_py
from lariat_amd import capi
lib = capi.load_library()
oracle = oracle_py.load()
names, contigs = helpers.small_genome()
oidx = oracle.index_build_naive(names, contigs)
idx = lib.index_from_arrays(oidx.arrays())
rs = helpers.small_reads(names, contigs, n_barcodes=int(os.environ.get("NBC", "6")), pairs=80, junk=0.03, seed=23)
b = helpers.batch_of(rs)
res = idx.context(rs.n_pairs).align_barcodes(b)
helpers.assert_same_result(res, oidx.align_barcodes(b, threads=8), inference=True)
print("ok")

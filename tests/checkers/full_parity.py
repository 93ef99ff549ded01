"""bit-exact comparison of the HIP path with the oracle at bench scale (BASELINE.json configs[1] shape): every candidate field,
CIGAR, mismatch locus, RFA pick, MAPQ (development aid; the pytest suite compares at sizes the oracle finishes in seconds)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers, oracle_py
from lariat_amd import capi, synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 64
nbc = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rep = len(sys.argv) > 3 and sys.argv[3] == "repeats"   # configs[4]-like: planted segmental duplications and repeat families
lib = capi.load_library(); oracle = oracle_py.load()
contigs = synth.make_genome([int(mb * 1e6)], seed=20261002, **(dict(n_dup=200, dup_len=20000, dup_identity=0.99, n_rep_family=30, rep_len=300, rep_copies=80) if rep else {}))
prefix = "/tmp/lh_genome_%g%s" % (mb, "_rep" if rep else "")
if not os.path.exists(prefix + ".bwt"):
    lib.index_build(prefix, ["chr20"], contigs, threads=0)
idx = lib.index_load(prefix); oidx = oracle.index_load(prefix)
rs = synth.make_reads(contigs, ["chr20"], n_barcodes=nbc, pairs_per_barcode=100, seed=20261005, with_names=False)
b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed)
t = time.time(); res = idx.context(rs.n_pairs).align_barcodes(b); t_gpu = time.time() - t
t = time.time(); ores = oidx.align_barcodes(b, threads=min(os.cpu_count(), 128)); t_cpu = time.time() - t
helpers.assert_same_result(res, ores, inference=True)
for k in ("n_sa", "glob_cells", "n_rescue", "rescue_cells"):
    assert res.counters[k] == ores.counters[k], k
assert res.counters["n_ext"] <= ores.counters["n_ext"]   # K1's sweep filter / collapsed sweeps skip bwt_extend calls that cannot give a seed
print("bwt_extend: HIP %d, reference count %d" % (res.counters["n_ext"], ores.counters["n_ext"]))
print("genome %g Mb%s" % (mb, ", 200 duplications of 20 kb at 99 %, 30 repeat families x 80 copies" if rep else ""))
print("full parity ok: %d pairs, %d candidates, every field equal (HIP %.1f s incl. transfers, oracle %.1f s on %d threads)" % (rs.n_pairs, res.n_cand, t_gpu, t_cpu, min(os.cpu_count(), 128)))

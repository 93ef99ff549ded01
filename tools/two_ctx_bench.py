"""experiment: one batch as two half-batches on two contexts (own streams) aligned concurrently from two host threads"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lariat_amd import capi, synth
lib = capi.load_library()
mb, nbc, parts = 64, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 2
contigs = synth.make_genome([int(mb * 1e6)], seed=20261002)
prefix = "/tmp/lh_genome_%g" % mb
if not os.path.exists(prefix + ".bwt"):
    lib.index_build(prefix, ["chr20"], contigs, threads=0)
idx = lib.index_load(prefix)
rs = synth.make_reads(contigs, ["chr20"], n_barcodes=nbc, pairs_per_barcode=100, with_names=False)
subs = [rs.slice_barcodes(nbc * k // parts, nbc * (k + 1) // parts) for k in range(parts)]
ctxs = []
for s in subs:
    c = idx.context(s.n_pairs); c.upload(capi.Batch.from_arrays(s.seq, s.seq_off, s.bc_pair_off, s.name_seed)); ctxs.append(c)
o = lib.opts()
for rep in range(3):
    t = time.perf_counter()
    skew = float(os.environ.get("SKEW_MS", "0")) * 1e-3
    def run(k):
        if k: time.sleep(skew * k)
        ctxs[k].align_resident(o)
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(ctxs))]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
    print("parts=%d rep %d: %.1f ms  %.2f M pairs/s" % (parts, rep, dt * 1e3, rs.n_pairs / dt / 1e6), flush=True)

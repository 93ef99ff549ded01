"""quick GPU timing of the resident pipeline on a synthetic genome (development aid; bench.py is the contract)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lariat_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--genome-mb", type=float, default=16)
ap.add_argument("--barcodes", type=int, default=1000)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
lib = capi.load_library(os.environ.get('LH_LIB'))
t = time.time()
contigs = synth.make_genome([int(a.genome_mb * 1e6)], seed=20261002)
prefix = "/tmp/lh_genome_%g" % a.genome_mb
if not os.path.exists(prefix + ".bwt"):
    lib.index_build(prefix, ["chr20"], contigs, threads=0)
print("genome+index %.1fs" % (time.time() - t), flush=True)
idx = lib.index_load(prefix)
t = time.time()
rs = synth.make_reads(contigs, ["chr20"], n_barcodes=a.barcodes, pairs_per_barcode=100, with_names=False)
print("reads %.1fs  pairs=%d" % (time.time() - t, rs.n_pairs), flush=True)
b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed)
ctx = idx.context(rs.n_pairs)
ctx.upload(b)
o = lib.opts()
for i in range(a.reps):
    t = time.time()
    ctx.align_resident(o)
    dt = time.time() - t
    print("rep %d: %.1f ms  %.0f pairs/s  %s" % (i, dt * 1e3, rs.n_pairs / dt, ["%s=%.2f" % x for x in ctx.timings()]), flush=True)
print("counters", ctx.download().counters, flush=True)

#!/usr/bin/env python3
"""development aid: what kind of candidates need K7's wave DP on the bench workload (a 256 Mb genome is enough: the read model is the same)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi, workload
lib = capi.load_library()
ctg = workload.hg38_like_contigs(int(256e6))
l_pac = sum(c[1] for c in ctg)
pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
idx = lib.index_build_device(pac, l_pac, ctg)
r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED, n_barcodes=2000, pairs_per_barcode=100)
b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
res = idx.context(200000).align_barcodes(b)
ok = res.rid >= 0
ind = res.indels[ok]; mm = res.mismatches[ok]; sc = res.soft_clipped[ok]
n = int(ok.sum())
print("candidates %d; with indels %d (%.2f %%); no indel and >= 4 mismatches %d (%.2f %%); no indel, >= 4 mismatches, clipped %d" % (
    n, int((ind > 0).sum()), 100.0 * (ind > 0).mean(), int(((ind == 0) & (mm >= 4)).sum()), 100.0 * ((ind == 0) & (mm >= 4)).mean(), int(((ind == 0) & (mm >= 4) & (sc > 0)).sum())))
print("glob_cells", res.counters["glob_cells"], "per slow candidate ~", res.counters["glob_cells"] / max(1, int((ind > 0).sum()) + int(((ind == 0) & (mm >= 4)).sum())))
hist = np.bincount(np.minimum(mm[ind == 0], 12))
print("mismatches of indel-free candidates:", hist.tolist())

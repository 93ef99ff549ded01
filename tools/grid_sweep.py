"""development aid: K1 time against the number of persistent waves (lh_context_opts.smem_grid) at hg38 scale"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lariat_amd import capi, workload
lib = capi.load_library()
ctg = workload.hg38_like_contigs(int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3100000000)
l_pac = sum(c[1] for c in ctg)
pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
idx = lib.index_build_device(pac, l_pac, ctg)
r = lib.synth_reads(pac, l_pac, ctg, seed=5, n_barcodes=10000, pairs_per_barcode=100)
b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
for g in (2048, 3072, 4096, 5120, 6144, 8192, 12288):
    ctx = idx.context(r["n_pairs"], smem_grid=g)
    ctx.upload(b)
    opts = lib.opts()
    ctx.align_resident(opts)
    ks = {}
    for _ in range(3):
        ctx.align_resident(opts)
        for n, ms in ctx.timings():
            ks.setdefault(n, []).append(ms)
    print(g, {k: round(float(np.mean(v)), 2) for k, v in ks.items() if "smem4" in k}, flush=True)
    ctx.close()

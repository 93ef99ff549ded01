#!/usr/bin/env python3
"""Development probe: one hg38-scale index, one 2 M-pair batch, the kernel times under different lh_context_opts
(launch geometry of K1 / K7 / K8, K8's slab size).  Usage: python tools/ctx_sweep.py [--genome-mb 3100]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lariat_amd import capi, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3100)
    ap.add_argument("--barcodes", type=int, default=20000)
    a = ap.parse_args()
    lib = capi.load_library()
    ctg = workload.hg38_like_contigs(int(a.genome_mb * 1e6))
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg)
    r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED, n_barcodes=a.barcodes, pairs_per_barcode=100)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    opts = lib.opts()
    for kw in ({}, {"aln_grid": 2560}, {"aln_grid": 10240}, {"rfa_grid": 2048}, {"rfa_grid": 8192}, {"rfa_slab_kb": 1024, "rfa_grid": 8192}, {"smem_grid": 4096}, {"smem_grid": 8192}):
        ctx = idx.context(r["n_pairs"], **kw)
        ctx.upload(b)
        kern, steps = {}, []
        import time
        for s in range(4):
            t = time.perf_counter()
            ctx.align_resident(opts)
            if s:
                steps.append((time.perf_counter() - t) * 1e3)
                for name, ms in ctx.timings():
                    kern.setdefault(name, []).append(ms)
        k = {n: round(float(np.mean(v)), 2) for n, v in kern.items()}
        print(json.dumps({"opts": kw, "ms_per_step": round(float(np.mean(steps)), 2), "k_smem4": k.get("k_smem4"), "p2": k.get("k_smem4_p2"), "p3": k.get("k_smem4_p3"),
                          "k_aln_fast": k.get("k_aln_fast"), "k_aln": k.get("k_aln"), "k_rfa": k.get("k_rfa")}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()

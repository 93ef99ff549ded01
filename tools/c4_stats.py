#!/usr/bin/env python3
"""configs[4] on one GPU, development probe: the distributions the repeat-regime kernels are shaped by — SMEM intervals, seeds, chains (all / kept),
regions and candidates per read, molecules per barcode — from a stage dump of a few hundred barcodes, then per-kernel times of a full batch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lariat_amd import capi, workload  # noqa: E402


def dist(name, x):
    x = np.asarray(x)
    print("%-28s mean %8.1f  p50 %6d  p90 %6d  p99 %6d  max %7d   sum %d" % (name, x.mean(), np.percentile(x, 50), np.percentile(x, 90), np.percentile(x, 99), x.max(), x.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3060.0)
    ap.add_argument("--dump-barcodes", type=int, default=200)
    ap.add_argument("--barcodes", type=int, default=4000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--pairs-per-barcode", type=int, default=100, help="the timed steps' barcode size (the same pairs in fewer, larger barcodes: K8 works one barcode per wave)")
    ap.add_argument("--lib", default=None, help="another build of the library (e.g. the -DLH_RFA_PROF one of tools/prof_rfa.sh)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--rfa-slab-kb", type=int, default=0, help="K8's regular slab per wave (default 2048): a small value sends every barcode to the next tier")
    ap.add_argument("--frac", type=float, default=1.0, help="share of every barcode's pairs drawn on the repeat copies (bench.py's mixed leg: 0.05)")
    a = ap.parse_args()
    lib = capi.load_library(a.lib)
    g = workload.config4_genome(lib, a.genome_mb * 1e6, quiet=False)
    idx = lib.index_build_device(g["pac"], g["l_pac"], g["contigs"])
    idx.set_alt(g["alt_flags"])
    r = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 400, n_barcodes=a.dump_barcodes, pairs_per_barcode=100)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    ctx = idx.context(r["n_pairs"])
    d = ctx.stage_dump(b)
    dist("intervals per read", np.diff(d.intv_off))
    ns = np.diff(d.seed_off)
    dist("seeds per read", ns)
    for t in (32, 64, 128, 256, 512):
        print("   reads with > %3d seeds: %.3f   (their share of all seeds %.3f)" % (t, (ns > t).mean(), ns[ns > t].sum() / max(1, ns.sum())))
    nch = np.diff(d.chain_off)
    dist("kept chains per read", nch)
    dist("seeds per kept chain", d.chain_nseeds)
    dist("regions per read", np.diff(d.reg_off))
    # clusters of seeds: seeds of a read sorted by reference position, a new cluster where the gap exceeds 1,000
    ncl, big = [], []
    for i in range(min(len(ns), 4000)):
        s0, s1 = int(d.seed_off[i]), int(d.seed_off[i + 1])
        if s1 - s0 == 0:
            ncl.append(0)
            continue
        rb = np.sort(d.seed_rbeg[s0:s1])
        cut = np.nonzero(np.diff(rb) > 1000)[0]
        ncl.append(len(cut) + 1)
        sizes = np.diff(np.concatenate([[0], cut + 1, [len(rb)]]))
        big.append(sizes.max())
    dist("position clusters per read", ncl)
    dist("largest cluster (seeds)", big)
    res = ctx.align_barcodes(b)
    dist("candidates per read", np.diff(res.cand_off))
    ctx.close()
    n_pairs = a.barcodes * a.pairs_per_barcode
    ctx = idx.context(n_pairs, **({"rfa_slab_kb": a.rfa_slab_kb} if a.rfa_slab_kb else {}))
    opts = lib.opts(flags=a.flags)
    for s in range(a.steps):
        if a.frac >= 1.0:
            r = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 400 + s, n_barcodes=a.barcodes, pairs_per_barcode=a.pairs_per_barcode)
        else:
            uniq = workload.outside_windows(g["contigs"], g["alt_flags"], g["windows"])
            n_rep = int(round(a.pairs_per_barcode * a.frac))
            ra = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 700 + s, n_barcodes=a.barcodes, pairs_per_barcode=n_rep)
            rb = lib.synth_reads(g["pac"], g["l_pac"], uniq, seed=workload.READS_SEED + 100700 + s, n_barcodes=a.barcodes, pairs_per_barcode=a.pairs_per_barcode - n_rep)
            r = workload.interleave_reads(ra, rb)
        ctx.upload_slot(s, capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"]))
    ctx.select(0)
    ctx.align_resident(opts)
    for s in range(a.steps):
        ctx.select(s)
        t0 = time.perf_counter()
        ctx.align_resident(opts)
        dt = time.perf_counter() - t0
        print("step %d: %.1f ms  (%.0f pairs/s)" % (s, dt * 1e3, n_pairs / dt))
        for name, ms in ctx.timings():
            print("   %-28s %10.3f ms" % (name, ms))
    res = ctx.download()
    print("counters:", res.counters)


if __name__ == "__main__":
    main()

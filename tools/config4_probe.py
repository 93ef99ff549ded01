#!/usr/bin/env python3
"""configs[4] on one GPU, development probe: the hg38-scale genome of workload.config4_genome, one batch of reads drawn on the copies;
prints candidates per read, work counters and per-kernel times (and, with --oracle N, compares the first N barcodes with the oracle)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lariat_amd import capi, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3060.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--barcodes", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--oracle", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--rfa-slab-kb", type=int, default=0)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--no-download", action="store_true", help="timings only (a library of another ABI minor for an A/B)")
    ap.add_argument("--big", type=int, default=0, help="one more barcode of this many pairs (1,200 molecules)")
    a = ap.parse_args()
    lib = capi.load_library(a.lib)
    t0 = time.time()
    g = workload.config4_genome(lib, a.genome_mb * 1e6, scale=a.scale, quiet=False)
    print("genome %.1f s" % (time.time() - t0)); t0 = time.time()
    idx = lib.index_build_device(g["pac"], g["l_pac"], g["contigs"])
    idx.set_alt(g["alt_flags"])
    print("index %.1f s" % (time.time() - t0)); t0 = time.time()
    n_pairs = a.barcodes * 100
    ctx = idx.context(n_pairs + a.big, **({"rfa_slab_kb": a.rfa_slab_kb} if a.rfa_slab_kb else {}))
    opts = lib.opts(flags=a.flags)
    rs = []
    for s in range(a.steps):
        r = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 400 + s, n_barcodes=a.barcodes, pairs_per_barcode=100)
        if a.big:
            big = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 41, n_barcodes=1, pairs_per_barcode=a.big, mol_min=1200, mol_max=1200)
            r = dict(seq=np.concatenate([r["seq"], big["seq"]]), seq_off=np.concatenate([r["seq_off"], big["seq_off"][1:] + r["seq_off"][-1]]),
                     bc_pair_off=np.concatenate([r["bc_pair_off"], big["bc_pair_off"][1:] + r["bc_pair_off"][-1]]).astype(np.int32), name_seed=np.concatenate([r["name_seed"], big["name_seed"]]))
        rs.append(r)
        ctx.upload_slot(s, capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"]))
    print("reads %.1f s" % (time.time() - t0))
    for s in range(a.steps):
        ctx.select(s)
        t0 = time.perf_counter()
        ctx.align_resident(opts)
        dt = time.perf_counter() - t0
        print("step %d: %.1f ms  (%.0f pairs/s)" % (s, dt * 1e3, n_pairs / dt))
        for name, ms in ctx.timings():
            print("   %-28s %10.3f ms" % (name, ms))
    if a.no_download:
        return
    res = ctx.download()
    nc = np.diff(res.cand_off)
    print("candidates per read: mean %.1f median %d p90 %d p99 %d max %d" % (nc.mean(), np.median(nc), np.percentile(nc, 90), np.percentile(nc, 99), nc.max()))
    print("counters:", res.counters)
    print("per pair: rescues %.2f, rescue cells %.0f, ext cells %.0f, glob cells %.0f" % tuple(res.counters[k] / n_pairs for k in ("n_rescue", "rescue_cells", "ext_cells", "glob_cells")))
    if a.oracle:
        import helpers
        import oracle_py
        o = oracle_py.load()
        oidx = o.index_from_arrays(idx.export(), g["pac"])
        oidx.set_alt(g["alt_flags"])
        r = rs[-1]
        p1 = int(r["bc_pair_off"][a.oracle])
        sub = capi.Batch.from_arrays(r["seq"][: r["seq_off"][2 * p1]], r["seq_off"][: 2 * p1 + 1], r["bc_pair_off"][: a.oracle + 1], r["name_seed"][:p1])
        t0 = time.time()
        ref = oidx.align_barcodes(sub, threads=min(os.cpu_count() or 8, 128))
        print("oracle: %d pairs in %.1f s" % (p1, time.time() - t0))
        got = idx.context(p1).align_barcodes(sub)
        helpers.assert_same_result(got, ref, inference=True)
        print("HIP == oracle on %d barcodes (%d candidates)" % (a.oracle, ref.n_cand))


if __name__ == "__main__":
    main()

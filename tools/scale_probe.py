#!/usr/bin/env python3
"""Development probe: device index build + one batch at a given genome size, with stage timings, the index self-check and an
oracle comparison on a sample of barcodes.  Usage: python tools/scale_probe.py --genome-mb 3100 [--pairs 1000000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from lariat_amd import capi, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=256)
    ap.add_argument("--barcodes", type=int, default=10000)
    ap.add_argument("--oracle-barcodes", type=int, default=200)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--chunk-log2", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--smem-grid", type=int, default=0)
    ap.add_argument("--rfa-grid", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0, help="lh_opts.flags (e.g. 32 = K4 classes one after the other)")
    ap.add_argument("--lib", default=None, help="a development build of the library (e.g. with -DLH_SMEM_TURNS)")
    a = ap.parse_args()
    info = {"genome_mb": a.genome_mb, "nproc": os.cpu_count()}
    for line in open("/proc/meminfo"):
        if line.startswith(("MemTotal", "MemAvailable")):
            info[line.split(":")[0]] = line.split()[1] + " kB"
    lib = capi.load_library(a.lib)
    t = time.time()
    ctg = workload.hg38_like_contigs(int(a.genome_mb * 1e6))
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=20261002)
    info["t_genome_s"] = round(time.time() - t, 2)
    t = time.time()
    idx = lib.index_build_device(pac, l_pac, ctg, build_chunk_log2=a.chunk_log2)
    info["t_index_build_s"] = round(time.time() - t, 2)
    info["sa_interval"] = idx.sa_interval
    t = time.time()
    info["index_check(rows,bad_order,bad_lf)"] = idx.check(stride=4096)
    info["t_index_check_s"] = round(time.time() - t, 2)
    print(json.dumps(info), flush=True)
    t = time.time()
    r = lib.synth_reads(pac, l_pac, ctg, seed=20261005, n_barcodes=a.barcodes, pairs_per_barcode=100)
    info["t_reads_s"] = round(time.time() - t, 2)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    ctx = idx.context(r["n_pairs"], smem_grid=a.smem_grid, rfa_grid=a.rfa_grid)
    ctx.upload(b)
    opts = lib.opts(flags=a.flags)
    kern = {}
    for s in range(a.steps + 1):
        t = time.perf_counter()
        ctx.align_resident(opts)
        dt = time.perf_counter() - t
        if s:
            info.setdefault("ms_per_step", []).append(round(dt * 1e3, 2))
            for name, ms in ctx.timings():
                kern.setdefault(name, []).append(ms)
    info["kernel_ms"] = {k: round(float(np.mean(v)), 3) for k, v in kern.items()}
    res = ctx.download()
    info["counters"] = res.counters
    info["pairs_per_s"] = round(r["n_pairs"] / (np.mean(info["ms_per_step"]) * 1e-3))
    act = res.active_idx[0::2]
    ok = (res.rid[act] == r["truth_rid"]) & (np.abs(res.pos[act] - r["truth_pos1"]) < 20)
    info["read1_placed_at_truth"] = round(float(ok.mean()), 5)
    print(json.dumps(info), flush=True)
    if a.oracle_barcodes > 0:
        import helpers
        import oracle_py
        o = oracle_py.load()
        t = time.time()
        oidx = o.index_from_arrays(idx.export(), pac)
        info["t_export_to_oracle_s"] = round(time.time() - t, 2)
        nb = a.oracle_barcodes
        p1 = int(r["bc_pair_off"][nb])
        sub = capi.Batch.from_arrays(r["seq"][: r["seq_off"][2 * p1]], r["seq_off"][: 2 * p1 + 1], r["bc_pair_off"][: nb + 1], r["name_seed"][:p1])
        t = time.time()
        ores = oidx.align_barcodes(sub, threads=min(os.cpu_count(), 64))
        info["t_oracle_s"] = round(time.time() - t, 2)
        info["oracle_pairs_per_s"] = round(p1 / (time.time() - t))
        ctx2 = idx.context(p1)
        helpers.assert_same_result(ctx2.align_barcodes(sub), ores, inference=True)
        info["oracle_parity"] = "ok: %d pairs, %d candidates, every field" % (p1, ores.n_cand)
    print(json.dumps(info), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(info, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

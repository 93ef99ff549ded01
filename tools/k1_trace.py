#!/usr/bin/env python3
"""what bounds K1's pass 1 — measured, not argued.  With the -DLH_K1_TRACE build (tools/prof_rfa.sh) and LH_K1_TRACE=1 in the environment,
k_smem_pass<1> records every memory request it makes (table, address, bytes; k_smem4.h: K1_REQ) while it aligns one headline batch (BASELINE configs[2]: 2 M
pairs against the hg38-scale index), and k_k1_replay then issues exactly those sequences with the same launch geometry and no bookkeeping in between: its
time is the floor of THIS request stream on this chip.  Prints one JSON object (-> profiles/r05_k1_request_floor.json, which bench.py's roofline reads):
requests and bytes per table, the replay's time, pass 1's own time from the product build on the same batch."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    from lariat_amd import capi, workload
    lib = capi.load_library(a.lib)
    ctg = workload.hg38_like_contigs(int(a.genome_mb * 1e6))
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg)
    r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED, n_barcodes=a.barcodes, pairs_per_barcode=100)
    ctx = idx.context(r["n_pairs"])
    ctx.upload_slot(0, capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"]))
    ctx.select(0)
    opts = lib.opts()
    for _ in range(a.steps):
        ctx.align_resident(opts)
        print("TIMINGS " + json.dumps({k: v for k, v in ctx.timings()}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3100.0)
    ap.add_argument("--barcodes", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--commit", default=None, help="recorded in the output (the GPU box has no git)")
    ap.add_argument("--kernel-stats", default=None, help="a tools/rocpd_stats.py CSV of the same build: k_smem_first's average duration goes into the output")
    a = ap.parse_args()
    if a.child:
        return child(a)
    out = {}
    prof = os.path.join(ROOT, "lariat_amd", "_build", "liblariat_hip_prof.so")
    for tag, lib, env in (("product", None, {}), ("trace", prof, {"LH_K1_TRACE": "1"})):
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--genome-mb", str(a.genome_mb), "--barcodes", str(a.barcodes), "--steps", str(a.steps if lib is None else 2)]
        if lib:
            cmd += ["--lib", lib]
        p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        if p.returncode:
            sys.stderr.write(p.stderr[-4000:])
            raise SystemExit("child failed (%s)" % tag)
        tim = [json.loads(l[8:]) for l in p.stdout.splitlines() if l.startswith("TIMINGS ")]
        if tag == "product":
            out["pass1_bracket_ms"] = round(min(t["k_smem4"] for t in tim[1:]), 3)   # k_smem_first + k_smem_pass<1>, HIP events
            out["kernel_ms_product"] = {k: round(v, 3) for k, v in tim[-1].items() if k.startswith("k_smem")}
        else:
            m = [l for l in p.stderr.splitlines() if "K1TRACE" in l]
            tr = json.loads(m[-1].split("K1TRACE", 1)[1])
            out.update(tr)
            out["pass1_bracket_ms_while_tracing"] = round(tim[-1]["k_smem4"], 3)
    out["commit"] = a.commit
    if a.kernel_stats and os.path.exists(a.kernel_stats):
        import csv
        for row in csv.DictReader(open(a.kernel_stats)):
            if row["Name"].startswith("k_smem_first"):
                out["k_smem_first_ms"] = round(float(row["AverageNs"]) / 1e6, 3)
    rq = out["requests_by_table"]
    out["requests_total"] = sum(v[0] for v in rq.values())
    out["bytes_total"] = sum(v[1] for v in rq.values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()

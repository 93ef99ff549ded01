#!/usr/bin/env python3
"""Host-to-host rate of the C-ABI when the host keeps TWO batches in flight: two contexts on one index, two host threads, each doing
upload -> lh_align_resident -> lh_result_download for its share of the batches (what a Go driver would do with two goroutines).  One
thread's PCIe copies run beside the other's kernels.  Usage: python tools/host_pipeline_bench.py [--genome-mb 3100] [--threads 2]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3100)
    ap.add_argument("--barcodes", type=int, default=10000)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--threads", type=int, default=2)
    a = ap.parse_args()
    lib = capi.load_library()
    ctg = workload.hg38_like_contigs(int(a.genome_mb * 1e6))
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg)
    batches = []
    for g in range(a.batches):
        r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED + g, n_barcodes=a.barcodes, pairs_per_barcode=100)
        batches.append(capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"]))
    n_pairs = a.barcodes * 100
    opts = lib.opts()
    out = {"genome_mb": a.genome_mb, "pairs_per_batch": n_pairs, "batches": a.batches}
    for nt in (1, a.threads):
        ctxs = [idx.context(n_pairs) for _ in range(nt)]
        for c in ctxs:   # warm-up: pools sized, pinned result block allocated
            c.upload(batches[0]); c.align_resident(opts); c.download_raw()

        def work(k):
            c = ctxs[k]
            for g in range(k, a.batches, nt):
                c.upload(batches[g])
                c.align_resident(opts)
                c.download_raw()

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.perf_counter() - t0
        out["%d_in_flight_pairs_per_s" % nt] = round(a.batches * n_pairs / dt)
        for c in ctxs:
            c.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

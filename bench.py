#!/usr/bin/env python3
"""bench.py — read-pairs/sec through the MI355X-native per-barcode align loop (BASELINE.json metric).

A "step" is ONE pass of the whole hot path (lariat.go:461-547 minus DumpToBams: SMEM seeding, SA lookup, chaining,
banded SW extension, dedup/patch, mate rescue, region->CIGAR, tagBest/molecule inference/RFA/MAPQ/duplicates/split reads)
over one resident batch of synthetic barcode-sorted read pairs.  Workload = BASELINE.json configs[1]: a 64 Mb
"chr20-like" synthetic genome (no real genome exists offline), 1M 2x150 pairs in 10k barcodes per GPU.  Inputs are in
HBM when the timed region starts (lh_batch_upload before, lh_result_download after).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: barcodes are independent (lariat.go:348-350), so every rank aligns its own barcode range with the index
replicated in its HBM; there is NO collective on the data path (weak scaling: per-GPU work fixed).  torch.distributed is
used only for the barrier and the max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np


# HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KB -> B), collected separately and committed
# under profiles/ (see DESIGN.md section 4 for the calibration of FETCH_SIZE on this access pattern); None = not measured.
TRAFFIC = {"k_smem3": (1.455e8 + 5.255e7) * 1024.0, "k_smem4": (1.204e7 + 2.556e6 + 8.317e6 + 1.26e6 + 1.172e5 + 4.328e5) * 1024.0}   # FETCH_SIZE + WRITE_SIZE (KB) of the three K1 launches, profiles/r01_pmc_summary_v10.txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=float, default=64.0, help="synthetic chr20-like genome size (configs[1]: 64 Mb)")
    ap.add_argument("--barcodes", type=int, default=10000, help="barcodes per GPU (configs[1]: 10k, 100 pairs each)")
    ap.add_argument("--pairs-per-barcode", type=int, default=100)
    ap.add_argument("--cpu-sample-barcodes", type=int, default=600, help="barcodes of the same workload timed on the host cores (cpu_baseline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the double-buffered upload/align/download measurement")
    ap.add_argument("--index-dir", default=os.environ.get("LH_INDEX_DIR", "/tmp/lariat_amd_bench"))
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # LH_BENCH_SHARE_GPU=1 (development aid for 1-GPU boxes): all ranks use device 0 and the control collectives run over gloo,
    # which exercises everything of the N>1 path except RCCL itself
    share = os.environ.get("LH_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    from lariat_amd import capi, synth
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    lib = capi.load_library()
    if lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: liblariat_hip has no CPU fallback")

    # ---- synthetic genome + FM-index (built once per node by local rank 0, BWA-compatible files) ----
    t0 = time.time()
    glen = int(a.genome_mb * 1e6)
    contigs = synth.make_genome([glen], seed=20261002)
    names = ["chr20"]
    os.makedirs(a.index_dir, exist_ok=True)
    prefix = os.path.join(a.index_dir, "chr20like_%d.fa" % glen)
    if (rank == 0 if share else local_rank == 0) and not os.path.exists(prefix + ".done"):
        lib.index_build(prefix, names, contigs, threads=0)
        open(prefix + ".done", "w").write("ok\n")
    if dist is not None:
        dist.barrier()
    while not os.path.exists(prefix + ".done"):
        time.sleep(0.2)
    idx = lib.index_load(prefix, device=local_rank)
    idx_bwt_bytes = os.path.getsize(prefix + ".bwt")
    t_index = time.time() - t0

    # ---- this rank's barcode range: weak scaling, barcodes [rank*B, (rank+1)*B) of the sorted input ----
    t0 = time.time()
    rs = synth.make_reads(contigs, names, n_barcodes=a.barcodes, pairs_per_barcode=a.pairs_per_barcode, seed=20261003 + 2 + 1000 * rank, with_names=False)
    batch = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed)
    t_reads = time.time() - t0
    n_pairs = rs.n_pairs
    ctx = idx.context(n_pairs)
    opts = lib.opts()
    t0 = time.time()
    ctx.upload(batch)
    t_upload = time.time() - t0

    def sync_all():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        ctx.align_resident(opts)
    sync_all()
    t0 = time.perf_counter()
    kern = {}
    for _ in range(a.steps):
        ctx.align_resident(opts)   # enqueues every kernel on the context's stream and synchronises it
        for name, ms in ctx.timings():   # HIP events recorded on that stream around each launch
            kern.setdefault(name, []).append(ms)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    ctx.download_raw()   # first call sizes the pinned result block
    t0 = time.time()
    ctx.download_raw()   # the C-ABI cost (lh_result_download + lh_result_free): a cgo / C host reads the SoA block in place
    t_download = time.time() - t0
    t0 = time.time()
    res = ctx.download()
    t_download_py = time.time() - t0   # the same plus this harness's numpy copies of every column
    cnt = res.counters
    # the reference's own bwt_extend count on this batch (the algorithmic bytes are the reference's bookkeeping): one
    # untimed pass with K1's sweep filter off — that pass performs (or accounts for, in the unique runs and the 12-mer jump)
    # every call the reference makes; the parity tests check it against the oracle's counter
    # (run as the fused instance k_smem4_t<0>, so that a profile of this command keeps the timed launches' averages apart)
    os.environ["LH_NO_SWEEP_FILTER"] = "1"
    os.environ["LH_SMEM4_FUSED"] = "1"
    try:
        ctx.align_resident(opts)
        ref_n_ext = ctx.download().counters["n_ext"]
    finally:
        del os.environ["LH_NO_SWEEP_FILTER"]
        del os.environ["LH_SMEM4_FUSED"]

    if rank == 0:
        total_pairs = n_pairs * world * a.steps
        value = total_pairs / elapsed
        avg = {k: float(np.mean(v)) for k, v in kern.items()}
        # K1 (mem_collect_intv) is three launches of one templated kernel (passes 1, 2, 3: k_smem4_t<3>, <4>, <2>): the roofline is
        # taken over the stage, its bytes (the reference's bwt_extend count) over the sum of the three durations
        stage = dict(avg)
        k1_parts = [k for k in avg if k.startswith("k_smem4")]
        if k1_parts:
            for k in k1_parts:
                del stage[k]
            stage["k_smem4"] = sum(avg[k] for k in k1_parts)
        dom = max(stage, key=stage.get)
        avg_dom = stage[dom]
        # roofline of the dominant kernel.  For K1 (k_smem4): every bwt_extend reads two 32-B occurrence records of the
        # re-laid-out FM-index (the .bwt file's own layout would be two 64-B blocks) and every read's bases once;
        # n_ext = the reference's call count (see above); the timed passes execute fewer (cnt["n_ext"]).
        smem_bytes = 64.0 * ref_n_ext + 1.0 * int(rs.seq_off[-1])
        alg_bytes = {
            "k_smem3": smem_bytes, "k_smem4": smem_bytes,
            "k_seed": 64.0 * cnt["n_lf"] + 8.0 * cnt["n_sa"],
        }.get(dom, 0.0)
        achieved = alg_bytes / (avg_dom * 1e-3) / 1e9 if avg_dom > 0 else 0.0
        # measured ceiling for this access pattern: independent random 32-B record reads from a table the size of the occurrence table
        try:
            ceiling, _ = lib.diag_random_read(max(int(idx_bwt_bytes), 1 << 20), 32, 1 << 27, device=local_rank)
            ceiling = round(ceiling, 1)
        except Exception:
            ceiling = None
        roofline = {"bound": "hbm", "kernel": dom if dom != "k_smem4" else "k_smem4_t<3>+<4>+<2> (K1: passes 1-3 of mem_collect_intv)", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                    "traffic": TRAFFIC.get(dom), "algorithmic_bytes_per_launch": alg_bytes,
                    "bwt_extend_reference": ref_n_ext, "bwt_extend_performed_or_accounted": cnt["n_ext"], "practical_ceiling_GBps": ceiling, "avg_launch_ms": round(avg_dom, 4),
                    "kernel_ms": {k: round(v, 3) for k, v in avg.items()}}
        out = {
            "metric": "read-pairs/sec aligned (per-barcode align loop: seeding + SW + RFA/MAPQ), synthetic chr20-like reference",
            "value": round(value, 1), "unit": "read-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64/i32 (FM-index + integer DP), f64 (RFA scores)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: %.0f Mb chr20-like synthetic genome, %d pairs 2x150 (143+150 post-trim) / %d barcodes per GPU, "
                                   "RFA on device" % (a.genome_mb, n_pairs, a.barcodes),
                       "pairs_per_gpu": n_pairs, "barcodes_per_gpu": a.barcodes, "parallelism": "barcode-range shards, no collective"},
            "roofline": roofline,
            "setup_s": {"genome+index": round(t_index, 1), "reads": round(t_reads, 1), "upload_h2d": round(t_upload, 3), "download_d2h": round(t_download, 3), "download_d2h_plus_numpy_copies": round(t_download_py, 3)},
            "pcie_inclusive_pairs_per_s": round(n_pairs / (elapsed / a.steps + t_upload + t_download), 1),
        }
        if not a.no_pipeline:
            out["pipelined_pcie_inclusive_pairs_per_s"] = pipelined(idx, batch, n_pairs, opts)
            out["two_context_resident_pairs_per_s"] = two_contexts(idx, rs, opts)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, rs, prefix)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pipelined(idx, batch, n_pairs, opts, workers=2, rounds=3):
    """the production shape of the drop-in: host buffers in, host result SoA out, with `workers` contexts (own streams)
    double-buffering so that one batch's PCIe transfers and host-side result assembly overlap another's kernels.
    Informational (the headline metric is the resident rate above)."""
    import threading
    ctxs = [idx.context(n_pairs) for _ in range(workers)]
    for c in ctxs:   # warm
        c.align_barcodes(batch, opts)
    t0 = time.perf_counter()

    def work(c):
        for _ in range(rounds):
            c.align_barcodes(batch, opts, raw=True)

    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return round(workers * rounds * n_pairs / dt, 1)


def two_contexts(idx, rs, opts, rounds=3):
    """informational: the same batch as two half-batches (barcodes are independent) resident in two contexts and aligned
    concurrently from two host threads — kernels of one half fill the gaps of the other (K1 is memory-request bound, K4 is
    VALU bound).  The headline `value` stays the single-context rate, whose per-kernel times are the ones profiled."""
    import threading
    from lariat_amd import capi
    nbc = len(rs.bc_pair_off) - 1
    ctxs = []
    for k in range(2):
        s = rs.slice_barcodes(nbc * k // 2, nbc * (k + 1) // 2)
        c = idx.context(s.n_pairs)
        c.upload(capi.Batch.from_arrays(s.seq, s.seq_off, s.bc_pair_off, s.name_seed))
        ctxs.append(c)
    best = 0.0
    for _ in range(rounds + 1):
        t0 = time.perf_counter()
        th = [threading.Thread(target=c.align_resident, args=(opts,)) for c in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        best = max(best, rs.n_pairs / dt)
    for c in ctxs:
        c.close()
    return round(best, 1)


def cpu_baseline(a, rs, prefix):
    """the oracle (CPU restatement of the Go+BWA path; the reference itself cannot be built here) timed on the host cores,
    on a bounded sample of the same workload; threaded over barcodes like lariat's worker pool (lariat.go:348-350)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from lariat_amd import capi
    o = oracle_py.load()
    oidx = o.index_load(prefix)
    nb = min(a.cpu_sample_barcodes, len(rs.bc_pair_off) - 1)
    sub = rs.slice_barcodes(0, nb)
    b = capi.Batch.from_arrays(sub.seq, sub.seq_off, sub.bc_pair_off, sub.name_seed)
    cores = min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    oidx.time_align(b, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(sub.n_pairs / dt, 1), "unit": "read-pairs/s", "cores": cores, "kind": "port",
            "sample": "first %d barcodes (%d pairs) of the same batch, %.1f s wall" % (nb, sub.n_pairs, dt)}


if __name__ == "__main__":
    main()

"""BASELINE.json's configs on the GPU, against the oracle:

  configs[0]  4.6 Mb "E. coli-like" genome, 10 k pairs / 100 barcodes written as 9-line FASTQ, through lh_ingest_* -> lh_align_barcodes
  configs[1]  64 Mb "chr20-like" genome, 1 M pairs / 10 k barcodes: every field of every candidate
  work units of the reference's own size: barcodes of 5,000 and 30,000 pairs (fastqreader/reader.go:205 caps a set at 30,000)
  configs[2]  hg38-scale genome (3.1 Gb, index built in HBM): index self-check, oracle parity on a sample of barcodes, and
              size-independent properties on a whole 1 M-pair batch (idempotence; barcode-range split == whole batch)
  configs[4]  (one GPU) the same size with planted segmental duplications and repeat families, reads drawn from on and around them

Integer / index fields bit-exact, MAPQ within +-1, float scores within 1e-9 relative (helpers.assert_same_result).
"""
import os
import time

import numpy as np
import pytest

import helpers
from lariat_amd import capi, synth, workload

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 8, 128)


@pytest.fixture(scope="module")
def lib():
    L = capi.load_library()
    assert L.device_count() >= 1
    return L


def synthetic(lib, oracle, total_bases, **index_opts):
    ctg = workload.hg38_like_contigs(total_bases)
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg, **index_opts)
    return ctg, l_pac, pac, idx


def unpack(pac, ctg):
    out = []
    for _, ln, off in ctg:
        i = np.arange(off, off + ln)
        out.append(((pac[i >> 2] >> ((~i & 3) << 1)) & 3).astype(np.uint8))
    return out


def test_config0_ecoli_scale_fastq_to_alignments(lib, oracle, tmp_path):
    """configs[0]: the plumbing case — 9-line barcode-sorted FASTQ in, every alignment field out, equal to the oracle's"""
    ctg = [("ecoli_like", 4600000, 0)]
    l_pac = 4600000
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg)
    oidx = oracle.index_from_arrays(idx.export(), pac)
    contigs = unpack(pac, ctg)
    rs = synth.make_reads(contigs, ["ecoli_like"], n_barcodes=100, pairs_per_barcode=100, seed=workload.READS_SEED)
    path = str(tmp_path / "reads.fastq")
    open(path, "w").write(synth.to_fastq9(rs, trim_prefix=7))
    n_pairs = n_sets = 0
    for b in lib.ingest(path, trim=7, max_pairs=2500):
        res = idx.context(b.n_pairs).align_barcodes(b)
        helpers.assert_same_result(res, oidx.align_barcodes(b, threads=THREADS), inference=True)
        n_pairs += b.n_pairs
        n_sets += b.n_sets
    assert n_pairs == 10000 and n_sets == 100


def test_config1_chr20_scale_every_field(lib, oracle):
    """configs[1]: 64 Mb, 1 M pairs in 10 k barcodes — the whole batch against the oracle"""
    ctg = [("chr20", 64000000, 0)]
    l_pac = 64000000
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
    idx = lib.index_build_device(pac, l_pac, ctg)
    oidx = oracle.index_from_arrays(idx.export(), pac)
    r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED + 1, n_barcodes=10000, pairs_per_barcode=100, junk_frac=0.002)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    res = idx.context(r["n_pairs"]).align_barcodes(b)
    ref = oidx.align_barcodes(b, threads=THREADS)
    assert ref.n_cand > 2 * r["n_pairs"]
    helpers.assert_same_result(res, ref, inference=True)


def test_reference_sized_work_units(lib, oracle):
    """barcodes of 5,000 and 30,000 pairs (the reader's cap, reader.go:205) between ordinary ones: K8 runs them through its
    large-slab second pass; every field equals the oracle's"""
    ctg = [("c0", 1200000, 0), ("c1", 800000, 1200000)]
    l_pac = 2000000
    pac = lib.synth_genome(l_pac, seed=77)
    idx = lib.index_build_device(pac, l_pac, ctg)
    oidx = oracle.index_from_arrays(idx.export(), pac)
    sizes = [100, 5000, 60, 30000, 7]
    parts = [lib.synth_reads(pac, l_pac, ctg, seed=50 + k, n_barcodes=1, pairs_per_barcode=n, mol_min=8, mol_max=10, junk_frac=0.01) for k, n in enumerate(sizes)]
    seq = np.concatenate([p["seq"] for p in parts])
    offs, base = [np.zeros(1, dtype=np.int64)], 0
    for p in parts:
        offs.append(p["seq_off"][1:] + base)
        base += int(p["seq_off"][-1])
    seq_off = np.concatenate(offs)
    bco = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    seeds = np.concatenate([p["name_seed"] for p in parts])
    b = capi.Batch.from_arrays(seq, seq_off, bco, seeds)
    ctx = idx.context(int(bco[-1]))
    t0 = time.time()
    res = ctx.align_barcodes(b)
    dt = time.time() - t0
    ref = oidx.align_barcodes(b, threads=THREADS)
    helpers.assert_same_result(res, ref, inference=True)
    print("reference-sized work units: %d pairs aligned in %.2f s (k_rfa %.1f ms)" % (int(bco[-1]), dt, dict(ctx.timings()).get("k_rfa", -1)))
    assert dt < 20.0


def test_config2_hg38_scale(lib, oracle):
    """configs[2]'s reference: 3.1 Gb in 24 contigs — 6.2 G suffixes, past 2^31 symbols (occurrence super-blocks), dense SA + ISA"""
    ctg, l_pac, pac, idx = synthetic(lib, oracle, 3100000000)
    assert 2 * l_pac > 1 << 32 and idx.sa_interval == 1
    rows, bad_order, bad_lf = idx.check(stride=997)
    assert rows > 6000000 and bad_order == 0 and bad_lf == 0
    r = lib.synth_reads(pac, l_pac, ctg, seed=workload.READS_SEED + 2, n_barcodes=10000, pairs_per_barcode=100)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    ctx = idx.context(r["n_pairs"])
    res = ctx.align_barcodes(b)
    # (1) a sample of barcodes against the oracle, on the SAME index (exported to the layout of bwa's files)
    oidx = oracle.index_from_arrays(idx.export(), pac)
    nb = 400
    p1 = int(r["bc_pair_off"][nb])
    sub = capi.Batch.from_arrays(r["seq"][: r["seq_off"][2 * p1]], r["seq_off"][: 2 * p1 + 1], r["bc_pair_off"][: nb + 1], r["name_seed"][:p1])
    ref = oidx.align_barcodes(sub, threads=THREADS)
    got = idx.context(p1).align_barcodes(sub)
    helpers.assert_same_result(got, ref, inference=True)
    # (2) size-independent properties on the whole batch
    assert (res.active_idx >= 0).all() and (np.add.reduceat(res.active.astype(np.int64), res.cand_off[:-1]) == 1).all()   # one active candidate per read
    act = res.active_idx[0::2]
    placed = (res.rid[act] == r["truth_rid"]) & (np.abs(res.pos[act] - r["truth_pos1"]) < 20)
    assert placed.mean() > 0.995
    again = ctx.align_barcodes(b)                                                                                          # idempotence
    helpers.assert_same_result(again, res, inference=True, mapq_tol=0, rel=0)
    half = 5000                                                                                                            # barcodes are independent:
    ph = int(r["bc_pair_off"][half])                                                                                       # a barcode-range shard == that range of the whole
    lo = capi.Batch.from_arrays(r["seq"][: r["seq_off"][2 * ph]], r["seq_off"][: 2 * ph + 1], r["bc_pair_off"][: half + 1], r["name_seed"][:ph])
    rl = ctx.align_barcodes(lo)
    nc = int(res.cand_off[2 * ph])
    assert rl.n_cand == nc
    for f in helpers.INT_FIELDS + ["active", "is_proper", "duplicate", "molecule_id", "mapq"]:
        a, w = getattr(rl, f), getattr(res, f)
        n = {"cand_off": 2 * ph + 1, "cigar_off": nc + 1, "mm_off": nc + 1, "cigar": int(res.cigar_off[nc]), "mm_ref_loc": int(res.mm_off[nc]),
             "mm_read_loc": int(res.mm_off[nc])}.get(f, nc)
        assert np.array_equal(a[:n], w[:n]), f


def test_config4_like_segdup_biased(lib, oracle):
    """configs[4] on one GPU: the hg38-scale genome with planted segmental duplications (1,500 x 20 kb at 99 %, 300 x 20 kb identical),
    interspersed repeat families and 40 ALT contigs (1 Mb copies of primary regions at 99.7 %, flagged is_alt as <prefix>.alt would),
    every read drawn from on and around them — several candidates per read, equal pair scores (Go's generator decides), chains on ALT
    and primary copies competing in mem_chain_flt, the RFA stress case: a sample of barcodes against the oracle on the same index,
    every field"""
    ctg = workload.hg38_like_contigs(3060000000)
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED + 4)
    dups = workload.plant_segdups(pac, ctg, 1500, 20000, 0.99, seed=41, n_families=40, family_len=320, family_copies=60)
    dups += workload.plant_segdups(pac, ctg, 300, 20000, 1.0, seed=42)
    pac, l_pac, ctg_all, alt_flags, alts = workload.add_alt_contigs(pac, ctg, 40, 1000000, 0.997, seed=43)
    idx = lib.index_build_device(pac, l_pac, ctg_all)
    idx.set_alt(alt_flags)
    win = workload.repeat_windows(ctg_all, dups[:1500] + dups[-300:], flank=50000) + workload.repeat_windows(ctg_all, alts, flank=20000)
    r = lib.synth_reads(pac, l_pac, win, seed=workload.READS_SEED + 4, n_barcodes=3000, pairs_per_barcode=100, sub_hi=0.02, junk_frac=0.01)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    ctx = idx.context(r["n_pairs"])
    t0 = time.time()
    res = ctx.align_barcodes(b)
    dt = time.time() - t0
    nf = np.add.reduceat(res.in_filtered.astype(np.int64), res.cand_off[:-1])
    multi = float((nf >= 2).mean())
    on_alt = float((np.asarray(alt_flags)[res.rid[res.rid >= 0]] == 1).mean())
    print("segdup/ALT-biased reads: %d pairs in %.2f s, %.1f candidates per read, %.0f %% of the reads with two or more equally plausible ones, %.0f %% of the candidates on ALT contigs"
          % (r["n_pairs"], dt, res.n_cand / res.n_reads, 100 * multi, 100 * on_alt))
    assert multi > 0.15 and on_alt > 0.05
    oidx = oracle.index_from_arrays(idx.export(), pac)
    oidx.set_alt(alt_flags)
    nb = 300
    p1 = int(r["bc_pair_off"][nb])
    sub = capi.Batch.from_arrays(r["seq"][: r["seq_off"][2 * p1]], r["seq_off"][: 2 * p1 + 1], r["bc_pair_off"][: nb + 1], r["name_seed"][:p1])
    ref = oidx.align_barcodes(sub, threads=THREADS)
    got = idx.context(p1).align_barcodes(sub)
    helpers.assert_same_result(got, ref, inference=True)
    again = ctx.align_barcodes(b)
    helpers.assert_same_result(again, res, inference=True, mapq_tol=0, rel=0)


def test_hg38_scale_index_from_files(lib, oracle, tmp_path_factory):
    """the reference's only way to an index is bwa_idx_load(path, BWA_IDX_ALL) (gobwa.go:128-147): an hg38-scale index — with .amb holes and 40
    ALT contigs — is built on the device, saved in the layout of `bwa index`'s files (suffix array at interval 32), loaded again from
    those files, and must be the same index: the side tables derived from the loaded BWT and text (dense SA, inverse SA, LCP / PLCP, k-mer
    tree) digest to what the builder left, ALT flags and holes survive, and 40 k pairs align to the same results"""
    import shutil
    import tempfile
    ctg = workload.hg38_like_contigs(3060000000)
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED + 7)
    pac, l_pac, ctg_all, alt_flags, alts = workload.add_alt_contigs(pac, ctg, 40, 1000000, 0.997, seed=47)
    holes = [(int(ctg_all[k][2] + 1000 * (k + 1)), 50000 + 1000 * k, "N") for k in range(0, 24, 3)]   # (what bns_fasta2bntseq records for runs of N; their bases are random in .pac)
    built = lib.index_build_device(pac, l_pac, ctg_all)
    built.set_alt(alt_flags)
    built.set_holes(holes)
    r = lib.synth_reads(pac, l_pac, ctg_all, seed=workload.READS_SEED + 9, n_barcodes=400, pairs_per_barcode=100)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    ctx = built.context(r["n_pairs"])
    want = ctx.align_barcodes(b)
    ctx.close()
    want_digest = built.digest()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30) else None
    d = tempfile.mkdtemp(prefix="lh_idx_", dir=base)
    try:
        prefix = os.path.join(d, "ref.fa")
        t0 = time.time()
        built.save(prefix)
        t_save = time.time() - t0
        sizes = {e: os.path.getsize(prefix + e) for e in (".bwt", ".sa", ".pac", ".ann", ".amb", ".alt")}
        assert sizes[".sa"] == 56 + (2 * l_pac // 32) * 8 and sizes[".bwt"] > 2 * l_pac // 4
        built.close()   # two resident hg38-scale indexes do not fit one device
        t0 = time.time()
        loaded = lib.index_load(prefix)
        t_load = time.time() - t0
        print("hg38-scale index: saved in %.1f s (%.1f GB of files), loaded from files in %.1f s" % (t_save, sum(sizes.values()) / 1e9, t_load))
        assert loaded.sa_interval == 1 and loaded.l_pac == l_pac
        assert [c[0] for c in loaded.contigs()] == [c[0] for c in ctg_all]
        assert loaded.alt() == list(alt_flags)
        assert loaded.digest() == want_digest
        rows, bad_order, bad_lf = loaded.check(stride=4999)
        assert rows > 1000000 and bad_order == 0 and bad_lf == 0
        ctx = loaded.context(r["n_pairs"])
        got = ctx.align_barcodes(b)
        ctx.close()
        helpers.assert_same_result(got, want, inference=True, mapq_tol=0, rel=0)
        loaded.save(os.path.join(d, "again.fa"))   # holes and ALT names round-trip
        for e in (".amb", ".alt", ".ann"):
            assert open(prefix + e, "rb").read() == open(os.path.join(d, "again.fa") + e, "rb").read()
        loaded.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_config4_repeat_families_at_its_multiplicity(lib, oracle):
    """configs[4] as SURVEY 8d defines it: 120 segmental-duplication families of 50-200 copies x 20 kb at 98-99.5 %, LINE- and SINE-like families,
    40 ALT contigs, every read drawn ON the copies (workload.config4_genome) — tens to hundreds of candidates per read, up to 50 + 50 mate-rescue
    attempts per pair (gobwa.go:286-325), thousands of raw molecules per barcode (lariat.go:1370-1408).  200 ordinary barcodes and one of 2,500
    pairs whose reads come from 1,200 molecules (more than 1,000 candidate molecules survive scrapMolecules: the optimizer's 8 M^2 fastScore
    calls) against the oracle on the same index, every field; no capacity error anywhere."""
    g = workload.config4_genome(lib, 3060000000)
    idx = lib.index_build_device(g["pac"], g["l_pac"], g["contigs"])
    idx.set_alt(g["alt_flags"])
    small = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 40, n_barcodes=200, pairs_per_barcode=100)
    big = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 41, n_barcodes=1, pairs_per_barcode=2500, mol_min=1200, mol_max=1200)
    seq = np.concatenate([small["seq"], big["seq"]])
    seq_off = np.concatenate([small["seq_off"], big["seq_off"][1:] + small["seq_off"][-1]])
    bco = np.concatenate([small["bc_pair_off"], big["bc_pair_off"][1:] + small["bc_pair_off"][-1]]).astype(np.int32)
    seeds = np.concatenate([small["name_seed"], big["name_seed"]])
    b = capi.Batch.from_arrays(seq, seq_off, bco, seeds)
    n_pairs = int(bco[-1])
    ctx = idx.context(n_pairs)
    t0 = time.time()
    res = ctx.align_barcodes(b)
    dt = time.time() - t0
    nc = np.diff(res.cand_off)
    cnt = res.counters
    print("configs[4]: %d pairs in %.2f s; candidates per read: mean %.1f, median %d, 99th percentile %d, max %d; %.1f rescue attempts per pair (%.1f M cells per pair)"
          % (n_pairs, dt, nc.mean(), np.median(nc), np.percentile(nc, 99), nc.max(), cnt["n_rescue"] / n_pairs, cnt["rescue_cells"] / n_pairs / 1e6))
    assert nc.mean() >= 10 and np.percentile(nc, 99) >= 100
    # the 50-hit cap of the rescue loops binds: reads with more than 50 hits within rescue_score_delta of their best
    best = np.maximum.reduceat(res.score, res.cand_off[:-1])
    near = np.add.reduceat((res.score >= np.repeat(best, nc) - 25).astype(np.int64), res.cand_off[:-1])
    assert (near > 50).mean() > 0.05
    c0 = int(res.cand_off[2 * int(bco[-2])])
    n_mol_big = int(res.molecule_id[c0:].max()) + 1
    print("the 2,500-pair barcode: %d candidates, %d molecules after scrapMolecules" % (res.n_cand - c0, n_mol_big))
    assert n_mol_big >= 1000
    oidx = oracle.index_from_arrays(idx.export(), g["pac"])
    oidx.set_alt(g["alt_flags"])
    t0 = time.time()
    ref = oidx.align_barcodes(b, threads=THREADS)
    print("oracle: %.1f s on %d threads" % (time.time() - t0, THREADS))
    helpers.assert_same_result(res, ref, inference=True)
    assert cnt["n_rescue"] == ref.counters["n_rescue"] and cnt["rescue_cells"] == ref.counters["rescue_cells"]
    # (r06) the same batch with every rescue window run whole (LH_F_RESCUE_FULL): the same result from five to ten times the Smith-Waterman cells
    full = ctx.align_barcodes(b, lib.opts(flags=capi.LH_F_RESCUE_FULL))
    helpers.assert_same_result(full, ref, inference=True)
    print("rescue cells: the reference's %.3g, executed %.3g with the certificate, %.3g without" % (cnt["rescue_cells"], cnt["rescue_cells_exec"], full.counters["rescue_cells_exec"]))
    assert 0 < cnt["rescue_cells_exec"] < 0.25 * full.counters["rescue_cells_exec"]
    ctx.close()
    # (r06) barcodes of 200 and 400 pairs on the copies: their molecule x read tables outgrow the regular 2 MiB slabs, so they pass through K8's slab tiers at
    # their real sizes (16 MiB, 128 MiB; allocated when the batch lists them), 40 barcodes of each against the oracle, every field
    parts = [lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 44 + k, n_barcodes=40, pairs_per_barcode=ppb) for k, ppb in enumerate((200, 400))]
    seq2 = np.concatenate([p["seq"] for p in parts])
    so2 = np.concatenate([parts[0]["seq_off"], parts[1]["seq_off"][1:] + parts[0]["seq_off"][-1]])
    bco2 = np.concatenate([parts[0]["bc_pair_off"], parts[1]["bc_pair_off"][1:] + parts[0]["bc_pair_off"][-1]]).astype(np.int32)
    b2 = capi.Batch.from_arrays(seq2, so2, bco2, np.concatenate([p["name_seed"] for p in parts]))
    c2 = idx.context(int(bco2[-1]))
    try:
        free0 = lib.device_memory()[0]
        res2 = c2.align_barcodes(b2)
        tier_bytes = free0 - lib.device_memory()[0]
    finally:
        c2.close()
    print("200- and 400-pair barcodes: %d pairs, %.1f candidates per read; HBM taken during the batch (tier slabs among it): %.2f GiB" % (int(bco2[-1]), np.diff(res2.cand_off).mean(), tier_bytes / 2**30))
    helpers.assert_same_result(res2, oidx.align_barcodes(b2, threads=THREADS), inference=True)
    # the MIXED workload of bench.py's `mixed` leg on the same index: 5 of every barcode's 100 pairs drawn on the repeat copies, 95 on unique sequence (the primary
    # contigs outside the copies' windows), interleaved per barcode (workload.interleave_reads) — repeat-regime reads and unique reads side by side in every kernel's waves and in every
    # barcode's inference: 200 barcodes against the oracle, every field
    primary = workload.outside_windows(g["contigs"], g["alt_flags"], g["windows"])
    assert sum(c[1] for c in primary) > 0.2 * g["l_pac"]
    ra = lib.synth_reads(g["pac"], g["l_pac"], g["windows"], seed=workload.READS_SEED + 42, n_barcodes=200, pairs_per_barcode=5)
    rb = lib.synth_reads(g["pac"], g["l_pac"], primary, seed=workload.READS_SEED + 43, n_barcodes=200, pairs_per_barcode=95)
    m = workload.interleave_reads(ra, rb)
    assert m["n_pairs"] == 20000 and int(m["from_first"].sum()) == 1000 and (np.diff(m["bc_pair_off"]) == 100).all()
    bm = capi.Batch.from_arrays(m["seq"], m["seq_off"], m["bc_pair_off"], m["name_seed"])
    cm = idx.context(m["n_pairs"])
    try:
        got = cm.align_barcodes(bm)
    finally:
        cm.close()   # (its pools leave HBM before the oracle's pass: the module runs beside an hg38-scale index)
    ncm = np.diff(got.cand_off).reshape(-1, 2).sum(axis=1)
    print("mixed: %.1f candidates per pair on the copies, %.1f elsewhere" % (ncm[m["from_first"]].mean(), ncm[~m["from_first"]].mean()))
    assert ncm[m["from_first"]].mean() > 10 * ncm[~m["from_first"]].mean() and np.median(ncm[~m["from_first"]]) <= 4
    refm = oidx.align_barcodes(bm, threads=THREADS)
    helpers.assert_same_result(got, refm, inference=True)
    assert got.counters["n_rescue"] == refm.counters["n_rescue"] and got.counters["rescue_cells"] == refm.counters["rescue_cells"]
    assert 0 < got.counters["rescue_cells_exec"] < 0.5 * got.counters["rescue_cells"]   # (r06) K6 ran a fraction of the reference's Smith-Waterman cells (k_rescue3.h)

"""GPU parity of the whole candidate-generation path (K1..K7: SMEM, SA, chaining, banded SW extension, dedup/patch, mate
rescue, region->CIGAR + lariat's GetAlignments walk) against the oracle, through the C-ABI."""
import os

import numpy as np
import pytest

import helpers
from lariat_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_library()
    assert L.device_count() >= 1
    return L


def test_gobwa1_on_gpu(lib):
    """go/src/test/gobwa_test.go:13-28 through the HIP path"""
    idx = lib.index_load(helpers.PHIX)
    b = capi.Batch([capi.sequence_convert(helpers.PHIX_READ_A), capi.sequence_convert(helpers.PHIX_READ_B)], [0, 1])
    res = idx.context(8).align_barcodes(b, lib.opts(run_inference=0))
    a0 = list(res.cands_of_read(0))
    assert res.pos[a0[0]] == 210 and idx.contigs()[res.rid[a0[0]]][0] == "PhiX"
    assert len(list(res.cands_of_read(1))) == 1


def test_zero_length_read_on_gpu(lib, oracle):
    """go/src/test/lariat_test.go:12-24"""
    idx = lib.index_load(helpers.PHIX)
    recs = helpers.read_fastq9(helpers.GOLDEN + "/zero_length_read_test.fastq.gz", trim=7)
    reads = []
    for r in recs:
        reads += [capi.sequence_convert(r["r1"]), capi.sequence_convert(r["r2"])]
    b = capi.Batch(reads, [0, 3])
    res = idx.context(8).align_barcodes(b, lib.opts(improper_pair_penalty=-17.0, run_inference=0))
    assert (np.diff(res.cand_off) >= 1).all() and (res.rid == -1).all()
    helpers.assert_same_result(res, oracle.index_load(helpers.PHIX).align_barcodes(b, oracle.opts(improper_pair_penalty=-17.0, run_inference=0)),
                               inference=False)


@pytest.mark.parametrize("seed,junk", [(5, 0.03), (11, 0.15)])
def test_candidates_synthetic(lib, oracle, seed, junk):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=20, pairs=100, seed=seed, junk=junk)
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT + helpers.DUMP_REGS)
    res = ctx.align_barcodes(b, lib.opts(run_inference=0))
    ores = oidx.align_barcodes(b, oracle.opts(run_inference=0), threads=8)
    helpers.assert_same_result(res, ores, inference=False)
    for k in ("n_sa", "glob_cells", "n_rescue", "rescue_cells"):
        assert res.counters[k] == ores.counters[k], k
    # n_ext: K1's sweep filter leaves intervals out of the backward sweeps that cannot give a seed (k_smem4.h): fewer bwt_extend
    # than the reference; without the filter (honoured per call) the count is the reference's
    assert 0 < res.counters["n_ext"] < ores.counters["n_ext"]
    res_nf = ctx.align_barcodes(b, lib.opts(run_inference=0, flags=capi.LH_F_NO_SWEEP_FILTER))
    helpers.assert_same_result(res_nf, ores, inference=False)
    assert res_nf.counters["n_ext"] == ores.counters["n_ext"]
    # ext_cells counts the DP cells the device evaluated: extensions that are provably ungapped (k_extend2.h) skip their DP
    assert 0 < res.counters["ext_cells"] <= ores.counters["ext_cells"]
    # the suffix array is re-sampled densely on load (every row for a genome this small): no BWT walk is left in bwt_sa
    assert idx.sa_interval == 1 and res.counters["n_lf"] == 0
    # back at the .sa file's interval (sub-sampling path) the walk lengths are the oracle's, and nothing else changes
    idx.resample_sa(oidx.arrays()["sa_intv"])
    res2 = ctx.align_barcodes(b, lib.opts(run_inference=0))
    helpers.assert_same_result(res2, ores, inference=False)
    assert res2.counters["n_lf"] == ores.counters["n_lf"]
    idx.resample_sa(4)   # densifying path from a sparser table
    res3 = ctx.align_barcodes(b, lib.opts(run_inference=0))
    helpers.assert_same_result(res3, ores, inference=False)
    assert 0 < res3.counters["n_lf"] < ores.counters["n_lf"]


def test_repeat_rich_genome(lib, oracle):
    """high candidate multiplicity (config-5 flavour): many dups and repeat families"""
    from lariat_amd import synth
    names = ["c1", "c2"]
    contigs = synth.make_genome([150000, 150000], seed=9, n_dup=30, dup_len=3000, dup_identity=0.995, n_rep_family=6, rep_len=300, rep_copies=40)
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=10, pairs_per_barcode=60, seed=3, junk_frac=0.05)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b, lib.opts(run_inference=0))
    helpers.assert_same_result(res, oidx.align_barcodes(b, oracle.opts(run_inference=0), threads=8), inference=False)
    assert np.diff(res.cand_off).max() >= 4


def test_long_noisy_reads(lib, oracle):
    """240-bp reads with 3 % substitutions and indels: first extensions of 128+ columns (the 256-column LDS class of
    k_extend_lane), deferred reads, gapped global alignments with wide bands, long CIGARs and many mismatch loci"""
    from lariat_amd import synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=12, pairs_per_barcode=80, seed=31, len1=240, len2=236, sub_lo=0.005, sub_hi=0.03, indel_rate=0.003, junk_frac=0.02)
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT + helpers.DUMP_REGS)
    res = ctx.align_barcodes(b)
    ores = oidx.align_barcodes(b, threads=8)
    helpers.assert_same_result(res, ores, inference=True)
    for k in ("glob_cells", "n_rescue", "rescue_cells"):
        assert res.counters[k] == ores.counters[k], k
    assert res.counters["ext_cells"] <= ores.counters["ext_cells"] and res.counters["n_ext"] <= ores.counters["n_ext"]


def test_pools_grow_on_demand(lib, oracle):
    """a repeat-dense 62 kb genome gives far more than the 32 seeds / 6 candidates per read the pools start with: the seed,
    region and candidate pools are re-sized from the device-side totals instead of failing (found by tests/checkers/fuzz_gpu.py, seed 2892)"""
    from lariat_amd import synth
    seed = 2892
    rng = np.random.default_rng(seed)
    ncont = int(rng.integers(1, 5))
    lens = [int(rng.integers(60000, 400000)) for _ in range(ncont)]
    names = ["c%d" % i for i in range(ncont)]
    contigs = synth.make_genome(lens, seed=seed, n_dup=int(rng.integers(0, 25)), dup_len=int(rng.integers(500, 6000)), dup_identity=float(rng.uniform(0.97, 1.0)),
                                n_rep_family=int(rng.integers(0, 6)), rep_len=int(rng.integers(100, 400)), rep_copies=int(rng.integers(5, 60)))
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    if rng.random() < 0.3:
        idx.resample_sa(int(rng.choice([2, 8, 32])))
    l1, l2 = int(rng.integers(50, 240)), int(rng.integers(50, 240))
    rs = synth.make_reads(contigs, names, n_barcodes=int(rng.integers(1, 12)), pairs_per_barcode=int(rng.integers(1, 120)), seed=seed + 7, len1=l1, len2=l2,
                          sub_lo=0.0, sub_hi=float(rng.uniform(0.0, 0.06)), indel_rate=float(rng.choice([0.0, 0.001, 0.01])), junk_frac=float(rng.choice([0.0, 0.05, 0.3])))
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    d = ctx.stage_dump(b)
    assert d.seed_off[-1] > 32 * 2 * rs.n_pairs   # really beyond the initial pool
    helpers.assert_same_dump(d, oidx.stage_dump(b), helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b, threads=8), inference=True)


def test_low_complexity_reads_are_never_refused(lib, oracle):
    """BWA never refuses a read (its vectors grow: gobwa.go:244-260 -> mem_align1_core); nor does the product.  A genome of poly-A tracts,
    microsatellites and a 200-copy tandem repeat, noisy reads with indels drawn from in and around them (up to ~2,500 seeds and ~1,000
    chains per read): no LH_E_LIMIT / LH_E_CAPACITY, every field of every candidate equal to the oracle's."""
    from lariat_amd import synth
    names, contigs = helpers.low_complexity_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=6, pairs_per_barcode=120, seed=3, sub_lo=0.002, sub_hi=0.03, indel_rate=0.002, mol_min=2, mol_max=3)
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    od = oidx.stage_dump(b)
    assert np.diff(od.seed_off).max() > 2000 and np.diff(od.chain_off).max() > 500
    helpers.assert_same_dump(ctx.stage_dump(b), od, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b, threads=8), inference=True)


def test_overlapped_download_and_staged_upload(lib, oracle):
    """one context, four batches, a staging thread: batch k + 1 is uploaded (lh_batch_stage_slot) while batch k is aligned, batch k's result is
    collected (lh_result_download_end) after batch k + 1's kernels ran beside its copies — each result equals the oracle's for its own batch"""
    import threading
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    sets = [helpers.small_reads(names, contigs, n_barcodes=6, pairs=60 + 20 * k, junk=0.05, seed=60 + k) for k in range(4)]
    batches = [helpers.batch_of(rs) for rs in sets]
    ctx = idx.context(max(rs.n_pairs for rs in sets))
    opts = lib.opts()
    ctx.upload_slot(1, batches[0])
    got = []
    for k in range(4):
        ctx.select(1 + k % 2)
        th = None
        if k + 1 < 4:
            th = threading.Thread(target=lambda kk=k + 1: ctx.stage_slot(1 + kk % 2, batches[kk]))
            th.start()
        ctx.align_resident(opts)
        if k:
            got.append(ctx.download_end())
        ctx.download_begin()
        if th:
            th.join()
    got.append(ctx.download_end())
    for k in range(4):
        helpers.assert_same_result(got[k], oidx.align_barcodes(batches[k], threads=8), inference=True)


@pytest.mark.parametrize("g,ins_first", [(1, False), (1, True), (2, False), (2, True)])
def test_gapped_path_beats_a_four_mismatch_diagonal(lib, oracle, g, ins_first):
    """K7's shifted-diagonal check on the device (the construction and the argument: tests/test_emu_front.py, helpers.k7_shift_case)"""
    names, contigs, b, p = helpers.k7_shift_case(g, ins_first)
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    res = idx.context(8).align_barcodes(b, lib.opts(run_inference=0))
    ores = oidx.align_barcodes(b, oracle.opts(run_inference=0))
    helpers.assert_same_result(res, ores, inference=False)
    c = list(res.cands_of_read(0))[0]
    assert int(res.indels[c]) == 2 and int(res.mismatches[c]) == 0 and int(res.pos[c]) == p - 66


@pytest.mark.parametrize("g,ins_first", [(3, True), (7, False), (8, True), (8, False), (12, True)])
def test_k7_narrow_band_and_its_proof(lib, oracle, g, ins_first):
    """k_aln_grp's band of 7 and the bound that decides whether its result stands, on the device (the construction: helpers.k7_band_case,
    the argument: tests/test_emu_front.py)"""
    names, contigs, b, p = helpers.k7_band_case(g, ins_first)
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    res = idx.context(8).align_barcodes(b, lib.opts(run_inference=0))
    ores = oidx.align_barcodes(b, oracle.opts(run_inference=0))
    helpers.assert_same_result(res, ores, inference=False)
    assert res.counters["glob_cells"] == ores.counters["glob_cells"]
    c = list(res.cands_of_read(0))[0]
    assert int(res.indels[c]) == 2 and int(res.pos[c]) == p


def test_k7_second_look(lib, oracle):
    """K7's second look on the device (k_aln.h, aln_deep_check; the argument and the builds with one check left out: tests/test_emu_front.py): four adversarial
    batches of helpers.k7_deep_batch — five to seven mismatches beside an excursion of two to four gap runs through low-complexity sequence — and reads with
    3.5-5 % substitutions on unique sequence, every field against the oracle; the second look settles most candidates with five or six mismatches there"""
    from lariat_amd import capi, synth
    for seed in (1, 2, 3, 4):
        names, contigs, reads = helpers.k7_deep_batch(seed, 192)
        oidx = oracle.index_build_naive(names, contigs)
        b = capi.Batch(reads, [0, len(reads) // 2])
        ref = oidx.align_barcodes(b, oracle.opts(run_inference=0), threads=8)
        res = lib.index_from_arrays(oidx.arrays()).context(len(reads) // 2).align_barcodes(b, lib.opts(run_inference=0))
        helpers.assert_same_result(res, ref, inference=False)
        ok = ref.rid >= 0
        assert int((ref.indels[ok] > 0).sum()) > 60 and res.counters["n_glob_exec"] < res.counters["n_glob_listed"] <= ref.counters["n_glob_exec"]
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=8, pairs_per_barcode=100, seed=31, sub_lo=0.035, sub_hi=0.05, indel_rate=0.0, mol_min=3, mol_max=5)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b, threads=8)
    res = lib.index_from_arrays(oidx.arrays()).context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, ref, inference=True)
    ok = ref.rid >= 0
    n56 = int(((ref.indels[ok] == 0) & (ref.mismatches[ok] >= 5) & (ref.mismatches[ok] <= 6)).sum())
    assert n56 > 200 and res.counters["n_glob_listed"] - res.counters["n_glob_exec"] > 0.7 * n56, (n56, res.counters["n_glob_listed"], res.counters["n_glob_exec"])


def test_context_moves_between_regimes(lib, oracle):
    """K4's choice of path for the wave-chained reads follows the context's previous batch (tests/test_emu_front.py has the argument): repeat families,
    unique sequence, repeat families twice — every result equal to the oracle's"""
    names, contigs, rs_rep = helpers.repeat_family_case(13, 24)
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs_uni = helpers.small_reads(names, contigs, n_barcodes=24, pairs=30, junk=0.03, seed=77)
    ctx = idx.context(max(rs_rep.n_pairs, rs_uni.n_pairs))
    for rs in (rs_rep, rs_uni, rs_rep, rs_rep):
        b = helpers.batch_of(rs)
        helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b, threads=8), inference=True)

"""GPU parity of the complete per-barcode align loop (DoRFAForOneBarcode, lariat.go:461-547, minus DumpToBams)
against the oracle: integer/index fields bit-exact, MAPQ within +-1, float scores within 1e-9 relative."""
import numpy as np
import pytest

import helpers
from lariat_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_library()
    assert L.device_count() >= 1
    return L


@pytest.fixture(scope="module")
def small(lib, oracle):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    return names, contigs, oidx, lib.index_from_arrays(oidx.arrays())


@pytest.mark.parametrize("seed,junk,pairs", [(5, 0.03, 100), (7, 0.2, 60), (9, 0.0, 5)])
def test_full_path_synthetic(lib, oracle, small, seed, junk, pairs):
    names, contigs, oidx, idx = small
    rs = helpers.small_reads(names, contigs, n_barcodes=16, pairs=pairs, seed=seed, junk=junk)
    rfa = np.ones(16, dtype=np.uint8)
    rfa[3] = 0   # a barcode that fails worthRunningRFA
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, bc_do_rfa=rfa)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b, threads=8), inference=True)


def test_centromere_zeroes_mapq(lib, oracle, small):
    names, contigs, oidx, idx = small
    rs = helpers.small_reads(names, contigs, n_barcodes=6, pairs=50, seed=2, junk=0.0)
    cs = np.array([0, -1, -1], dtype=np.int64)
    ce = np.array([150000, -1, -1], dtype=np.int64)
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, cen_start=cs, cen_end=ce)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b, threads=8), inference=True)
    act = res.active_idx
    in_cen = (res.rid[act] == 0) & (res.pos[act] > 0) & (res.pos[act] <= 150000)
    assert in_cen.any() and (res.mapq[act][in_cen] == 0).all()


def test_repeat_rich_inference(lib, oracle):
    names = ["c1", "c2"]
    contigs = synth.make_genome([150000, 150000], seed=9, n_dup=30, dup_len=3000, dup_identity=0.995, n_rep_family=6, rep_len=300, rep_copies=40)
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=10, pairs_per_barcode=60, seed=3, junk_frac=0.05)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b, threads=8), inference=True)


def test_improper_penalty_not_dyadic(lib, oracle, small):
    """a penalty that is not a multiple of 0.5 exercises the floating-point summation order of fastScore"""
    names, contigs, oidx, idx = small
    rs = helpers.small_reads(names, contigs, n_barcodes=6, pairs=60, seed=13, junk=0.1)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b, lib.opts(improper_pair_penalty=-4.3))
    helpers.assert_same_result(res, oidx.align_barcodes(b, oracle.opts(improper_pair_penalty=-4.3), threads=8), inference=True)


def test_round_trip_properties_large(lib):
    """size-independent properties on a batch too large for the oracle to be worth running: one active alignment per
    read, mates linked symmetrically, candidates sorted by score, MAPQ in range, placement accuracy"""
    names = ["chrS"]
    contigs = synth.make_genome([4000000], seed=20261002)
    import tempfile, os
    d = tempfile.mkdtemp()
    lib.index_build(os.path.join(d, "g"), names, contigs, threads=0)
    idx = lib.index_load(os.path.join(d, "g"))
    rs = synth.make_reads(contigs, names, n_barcodes=500, pairs_per_barcode=100, with_names=False)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    act = res.active_idx
    assert (act >= 0).all()
    assert (res.active[act] == 1).all() and res.active.sum() == res.n_reads
    mate = res.mate_idx[act]
    assert (res.mate_idx[mate] == act).all()
    assert ((res.mapq[act] >= 0) & (res.mapq[act] <= 60)).all()
    for r in range(0, res.n_reads, 997):
        s = res.score[res.cand_off[r]:res.cand_off[r + 1]]
        assert (np.diff(s) <= 0).all()
    truth = np.empty(res.n_reads, dtype=np.int64)
    truth[0::2] = rs.truth_pos1
    truth[1::2] = rs.truth_pos2
    ok = np.abs(res.pos[act] - truth) < 20
    assert ok.mean() > 0.99


def test_slab_overflow_second_pass(lib, oracle):
    """barcodes whose tables outgrow a wave's slab are redone by the second k_rfa launch with large slabs"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=6, pairs=80, junk=0.03, seed=23)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs, rfa_slab_kb=8).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b, threads=8), inference=True)


def test_two_contexts_concurrently(lib, oracle):
    """double buffering as a driver would do it: two contexts of one index (own streams, own pools) aligned from two host
    threads at the same time give the results of running them one after the other"""
    import threading
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    batches = [helpers.batch_of(helpers.small_reads(names, contigs, n_barcodes=6, pairs=60, junk=0.03, seed=41 + k)) for k in range(2)]
    ctxs = [idx.context(360) for _ in range(2)]
    seq = [ctxs[k].align_barcodes(batches[k]) for k in range(2)]
    out = [None, None]

    def work(k):
        for _ in range(3):
            out[k] = ctxs[k].align_barcodes(batches[k])

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        helpers.assert_same_result(out[k], seq[k], inference=True)


def test_fastq_ingest_to_alignment(lib, oracle, tmp_path):
    """the N2 reader feeding the hot path: a 9-line FASTQ written from synthetic reads, read back in batches of whole barcode
    sets, gives the alignments of the same reads handed over as arrays (trim 7, md5 tie-break seeds, nt4 conversion)"""
    from lariat_amd import synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=5, pairs=40, junk=0.03, seed=51)
    p = tmp_path / "reads.fastq"
    p.write_text(synth.to_fastq9(rs, trim_prefix=7))
    direct = idx.context(rs.n_pairs).align_barcodes(helpers.batch_of(rs))
    ctx = idx.context(rs.n_pairs)
    r0 = 0
    for b in lib.ingest(str(p), trim=7, max_pairs=90):
        res = ctx.align_barcodes(b)
        n = 2 * b.n_pairs
        c0, c1 = int(direct.cand_off[r0]), int(direct.cand_off[r0 + n])
        assert np.array_equal(res.cand_off, direct.cand_off[r0:r0 + n + 1] - c0)
        for f in ("rid", "pos", "reversed", "score", "nm", "mapq", "active", "duplicate"):
            assert np.array_equal(getattr(res, f), getattr(direct, f)[c0:c1]), f
        r0 += n
    assert r0 == 2 * rs.n_pairs


def test_fastq_to_bam_records(lib, oracle, tmp_path):
    """N2 -> K1..K8 -> N1: the BAM record content the product derives from the HIP result equals what the Python restatement of
    AppendBam (oracle/bam_oracle.py) derives from the ORACLE's result: independent implementation on independent input"""
    from lariat_amd import synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=4, pairs_per_barcode=50, seed=71, sub_hi=0.03, indel_rate=0.002, junk_frac=0.05)
    p = tmp_path / "reads.fastq"
    p.write_text(synth.to_fastq9(rs, trim_prefix=7))
    ctx = idx.context(rs.n_pairs)
    n = 0
    import os
    import sys
    sys.path.insert(0, os.path.join(helpers.ROOT, "oracle"))
    import bam_oracle   # the independent restatement of bamwriter.go:286-568 (flags, the unmapping rule's side effects, TLEN, tags)
    cols_wanted = ("name", "rgid", "qual1", "qual2", "trim_bases", "trim_quals", "bc", "rawbc", "bcqual", "si", "siqual")
    for b in lib.ingest(str(p), trim=7, max_pairs=120):
        got = lib.records_text(ctx.align_barcodes(b), b, names)                     # product records from the HIP result
        ores = oidx.align_barcodes(b, threads=8)
        cols = {c: b.column(c) for c in cols_wanted}
        want = bam_oracle.records_text(ores, cols, b.seq, b.seq_off, b.bc_pair_off, b.set_complete, names)   # oracle records from the oracle's result
        assert got == want
        # -debugBamTags: product records from the HIP result against the MapQData of the oracle's molecules (bamwriter.go:498-558)
        got_d = lib.records_text(ctx.align_barcodes(b), b, names, debug_tags=True)
        want_d = bam_oracle.records_text(ores, cols, b.seq, b.seq_off, b.bc_pair_off, b.set_complete, names, debug_tags=True, md_int=ores.md_int, md_sb_conf=ores.md_sb_conf)
        assert got_d == want_d
        n += len(got.splitlines())
    assert n >= 2 * rs.n_pairs


def test_simulated_accounting_on_hip_result(lib, oracle, tmp_path):
    """N4: lariat's -simulated counters and the check.py report, computed from the HIP result, equal those from the oracle's"""
    from lariat_amd import simulated, synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=5, pairs_per_barcode=40, seed=79, junk_frac=0.05)
    p = tmp_path / "reads.fastq"
    p.write_text(synth.to_fastq9(rs, trim_prefix=7))
    ctx = idx.context(rs.n_pairs)
    sg, so = simulated.SimulatedStats(), simulated.SimulatedStats()
    tg, to = [], []
    for b in lib.ingest(str(p), trim=7):
        nm = b.column("name")
        rg, ro = ctx.align_barcodes(b), oidx.align_barcodes(b, threads=8)
        sg.add(rg, nm, b.bc_pair_off, b.bc_do_rfa)
        so.add(ro, nm, b.bc_pair_off, b.bc_do_rfa)
        tg += lib.records_text(rg, b, names).splitlines()
        to += lib.records_text(ro, b, names).splitlines()
    assert sg.as_dict() == so.as_dict() and sg.total > 0
    assert simulated.check_report(tg, mate_aware=True) == simulated.check_report(to, mate_aware=True)


def test_two_lanes_equal_one(lib, oracle, small):
    """lh_context_opts.lanes = 2 on the device: two halves of a batch side by side from two host threads; merged result == oracle"""
    names, contigs, oidx, idx = small
    rs = helpers.small_reads(names, contigs, n_barcodes=12, pairs=70, seed=37, junk=0.04)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b, threads=8)
    ctx = idx.context(rs.n_pairs, lanes=2)
    for _ in range(2):
        helpers.assert_same_result(ctx.align_barcodes(b), ref, inference=True)
    one = helpers.batch_of(rs.slice_barcodes(4, 5))   # a single barcode cannot be cut: first lane only
    helpers.assert_same_result(ctx.align_barcodes(one), oidx.align_barcodes(one, threads=2), inference=True)

"""Kernel SOURCES of the front end run under the CPU SPMD emulator (tests/hipemu) and diffed against the oracle.
This is a logic check that works without a GPU; the `-m gpu` tests are the parity tests proper."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from lariat_amd import capi

EMU = os.environ.get("LH_EMU_LIB") or os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")   # (LH_EMU_LIB: e.g. an AddressSanitizer build, tests/hipemu/Makefile asan)


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    return capi.Library(EMU)


def test_emu_phix_front(emu, oracle):
    idx = emu.index_load(helpers.PHIX)
    oidx = oracle.index_load(helpers.PHIX)
    b = capi.Batch([capi.sequence_convert(helpers.PHIX_READ_A), capi.sequence_convert(helpers.PHIX_READ_B)], [0, 1])
    helpers.assert_same_dump(idx.context(8).stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT)


def test_emu_front_synthetic(emu, oracle):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=3, pairs=30, junk=0.05)
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT + helpers.DUMP_REGS)
    res = ctx.align_barcodes(b, emu.opts(run_inference=0))
    helpers.assert_same_result(res, oidx.align_barcodes(b, oracle.opts(run_inference=0)), inference=False)


def test_emu_occ_superblocks_and_sparse_sa(emu, oracle):
    """the occurrence table's u32 counts are relative to super-block bases (needed past 2^31 symbols, e.g. hg38): force
    tiny super-blocks; and keep the .sa file's sampling so that bwt_sa really walks the re-laid-out BWT"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays(), sb_shift=14, sa_intv=32)
    assert idx.sa_interval == 32
    rs = helpers.small_reads(names, contigs, n_barcodes=2, pairs=30, junk=0.05, seed=11)
    b = helpers.batch_of(rs)
    helpers.assert_same_dump(idx.context(rs.n_pairs).stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT)


def test_emu_full_inference(emu, oracle):
    """whole hot path (candidates + tagBest + molecules + RFA + MAPQ + duplicates + split reads) under emulation"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=3, pairs=40, junk=0.05)
    rfa = np.array([1, 0, 1], dtype=np.uint8)   # the middle barcode takes the no-RFA path (lariat.go:489-496)
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, bc_do_rfa=rfa)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b), inference=True)
    assert (res.active_idx >= 0).all()


def test_emu_inference_slab_overflow_pass(emu, oracle):
    """barcodes whose tables outgrow a wave's slab are redone by the second launch with large slabs: force that path"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=3, pairs=30, junk=0.05, seed=21)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs, rfa_slab_kb=4).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b), inference=True)


def test_emu_inference_slab_tiers(emu, oracle):
    """(r05) between the regular slabs and the few large ones there are two tiers; a barcode that outgrows one is listed for the next.  With tiers of 48 and 256 KiB
    (lh_context_opts.rfa_tier_kb) and 4-KiB regular slabs the suite's barcodes go through all of them: some finish in the second tier, the largest in the last launch.
    (r06) The tiers' slabs are allocated when a batch first needs them, as many as it lists barcodes (here: at most 2 and 1, so that the lists are longer than the tiers
    have waves), and grown by a later batch that lists more."""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    from lariat_amd import synth
    rs = synth.make_reads(contigs, names, n_barcodes=5, pairs_per_barcode=30, seed=77, junk_frac=0.05)
    big = synth.make_reads(contigs, names, n_barcodes=1, pairs_per_barcode=400, seed=78, junk_frac=0.05)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs, rfa_slab_kb=4, rfa_tier_kb=(48, 256)).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b), inference=True)
    b2 = helpers.batch_of(big)
    res2 = idx.context(big.n_pairs, rfa_slab_kb=4, rfa_tier_kb=(48, 256)).align_barcodes(b2)
    helpers.assert_same_result(res2, oidx.align_barcodes(b2), inference=True)
    # one context through a batch that needs no tier, one that needs them, and a larger one (the tiers grow); few waves per tier; one tier switched off
    both = synth.make_reads(contigs, names, n_barcodes=9, pairs_per_barcode=40, seed=79, junk_frac=0.05)
    ctx = idx.context(both.n_pairs, rfa_slab_kb=4, rfa_tier_kb=(48, 256), rfa_tier_grid=(2, 1))
    for batch in (rs, both, rs):
        bb = helpers.batch_of(batch)
        helpers.assert_same_result(ctx.align_barcodes(bb), oidx.align_barcodes(bb), inference=True)
    ctx.close()
    bb = helpers.batch_of(both)
    helpers.assert_same_result(idx.context(both.n_pairs, rfa_slab_kb=4, rfa_tier_kb=(-1, 256)).align_barcodes(bb), oidx.align_barcodes(bb), inference=True)


@pytest.mark.parametrize("build", ["default", "small"])
def test_emu_position_sort_long_lists_with_equal_positions(emu, oracle, build):
    """K8's position sort of a large barcode (k_rfa.h; lariat.go:1545-1547 sort.Sort(ByPosition), whose order of EQUAL positions is part of the result): one barcode
    of 480 pairs on two contigs, forty of the pairs present twice — contig lists of several hundred candidates, dozens of equal positions in each.  The default build
    sorts such a list by the network in one LDS block and, for the ties, Go's algorithm on (rank, place) words in LDS with its long ranges partitioned by the whole
    wave (lh_sort.h wave_go_pivot); the `small` build runs the network 64 places at a time with passes in the slab between the blocks, and Go's algorithm split
    between the slab (long ranges, by the wave) and LDS (ranges of up to 64, with the depth the long sort has left them) — the forms a 400-pair barcode on repeat
    families takes on the device.  Every field against the oracle."""
    from lariat_amd import synth
    lib = emu
    if build == "small":
        subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu"), "small"])
        lib = capi.Library(os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu_small.so"))
    rng = np.random.default_rng(3)
    names, contigs = ["chrP", "chrQ"], [rng.integers(0, 4, size=150000).astype(np.uint8), rng.integers(0, 4, size=90000).astype(np.uint8)]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=1, pairs_per_barcode=440, seed=9, junk_frac=0.02, mol_min=6, mol_max=9)
    n_dup = 40   # the first forty pairs once more, at the end of the barcode
    cut = int(rs.seq_off[2 * n_dup])
    seq = np.concatenate([rs.seq, rs.seq[:cut]])
    seq_off = np.concatenate([rs.seq_off, rs.seq_off[1:2 * n_dup + 1] + rs.seq_off[-1]])
    name_seed = np.concatenate([rs.name_seed, rs.name_seed[:n_dup] + np.uint64(12345)])
    b = capi.Batch.from_arrays(seq, seq_off, np.array([0, rs.n_pairs + n_dup], dtype=np.int32), name_seed)
    ref = oidx.align_barcodes(b, threads=8)
    ok = ref.rid >= 0
    assert int(ok.sum()) > 900   # (more candidates than the barcode-wide LDS sort holds: contig by contig)
    res = lib.index_from_arrays(oidx.arrays()).context(rs.n_pairs + n_dup).align_barcodes(b)
    helpers.assert_same_result(res, ref, inference=True)


def test_emu_fuzz_regression_clipped_pairs(emu, oracle):
    """a case the differential fuzzer found on the device (tests/checkers/fuzz_gpu.py, seed 95343; r06): repeat families, reads of 195 and 147 bases, gap-open
    penalties 3 / 8 — soft-clipped alignments compete inside a molecule.  markBest's first step scores a pair from integers (twice the entries' own parts,
    k_rfa.h MEnt::s2); with the clip term doubled a read's best alignment in one molecule changed, and a probability sum with it by 3e-8.  The same case through
    the kernel sources here, every field."""
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests", "checkers"))
    import fuzz_gpu
    fuzz_gpu.run_case(emu, oracle, 95343)


def test_emu_barcode_with_more_than_64_molecules(emu, oracle):
    """K8's optimizer and probability sums ask fastScore about 64 sinks at a time (k_rfa.h): with more than 64 molecules in a barcode the source of a turn is staged
    once for all groups, and a group in which no read of the source has an alignment is skipped (a word of group bits per read, set where the molecule x read table
    is filled).  One barcode of 300 pairs on 150 contigs of 3 kb, a third of them copies of each other with a few substitutions (reads with alignments in several
    molecules: moves to score, sums to add up): well over 64 molecules.  Every field against the oracle."""
    from lariat_amd import synth
    rng = np.random.default_rng(8)
    base = [rng.integers(0, 4, size=3000).astype(np.uint8) for _ in range(100)]
    contigs = list(base)
    for k in range(50):   # copies of the first 25 contigs, 1 % off
        c = base[k % 25].copy()
        m = rng.random(3000) < 0.01
        c[m] = (c[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        contigs.append(c)
    names = ["u%03d" % i for i in range(len(contigs))]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=1, pairs_per_barcode=300, seed=4, junk_frac=0.02, mol_min=140, mol_max=140)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b, threads=8)
    n_mol = len(set(int(x) for x in ref.rid[(ref.active != 0) & (ref.rid >= 0)]))   # (a contig with an active alignment holds at least one molecule)
    assert n_mol >= 80, n_mol
    res = emu.index_from_arrays(oidx.arrays()).context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, ref, inference=True)


def test_emu_long_noisy_reads(emu, oracle):
    """240-bp reads with substitutions and indels: the 128/256-column classes of the lane-per-read extension, deferred reads,
    gapped global alignments"""
    from lariat_amd import synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=2, pairs_per_barcode=40, seed=31, len1=240, len2=236, sub_lo=0.005, sub_hi=0.03, indel_rate=0.003, junk_frac=0.02)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, oidx.align_barcodes(b), inference=True)


def test_emu_k1_sweep_filter_and_text_shortcuts(emu, oracle):
    """K1 with and without its exact shortcuts (sweep filter, collapsed sweeps, calls by text): the same intervals / seeds / chains as
    the oracle both ways; with them fewer bwt_extend calls are accounted for, without them (LH_F_NO_SWEEP_FILTER) the count is the
    reference's"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=2, pairs=25, junk=0.05, seed=41)
    rs.seq[np.arange(7, len(rs.seq), 211)] = 4   # ambiguous bases: the filter's window logic, runs that stop at a non-base
    b = helpers.batch_of(rs)
    want = oidx.stage_dump(b)
    want_ext = oidx.align_barcodes(b, oracle.opts(run_inference=0)).counters["n_ext"]
    ctx = idx.context(rs.n_pairs)
    seen = {}
    NOF = capi.LH_F_NO_SWEEP_FILTER
    for flags in (0, NOF):
        helpers.assert_same_dump(ctx.stage_dump(b, emu.opts(flags=flags)), want, helpers.DUMP_FRONT)
        seen[flags] = ctx.align_barcodes(b, emu.opts(run_inference=0, flags=flags)).counters["n_ext"]
    assert seen[NOF] == want_ext
    assert seen[0] < want_ext


def test_emu_many_candidate_pairs_go_rng_ring(emu, oracle):
    """pairs with 20 x 20 equally good (alignment, mate) combinations: tagBestAlignments draws 400 jitter values per pair from
    Go's generator (lariat.go:1499) — past the 273 draws the device serves without materialising the generator state"""
    names, contigs, unit, spacer = helpers.exact_repeat_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.repeat_unit_reads(contigs, unit, spacer, n_pairs=6)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b)
    nf = np.add.reduceat(ref.in_filtered.astype(np.int64), ref.cand_off[:-1])
    assert (nf[0::2] * nf[1::2]).max() > 273
    res = idx.context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, ref, inference=True)


def test_emu_chance_match_second_chains(emu, oracle):
    """one-seed chains from chance matches (the hg38-scale case): extended with the full band over query sides of 64 columns and
    more — k_extend_lane's live-interval window — with every region, CIGAR and pick equal to the oracle's"""
    names, contigs, rs = helpers.chance_match_genome_and_reads()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    b = helpers.batch_of(rs)
    od = oidx.stage_dump(b)
    n_chains = np.diff(od.chain_off)
    assert (n_chains >= 2).sum() >= rs.n_pairs   # the construction works: most reads 1 carry a second chain
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), od, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b), inference=True)


def test_emu_kmer_tree_levels(emu, oracle):
    """K1 reads the result of a bwt_extend that yields a short match from the k-mer tree table: the same intervals, seeds, chains and
    alignments with the table at several depths (incl. ambiguous bases and reads shorter than a level) and without it"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = helpers.small_reads(names, contigs, n_barcodes=2, pairs=25, junk=0.05, seed=43)
    rs.seq[np.arange(5, len(rs.seq), 97)] = 4
    b = helpers.batch_of(rs)
    want = oidx.stage_dump(b)
    ref = oidx.align_barcodes(b)
    used = {}
    for levels in (-1, 2, 7, 11):
        idx = emu.index_from_arrays(oidx.arrays(), ktree_levels=levels)
        ctx = idx.context(rs.n_pairs)
        helpers.assert_same_dump(ctx.stage_dump(b), want, helpers.DUMP_FRONT)
        res = ctx.align_barcodes(b)
        helpers.assert_same_result(res, ref, inference=True)
        assert res.counters["n_ext"] < ref.counters["n_ext"]
        used[levels] = sum(res.counters["n_ktree_p%d" % k] for k in (1, 2, 3))
    assert used[-1] == 0 < used[2] < used[7] < used[11]


def test_emu_two_lanes(emu, oracle):
    """lh_context_opts.lanes = 2: the batch is cut at a barcode boundary, the two parts run side by side from two host threads on
    two pipelines, the merged result equals the oracle's for the whole batch (and a one-barcode batch stays on the first lane)"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=5, pairs=24, junk=0.05, seed=17)
    rfa = np.array([1, 1, 0, 1, 1], dtype=np.uint8)
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, bc_do_rfa=rfa)
    ref = oidx.align_barcodes(b)
    for lanes in (3, 2):
        ctx = idx.context(rs.n_pairs, lanes=lanes)
        res = ctx.align_barcodes(b)
        helpers.assert_same_result(res, ref, inference=True)
    for k in ("n_sa", "glob_cells", "n_rescue", "rescue_cells"):
        assert res.counters[k] == ref.counters[k], k
    # slots: two different batches resident, selected in turn
    one = rs.slice_barcodes(1, 2)
    b1 = helpers.batch_of(one)
    ctx.upload_slot(3, b1)
    ctx.upload_slot(1, b)
    ctx.select(3)
    ctx.align_resident(emu.opts())
    helpers.assert_same_result(ctx.download(), oidx.align_barcodes(b1), inference=True)
    ctx.select(1)
    ctx.align_resident(emu.opts())
    helpers.assert_same_result(ctx.download(), ref, inference=True)
    with pytest.raises(capi.LhError):
        ctx.stage_dump()


def _check_get_seq(lib, oracle):
    """GoBwaReference.GetSeq (gobwa.go:50-80) through lh_get_seq: the reference's own vector (gobwa_test.go:13-28 reads PhiX
    210..280), then windows that cross contig ends (bns_fetch_seq clips to the contig of the midpoint) and reversed ones"""
    idx = lib.index_load(helpers.PHIX); oidx = oracle.index_load(helpers.PHIX)
    assert idx.get_seq(0, 210, 280, False).decode() == helpers.PHIX_READ_A
    assert idx.get_seq(0, 100, 160, True) == oidx.get_seq(0, 100, 160, True)
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rng = np.random.default_rng(5)
    for rid in range(len(names)):
        n = len(contigs[rid])
        for _ in range(40):
            s = int(rng.integers(-50, n)); e = s + int(rng.integers(0, 300))
            if s < 0 and rng.random() < 0.5: s = 0
            for rev in (False, True):
                if s < 0: continue   # GetSeq is never called with a negative start (lariat.go:640-647 clamps)
                assert idx.get_seq(rid, s, e, rev) == oidx.get_seq(rid, s, e, rev), (rid, s, e, rev)


def test_emu_get_seq(emu, oracle):
    _check_get_seq(emu, oracle)


def test_emu_large_barcode_position_sort(emu, oracle):
    """a barcode with more filtered candidates than K8 stages in LDS: the position sort of inferMolecules (lariat.go:1545-1547, Go's
    unstable sort.Sort) runs as the wave-wide restatement (lh_sort.h wave_gosort); duplicated pairs give equal positions, whose
    order is the algorithm's"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=1, pairs=460, junk=0.02, seed=23)
    for p in range(5, rs.n_pairs, 7):   # PCR duplicates: the pair before, again
        for k in (0, 1):
            d, s = 2 * p + k, 2 * (p - 1) + k
            n = min(int(rs.seq_off[d + 1] - rs.seq_off[d]), int(rs.seq_off[s + 1] - rs.seq_off[s]))
            rs.seq[rs.seq_off[d]:rs.seq_off[d] + n] = rs.seq[rs.seq_off[s]:rs.seq_off[s] + n]
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b)
    assert int((ref.in_filtered != 0).sum()) > 800   # past LH_RFA_SORT_LDS
    helpers.assert_same_result(idx.context(rs.n_pairs).align_barcodes(b), ref, inference=True)



def test_emu_alt_contigs(emu, oracle, tmp_path):
    """ALT contigs (bwa_idx_load BWA_IDX_ALL restores <prefix>.alt: gobwa.go:130): mem_chain_flt does not let an ALT chain shadow a
    primary one.  Same stage dumps and results as the oracle with the flags on; the flags change which chains are kept; and the .alt
    file is read the way bns_restore reads it (first token per line, '@' lines skipped, unknown names ignored, a last line without a
    newline not seen) by both loaders, and written back by lh_index_save."""
    names, contigs, rs = helpers.alt_genome_and_reads()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    b = helpers.batch_of(rs)
    plain = oidx.stage_dump(b)
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), plain, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    flags = [0, 0, 1]
    oidx.set_alt(flags); idx.set_alt(flags)
    assert idx.alt() == flags and oidx.alt() == flags
    want = oidx.stage_dump(b)
    helpers.assert_same_dump(ctx.stage_dump(b), want, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b), inference=True)
    # the ALT branch matters here: reads whose kept chains differ with and without the flags
    changed = 0
    for r in range(2 * rs.n_pairs):
        a = want.chain_kept[want.chain_off[r]:want.chain_off[r + 1]]
        p = plain.chain_kept[plain.chain_off[r]:plain.chain_off[r + 1]]
        changed += len(a) != len(p) or not np.array_equal(a, p)
    assert changed >= 5, changed
    # the file
    prefix = str(tmp_path / "alt.fa")
    idx.save(prefix)
    assert open(prefix + ".alt").read() == "chrP1_alt1\n"
    open(prefix + ".alt", "w").write("@SQ\tSN:chrP2\nchrP1_alt1\t0\tchrP1\t40001\nnot_a_contig\t1\nchrP2")
    li, lo = emu.index_load(prefix), oracle.index_load(prefix)
    assert li.alt() == flags and lo.alt() == flags


def test_emu_second_chance_paths(oracle):
    """no read may fail a batch for a slot limit (BWA's vectors grow, gobwa.go:244-260): a build with FOUR regular interval slots per read
    and two extension rounds sends nearly every read through the second chances — the three seeding passes again into the big slab, its
    own sort, the seeds read from there; K4's leftover reads in the wave kernel — with the same stage dumps and results as the oracle"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu"), "small"])
    small = capi.Library(os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu_small.so"))
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = small.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=3, pairs=30, junk=0.05, seed=29)
    b = helpers.batch_of(rs)
    od = oidx.stage_dump(b)
    assert (np.diff(od.intv_off) > 4).mean() > 0.5   # most reads have more intervals than the regular slots of this build
    ctx = idx.context(rs.n_pairs)
    helpers.assert_same_dump(ctx.stage_dump(b), od, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b), inference=True)
    # ... and when more reads need a slot of that slab than it has, the slab grows and the reads left out run again (three slots for ~100 reads)
    tiny = idx.context(rs.n_pairs, big_slots=3)
    helpers.assert_same_dump(tiny.stage_dump(b), od, helpers.DUMP_FRONT + helpers.DUMP_REGS)
    helpers.assert_same_result(tiny.align_barcodes(b), oidx.align_barcodes(b), inference=True)


def test_emu_overlapped_download_and_staged_upload(emu, oracle):
    """one context, three batches: batch k's result is collected (lh_result_download_end) after batch k + 1 has been aligned, and batch
    k + 1 was staged (lh_batch_stage_slot) while batch k was the selected one — each result equals the oracle's for its own batch"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    sets = [helpers.small_reads(names, contigs, n_barcodes=2, pairs=16 + 6 * k, junk=0.05, seed=50 + k) for k in range(3)]
    batches = [helpers.batch_of(rs) for rs in sets]
    ctx = idx.context(max(rs.n_pairs for rs in sets))
    opts = emu.opts()
    ctx.upload_slot(1, batches[0])
    got = []
    for k in range(3):
        ctx.select(1 + k % 2)
        if k + 1 < 3:
            ctx.stage_slot(1 + (k + 1) % 2, batches[k + 1])   # (a second host thread would do this while the align below runs)
        ctx.align_resident(opts)
        if k:
            got.append(ctx.download_end())                     # batch k - 1: its copies ran under this batch's kernels
        ctx.download_begin()
    got.append(ctx.download_end())
    with pytest.raises(capi.LhError):
        ctx.download_end()                                     # nothing in flight
    for k in range(3):
        helpers.assert_same_result(got[k], oidx.align_barcodes(batches[k]), inference=True)
    with pytest.raises(capi.LhError):
        ctx.stage_slot(1 + 2 % 2, batches[0])                  # the selected slot cannot be staged into


def test_emu_k7_four_mismatches_and_shifted_diagonals(emu, oracle):
    """K7 settles a candidate with equal spans without the DP when no path with gaps can reach the diagonal (k_aln.h: bounds + the diagonals
    shifted by one and two bases).  Reads with 3-5 % substitutions and no indels, half of them from microsatellites and tandem repeats — where
    shifted diagonals DO match and the DP has to run — against the oracle, every field (the GPU suite has the same on the 80-kb low-complexity genome)"""
    from lariat_amd import synth
    rng = np.random.default_rng(17)
    rnd = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)
    tandem = lambda unit, n: np.tile(np.asarray(unit, dtype=np.uint8), n)
    names, contigs = ["chrL"], [np.concatenate([rnd(2500), tandem([0], 300), rnd(700), tandem([1, 0], 150), rnd(700), tandem(rnd(37), 30), rnd(700), tandem(rnd(5), 80), rnd(2500)])]
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs = synth.make_reads(contigs, names, n_barcodes=1, pairs_per_barcode=16, seed=5, sub_lo=0.03, sub_hi=0.05, indel_rate=0.0, mol_min=2, mol_max=3)
    b = helpers.batch_of(rs)
    res = idx.context(rs.n_pairs).align_barcodes(b)
    ores = oidx.align_barcodes(b, threads=8)
    helpers.assert_same_result(res, ores, inference=True)
    ok = ores.rid >= 0
    assert int(((ores.indels[ok] == 0) & (ores.mismatches[ok] >= 4)).sum()) > 8   # the proof's (and the DP's) cases are there


@pytest.mark.parametrize("g,ins_first", [(1, False), (1, True), (2, False), (2, True)])
def test_emu_k7_gapped_path_beats_a_four_mismatch_diagonal(emu, oracle, g, ins_first):
    """the case the shifted-diagonal check of K7 exists for: a read that lost g bases before a stretch of A's with two interruptions and gained g
    behind it (or the other way round) has FOUR mismatches on the diagonal (loss 20) — and a path with one deletion and one insertion (cost
    12 + 2g + g pairs) that has none.  Equal spans, so bwa_gen_cigar2's shortcut does not apply; the bounds of k_aln.h do not exclude the path;
    the diagonal shifted by g matches through all four mismatches: the DP must run (CIGAR with I and D).  A kernel that skipped the check
    would report 150M (checked by disabling it: this test fails)."""
    names, contigs, b, p = helpers.k7_shift_case(g, ins_first)
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    res = idx.context(8).align_barcodes(b, emu.opts(run_inference=0))
    ores = oidx.align_barcodes(b, oracle.opts(run_inference=0))
    helpers.assert_same_result(res, ores, inference=False)
    c = list(ores.cands_of_read(0))[0]
    assert int(ores.indels[c]) == 2 and int(ores.mismatches[c]) == 0 and int(ores.pos[c]) == p - 66


@pytest.mark.parametrize("g,ins_first", [(3, True), (6, False), (7, True), (7, False), (8, True), (8, False), (9, False), (12, True)])
def test_emu_k7_narrow_band_and_its_proof(emu, oracle, g, ins_first):
    """K7's four-candidates-per-wave kernel runs a band of 7 whatever band mem_reg2aln asks for (here 20 and more) and keeps the result only when
    no path outside that band can reach its score (k_aln.h, k_aln_grp).  A read that runs g columns off its diagonal for 60 bases: up to
    g = 7 the path is inside the band and the bound holds (a path further out loses at least 8 pairs and two longer gaps); from g = 8 on the
    band of 7 cannot hold the path, the bound says so, and the candidate takes the wave kernel with the full band.  Every field against the
    oracle's ksw_global2 with BWA's own band."""
    names, contigs, b, p = helpers.k7_band_case(g, ins_first)
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    res = idx.context(8).align_barcodes(b, emu.opts(run_inference=0))
    ores = oidx.align_barcodes(b, oracle.opts(run_inference=0))
    helpers.assert_same_result(res, ores, inference=False)
    assert res.counters["glob_cells"] == ores.counters["glob_cells"]
    c = list(ores.cands_of_read(0))[0]
    assert int(ores.indels[c]) == 2 and int(ores.pos[c]) == p


def _k7_differing_reads(res, ref):
    """reads whose candidates differ in place, score, counts or CIGAR"""
    bad = []
    for r in range(ref.n_reads):
        ca, cb = list(res.cands_of_read(r)), list(ref.cands_of_read(r))
        same = len(ca) == len(cb)
        for x, y in zip(ca, cb):
            same = same and all(int(getattr(res, f)[x]) == int(getattr(ref, f)[y]) for f in ("pos", "aend", "mismatches", "indels", "nm", "score")) \
                and np.array_equal(res.cigar_of(x), ref.cigar_of(y))
        if not same:
            bad.append(r)
    return bad


def test_emu_k7_second_look_and_its_two_checks(emu, oracle):
    """K7's second look (k_aln.h, aln_deep_check; r06): a candidate with equal spans and five or six mismatches is settled without the DP when no excursion from
    the diagonal gains — two gap runs of up to g0 - 1 bases by an exact running maximum over the shifted diagonals (C1), three or more by the largest saving B of
    any stretch of those diagonals (C2).
    (1) Reads with 3.5-5 % substitutions on unique sequence: every field equal to the oracle's, and the second look settles most of what the first listed.
    (2) The adversarial batch (helpers.k7_deep_batch): five to seven mismatches beside an excursion of two to four gap runs through low-complexity sequence,
    where the DP's answer is a CIGAR with gaps about as often as not.  The product equals the oracle on every read; the SAME kernels built with C1 cut down to
    shifts of one and two bases (`weak1`) or with C2's B taken as zero (`weak2`) report plain diagonals where the reference has gaps — each check carries cases
    of its own."""
    from lariat_amd import synth
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=2, pairs_per_barcode=60, seed=31, sub_lo=0.035, sub_hi=0.05, indel_rate=0.0, mol_min=3, mol_max=5)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b, threads=8)
    res = emu.index_from_arrays(oidx.arrays()).context(rs.n_pairs).align_barcodes(b)
    helpers.assert_same_result(res, ref, inference=True)
    ok = ref.rid >= 0
    n57 = int(((ref.indels[ok] == 0) & (ref.mismatches[ok] >= 5) & (ref.mismatches[ok] <= 6)).sum())
    listed, left = res.counters["n_glob_listed"], res.counters["n_glob_exec"]
    assert n57 > 40 and listed <= ref.counters["n_glob_exec"] and listed - left > 0.7 * n57, (n57, listed, left, ref.counters["n_glob_exec"])

    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu"), "weak1", "weak2"])
    names, contigs, reads = helpers.k7_deep_batch(1, 192)
    oidx = oracle.index_build_naive(names, contigs)
    b = capi.Batch(reads, [0, len(reads) // 2])
    ref = oidx.align_barcodes(b, oracle.opts(run_inference=0), threads=8)
    ok = ref.rid >= 0
    assert int((ref.indels[ok] > 0).sum()) > 60 and int(((ref.indels[ok] == 0) & (ref.mismatches[ok] >= 5)).sum()) > 60   # both answers are there
    res = emu.index_from_arrays(oidx.arrays()).context(len(reads) // 2).align_barcodes(b, emu.opts(run_inference=0))
    helpers.assert_same_result(res, ref, inference=False)
    assert res.counters["n_glob_exec"] < res.counters["n_glob_listed"]
    for weak in ("weak1", "weak2"):
        wl = capi.Library(os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu_%s.so" % weak))
        wres = wl.index_from_arrays(oidx.arrays()).context(len(reads) // 2).align_barcodes(b, wl.opts(run_inference=0))
        bad = _k7_differing_reads(wres, ref)
        assert len(bad) >= 2, (weak, bad)
        for r in bad:   # what goes wrong is what the check is there to stop: a diagonal reported where the reference's alignment has gaps
            c = list(ref.cands_of_read(r))
            assert any(int(ref.indels[x]) > 0 for x in c) and all(int(wres.indels[x]) <= int(ref.indels[y]) for x, y in zip(wres.cands_of_read(r), c)), (weak, r)


@pytest.mark.parametrize("build", ["default", "small"])
def test_emu_repeat_families(emu, oracle, build):
    """reads on the copies of repeat families (helpers.repeat_family_case: 6-7 candidates per read, 800-3,000 rescue attempts per batch), the paths r04 built
    for them: K4's long queue (a read's chains as units, the per-read check, the calls a lane cannot hold by a wave), K7's lane per candidate and
    its four-per-wave DP in a proved band of 7, K6's replay on an unsorted LDS list (two kernel instances by list length), K3's chain keys.
    The default build hands the wave-chained reads of so small a batch to the wave kernel after round 0 (fewer than 1,024 jobs); the `small`
    build runs the rounds to the end, keeps 64 / 128 regions in the replay's lists (most pairs take the second instance, some neither).  Results, rescue counts and rescue cells against the oracle."""
    lib = emu
    if build == "small":
        subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu"), "small"])
        lib = capi.Library(os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu_small.so"))
    for seed, nb in ((13, 12), (17, 16)):
        names, contigs, rs = helpers.repeat_family_case(seed, nb)
        oidx = oracle.index_build_naive(names, contigs)
        idx = lib.index_from_arrays(oidx.arrays())
        b = helpers.batch_of(rs)
        ref = oidx.align_barcodes(b, threads=8)
        assert ref.n_cand / ref.n_reads > 4 and ref.counters["n_rescue"] > 500
        res = idx.context(rs.n_pairs).align_barcodes(b)
        helpers.assert_same_result(res, ref, inference=True)
        for k in ("glob_cells", "n_rescue", "rescue_cells"):
            assert res.counters[k] == ref.counters[k], k
        # (r06) the forward passes ran the rows k_resc_cert left them (k_rescue3.h): the same results from a fraction of the reference's cells; with
        # LH_F_RESCUE_FULL every window is run whole, as before
        assert 0 < res.counters["rescue_cells_exec"] < 0.6 * res.counters["rescue_cells"], (res.counters["rescue_cells_exec"], res.counters["rescue_cells"])
        full = idx.context(rs.n_pairs).align_barcodes(b, lib.opts(flags=capi.LH_F_RESCUE_FULL))
        helpers.assert_same_result(full, ref, inference=True)
        assert full.counters["rescue_cells"] == ref.counters["rescue_cells"]
        assert 0.9 * full.counters["rescue_cells"] < full.counters["rescue_cells_exec"] < 1.1 * full.counters["rescue_cells"]   # (jobs dropped by the replay ran; attempts without a job ran in place)


def test_emu_k3_cluster_kernel_and_wave_kernel_agree(emu, oracle):
    """K3's two ways for a read with many seeds — k_chain_cl (seeds sorted by position, one lane per cluster, mem_chain_flt's greedy scan in rounds) and
    k_chain (one wave-wide look-up per seed, the scan as written), chosen by LH_F_CHAIN_WAVE — give the chains the oracle gives: seeds, weights, kept marks,
    order (stage dump), and the same final result."""
    names, contigs, rs = helpers.repeat_family_case(41, 8)
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    b = helpers.batch_of(rs)
    want = oidx.stage_dump(b)
    ref = oidx.align_barcodes(b, threads=8)
    for flags in (0, capi.LH_F_CHAIN_WAVE, capi.LH_F_P2_TASKS):   # (... and K1's pass 2 with a read's re-seeding calls dealt out to several lanes)
        ctx = idx.context(rs.n_pairs)
        helpers.assert_same_dump(ctx.stage_dump(b, emu.opts(flags=flags)), want, helpers.DUMP_FRONT + helpers.DUMP_REGS)
        helpers.assert_same_result(ctx.align_barcodes(b, emu.opts(flags=flags)), ref, inference=True)


def test_emu_context_moves_between_regimes(emu, oracle):
    """which way K4 takes the wave-chained reads rests on the PREVIOUS batch of the context (lh_host_stage2.inc: the long queue when that batch had many of
    them, the wave-per-read kernel at once when it had few) — a choice of path, never of result.  One context, a repeat-family batch, then reads on unique
    sequence, then repeat families twice more: the third batch takes the wave kernel with hundreds of wave-chained reads, the fourth the long queue again;
    each result equals the oracle's."""
    names, contigs, rs_rep = helpers.repeat_family_case(13, 10)
    oidx = oracle.index_build_naive(names, contigs)
    idx = emu.index_from_arrays(oidx.arrays())
    rs_uni = helpers.small_reads(names, contigs, n_barcodes=10, pairs=30, junk=0.03, seed=77)
    _, _, rs_rep2 = helpers.repeat_family_case(13, 10)
    ctx = idx.context(max(rs_rep.n_pairs, rs_uni.n_pairs))
    for rs in (rs_rep, rs_uni, rs_rep2, rs_rep):
        b = helpers.batch_of(rs)
        helpers.assert_same_result(ctx.align_barcodes(b), oidx.align_barcodes(b, threads=8), inference=True)

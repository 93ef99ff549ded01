"""GPU parity of the candidate-generation front end (K1 SMEM, K2 SA lookup, K3 chaining) against the oracle,
through the C-ABI.  Runs on the MI355X box only."""
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


def test_phix_kat_front(lib, oracle):
    idx = lib.index_load(helpers.PHIX)
    oidx = oracle.index_load(helpers.PHIX)
    b = capi.Batch([capi.sequence_convert(helpers.PHIX_READ_A), capi.sequence_convert(helpers.PHIX_READ_B)], [0, 1])
    d = idx.context(8).stage_dump(b)
    helpers.assert_same_dump(d, oidx.stage_dump(b), helpers.DUMP_FRONT)
    assert list(d.seed_rbeg[:2]) == [210, 210]


def test_front_synthetic(lib, oracle):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays())
    rs = helpers.small_reads(names, contigs, n_barcodes=20, pairs=100, junk=0.03)
    b = helpers.batch_of(rs)
    rs.seq[np.arange(5, len(rs.seq), 397)] = 4   # ambiguous bases in some reads
    b = helpers.batch_of(rs)
    ctx = idx.context(rs.n_pairs)
    want = oidx.stage_dump(b)
    helpers.assert_same_dump(ctx.stage_dump(b), want, helpers.DUMP_FRONT)


@pytest.mark.gpu
def test_occ_superblocks_and_sparse_sa(lib, oracle):
    """u32 occurrence counts relative to super-block bases (forced small here; real use: > 2^31 symbols) and bwt_sa walking
    the re-laid-out BWT at the .sa file's own sampling interval"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = lib.index_from_arrays(oidx.arrays(), sb_shift=14, sa_intv=32)
    assert idx.sa_interval == 32
    rs = helpers.small_reads(names, contigs, n_barcodes=8, pairs=60, junk=0.05, seed=11)
    b = helpers.batch_of(rs)
    helpers.assert_same_dump(idx.context(rs.n_pairs).stage_dump(b), oidx.stage_dump(b), helpers.DUMP_FRONT)


def test_get_seq(lib, oracle):
    """lh_get_seq == GoBwaReference.GetSeq (gobwa.go:50-80): reference vector + windows over contig ends, both strands"""
    from test_emu_front import _check_get_seq
    _check_get_seq(lib, oracle)


def test_fuzz_regressions(lib, oracle):
    """cases the differential fuzzer (tests/checkers/fuzz_gpu.py) once found: seed 72337 holds a 235-base read with a candidate of 89
    mismatch loci — past the 64 slots per candidate, continued in the batch's pool (k_aln.h DCand::mm_x*); seed 95343 (r06) a barcode in which
    soft-clipped alignments compete inside a molecule: markBest's integer pair score had the clip term doubled, one read's best alignment in a
    molecule changed and with it a probability sum by 3e-8 (k_rfa.h, MEnt::s2)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests", "checkers"))
    import fuzz_gpu
    for seed in (72337, 95343):
        fuzz_gpu.run_case(lib, oracle, seed)


def test_gosort_serial_and_wave(lib, oracle):
    """K8's two restatements of Go's sort.Sort (equal keys end up where Go leaves them) against the oracle's, on the GPU"""
    import test_sort
    test_sort.check(lib, oracle)


def test_gosort_split(lib, oracle):
    """Go's sort of a list too long for LDS, split between the whole wave (long ranges) and per-range sorts with the depth left — on the GPU"""
    import test_sort
    test_sort.check_split(lib, oracle)


def test_bitonic_network(lib):
    """K8's sorting network for contig lists without equal positions: one LDS block, block by block, all in memory — on the GPU"""
    import test_sort
    test_sort.check_bitonic(lib)


def test_introsort_one_lane_and_wave(lib, oracle):
    """K5 / K6's two restatements of klib's ks_introsort (equal keys end up where klib leaves them) against the oracle's, on the GPU"""
    import test_sort
    test_sort.check_introsort(lib, oracle)

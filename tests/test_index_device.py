"""lh_index_build_device (FM-index construction in HBM, k_index_build.h): the index it leaves resident, exported back to the
layout of `bwa index`'s files, must be byte-identical to the reference's PhiX fixture (go/src/test/inputs/phix/PhiX.fa.*) and to
the oracle's builder on genomes with repeats; CPU runs use the kernel sources under the emulator, `-m gpu` the product."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from lariat_amd import capi

EMU = os.environ.get("LH_EMU_LIB") or os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")   # (LH_EMU_LIB: e.g. an AddressSanitizer build, tests/hipemu/Makefile asan)


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    return capi.Library(EMU)


def pac_of(contigs):
    g = np.concatenate(contigs).astype(np.uint8)
    l_pac = len(g)
    pad = np.zeros((l_pac // 4 + 1) * 4, dtype=np.uint8)
    pad[:l_pac] = g
    q = pad.reshape(-1, 4)
    pac = (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(c) for c in contigs])])
    return pac, l_pac, offs


def check_against(lib, oidx, names, contigs, **index_opts):
    pac, l_pac, offs = pac_of(contigs)
    ctg = [(names[i], len(contigs[i]), int(offs[i])) for i in range(len(names))]
    idx = lib.index_build_device(pac, l_pac, ctg, **index_opts)
    want = oidx.arrays()
    if index_opts.get("sa_intv", 0) in (0, 1):   # LCP array and k-mer tree: from the builder's sort keys == from the text of an uploaded index
        up = lib.index_from_arrays(want, **{k: v for k, v in index_opts.items() if k != "build_chunk_log2"})
        assert idx.digest() == up.digest() and idx.digest()[2] > 0 and idx.digest()[0] > 0
    got = idx.export(sa_intv=32)
    assert got["primary"] == want["primary"]
    assert list(got["L2"]) == list(want["L2"])
    assert np.array_equal(got["bwt"], want["bwt"])
    assert np.array_equal(got["sa"], want["sa"])
    return idx


def phix(oracle):
    names, seqs = helpers.read_fasta(helpers.PHIX)
    contigs = [capi.sequence_convert(s) for s in seqs]
    return names, contigs, oracle.index_load(helpers.PHIX)


def test_emu_device_build_reproduces_phix_fixture(emu, oracle, tmp_path):
    names, contigs, oidx = phix(oracle)
    for opts in ({}, {"build_chunk_log2": 9}, {"build_chunk_log2": 6, "sa_intv": 4}):   # one chunk; many chunks; sparse resident SA
        idx = check_against(emu, oidx, names, contigs, **opts)
    # the files lh_index_save writes are the fixture's, byte for byte
    idx = check_against(emu, oidx, names, contigs)
    prefix = str(tmp_path / "PhiX.fa")
    idx.save(prefix)
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        assert open(prefix + ext, "rb").read() == open(helpers.PHIX + ext, "rb").read(), ext


def test_emu_device_build_with_repeats_and_aligns(emu, oracle):
    """exact repeats (ties past the 32-base sort key, groups of 20+), several contigs, and the index is then USED"""
    names, contigs, unit, spacer = helpers.exact_repeat_genome(copies=12, unit=620, spacer=150)
    contigs = contigs + [np.zeros(700, dtype=np.uint8), contigs[0][500:3000][::-1].copy()]   # poly-A; a reversed copy
    names = names + ["polyA", "rev"]
    oidx = oracle.index_build_naive(names, contigs)
    idx = check_against(emu, oidx, names, contigs, build_chunk_log2=11)
    rs = helpers.repeat_unit_reads(contigs, unit, spacer, n_pairs=4, len1=100, len2=100)
    b = helpers.batch_of(rs)
    helpers.assert_same_result(idx.context(rs.n_pairs).align_barcodes(b), oidx.align_barcodes(b), inference=True)


def tandem_genome():
    """what a real reference holds and an iid one does not: a long tandem repeat (5-mer x 4 kb: tie groups of ~800 suffixes that share up
    to 4 kb), a long exact duplication, a palindrome (a segment followed by its reverse complement) and a poly-A tract"""
    rng = np.random.default_rng(21)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)

    def rnd(n):
        return rng.integers(0, 4, size=n).astype(np.uint8)

    seg = rnd(3000)
    pal = rnd(1500)
    c0 = np.concatenate([rnd(2000), np.tile(rnd(5), 800), rnd(1000), seg, rnd(500), pal, comp[pal[::-1]], rnd(700)])
    c1 = np.concatenate([rnd(1500), seg, rnd(300), np.zeros(900, dtype=np.uint8), rnd(1200)])
    return ["t0", "t1"], [c0, c1]


def test_emu_device_build_with_large_tie_groups(emu, oracle):
    """tie groups far beyond what one lane sorts (k_ib_sort_big: a block per group, merge sort through the text): the same suffix array,
    BWT, LCP array and k-mer tree as the oracle's builder and the text-derived tables, with one chunk and with many"""
    names, contigs = tandem_genome()
    oidx = oracle.index_build_naive(names, contigs)
    check_against(emu, oidx, names, contigs)
    check_against(emu, oidx, names, contigs, build_chunk_log2=12)


def test_emu_export_of_a_loaded_index_is_the_file(emu):
    idx = emu.index_load(helpers.PHIX, sb_shift=9)
    got = idx.export(sa_intv=32)
    raw = np.fromfile(helpers.PHIX + ".bwt", dtype=np.uint8)
    assert np.array_equal(got["bwt"], raw[40:].view(np.uint32))
    sa = np.fromfile(helpers.PHIX + ".sa", dtype=np.uint8)[56:].view(np.uint64)
    assert np.array_equal(got["sa"][1:], sa)


@pytest.mark.gpu
def test_gpu_device_build_phix_and_repeats(oracle):
    lib = capi.load_library()
    names, contigs, oidx = phix(oracle)
    for opts in ({}, {"build_chunk_log2": 8}):
        check_against(lib, oidx, names, contigs, **opts)
    names, contigs = tandem_genome()
    check_against(lib, oracle.index_build_naive(names, contigs), names, contigs)
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    idx = check_against(lib, oidx, names, contigs, build_chunk_log2=16)
    rs = helpers.small_reads(names, contigs, n_barcodes=8, pairs=50, junk=0.03, seed=9)
    b = helpers.batch_of(rs)
    helpers.assert_same_result(idx.context(rs.n_pairs).align_barcodes(b), oidx.align_barcodes(b, threads=8), inference=True)


def test_emu_index_self_check_and_synth(emu, oracle):
    """lh_synth_genome / lh_synth_reads feed the device builder; lh_diag_index_check passes on a sound index"""
    l_pac = 40000
    pac = emu.synth_genome(l_pac, seed=5)
    assert np.array_equal(pac, emu.synth_genome(l_pac, seed=5, threads=3))
    ctg = [("c0", 25000, 0), ("c1", 15000, 25000)]
    idx = emu.index_build_device(pac, l_pac, ctg, build_chunk_log2=12)
    checked, bad_order, bad_lf = idx.check(stride=7)
    assert checked > 10000 and bad_order == 0 and bad_lf == 0
    r = emu.synth_reads(pac, l_pac, ctg, seed=9, n_barcodes=3, pairs_per_barcode=20, indel_rate=0.002, junk_frac=0.05)
    r2 = emu.synth_reads(pac, l_pac, ctg, seed=9, n_barcodes=3, pairs_per_barcode=20, indel_rate=0.002, junk_frac=0.05, threads=2)
    assert np.array_equal(r["seq"], r2["seq"]) and np.array_equal(r["seq_off"], r2["seq_off"])
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    res = idx.context(r["n_pairs"]).align_barcodes(b)
    oidx = oracle.index_from_arrays(idx.export(), pac)
    helpers.assert_same_result(res, oidx.align_barcodes(b), inference=True)
    # most reads land where they were drawn from
    act = res.active_idx[0::2]
    ok = (res.rid[act] == r["truth_rid"]) & (np.abs(res.pos[act] - r["truth_pos1"]) < 20)
    assert ok.mean() > 0.8


def test_emu_device_build_of_a_reference_with_holes(emu, oracle, tmp_path):
    """N runs: lh_reference_pack (bwa's replacement rule) -> device build -> lh_index_set_holes -> lh_index_save gives the
    files the oracle's builder writes, .amb and .ann included; reads across the replaced stretch align as the oracle's do"""
    rng = np.random.default_rng(12)
    g = "".join("ACGT"[v] for v in rng.integers(0, 4, size=40000))
    c1 = (g[:9000] + "N" * 40 + g[9040:25000]).encode()
    c2 = ("NN" + g[25002:33000] + "KKKK" + g[33004:]).encode()
    contigs = [np.frombuffer(c1, dtype=np.uint8), np.frombuffer(c2, dtype=np.uint8)]
    pac, l_pac, n_ambs, holes = emu.reference_pack(contigs)
    ctg = [("c1", len(c1), 0), ("c2", len(c2), len(c1))]
    idx = emu.index_build_device(pac, l_pac, ctg, build_chunk_log2=13)
    idx.set_holes(holes)
    oidx = oracle.index_build_naive(["c1", "c2"], contigs)
    prefix = str(tmp_path / "h.fa")
    idx.save(prefix)
    for k, ext in enumerate((".bwt", ".sa", ".pac", ".ann", ".amb")):
        assert open(prefix + ext, "rb").read() == oidx.image(k), ext
    again = emu.index_load(prefix)   # holes survive a load / save round trip
    again.save(str(tmp_path / "h2.fa"))
    assert open(str(tmp_path / "h2.fa.amb")).read() == open(prefix + ".amb").read()
    r = emu.synth_reads(pac, l_pac, ctg, seed=3, n_barcodes=2, pairs_per_barcode=30)
    b = capi.Batch.from_arrays(r["seq"], r["seq_off"], r["bc_pair_off"], r["name_seed"])
    helpers.assert_same_result(idx.context(r["n_pairs"]).align_barcodes(b), oidx.align_barcodes(b), inference=True)

"""The inference half a second time: tests/lariat_py.py follows lariat.go / split.go / ordered_*map.go /
optimizer.go with Go's own data structures (objects, ordered maps with swap-delete, 8 * M GenerateMove calls) and must arrive at what
the oracle's dense-table restatement (oracle/lariat_oracle.cpp) — and with it the device's k_rfa — computes from the same candidate lists:
active picks, molecule ids, sum_move_probability_change, MAPQ (exactly), duplicates, split reads.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import helpers
import lariat_py
from lariat_amd import capi, synth


def test_go_sort_matches_the_oracles_restatement(oracle):
    """Go 1.9 sort.Sort written out twice (lariat_py.go_sort; oracle/gosort_impl.h): the same permutation, ties and all"""
    rng = np.random.default_rng(1)
    first, keys = [0], []
    for n in list(range(0, 60)) + [100, 257, 1000, 5000]:
        for spread in (3, 50, 1 << 40):
            keys += list(rng.integers(0, spread, size=n))
            first.append(len(keys))
        keys += sorted(rng.integers(0, 9, size=n))            # presorted with ties, reversed, all equal
        first.append(len(keys))
        keys += sorted(rng.integers(0, 9, size=n))[::-1]
        first.append(len(keys))
        keys += [7] * n
        first.append(len(keys))
    first = np.array(first, dtype=np.int32)
    k = np.array(keys, dtype=np.int64)
    perm = np.concatenate([np.arange(first[i + 1] - first[i], dtype=np.int32) for i in range(len(first) - 1)])
    kk = k.copy()
    oracle.L.lo_gosort.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    oracle.L.lo_gosort.restype = None
    oracle.L.lo_gosort(len(first) - 1, first.ctypes.data_as(C.POINTER(C.c_int32)), kk.ctypes.data_as(C.POINTER(C.c_int64)), perm.ctypes.data_as(C.POINTER(C.c_int32)))
    for s in range(len(first) - 1):
        a, b = int(first[s]), int(first[s + 1])
        items = [(int(k[a + i]), i) for i in range(b - a)]
        lariat_py.go_sort(len(items), lambda i, j: items[i][0] < items[j][0], lambda i, j: lariat_py._swap(items, i, j))
        assert [it[1] for it in items] == list(perm[a:b]), (s, b - a)


def derive(ref, contig_names, batch_arrays, improper=-4.0, cen=None):
    """run tests/lariat_py.py over every barcode of the oracle result `ref` and compare every inference field"""
    seq_off, bco, name_seed, do_rfa = batch_arrays
    read_lens = np.diff(seq_off)
    n_bc = len(bco) - 1
    checked = {"barcodes": 0, "cands": 0, "moved": 0, "mols": 0}
    for bc in range(n_bc):
        p0, p1 = int(bco[bc]), int(bco[bc + 1])
        r0, r1 = 2 * p0, 2 * p1
        alignments, full = lariat_py.barcode_from_result(ref, contig_names, read_lens, r0, r1)
        seeds = [int(np.int64(np.uint64(name_seed[p]))) for p in range(p0, p1)]
        inf = lariat_py.Inference(improper=improper, centromeres=cen)
        rfa = bool(do_rfa[bc]) if do_rfa is not None else True
        mols = inf.run_barcode(alignments, full, seeds, rfa)
        c0, c1 = int(ref.cand_off[r0]), int(ref.cand_off[r1])
        for arr in full:
            for a in arr:
                c = a.idx
                where = (bc, a.read_id, c - c0)
                assert bool(ref.active[c]) == a.active, ("active",) + where
                assert bool(ref.bwa_pick[c]) == a.bwa_pick, ("bwa_pick",) + where
                assert bool(ref.is_proper[c]) == a.is_proper, ("is_proper",) + where
                assert bool(ref.active_molecule[c]) == a.active_molecule, ("active_molecule",) + where
                assert bool(ref.duplicate[c]) == a.duplicate, ("duplicate",) + where
                assert int(ref.molecule_id[c]) == a.molecule_id, ("molecule_id",) + where
                assert int(ref.mapq[c]) == a.mapq, ("mapq", int(ref.mapq[c]), a.mapq) + where
                assert int(ref.mate_idx[c]) == (a.mate_alignment.idx if a.mate_alignment is not None else -1), ("mate",) + where
                for name, got in (("molecule_difference", a.molecule_difference), ("molecule_confidence", a.molecule_confidence),
                                  ("sum_move_probability_change", a.sum_move_probability_change)):
                    want = float(getattr(ref, name)[c])
                    assert (want != want and got != got) or abs(want - got) <= 1e-9 * max(1.0, abs(want)), (name, want, got) + where
                if a.bwa_pick and not a.active:
                    checked["moved"] += 1
        for k, arr in enumerate(full):
            r = r0 + k
            act = [a for a in arr if a.active]
            assert len(act) == 1 and int(ref.active_idx[r]) == act[0].idx
            a = act[0]
            assert int(ref.second_best_idx[r]) == (a.md_second_best.idx if a.md_second_best is not None else -1), ("second_best", bc, k)
            assert abs(float(ref.second_best_score[r]) - a.md_second_best_score) < 1e-9 and abs(float(ref.as_score[r]) - a.md_score) < 1e-9, ("scores", bc, k)
            assert int(ref.split_idx[r]) == (a.secondary.idx if a.secondary is not None else -1), ("split", bc, k)
            if a.secondary is not None:
                s = a.secondary
                assert int(ref.split_mapq[r]) == s.mapq
                assert abs(float(ref.split_second_best[r]) - s.split_md_second_best_score) < 1e-9 and abs(float(ref.split_score[r]) - s.split_md_score) < 1e-9
        checked["barcodes"] += 1
        checked["cands"] += c1 - c0
        checked["mols"] += len(mols) if mols else 0
    return checked


def arrays_of(b, rs_seq_off, rs_bco, rs_seed, do_rfa=None):
    return (np.asarray(rs_seq_off), np.asarray(rs_bco), np.asarray(rs_seed), do_rfa)


def test_small_barcodes_with_repeats(oracle):
    """240 barcodes of 8-40 pairs on a genome with duplications and repeat families: junk reads (placeholders), unpaired best hits,
    barcodes too small for RFA, barcodes that skip it"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    tot = {"barcodes": 0, "cands": 0, "moved": 0, "mols": 0}
    for k, (nb, pairs) in enumerate([(120, 12), (80, 25), (40, 40)]):
        rs = synth.make_reads(contigs, names, n_barcodes=nb, pairs_per_barcode=pairs, seed=100 + k, junk_frac=0.05, mol_min=1, mol_max=4)
        rfa = (np.arange(nb) % 7 != 3).astype(np.uint8)
        b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, bc_do_rfa=rfa)
        ref = oidx.align_barcodes(b, threads=8)
        got = derive(ref, names, arrays_of(b, rs.seq_off, rs.bc_pair_off, rs.name_seed, rfa))
        for key in tot:
            tot[key] += got[key]
    print("second derivation:", tot)
    assert tot["barcodes"] == 240 and tot["moved"] >= 5 and tot["mols"] > 400


def test_repeat_rich_barcodes(oracle):
    """exact 20-copy repeats (400 equally good pair combinations per read: Go's generator decides) and reads whose candidates spread over
    many molecules: the optimizer's moves and the molecule-move sums at work"""
    names, contigs, unit, spacer = helpers.exact_repeat_genome(copies=20)
    oidx = oracle.index_build_naive(names, contigs)
    tot = 0
    for seed in range(4):
        rs = helpers.repeat_unit_reads(contigs, unit, spacer, n_pairs=14, seed=seed + 4, copy=3 + seed)
        b = helpers.batch_of(rs)
        ref = oidx.align_barcodes(b, threads=8)
        got = derive(ref, names, arrays_of(b, rs.seq_off, rs.bc_pair_off, rs.name_seed))
        tot += got["cands"]
    assert tot > 4 * 14 * 2 * 15


def test_repeat_families(oracle):
    """configs[4]'s regime in small: families of diverged copies (workload.plant_family), reads drawn on the copies — tens of candidates
    per read in dozens of molecules per barcode, many optimizer moves"""
    from lariat_amd import workload
    rng = np.random.default_rng(5)
    g = rng.choice(4, size=600000, p=[0.295, 0.205, 0.205, 0.295]).astype(np.uint8)
    q = g.reshape(-1, 4)
    pac = np.concatenate([(q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
    ctg = [("c0", 400000, 0), ("c1", 200000, 400000)]
    fam = workload.plant_family(pac, ctg, rng, 3000, 12, 0.002, 0.012, indel_per_base=1 / 1500.0)
    fam2 = workload.plant_family(pac, ctg, rng, 300, 30, 0.02, 0.08, indel_per_base=1 / 300.0)
    names = [c[0] for c in ctg]
    contigs = [workload._unpack(pac, off, ln) for _, ln, off in ctg]
    oidx = oracle.index_build_naive(names, contigs)
    win = workload.windows_on(ctg, fam, 1000) + workload.windows_on(ctg, fam2, 850)
    wnames = ["w%d" % i for i in range(len(win))]
    wcontigs = [workload._unpack(pac, off // 4 * 4, (ln + off % 4 + 3) // 4 * 4)[off % 4: off % 4 + ln] for _, ln, off in win]
    # reads drawn from the windows (as separate little contigs), aligned against the whole genome
    rs = synth.make_reads(wcontigs, wnames, n_barcodes=10, pairs_per_barcode=40, seed=9, mol_min=2, mol_max=4)
    b = helpers.batch_of(rs)
    ref = oidx.align_barcodes(b, threads=8)
    assert ref.n_cand > 5 * ref.n_reads
    got = derive(ref, names, arrays_of(b, rs.seq_off, rs.bc_pair_off, rs.name_seed))
    print("repeat families:", got, "%.1f candidates per read" % (ref.n_cand / ref.n_reads))
    assert got["moved"] >= 20


def test_centromeres_and_penalty(oracle):
    """a centromere region zeroes MAPQ and split MAPQ inside it; a non-dyadic improper-pair penalty"""
    names, contigs = helpers.small_genome(seed=3)
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=12, pairs_per_barcode=30, seed=77, junk_frac=0.03, mol_min=2, mol_max=5)
    cs = np.array([50000, -1, 10000], dtype=np.int64)
    ce = np.array([150000, -1, 60000], dtype=np.int64)
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, cen_start=cs, cen_end=ce)
    ref = oidx.align_barcodes(b, oracle.opts(improper_pair_penalty=-3.7), threads=8)
    cen = {names[i]: (int(cs[i]), int(ce[i])) for i in range(3) if cs[i] >= 0}
    got = derive(ref, names, arrays_of(b, rs.seq_off, rs.bc_pair_off, rs.name_seed), improper=-3.7, cen=cen)
    assert got["barcodes"] == 12

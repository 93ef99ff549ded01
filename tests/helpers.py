"""shared helpers for the test-suite"""
import gzip
import os

import numpy as np

from lariat_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PHIX = os.path.join(GOLDEN, "phix", "PhiX.fa")

# go/src/test/gobwa_test.go:18,23
PHIX_READ_A = "TCAAAAACTGACGCGTTGGATGAGGAGAAGTGGCTTAATATGCTTGGCACGTTCGTCAAGGACTGGTTTA"
PHIX_READ_B = "TATGACCAGTGTTTCCAGTCCGTTCAGTTGTTGCAGTGGAATAGTCAGGTTAAATTTAATGTGACCGCTT"


def read_fasta(path):
    names, seqs = [], []
    for line in open(path):
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append([])
        else:
            seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]


def read_fastq9(path, trim):
    """minimal 9-line reader (reader.go:91-174 semantics that matter here: trim the first `trim` bases of read 1)"""
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    lines = [l.rstrip(b"\n") for l in op(path, "rb").read().split(b"\n")]
    recs = []
    for i in range(0, len(lines) - 8, 9):
        name, r1, q1, r2, q2, bc = lines[i:i + 6]
        recs.append(dict(name=name[1:].decode(), r1=r1[trim:].decode(), r2=r2.decode(), bc=bc.decode()))
    return recs


def small_genome(seed=1):
    names = ["chrA", "chrB", "chrC"]
    contigs = synth.make_genome([300000, 200000, 100000], seed=seed, n_dup=6, dup_len=5000, dup_identity=0.99, n_rep_family=2, rep_copies=20)
    return names, contigs


def small_reads(names, contigs, n_barcodes=8, pairs=40, seed=5, junk=0.02):
    return synth.make_reads(contigs, names, n_barcodes=n_barcodes, pairs_per_barcode=pairs, seed=seed, junk_frac=junk)


def batch_of(rs):
    return capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed)


DUMP_FRONT = ["intv_off", "intv", "seed_off", "seed_rbeg", "seed_qbeg", "seed_len", "seed_rid", "chain_off", "chain_nseeds", "chain_rid", "chain_w",
              "chain_kept", "chain_pos"]
DUMP_REGS = ["reg_off", "reg_rb", "reg_re", "reg_qb", "reg_qe", "reg_rid", "reg_score", "reg_truesc", "reg_w", "reg_seedcov", "reg_seedlen0", "reg_csub",
             "reg_secondary"]


def assert_same_dump(d, od, fields):
    for f in fields:
        a, b = getattr(d, f), getattr(od, f)
        assert a.shape == b.shape, (f, a.shape, b.shape)
        if not (a == b).all():
            bad = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
            raise AssertionError("stage dump field %s differs at %s (got %s want %s)" % (f, bad[:5], a[bad[:3]], b[bad[:3]]))


INT_FIELDS = ["cand_off", "rid", "pos", "aend", "rb", "re", "reversed", "score", "qb", "qe", "nm", "matches", "mismatches", "indels", "soft_clipped",
              "soft_clipped_length", "in_filtered", "cigar_off", "cigar", "mm_off", "mm_ref_loc", "mm_read_loc"]
INF_FIELDS = ["active", "is_proper", "bwa_pick", "active_molecule", "duplicate", "molecule_id", "mate_idx", "active_idx", "second_best_idx", "split_idx",
              "split_mapq"]
F64_FIELDS = ["log_alignment_probability", "molecule_difference", "molecule_confidence", "sum_move_probability_change", "second_best_score", "as_score",
              "split_second_best", "split_score"]


def assert_same_result(r, ref, inference=True, mapq_tol=1, rel=1e-9):
    """bit-exact for integer/index fields; MAPQ within +-1; float scores within 1e-9 relative (BASELINE.json north_star)"""
    for f in INT_FIELDS + (INF_FIELDS if inference else []):
        a, b = getattr(r, f), getattr(ref, f)
        assert a.shape == b.shape, (f, a.shape, b.shape)
        if not (a == b).all():
            bad = np.nonzero(a != b)[0]
            raise AssertionError("result field %s differs at %s (got %s want %s)" % (f, bad[:5], a[bad[:5]], b[bad[:5]]))
    fl = ["log_alignment_probability"] + (F64_FIELDS[1:] if inference else [])
    for f in fl:
        a, b = getattr(r, f), getattr(ref, f)
        assert a.shape == b.shape, f
        ok = np.isclose(a, b, rtol=rel, atol=1e-12, equal_nan=True)
        assert ok.all(), (f, np.nonzero(~ok)[0][:5], a[~ok][:5], b[~ok][:5])
    if inference:
        d = np.abs(r.mapq.astype(np.int64) - ref.mapq.astype(np.int64))
        assert (d <= mapq_tol).all(), ("mapq", np.nonzero(d > mapq_tol)[0][:5])


def exact_repeat_genome(copies=20, unit=900, spacer=400, seed=3):
    """one contig holding `copies` EXACT copies of a unit between random spacers + a unique tail: reads from a unit have
    `copies` equally good candidates on both mates (tagBestAlignments then draws copies^2 jitter values per pair)"""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 4, size=unit).astype(np.uint8)
    parts = []
    for _ in range(copies):
        parts += [rng.integers(0, 4, size=spacer).astype(np.uint8), u]
    parts.append(rng.integers(0, 4, size=20000).astype(np.uint8))
    return ["chrR"], [np.concatenate(parts)], unit, spacer


def repeat_unit_reads(contigs, unit, spacer, n_pairs, seed=4, copy=3, len1=120, len2=120):
    """FR pairs that lie inside copy `copy` of exact_repeat_genome's unit; one barcode"""
    rng = np.random.default_rng(seed)
    g = contigs[0]
    base = copy * (unit + spacer) + spacer
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    reads, names = [], []
    for i in range(n_pairs):
        ins = int(rng.integers(300, 500))
        s = base + int(rng.integers(0, unit - ins))
        r1 = g[s:s + len1].copy()
        r2 = comp[g[s + ins - len2:s + ins][::-1]]
        if rng.random() < 0.5:
            r1[int(rng.integers(len1))] ^= 1
        reads += [r1, r2]
        names.append("rep:%d" % i)
    rs = synth.ReadSet()
    lens = np.array([len(x) for x in reads], dtype=np.int64)
    rs.seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rs.seq = np.concatenate(reads)
    rs.bc_pair_off = np.array([0, n_pairs], dtype=np.int32)
    rs.names = names
    rs.name_seed = synth._name_seeds(names)
    return rs


def chance_match_genome_and_reads(n_pairs=24, seed=8, plant=21, rlen=150):
    """a random contig in which the MIDDLE `plant` bases of every read 1 (and of some reads 2) also occur at an unrelated place:
    re-seeding (bwt_smem1 from the middle of a long SMEM with min_intv 2) then yields a second, one-seed chain per read that
    mem_chain_flt keeps as the first shadowed chain and mem_chain2aln extends with the full band over long query sides — at
    human-genome scale chance matches do that to every third read"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=260000).astype(np.uint8)
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    loci = []
    for i in range(n_pairs):
        p = 5000 + i * 4000 + int(rng.integers(0, 1000))
        ins = int(rng.integers(320, 480))
        loci.append((p, ins))
        mid = p + rlen // 2 - plant // 2
        g[150000 + i * 300: 150000 + i * 300 + plant] = g[mid: mid + plant]
        if i % 3 == 0:   # read 2's middle as well (reverse strand)
            mid2 = p + ins - rlen // 2 - plant // 2
            g[200000 + i * 300: 200000 + i * 300 + plant] = g[mid2: mid2 + plant]
    reads, names = [], []
    for i, (p, ins) in enumerate(loci):
        r1 = g[p:p + rlen].copy()
        r2 = comp[g[p + ins - rlen:p + ins][::-1]]
        if i % 2:
            r1[int(rng.integers(5, 40))] ^= 2      # a mismatch off the middle: the planted match still covers the re-seeding point
        if i % 4 == 0:
            r2[int(rng.integers(100, 140))] ^= 1
        reads += [r1, r2]
        names.append("chance:%d" % i)
    rs = synth.ReadSet()
    lens = np.array([len(x) for x in reads], dtype=np.int64)
    rs.seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rs.seq = np.concatenate(reads)
    rs.bc_pair_off = np.array([0, n_pairs // 2, n_pairs], dtype=np.int32)
    rs.names = names
    rs.name_seed = synth._name_seeds(names)
    return ["chrC"], [g], rs


def alt_genome_and_reads(seed=12, n_pairs=60):
    """two primary contigs + one ALT contig (configs[4]: "hg38 + ALT/decoy"): the ALT is a copy of 30 kb of chrP1 in which novel 90-base
    insertions alternate with 60 kept bases over a few kb (plus point mutations elsewhere), and chrP2 holds an exact copy of the same
    30 kb.  Reads from the ALT's inserted stretch have a heavy chain on the ALT and two light ones on the primaries (the kept 60 bases):
    with is_alt the light primary chains are not compared with the ALT chain in mem_chain_flt."""
    rng = np.random.default_rng(seed)
    p1 = rng.integers(0, 4, size=120000).astype(np.uint8)
    p2 = rng.integers(0, 4, size=90000).astype(np.uint8)
    src = p1[40000:70000].copy()
    p2[20000:50000] = src
    parts, pos = [], 0
    while pos < len(src):
        if 8000 <= pos < 16000:
            parts += [src[pos:pos + 60], rng.integers(0, 4, size=90).astype(np.uint8)]
            pos += 60
        else:
            seg = src[pos:pos + 500].copy()
            m = rng.random(len(seg)) < 0.004
            seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
            parts.append(seg)
            pos += 500
    alt = np.concatenate(parts)
    names = ["chrP1", "chrP2", "chrP1_alt1"]
    contigs = [p1, p2, alt]
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    lo = 8000 * 150 // 60   # the inserted stretch in ALT coordinates
    reads, rnames = [], []
    for i in range(n_pairs):
        ins = int(rng.integers(300, 480))
        s = lo + int(rng.integers(0, 8000 * 150 // 60 - ins)) if i % 4 else int(rng.integers(0, len(alt) - ins))
        r1 = alt[s:s + 143].copy()
        r2 = comp[alt[s + ins - 150:s + ins][::-1]]
        if i % 3 == 0:
            r1[int(rng.integers(143))] ^= 1
        reads += [r1, r2]
        rnames.append("alt:%d" % i)
    rs = synth.ReadSet()
    lens = np.array([len(x) for x in reads], dtype=np.int64)
    rs.seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rs.seq = np.concatenate(reads)
    rs.bc_pair_off = np.array([0, n_pairs // 2, n_pairs], dtype=np.int32)
    rs.names = rnames
    rs.name_seed = synth._name_seeds(rnames)
    return names, contigs, rs


def low_complexity_genome(seed=9):
    """poly-A tracts, di- and penta-nucleotide microsatellites and a 200-copy tandem repeat of a 37-base unit between random flanks: reads from
    them have intervals with thousands of occurrences, hundreds of chains per read and long runs of equal candidates"""
    rng = np.random.default_rng(seed)

    def rnd(n):
        return rng.integers(0, 4, size=n).astype(np.uint8)

    def tandem(unit, n):
        return np.tile(np.asarray(unit, dtype=np.uint8), n)

    parts = [rnd(20000), tandem([0], 5000), rnd(3000), tandem([1, 0], 1500), rnd(3000), tandem(rnd(37), 200), rnd(3000), tandem(rnd(5), 800), rnd(3000),
             tandem(rnd(2), 400), rnd(500), tandem([3], 600), rnd(20000)]
    return ["chrL"], [np.concatenate(parts)]


def k7_shift_case(g, ins_first):
    """a read pair whose read 1 has four mismatches on the diagonal of its alignment and NONE on a path with one insertion and one deletion of g
    bases: a stretch of A's with two interruptions, shifted by g inside its own span (K7's shifted-diagonal check, k_aln.h).  Returns
    (names, contigs, batch, position of the stretch)."""
    from lariat_amd import capi
    rng = np.random.default_rng(23)
    rnd = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)
    seg = np.array([0] * 6 + [2] + [0] * 6 + [3] + [0] * 6, dtype=np.uint8)                 # AAAAAA G AAAAAA T AAAAAA
    left, right = rnd(4000), rnd(4000)
    left[-1] = 1; right[0] = 1                                                                # (no A next to the stretch: the indels cannot slide out of it)
    contig = np.concatenate([left, seg, right])
    p = len(left)
    body = np.concatenate([[0] * g, seg[:-g]]) if ins_first else np.concatenate([seg[g:], [0] * g])   # the stretch shifted by g inside its own span
    r1 = np.concatenate([contig[p - 66:p], body, contig[p + len(seg):p + len(seg) + 150 - 66 - len(seg)]]).astype(np.uint8)
    assert len(r1) == 150 and int((r1 != contig[p - 66:p + 84]).sum()) == 4
    mate = contig[p + 250:p + 400]
    r2 = (3 - mate[::-1]).astype(np.uint8)
    return ["chrK"], [contig], capi.Batch([r1, r2], [0, 1]), p


def k7_band_case(g, ins_first):
    """a read pair whose read 1 runs g columns off its main diagonal for 60 bases: g inserted bases after base 40 and g deleted reference bases
    60 bases later (or the other way round) — equal spans, a path of two gaps.  K7's four-per-wave kernel (k_aln_grp, k_aln.h) runs a band
    of 7 and proves it sufficient; from g = 8 on the path lies outside that band and the proof must fail.  Returns (names, contigs, batch, p)."""
    from lariat_amd import capi
    rng = np.random.default_rng(1000 + 2 * g + int(ins_first))
    contig = rng.integers(0, 4, size=8400).astype(np.uint8)
    p = 4000
    R = contig[p:p + 150]
    X = rng.integers(0, 4, size=g).astype(np.uint8)
    if ins_first:
        X[0] = (R[40] + 1) & 3; X[-1] = (R[39] + 2) & 3
        r1 = np.concatenate([R[:40], X, R[40:100], R[100 + g:]])
    else:
        r1 = np.concatenate([R[:40], R[40 + g:110], X, R[110:]])
    r1 = r1.astype(np.uint8)
    assert len(r1) == 150
    mate = contig[p + 250:p + 400]
    r2 = (3 - mate[::-1]).astype(np.uint8)
    return ["chrB"], [contig], capi.Batch([r1, r2], [0, 1]), p


def _k7_low_complexity(rng, n):
    kind = int(rng.integers(0, 4))
    if kind == 0:      # a homopolymer with interruptions
        s = np.full(n, int(rng.integers(0, 4)), dtype=np.uint8)
    elif kind == 1:    # a short tandem repeat
        s = np.tile(rng.integers(0, 4, size=int(rng.integers(2, 8))).astype(np.uint8), n)[:n]
    elif kind == 2:    # two homopolymers
        s = np.concatenate([np.full(n // 2, int(rng.integers(0, 4))), np.full(n - n // 2, int(rng.integers(0, 4)))]).astype(np.uint8)
    else:              # a repeat of a longer unit with a homopolymer inside
        u = rng.integers(0, 4, size=int(rng.integers(8, 14))).astype(np.uint8); u[2:6] = u[2]
        s = np.tile(u, n)[:n]
    for _ in range(int(rng.integers(0, 5))):
        k = int(rng.integers(0, n)); s[k] = (s[k] + int(rng.integers(1, 4))) & 3
    return s


def _k7_deep_case(rng):
    """one locus and a read 1 on it whose middle is an excursion of two or three gap runs through low-complexity sequence (equal spans), plus substitutions"""
    lc = _k7_low_complexity(rng, int(rng.integers(24, 70)))
    left, right = rng.integers(0, 4, size=400).astype(np.uint8), rng.integers(0, 4, size=700).astype(np.uint8)
    locus = np.concatenate([left, lc, right])
    p = len(left)
    ref = locus[p - 50:p + 100].copy()
    a = 50 + int(rng.integers(0, max(1, len(lc) // 3)))
    runs = int(rng.choice([2, 2, 3, 3, 4]))
    total = int(rng.integers(1, 6)) if runs == 2 else int(rng.integers(2, 5))
    read = list(ref)
    # deletions of reference bases first or insertions first; the later runs give the bases back
    first_del = bool(rng.integers(0, 2))
    cuts = sorted(int(x) for x in rng.choice(np.arange(a, min(a + len(lc), 140)), size=runs, replace=False))
    if runs == 2:
        sizes_a, sizes_b = [total], [total]
        pos_a, pos_b = cuts[:1], cuts[1:]
    elif runs == 3:
        k = int(rng.integers(1, total))
        if rng.integers(0, 2): sizes_a, sizes_b, pos_a, pos_b = [k, total - k], [total], cuts[:2], cuts[2:]
        else: sizes_a, sizes_b, pos_a, pos_b = [total], [k, total - k], cuts[:1], cuts[1:]
    else:
        k, k2 = int(rng.integers(1, total)), int(rng.integers(1, total))
        sizes_a, sizes_b, pos_a, pos_b = [k, total - k], [k2, total - k2], cuts[:2], cuts[2:]
    ops = [(q, "d" if first_del else "i", n) for q, n in zip(pos_a, sizes_a)] + [(q, "i" if first_del else "d", n) for q, n in zip(pos_b, sizes_b)]
    out, at = [], 0
    for q, kind, n in sorted(ops):
        out += list(ref[at:q]); at = q
        if kind == "d": at = min(len(ref), q + n)
        else: out += [int(ref[max(0, q - 1 - j)]) if rng.random() < 0.7 else int(rng.integers(0, 4)) for j in range(n)]
    out += list(ref[at:])
    read = np.array(out[:150], dtype=np.uint8)
    if len(read) < 150:
        read = np.concatenate([read, locus[p + 100:p + 100 + 150 - len(read)]])
    for _ in range(int(rng.integers(0, 4))):
        k = int(rng.integers(5, 145)); read[k] = (read[k] + int(rng.integers(1, 4))) & 3
    mate = locus[p + 300:p + 450]
    return locus, read, (3 - mate[::-1]).astype(np.uint8), int((read != ref).sum())


def k7_deep_batch(seed, n):
    """n read pairs for K7's second look (k_aln.h, aln_deep_check): read 1 of each has five to seven mismatches on its diagonal and an excursion of two to four gap runs
    through low-complexity sequence beside it; which of the two wins is the DP's to say — or the proof's.  Returns (names, contigs, reads)."""
    rng = np.random.default_rng(seed)
    loci, reads, at = [], [], 0
    while len(loci) < n:
        locus, r1, r2, mm = _k7_deep_case(rng)
        if not 5 <= mm <= 7:
            continue
        loci.append(locus); reads += [r1, r2]
    contig = np.concatenate(loci)
    return ["chrD"], [contig], reads


def repeat_family_case(seed, n_barcodes, pairs=30):
    """a 600-kb genome with a family of 12 copies x 3 kb (0.2-1.2 % off the consensus, indels), one of 30 copies x 300 bp (2-8 %) and five tandem copies
    of 1.2 kb; every read pair drawn on and around the copies: tens of chains, regions and rescue attempts per read — the regime of BASELINE
    configs[4] in miniature.  Returns (names, contigs, read set)."""
    from lariat_amd import synth, workload
    rng = np.random.default_rng(seed)
    g = rng.choice(4, size=600000, p=[0.295, 0.205, 0.205, 0.295]).astype(np.uint8)
    q = g.reshape(-1, 4)
    pac = np.concatenate([(q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
    ctg = [("c0", 400000, 0), ("c1", 200000, 400000)]
    fam = workload.plant_family(pac, ctg, rng, 3000, 12, 0.002, 0.012, indel_per_base=1 / 1500.0)
    fam2 = workload.plant_family(pac, ctg, rng, 300, 30, 0.02, 0.08, indel_per_base=1 / 300.0)
    tand = workload._unpack(pac, 100000, 1200)
    for k in range(1, 5):   # tandem copies: rescue windows that hold two alignments of the mate, regions next to each other
        seg = tand.copy(); m = rng.random(1200) < 0.01; seg[m] = (seg[m] + 1) & 3
        workload._repack(pac, 100000 + 1200 * k, seg)
    names = [c[0] for c in ctg]
    contigs = [workload._unpack(pac, off, ln) for _, ln, off in ctg]
    win = workload.windows_on(ctg, fam, 1000) + workload.windows_on(ctg, fam2, 850) + [("t", 9000, 98000)] * 6
    wnames = ["w%d" % i for i in range(len(win))]
    wcontigs = [workload._unpack(pac, off // 4 * 4, (ln + off % 4 + 3) // 4 * 4)[off % 4: off % 4 + ln] for _, ln, off in win]
    rs = synth.make_reads(wcontigs, wnames, n_barcodes=n_barcodes, pairs_per_barcode=pairs, seed=seed + 9, mol_min=2, mol_max=4, ins_mean=470, ins_sd=120, ins_max=900,
                          junk_frac=0.02)
    return names, contigs, rs

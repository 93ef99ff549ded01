"""The synthetic hg38-scale workload of BASELINE.json's configs: no human reference exists offline, so the
genome is generated (seed 20261002, iid ACGT with GC 0.41) with hg38's contig structure — 24 contigs in the proportions of
chr1..22, X, Y — and the reads by the linked-read model of lh_synth_reads."""

# hg38 primary assembly chromosome lengths, bases
HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
                114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
GENOME_SEED = 20261002
READS_SEED = 20261003


def hg38_like_contigs(total_bases):
    """[(name, len, offset)]: hg38's 24 primary contigs scaled so that they sum to about `total_bases` (each a multiple of 4)"""
    tot = float(sum(HG38_LENGTHS))
    out, off = [], 0
    for name, ln in zip(HG38_NAMES, HG38_LENGTHS):
        n = max(4000, int(ln / tot * total_bases) // 4 * 4)
        out.append((name, n, off))
        off += n
    return out


def plant_segdups(pac, contigs, n_dup, dup_len, identity, seed, n_families=0, family_len=300, family_copies=0):
    """configs[4]-like repeat structure on a packed genome (4 bases per byte, as lh_synth_genome writes it), in place: `n_dup`
    segmental duplications of `dup_len` bases at `identity` (substitutions only; sources and copies are multiples of 4 so that the copy
    is a byte copy), plus `n_families` interspersed-repeat families of `family_copies` copies each.  Returns [(src, dst, len)] in
    global coordinates (the windows that reads biased to repeats are drawn from, see repeat_windows)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = np.array([c[1] for c in contigs], dtype=np.int64)
    offs = np.array([c[2] for c in contigs], dtype=np.int64)
    w = lens / lens.sum()

    def place(n_bases):
        k = int(rng.choice(len(contigs), p=w))
        p = int(rng.integers(0, max(1, (lens[k] - n_bases) // 4))) * 4
        return int(offs[k] + p)

    def copy(src, dst, n, ident):
        seg = pac[src >> 2:(src + n) >> 2].copy()
        n_mut = int(round((1.0 - ident) * n))
        if n_mut:
            at = rng.integers(0, n, size=n_mut)
            delta = rng.integers(1, 4, size=n_mut).astype(np.uint8)            # a different base: add 1..3 mod 4 to the 2-bit code
            sh = ((~at & 3) << 1).astype(np.uint8)
            for a, d, s_ in zip(at, delta, sh):                                  # (few thousand per copy)
                b = int(seg[a >> 2])
                code = ((b >> int(s_)) + int(d)) & 3
                seg[a >> 2] = (b & ~(3 << int(s_)) & 0xff) | (code << int(s_))
        pac[dst >> 2:(dst + n) >> 2] = seg

    out = []
    dup_len = dup_len // 4 * 4
    for _ in range(n_dup):
        src, dst = place(dup_len), place(dup_len)
        if abs(src - dst) < dup_len:
            continue
        copy(src, dst, dup_len, identity)
        out.append((src, dst, dup_len))
    family_len = family_len // 4 * 4
    for _ in range(n_families):
        src = place(family_len)
        for _ in range(family_copies):
            dst = place(family_len)
            if abs(src - dst) >= family_len:
                copy(src, dst, family_len, float(rng.uniform(0.9, 1.0)))
                out.append((src, dst, family_len))
    return out


def repeat_windows(contigs, dups, flank=60000):
    """[(name, len, offset)] of windows around both copies of every planted repeat, clipped to their contigs: passed to lh_synth_reads as
    its contig list, the molecules (and so the reads) fall on and around the repeats"""
    out = []
    for k, (src, dst, n) in enumerate(dups):
        for tag, p in (("s", src), ("d", dst)):
            for name, ln, off in contigs:
                if off <= p < off + ln:
                    b = max(off, p - flank) // 4 * 4
                    e = min(off + ln, p + n + flank)
                    out.append(("%s%d" % (tag, k), int(e - b), int(b)))
                    break
    return out


def add_alt_contigs(pac, contigs, n_alt, alt_len, identity, seed):
    """configs[4]'s "ALT contigs = mutated copies of 1 Mb regions"  on a packed genome: appends `n_alt` contigs named
    <source contig>_alt<k>, each a copy of `alt_len` bases of a primary contig with substitutions at 1 - identity.  Returns (pac, l_pac,
    contigs, is_alt flags, [(src, dst, len)] in global coordinates)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = np.array([c[1] for c in contigs], dtype=np.int64)
    offs = np.array([c[2] for c in contigs], dtype=np.int64)
    alt_len = alt_len // 4 * 4
    l_pac = int(offs[-1] + lens[-1])
    assert l_pac % 4 == 0
    out = list(contigs)
    flags = [0] * len(contigs)
    body = [pac[:l_pac >> 2]]
    pairs = []
    for k in range(n_alt):
        ci = int(rng.choice(len(contigs), p=lens / lens.sum()))
        src = int(offs[ci] + int(rng.integers(0, max(1, (lens[ci] - alt_len) // 4))) * 4)
        seg = pac[src >> 2:(src + alt_len) >> 2].copy()
        n_mut = int(round((1.0 - identity) * alt_len))
        at = rng.integers(0, alt_len, size=n_mut)
        delta = rng.integers(1, 4, size=n_mut)
        sh = (~at & 3) << 1
        codes = (seg[at >> 2] >> sh) & 3                     # (a position drawn twice keeps one of its draws: still a substitution)
        seg[at >> 2] = (seg[at >> 2] & ~(3 << sh).astype(np.uint8)) | ((((codes + delta) & 3) << sh).astype(np.uint8))
        body.append(seg)
        out.append(("%s_alt%d" % (contigs[ci][0], k + 1), alt_len, l_pac))
        flags.append(1)
        pairs.append((src, l_pac, alt_len))
        l_pac += alt_len
    body.append(np.zeros(1, dtype=np.uint8))
    return np.concatenate(body), l_pac, out, flags, pairs


# ---- BASELINE.json configs[4]: "hg38 + ALT/decoy contigs, pairs biased to segdup/repeat loci (high candidate multiplicity per read)" ----
# segmental duplications (200 copies x 20 kb at 98-99.5 % identity), 300-bp / 6-kb interspersed repeat families, ALT contigs =
# mutated copies of 1 Mb regions.  What multiplies the candidates of a read (mem_align1_core regions, gobwa.go:244,253), the rescue attempts of
# a pair (gobwa.go:286-325: up to 50 hits of the mate each) and the molecules of a barcode (lariat.go:1135-1167: 8 M^2 fastScore calls) is a
# FAMILY: n copies that all resemble each other — n - 1 further candidates for a read from any of them — not n independent pairs of copies.

def _unpack(pac, p, n):
    """nt4 codes of the n bases from p (both multiples of 4) of a packed genome"""
    import numpy as np
    b = pac[p >> 2:(p + n) >> 2]
    return np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], axis=1).reshape(-1).astype(np.uint8)


def _repack(pac, p, seg):
    q = seg.reshape(-1, 4)
    pac[p >> 2:(p + len(seg)) >> 2] = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]


def plant_family(pac, contigs, rng, unit_len, n_copies, div_lo, div_hi, indel_per_base=0.0, revcomp=True, consensus=None):
    """one repeat family on a packed genome, in place: a consensus of `unit_len` bases (a stretch of the genome unless given) and `n_copies`
    copies of it at random places of the primary contigs, each with its own divergence d ~ U(div_lo, div_hi) from the consensus (substitutions at
    rate d, indels of 1-8 bases at `indel_per_base`, the copy cut or padded back to unit_len), on either strand.  Two copies differ from each
    other by about the sum of their divergences.  Returns [(global position, unit_len)]."""
    import numpy as np
    lens = np.array([c[1] for c in contigs], dtype=np.int64)
    offs = np.array([c[2] for c in contigs], dtype=np.int64)
    w = lens / lens.sum()
    unit_len = unit_len // 4 * 4

    def place():
        k = int(rng.choice(len(contigs), p=w))
        return int(offs[k] + int(rng.integers(0, max(1, (lens[k] - unit_len) // 4))) * 4)

    if consensus is None:
        consensus = _unpack(pac, place(), unit_len).copy()
    out = []
    for _ in range(n_copies):
        d = float(rng.uniform(div_lo, div_hi))
        seg = consensus.copy()
        m = rng.random(unit_len) < d
        seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        n_indel = int(rng.poisson(indel_per_base * unit_len)) if indel_per_base > 0 else 0
        if n_indel:
            parts, last = [], 0
            for at in np.sort(rng.integers(10, unit_len - 10, size=n_indel)):
                at, ln = int(at), 1 + int(rng.integers(0, 8))
                if at < last:
                    continue
                parts.append(seg[last:at])
                if rng.random() < 0.5:
                    last = min(unit_len, at + ln)                                   # deletion
                else:
                    parts.append(rng.integers(0, 4, size=ln).astype(np.uint8))      # insertion
                    last = at
            parts.append(seg[last:])
            seg = np.concatenate(parts)
            if len(seg) < unit_len:
                seg = np.concatenate([seg, rng.integers(0, 4, size=unit_len - len(seg)).astype(np.uint8)])
            seg = seg[:unit_len]
        if revcomp and rng.random() < 0.5:
            seg = (3 - seg[::-1]).astype(np.uint8)
        p = place()
        _repack(pac, p, np.ascontiguousarray(seg))
        out.append((p, unit_len))
    return out


def windows_on(contigs, copies, flank):
    """[(name, len, offset)] of windows over planted copies +- flank, clipped to their contigs: passed to lh_synth_reads as its contig
    list, so that every molecule (and so every read) lies ON a copy or within `flank` of one"""
    import numpy as np
    offs = np.array([c[2] for c in contigs], dtype=np.int64)
    out = []
    for k, (p, n) in enumerate(copies):
        ci = int(np.searchsorted(offs, p, side="right")) - 1
        off, ln = contigs[ci][2], contigs[ci][1]
        b = max(off, p - flank)
        e = min(off + ln, p + n + flank)
        if e - b >= 2000:
            out.append(("w%d" % k, int(e - b), int(b)))
    return out


def config4_genome(lib, total_bases, seed=GENOME_SEED + 4, scale=1.0, n_alt=40, alt_len=1000000, threads=0, flank=2000, quiet=True):
    """configs[4]'s reference and the windows its reads are drawn from.  At scale 1 (hg38 size):
      * 120 segmental-duplication families of 50-200 copies x 20 kb, each copy 0.25-1 % off its family's consensus (pairs of copies at 98-99.5 %),
        with an indel per ~4 kb, either strand: ~300 Mb, as much of the genome as real segmental duplications take (5-10 %);
      * 40 families of 100-400 copies x 6 kb at 1-4 % from the consensus (LINE-like), 80 families of 500-3,000 copies x 300 bp at 2-10 % (SINE-like);
      * `n_alt` ALT contigs: copies of `alt_len` bases of the primary assembly at 99.7 %, flagged is_alt as <prefix>.alt would.
    `scale` multiplies the numbers of families (small genomes of the tests).  Reads are drawn on the copies +- `flank` (<= 2 kb).
    Returns dict(pac, l_pac, contigs, alt_flags, windows, families=[(kind, [(pos, len)])])."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ctg = hg38_like_contigs(int(total_bases))
    l_pac = sum(c[1] for c in ctg)
    pac = lib.synth_genome(l_pac, seed=seed, threads=threads)
    fams = []
    n_sd, n_l, n_s = max(1, int(round(120 * scale))), max(1, int(round(40 * scale))), max(1, int(round(80 * scale)))
    for _ in range(n_s):   # the short ones first: later, longer copies overwrite some of them (old repeats inside young duplications)
        fams.append(("sine300", plant_family(pac, ctg, rng, 300, int(rng.integers(500, 3001)), 0.02, 0.10, indel_per_base=1 / 300.0)))
    for _ in range(n_l):
        fams.append(("line6k", plant_family(pac, ctg, rng, 6000, int(rng.integers(100, 401)), 0.01, 0.04, indel_per_base=1 / 1500.0)))
    for _ in range(n_sd):
        fams.append(("segdup20k", plant_family(pac, ctg, rng, 20000, int(rng.integers(50, 201)), 0.0025, 0.01, indel_per_base=1 / 4000.0)))
    alt_flags = [0] * len(ctg)
    ctg_all, alts = ctg, []
    if n_alt:
        pac, l_pac, ctg_all, alt_flags, alts = add_alt_contigs(pac, ctg, n_alt, min(alt_len, l_pac // 200 // 4 * 4), 0.997, seed=seed + 1)
    win = []
    for kind, copies in fams:
        win += windows_on(ctg_all, copies, flank if kind != "sine300" else min(flank, 850))
    for src, dst, n in alts:   # reads from the ALT contigs and from the primary regions they copy: 2 kb windows spread over them, one per 25 kb
        for p0 in (src, dst):
            win += windows_on(ctg_all, [(p0 + o, 2000) for o in range(0, n - 2000, 25000)], 1000)
    if not quiet:
        print("config4 genome: %d Mb, %d families, %d copies, %d windows (%.0f Mb)" % (l_pac // 1000000, len(fams), sum(len(c) for _, c in fams), len(win), sum(w[1] for w in win) / 1e6))
    return dict(pac=pac, l_pac=l_pac, contigs=ctg_all, alt_flags=alt_flags, windows=win, families=fams)


def interleave_reads(ra, rb, chunk_reads=400000):
    """two read sets of lh_synth_reads (dicts of seq / seq_off / bc_pair_off / name_seed [/ truth_*]) over the SAME number of barcodes as one
    barcode-sorted set: barcode k holds ra's pairs of barcode k, then rb's.  The mixed workload of bench.py / the tests: a few pairs of every
    barcode drawn on repeat copies, the rest on unique sequence."""
    import numpy as np
    nb = len(ra["bc_pair_off"]) - 1
    assert nb == len(rb["bc_pair_off"]) - 1
    ca, cb = np.diff(ra["bc_pair_off"]).astype(np.int64), np.diff(rb["bc_pair_off"]).astype(np.int64)
    npa, npb = int(ca.sum()), int(cb.sum())
    # source pair of every output pair: index into the concatenation [ra's pairs | rb's pairs]
    key = np.concatenate([np.repeat(np.arange(nb, dtype=np.int64) * 2, ca), np.repeat(np.arange(nb, dtype=np.int64) * 2 + 1, cb)])
    src_pair = np.argsort(key, kind="stable")
    bco = np.zeros(nb + 1, dtype=np.int32)
    np.cumsum(ca + cb, out=bco[1:])
    so_a, so_b = np.asarray(ra["seq_off"], dtype=np.int64), np.asarray(rb["seq_off"], dtype=np.int64)
    starts = np.concatenate([so_a[:-1], so_b[:-1] + so_a[-1]])          # start of every source read in the concatenated bases
    lens = np.concatenate([np.diff(so_a), np.diff(so_b)])
    src_read = np.stack([2 * src_pair, 2 * src_pair + 1], axis=1).reshape(-1)
    out_len = lens[src_read]
    seq_off = np.zeros(len(src_read) + 1, dtype=np.int64)
    np.cumsum(out_len, out=seq_off[1:])
    seq_all = np.concatenate([np.asarray(ra["seq"][: so_a[-1]], dtype=np.uint8), np.asarray(rb["seq"][: so_b[-1]], dtype=np.uint8)])
    seq = np.empty(int(seq_off[-1]), dtype=np.uint8)
    for r0 in range(0, len(src_read), chunk_reads):
        r1 = min(len(src_read), r0 + chunk_reads)
        o0, o1 = int(seq_off[r0]), int(seq_off[r1])
        idx = np.repeat(starts[src_read[r0:r1]] - seq_off[r0:r1], out_len[r0:r1]) + np.arange(o0, o1, dtype=np.int64)
        seq[o0:o1] = seq_all[idx]
    out = dict(seq=seq, seq_off=seq_off, bc_pair_off=bco, name_seed=np.concatenate([ra["name_seed"], rb["name_seed"]])[src_pair], n_pairs=npa + npb,
               from_first=(src_pair < npa))
    for k in ("truth_rid", "truth_pos1", "truth_pos2"):
        if k in ra and k in rb:
            out[k] = np.concatenate([ra[k], rb[k]])[src_pair]
    return out


def outside_windows(contigs, alt_flags, windows, min_len=20000):
    """[(name, len, offset)]: the stretches of the primary contigs that lie outside every window of `windows` and are at least `min_len` long (lh_synth_reads
    cuts a molecule to the stretch it falls on; config4_genome's copies lie 18 kb apart on average, so the stretches between them are short) — where
    config4_genome planted nothing: the "unique sequence" the mixed workload draws most of its pairs on"""
    import numpy as np
    w = sorted((int(o), int(o + n)) for _, n, o in windows)
    ws = np.array([a for a, _ in w], dtype=np.int64)
    we = np.maximum.accumulate(np.array([b for _, b in w], dtype=np.int64)) if w else np.zeros(0, np.int64)
    out = []
    for (name, ln, off), alt in zip(contigs, alt_flags):
        if alt:
            continue
        lo = int(np.searchsorted(we, off, side="right"))
        cur = off
        k = lo
        while cur < off + ln:
            nxt = int(ws[k]) if k < len(ws) and ws[k] < off + ln else off + ln
            if nxt - cur >= min_len:
                b = (cur + 3) // 4 * 4
                out.append(("%s_u%d" % (name, len(out)), int(nxt - b) // 4 * 4, int(b)))
            if k >= len(ws) or ws[k] >= off + ln:
                break
            cur = max(cur, int(we[k]))
            k += 1
    return out

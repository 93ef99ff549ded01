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

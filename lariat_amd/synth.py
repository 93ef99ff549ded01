"""Synthetic linked-read data: seeded genomes and barcode-sorted read pairs.

No genome other than PhiX exists offline, so every benchmark/test genome is generated here:
iid ACGT with GC 0.41, optional planted repeats (segmental duplications / interspersed repeat families) so that
multi-candidate reads, mate rescue and RFA moves are exercised.  Reads follow the linked-read model: per
barcode a few long molecules, FR pairs with insert ~ N(350,50), substitution error ramping 0.1%->1% along the
read, rare indels; names are ``mol:<bc>:<chrom>:<ms>:<me>:<pos1>:<pos2>`` (what lariat's -simulated parses,
lariat.go:528-530).
"""
import hashlib

import numpy as np

_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_genome(contig_lens, seed=20261002, gc=0.41, n_dup=0, dup_len=20000, dup_identity=0.99, n_rep_family=0, rep_len=300, rep_copies=50):
    """returns list of uint8 nt4 arrays"""
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    contigs = [rng.choice(4, size=int(n), p=p).astype(np.uint8) for n in contig_lens]
    total = [len(c) for c in contigs]

    def mutate(seg, identity):
        seg = seg.copy()
        m = rng.random(len(seg)) > identity
        seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        return seg

    for _ in range(n_dup):   # segmental duplications: copy a segment elsewhere at the given identity
        ci = int(rng.integers(len(contigs)))
        cj = int(rng.integers(len(contigs)))
        L = min(dup_len, total[ci] // 4, total[cj] // 4)
        a = int(rng.integers(0, total[ci] - L))
        b = int(rng.integers(0, total[cj] - L))
        seg = mutate(contigs[ci][a:a + L], dup_identity)
        if rng.random() < 0.5:
            seg = _COMP[seg[::-1]]
        contigs[cj][b:b + L] = seg
    for _ in range(n_rep_family):   # interspersed repeat family
        cons = rng.choice(4, size=rep_len, p=p).astype(np.uint8)
        for _ in range(rep_copies):
            cj = int(rng.integers(len(contigs)))
            b = int(rng.integers(0, total[cj] - rep_len))
            seg = mutate(cons, 0.97)
            if rng.random() < 0.5:
                seg = _COMP[seg[::-1]]
            contigs[cj][b:b + rep_len] = seg
    return contigs


class ReadSet:
    """barcode-sorted pairs: seq (nt4), seq_off[2*n_pairs+1], bc_pair_off, names, truth arrays"""

    def __init__(self):
        self.seq = None
        self.seq_off = None
        self.bc_pair_off = None
        self.barcodes = None
        self.names = None
        self.name_seed = None
        self.truth_contig = None
        self.truth_pos1 = None
        self.truth_pos2 = None

    @property
    def n_pairs(self):
        return (len(self.seq_off) - 1) // 2

    def read(self, r):
        return self.seq[self.seq_off[r]:self.seq_off[r + 1]]

    def slice_barcodes(self, b0, b1):
        """sub-batch of barcodes [b0,b1) (used to shard by barcode range across ranks)"""
        out = ReadSet()
        p0, p1 = int(self.bc_pair_off[b0]), int(self.bc_pair_off[b1])
        s0, s1 = int(self.seq_off[2 * p0]), int(self.seq_off[2 * p1])
        out.seq = self.seq[s0:s1]
        out.seq_off = self.seq_off[2 * p0:2 * p1 + 1] - s0
        out.bc_pair_off = (self.bc_pair_off[b0:b1 + 1] - p0).astype(np.int32)
        out.barcodes = self.barcodes[b0:b1] if self.barcodes is not None else None
        out.names = self.names[p0:p1] if self.names is not None else None
        out.name_seed = self.name_seed[p0:p1]
        out.truth_contig = self.truth_contig[p0:p1]
        out.truth_pos1 = self.truth_pos1[p0:p1]
        out.truth_pos2 = self.truth_pos2[p0:p1]
        return out


def _name_seeds(names):
    out = np.empty(len(names), dtype=np.uint64)
    for i, n in enumerate(names):
        out[i] = int.from_bytes(hashlib.md5(n.encode()).digest()[:8], "little")   # lariat.go:1483-1484
    return out


def make_reads(contigs, contig_names, n_barcodes, pairs_per_barcode=100, seed=20261004, len1=143, len2=150, sub_lo=0.001, sub_hi=0.01,
               indel_rate=0.0001, mol_min=4, mol_max=10, with_names=True, junk_frac=0.0, ins_mean=350.0, ins_sd=50.0, ins_max=700):
    rng = np.random.default_rng(seed)
    n_pairs = n_barcodes * pairs_per_barcode
    clen = np.array([len(c) for c in contigs], dtype=np.int64)
    coff = np.concatenate([[0], np.cumsum(clen)])
    genome = np.concatenate(contigs)
    # molecules per barcode
    nmol = rng.integers(mol_min, mol_max + 1, size=n_barcodes)
    pair_mol_contig = np.empty(n_pairs, dtype=np.int64)
    pair_mol_start = np.empty(n_pairs, dtype=np.int64)
    pair_mol_end = np.empty(n_pairs, dtype=np.int64)
    k = 0
    for b in range(n_barcodes):
        K = int(nmol[b])
        mlen = np.clip(np.exp(rng.normal(np.log(50000.0), 0.6, size=K)), 10000, 200000).astype(np.int64)
        mc = rng.choice(len(contigs), size=K, p=clen / clen.sum())
        mlen = np.minimum(mlen, clen[mc] - 1000)
        ms = (rng.random(K) * (clen[mc] - mlen)).astype(np.int64)
        which = rng.choice(K, size=pairs_per_barcode, p=mlen / mlen.sum())
        pair_mol_contig[k:k + pairs_per_barcode] = mc[which]
        pair_mol_start[k:k + pairs_per_barcode] = ms[which]
        pair_mol_end[k:k + pairs_per_barcode] = ms[which] + mlen[which]
        k += pairs_per_barcode
    insert = np.clip(rng.normal(ins_mean, ins_sd, size=n_pairs), max(200, len1, len2), ins_max).astype(np.int64)
    span = np.maximum(pair_mol_end - pair_mol_start - insert, 1)
    frag_s = pair_mol_start + (rng.random(n_pairs) * span).astype(np.int64)
    flip = rng.random(n_pairs) < 0.5
    g0 = coff[pair_mol_contig]
    # forward-strand windows of the two mates
    fwd_len = np.where(flip, len2, len1)   # mate lying on the forward strand at frag_s
    rev_len = np.where(flip, len1, len2)   # mate lying on the reverse strand at frag end
    pos_f = frag_s
    pos_r = frag_s + insert - rev_len

    def gather(sel, pos, L, rc):
        idx = (g0[sel] + pos[sel])[:, None] + np.arange(L)[None, :]
        s = genome[idx]
        if rc:
            s = _COMP[s[:, ::-1]]
        return s

    # R1: forward at pos_f unless flipped (then reverse at pos_r); R2 the other
    r1 = np.empty((n_pairs, len1), dtype=np.uint8)
    r2 = np.empty((n_pairs, len2), dtype=np.uint8)
    nf = ~flip
    if nf.any():
        r1[nf] = gather(nf, pos_f, len1, False)
        r2[nf] = gather(nf, pos_r, len2, True)
    if flip.any():
        r1[flip] = gather(flip, pos_r, len1, True)
        r2[flip] = gather(flip, pos_f, len2, False)
    # substitutions: ramp along the read
    for arr, L in ((r1, len1), (r2, len2)):
        pe = np.linspace(sub_lo, sub_hi, L)[None, :]
        m = rng.random(arr.shape) < pe
        arr[m] = (arr[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
    reads = []
    for i in range(n_pairs):
        reads.append(r1[i])
        reads.append(r2[i])
    # rare indels (per read, python loop only over affected reads)
    if indel_rate > 0:
        n_reads = 2 * n_pairs
        has = rng.random(n_reads) < indel_rate * 150
        for r in np.nonzero(has)[0]:
            s = reads[r]
            p = int(rng.integers(20, len(s) - 20))
            ln = int(rng.integers(1, 4))
            if rng.random() < 0.5:   # deletion from the read
                s = np.concatenate([s[:p], s[p + ln:]])
            else:
                s = np.concatenate([s[:p], rng.integers(0, 4, size=ln).astype(np.uint8), s[p:]])
            reads[r] = s
    if junk_frac > 0:   # unalignable reads (placeholders, rescue attempts)
        n_reads = 2 * n_pairs
        for r in np.nonzero(rng.random(n_reads) < junk_frac)[0]:
            reads[r] = rng.integers(0, 4, size=len(reads[r])).astype(np.uint8)
    rs = ReadSet()
    lens = np.fromiter((len(x) for x in reads), dtype=np.int64, count=len(reads))
    rs.seq_off = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=rs.seq_off[1:])
    rs.seq = np.concatenate(reads)
    rs.bc_pair_off = (np.arange(n_barcodes + 1) * pairs_per_barcode).astype(np.int32)
    # barcodes [ACGT]{16}-1, sorted
    bcs = set()
    while len(bcs) < n_barcodes:
        need = n_barcodes - len(bcs)
        raw = rng.integers(0, 4, size=(need, 16))
        for row in raw:
            bcs.add("".join("ACGT"[v] for v in row) + "-1")
    rs.barcodes = sorted(bcs)
    pos1 = np.where(flip, pos_r, pos_f)
    pos2 = np.where(flip, pos_f, pos_r)
    rs.truth_contig = pair_mol_contig
    rs.truth_pos1 = pos1
    rs.truth_pos2 = pos2
    if with_names:
        rs.names = ["mol:%s:%s:%d:%d:%d:%d" % (rs.barcodes[i // pairs_per_barcode], contig_names[pair_mol_contig[i]], pair_mol_start[i], pair_mol_end[i],
                                               pos1[i], pos2[i]) for i in range(n_pairs)]
        rs.name_seed = _name_seeds(rs.names)
    else:   # cheap deterministic seeds for very large benchmark batches (documented in bench.py)
        x = np.arange(1, n_pairs + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        rs.names = None
        rs.name_seed = x ^ (x >> np.uint64(29))
    return rs


def to_fastq9(rs, trim_prefix=7, seed=7):
    """9-line barcode-sorted FASTQ text (README.md:34-48 of the reference) with a random trim prefix on read 1"""
    rng = np.random.default_rng(seed)
    out = []
    for b in range(len(rs.bc_pair_off) - 1):
        for p in range(rs.bc_pair_off[b], rs.bc_pair_off[b + 1]):
            r1 = "".join("ACGTN"[v] for v in rs.read(2 * p))
            r2 = "".join("ACGTN"[v] for v in rs.read(2 * p + 1))
            pre = "".join("ACGT"[v] for v in rng.integers(0, 4, size=trim_prefix))
            bc = rs.barcodes[b]
            out += ["@" + rs.names[p], pre + r1, "I" * (len(r1) + trim_prefix), r2, "I" * len(r2), bc, "I" * 16, "ACGTACGT", "I" * 8]
    return "\n".join(out) + "\n"

"""Independent property checks of the oracle.  No reference test pins a gapped alignment, a rescue, a molecule or a
MAPQ, so GPU == oracle is two restatements by one author agreeing; these tests derive the same facts a SECOND way, from definitions
rather than from BWA's / lariat's procedures, on CPU-sized inputs:

  * every SMEM interval the oracle reports occurs exactly as often as its interval size says (brute-force substring counts over the
    text fwd || revcomp), every seed position really holds its seed's bases, and the super-maximal exact matches of a read found by
    brute force (longest match from every start, minus the contained ones) are all among the oracle's intervals;
  * a candidate's CIGAR consumes exactly its query and reference spans; NM, the mismatch loci and the match / mismatch / indel /
    soft-clip counts recomputed from CIGAR + text are the reported ones; the CIGAR's path score equals the optimum of an independent
    full-matrix global alignment (no band) of the same spans in all but a handful of candidates, and never exceeds it;
  * log_alignment_probability, the AS value and is_proper follow from lariat.go:599-624 / 1102-1133 applied to the reported counts and
    positions (a straight-line Python derivation), and a read's MAPQ never exceeds 60 and is 0-or-more for every active alignment.
"""
import numpy as np
import pytest

import helpers
from lariat_amd import capi, synth

COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


@pytest.fixture(scope="module")
def world(oracle):
    rng = np.random.default_rng(77)
    contigs = synth.make_genome([90000, 60000], seed=13, n_dup=3, dup_len=1500, dup_identity=0.98, n_rep_family=1, rep_len=200, rep_copies=8)
    names = ["pA", "pB"]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=3, pairs_per_barcode=30, seed=19, sub_lo=0.002, sub_hi=0.03, indel_rate=0.004, junk_frac=0.03, mol_min=2, mol_max=3)
    b = helpers.batch_of(rs)
    fwd = np.concatenate(contigs)
    text = np.concatenate([fwd, COMP[fwd[::-1]]])
    return dict(oidx=oidx, rs=rs, b=b, text=bytes(text + 65), l_pac=len(fwd), contigs=contigs, names=names, dump=oidx.stage_dump(b), res=oidx.align_barcodes(b))


def occurrences(text, pat):
    n, at = 0, text.find(pat)
    while at >= 0:
        n += 1
        at = text.find(pat, at + 1)
    return n


def test_smem_intervals_by_brute_force(world):
    text, rs, d = world["text"], world["rs"], world["dump"]
    checked = found = 0
    for r in range(0, 2 * rs.n_pairs, 3):
        read = rs.read(r)
        q = bytes(np.where(read > 3, 200, read + 65).astype(np.uint8))   # a non-base never matches the text
        iv = d.intv[d.intv_off[r]:d.intv_off[r + 1]]
        have = set()
        for x0, x1, x2, info in iv:
            s, e = int(info >> 32), int(info & 0xffffffff)
            assert e - s >= 19
            assert occurrences(text, q[s:e]) == int(x2), (r, s, e)      # the interval's size is the number of occurrences
            have.add((s, e))
            checked += 1
        # super-maximal exact matches by definition: the longest match from every start, minus the contained ones
        L = []
        for i in range(len(q)):
            lo, hi = 0, len(q) - i
            while lo < hi:
                mid = (lo + hi + 1) >> 1
                if q[i:i + mid] in text:
                    lo = mid
                else:
                    hi = mid - 1
            L.append(lo)
        for i in range(len(q)):
            if L[i] >= 19 and (i == 0 or i - 1 + L[i - 1] < i + L[i]):
                assert (i, i + L[i]) in have, (r, i, i + L[i])
                found += 1
    assert checked > 80 and found > 40


def test_seed_positions_hold_their_bases(world):
    text, rs, d = world["text"], world["rs"], world["dump"]
    n = 0
    for r in range(2 * rs.n_pairs):
        q = bytes(np.where(rs.read(r) > 3, 200, rs.read(r) + 65).astype(np.uint8))
        for k in range(int(d.seed_off[r]), int(d.seed_off[r + 1])):
            rb, qb, ln = int(d.seed_rbeg[k]), int(d.seed_qbeg[k]), int(d.seed_len[k])
            assert text[rb:rb + ln] == q[qb:qb + ln]
            n += 1
    assert n > 500


def full_matrix_global(q, t, a=1, b=4, o=6, e=1):
    """optimum of the global alignment of q and t with affine gaps (Gotoh, no band): an independent textbook implementation"""
    NEG = -10 ** 9
    n, m = len(q), len(t)
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64); E = H.copy(); F = H.copy()
    H[0, 0] = 0
    for i in range(1, n + 1):
        F[i, 0] = H[i, 0] = -(o + e * i)
    for j in range(1, m + 1):
        E[0, j] = H[0, j] = -(o + e * j)
    for i in range(1, n + 1):
        qi = q[i - 1]
        for j in range(1, m + 1):
            s = -1 if (qi > 3 or t[j - 1] > 3) else (a if qi == t[j - 1] else -b)
            E[i, j] = max(E[i, j - 1] - e, H[i, j - 1] - o - e)
            F[i, j] = max(F[i - 1, j] - e, H[i - 1, j] - o - e)
            H[i, j] = max(H[i - 1, j - 1] + s, E[i, j], F[i, j])
    return int(H[n, m])


def test_cigar_nm_and_loci_from_the_text(world):
    res, rs, l_pac = world["res"], world["rs"], world["l_pac"]
    fwd = np.concatenate(world["contigs"])
    coff = np.concatenate([[0], np.cumsum([len(c) for c in world["contigs"]])])
    n_checked = n_opt = n_gapped = 0
    for r in range(2 * rs.n_pairs):
        read = rs.read(r)
        for c in res.cands_of_read(r):
            if res.rid[c] < 0:
                continue
            ops = res.cigar[res.cigar_off[c]:res.cigar_off[c + 1]]
            rev = bool(res.reversed[c])
            # the reference bases of the alignment on the forward strand, and the read in the strand's orientation
            pos, aend = int(res.pos[c]), int(res.aend[c])
            g0 = int(coff[res.rid[c]]) + pos
            ref = fwd[g0:g0 + (aend - pos)]
            seq = COMP[read[::-1]] if rev else read
            x = y = nm = mm = ind = indlen = clips = cliplen = 0
            loci = []
            score = 0
            for k, cv in enumerate(ops):
                op, ln = int(cv) & 0xf, int(cv) >> 4
                if op == 0:
                    for t in range(ln):
                        qv, tv = int(seq[x + t]), int(ref[y + t])
                        if qv != tv:
                            mm += 1
                            loci.append(pos + y + t + (1 if rev else 0))   # lariat.go:1609 counts a reverse read's loci down from refEnd, one past the base
                        score += -1 if qv > 3 else (1 if qv == tv else -4)
                    x += ln; y += ln
                elif op == 1:
                    ind += 1; indlen += ln; x += ln; score -= 6 + ln
                elif op == 2:
                    ind += 1; indlen += ln; y += ln; score -= 6 + ln
                elif op == 3:
                    assert k in (0, len(ops) - 1)
                    clips += 1; cliplen += ln; x += ln
            assert x == len(read) and y == len(ref), (r, c)
            assert int(res.nm[c]) == mm + indlen and int(res.mismatches[c]) == mm and int(res.indels[c]) == ind
            assert int(res.soft_clipped[c]) == clips and int(res.soft_clipped_length[c]) == cliplen and int(res.matches[c]) == x - cliplen - indlen_query(ops) - mm
            got = sorted(int(v) for v in res.mm_ref_loc[res.mm_off[c]:res.mm_off[c + 1]])
            assert got == sorted(loci), (r, c)
            n_checked += 1
            if ind and n_gapped < 25:   # the path through the gaps is an optimal global alignment of the two spans
                qs = seq[cliplen_lead(ops):len(seq) - cliplen_tail(ops)]
                best = full_matrix_global(qs, ref)
                assert score <= best
                n_opt += score == best
                n_gapped += 1
    assert n_checked > 150 and n_gapped >= 5 and n_opt >= n_gapped - 1


def indlen_query(ops):
    return sum(int(c) >> 4 for c in ops if int(c) & 0xf == 1)


def cliplen_lead(ops):
    return int(ops[0]) >> 4 if int(ops[0]) & 0xf == 3 else 0


def cliplen_tail(ops):
    return int(ops[-1]) >> 4 if len(ops) > 1 and int(ops[-1]) & 0xf == 3 else 0


def test_pair_scores_and_flags_by_a_second_derivation(world):
    res, rs = world["res"], world["rs"]
    improper = -4.0   # lariat's default -improper_pair_penalty (main.go), what lo_opts_init sets

    def single(c):   # lariat.go:599-624, one side
        s = float(res.mismatches[c]) * -2.0 + float(res.indels[c]) * -3.0
        if res.soft_clipped[c] > 0:
            s -= 5.0 * float(res.soft_clipped[c]) + 0.5 * float(res.soft_clipped_length[c])
        return s

    def is_pair(a, m):   # lariat.go:1102-1133
        if res.reversed[a] == res.reversed[m] or res.rid[a] != res.rid[m]:
            return False
        f, v = (m, a) if res.reversed[a] else (a, m)
        return -35 <= int(res.pos[v]) - int(res.pos[f]) < 750

    n = 0
    for c in range(res.n_cand):
        assert res.log_alignment_probability[c] == single(c)   # scoreAlignment(aln, nil, 0) - improper
    for r in range(2 * rs.n_pairs):
        a = int(res.active_idx[r])
        assert a >= 0 and res.active[a] and res.in_filtered[a]
        assert 0 <= int(res.mapq[a]) <= 60
        m = int(res.mate_idx[a])
        if m >= 0 and res.rid[a] >= 0 and res.rid[m] >= 0:
            want = single(a) + single(m) + (0.0 if is_pair(a, m) else improper)
            assert abs(res.as_score[r] - want) < 1e-12, (r, res.as_score[r], want)   # mapq_data.score: the pair score without molecule penalty
            n += 1
    assert n > 100

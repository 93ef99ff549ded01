"""Definition-level checks of the MIDDLE of candidate generation: the stages that decide WHICH candidates exist — mem_chain /
mem_chain_flt, ksw_extend2's regions, mem_sort_dedup_patch, ksw_u8's local optimum (reached from gobwa.go:244,253 -> mem_align1_core and
gobwa.go:291,315 -> mem_matesw) — rest on one recollection of BWA 0.7.17 in the oracle and its twin on the GPU.  These tests derive
the same facts a second way, from what the stages are DEFINED to compute rather than from how BWA computes them:

  * chains: a from-scratch Python mem_chain (a sorted list for the B-tree, test_and_merge and mem_chain_weight from their descriptions) finds the
    oracle's kept chains among its own, with their seed counts and weights; every chain it has and the oracle dropped is shadowed by a kept chain of the
    oracle under mem_chain_flt's stated rule (query overlap >= mask_level of the shorter, weight < drop_ratio x the kept one's and >= 2 x min_seed_len
    lighter), and no two kept chains marked 3 shadow each other;
  * regions: every region of mem_align1_core holds a seed of its read through which a textbook anchored extension (full matrix, no band, no z-drop;
    left with h0 = seed length, right with h0 = the left result) reaches exactly the region's score, and ends on the region's end points whenever that
    optimum is unique — the precondition (the optimum's path inside BWA's band, never z-dropped) is counted, not assumed;
  * mem_sort_dedup_patch: no two regions of a read's final list are redundant under its 0.95 rule, none is an exact duplicate, the order is the stated
    (score desc, rb, qb); every kept chain's first seed lies inside a final region of its contig or inside one that is redundant with it;
  * ksw_u8 (mate rescue): on windows with planted diverged copies — gapped ones, tandem second copies that tie the score — the oracle's (score, te, qe)
    is the textbook local Smith-Waterman optimum (numpy, plain Gotoh recurrences, no striping, no saturation: scores < 250), te the first row that holds
    it and qe the smallest column of that row; (tb, qb) the same for the reversed prefixes.
"""
import bisect
import ctypes as C

import numpy as np
import pytest

import helpers
from lariat_amd import synth

A, B, O, E, W, MAXGAP, MINSEED = 1, 4, 6, 1, 100, 10000, 19
COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


@pytest.fixture(scope="module")
def world(oracle):
    contigs = synth.make_genome([120000, 80000], seed=31, n_dup=6, dup_len=2500, dup_identity=0.97, n_rep_family=2, rep_len=250, rep_copies=12)
    names = ["mA", "mB"]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=3, pairs_per_barcode=40, seed=23, sub_lo=0.002, sub_hi=0.03, indel_rate=0.004, junk_frac=0.02, mol_min=2, mol_max=3)
    b = helpers.batch_of(rs)
    fwd = np.concatenate(contigs)
    text = np.concatenate([fwd, COMP[fwd[::-1]]])
    coff = np.concatenate([[0], np.cumsum([len(c) for c in contigs])])
    return dict(oidx=oidx, rs=rs, b=b, text=text, l_pac=len(fwd), coff=coff, dump=oidx.stage_dump(b))


@pytest.fixture(scope="module")
def rep_world(oracle):
    """reads on the copies of repeat families and of a tandem array (helpers.repeat_family_case): tens of chains and regions per read, many of them close together"""
    names, contigs, rs = helpers.repeat_family_case(29, 3)
    oidx = oracle.index_build_naive(names, contigs)
    fwd = np.concatenate(contigs)
    return dict(oidx=oidx, rs=rs, l_pac=len(fwd), dump=oidx.stage_dump(helpers.batch_of(rs)))


# ------------------------------------------------------------------------------------------------ chains
def py_mem_chain(seeds, l_pac):
    """mem_chain's walk, written from its description: seeds = [(rbeg, qbeg, len, rid)] in the read's seed order"""
    keys, chains = [], []   # keys: sorted (pos, order of creation) -> the B-tree; chains[k] = dict
    for rbeg, qbeg, ln, rid in seeds:
        if rid < 0:
            continue
        at = bisect.bisect_right(keys, (rbeg, 1 << 60)) - 1   # the chain with the greatest pos <= rbeg, the one made last among equal positions
        merged = False
        if at >= 0:
            c = chains[keys[at][1]]
            first, last = c["seeds"][0], c["seeds"][-1]
            qend, rend = last[1] + last[2], last[0] + last[2]
            if rid != c["rid"]:
                pass
            elif qbeg >= first[1] and qbeg + ln <= qend and rbeg >= first[0] and rbeg + ln <= rend:
                merged = True   # contained: absorbed
            elif (last[0] < l_pac or first[0] < l_pac) and rbeg >= l_pac:
                pass            # the other strand
            else:
                x, y = qbeg - last[1], rbeg - last[0]
                if y >= 0 and x - y <= W and y - x <= W and x - last[2] < MAXGAP and y - last[2] < MAXGAP:
                    c["seeds"].append((rbeg, qbeg, ln))
                    merged = True
        if not merged:
            keys.insert(bisect.bisect_right(keys, (rbeg, 1 << 60)), (rbeg, len(chains)))
            chains.append(dict(pos=rbeg, rid=rid, seeds=[(rbeg, qbeg, ln)]))
    for c in chains:   # mem_chain_weight: covered query bases, covered reference bases, the smaller
        def cover(lo_len):
            w = end = 0
            for lo, ln in lo_len:
                if lo >= end:
                    w += ln
                elif lo + ln > end:
                    w += lo + ln - end
                end = max(end, lo + ln)
            return w
        c["w"] = min(cover([(s[1], s[2]) for s in c["seeds"]]), cover([(s[0], s[2]) for s in c["seeds"]]))
        c["beg"], c["end"] = c["seeds"][0][1], c["seeds"][-1][1] + c["seeds"][-1][2]
    return chains


def shadows(kept, c):
    """mem_chain_flt's rule: chain c is dropped because of kept chain `kept`"""
    b_max, e_min = max(kept["beg"], c["beg"]), min(kept["end"], c["end"])
    if e_min <= b_max:
        return False
    min_l = min(kept["end"] - kept["beg"], c["end"] - c["beg"])
    return e_min - b_max >= min_l * 0.5 and min_l < MAXGAP and c["w"] < kept["w"] * 0.5 and kept["w"] - c["w"] >= MINSEED * 2


@pytest.mark.parametrize("which", ["unique", "repeats"])
def test_chains_by_a_second_derivation(world, rep_world, which):
    wd = world if which == "unique" else rep_world
    d, l_pac = wd["dump"], wd["l_pac"]
    n_reads = n_dropped = n_multi = 0
    for r in range(d.n_reads):
        s0, s1 = int(d.seed_off[r]), int(d.seed_off[r + 1])
        seeds = [(int(d.seed_rbeg[k]), int(d.seed_qbeg[k]), int(d.seed_len[k]), int(d.seed_rid[k])) for k in range(s0, s1)]
        mine = py_mem_chain(seeds, l_pac)
        c0, c1 = int(d.chain_off[r]), int(d.chain_off[r + 1])
        theirs = [(int(d.chain_pos[k]), int(d.chain_rid[k]), int(d.chain_nseeds[k]), int(d.chain_w[k]), int(d.chain_kept[k])) for k in range(c0, c1)]
        pool = {}
        for c in mine:
            pool.setdefault((c["pos"], c["rid"], len(c["seeds"]), c["w"]), []).append(c)
        kept = []
        for pos, rid, n, w, kp in theirs:
            assert kp in (1, 2, 3)
            cands = pool.get((pos, rid, n, w))
            assert cands, ("the oracle kept a chain the second derivation does not have", r, pos, rid, n, w)
            c = cands.pop()
            c["kept"] = kp
            kept.append(c)
        assert len(kept) == len(theirs)
        for c in mine:
            if "kept" in c:
                continue
            n_dropped += 1   # dropped by mem_chain_flt: some kept chain shadows it under the stated rule
            assert any(shadows(k, c) for k in kept), ("a chain was dropped that no kept chain shadows", r, c["pos"], c["w"], c["beg"], c["end"])
        for i, a in enumerate(kept):   # kept = 3: no significant overlap with a heavier kept chain that would have shadowed it
            for b_ in kept[:i]:
                assert not (a["kept"] == 3 and shadows(b_, a) and b_["kept"] == 3 and a["w"] != b_["w"]), (r, a["pos"], b_["pos"])
        n_multi += len(mine) > 1
        n_reads += 1
    print("reads: %d, with more than one chain: %d, chains dropped by mem_chain_flt (each shadowed by a kept one): %d" % (n_reads, n_multi, n_dropped))
    assert n_reads > 150 and n_multi > 30 and n_dropped > 10


# ------------------------------------------------------------------------------------------------ regions
def anchored_extension(q, t, h0):
    """textbook extension from an anchor worth h0: the best score over all end cells (i rows of t, j columns of q) of h0 + a path from the anchor, affine gaps,
    a cell whose score is not positive is dead (ksw_extend2's semantics of an extension, without its band, z-drop or early exit).  Returns (max, the set of
    end cells (ti, qi) = consumed target / query bases that reach it, the best score that consumes all of q and where)."""
    n, m = len(t), len(q)
    NEG = -10 ** 9
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64)
    Eg = np.full((n + 1, m + 1), NEG, dtype=np.int64)   # gap in the query direction (insertion: consumes q)
    Fg = np.full((n + 1, m + 1), NEG, dtype=np.int64)   # gap in the target direction (deletion: consumes t)
    H[0, 0] = h0
    for j in range(1, m + 1):
        v = h0 - (O + E * j)
        if v > 0:
            Eg[0, j] = H[0, j] = v
    for i in range(1, n + 1):
        v = h0 - (O + E * i)
        if v > 0:
            Fg[i, 0] = H[i, 0] = v
        ti = t[i - 1]
        for j in range(1, m + 1):
            s = -1 if (q[j - 1] > 3 or ti > 3) else (A if q[j - 1] == ti else -B)
            e = max(Eg[i, j - 1] - E, H[i, j - 1] - O - E)
            f = max(Fg[i - 1, j] - E, H[i - 1, j] - O - E)
            dg = H[i - 1, j - 1] + s if H[i - 1, j - 1] > 0 else NEG
            h = max(dg, e, f)
            if h > 0:
                H[i, j] = h
                Eg[i, j] = e if e > 0 else NEG
                Fg[i, j] = f if f > 0 else NEG
    mx = int(H.max())
    ends = set(zip(*[x.tolist() for x in np.nonzero(H == mx)]))
    g = int(H[:, m].max()) if m else h0
    return mx, ends, g


def test_regions_are_anchored_extension_optima(world):
    d, text, rs = world["dump"], world["text"], world["rs"]
    n_regions = n_equal = n_ends = n_checked_ends = 0
    for r in range(0, d.n_reads, 2):
        read = rs.read(r)
        s0, s1 = int(d.seed_off[r]), int(d.seed_off[r + 1])
        for k in range(int(d.reg_off[r]), int(d.reg_off[r + 1])):
            rb, re, qb, qe, score = int(d.reg_rb[k]), int(d.reg_re[k]), int(d.reg_qb[k]), int(d.reg_qe[k]), int(d.reg_score[k])
            if int(d.reg_seedlen0[k]) == 0:
                continue
            n_regions += 1
            ok = False
            for s in range(s0, s1):   # a seed of the read inside the region, on a diagonal the region can hold
                sr, sq, sl = int(d.seed_rbeg[s]), int(d.seed_qbeg[s]), int(d.seed_len[s])
                if not (sr >= rb and sr + sl <= re and sq >= qb and sq + sl <= qe) or sl != int(d.reg_seedlen0[k]):
                    continue
                # left: the reversed prefixes; right: the suffixes (windows as wide as BWA's: the query flank + the gaps it could pay for)
                ql = read[:sq][::-1]
                tl = text[max(0, sr - len(ql) - 110):sr][::-1]
                sL, endsL, gL = anchored_extension(ql, tl, sl * A) if len(ql) else (sl * A, {(0, 0)}, sl * A)
                qr = read[sq + sl:]
                tr = text[sr + sl:sr + sl + len(qr) + 110]
                sR, endsR, gR = anchored_extension(qr, tr, sL) if len(qr) else (sL, {(0, 0)}, sL)
                if sR != score:
                    continue
                ok = True
                # the end points: local ends unless the end-to-end score is within the clipping penalty (pen_clip = 5)
                if len(endsL) == 1 and len(endsR) == 1 and len(ql) and len(qr):
                    (tiL, qiL), (tiR, qiR) = next(iter(endsL)), next(iter(endsR))
                    n_checked_ends += 1
                    loc_l = gL <= 0 or gL <= sL - 5
                    loc_r = gR <= 0 or gR <= sR - 5
                    if loc_l and loc_r:
                        n_ends += (sq - qiL == qb and sr - tiL == rb and sq + sl + qiR == qe and sr + sl + tiR == re)
                    else:
                        n_ends += 1   # (end-to-end on a side: that side's end is the read's, checked by the span test of test_oracle_properties)
                break
            n_equal += ok
    # the textbook optimum has no band and no z-drop: equality means the preconditions held; they do but for a handful (counted, not assumed)
    print("regions from seeds: %d, equal to the anchored-extension optimum through a seed: %d; unique end points checked: %d, equal: %d" % (n_regions, n_equal, n_checked_ends, n_ends))
    assert n_regions > 100 and n_equal >= 0.97 * n_regions and n_ends >= 0.95 * n_checked_ends


# ------------------------------------------------------------------------------------------------ mem_sort_dedup_patch
def redundant(p, q):
    """mem_sort_dedup_patch's test for two regions (rb, re, qb, qe) of one contig"""
    orr = min(p[1], q[1]) - max(p[0], q[0])
    oq = min(p[3], q[3]) - max(p[2], q[2])
    mr, mq = min(p[1] - p[0], q[1] - q[0]), min(p[3] - p[2], q[3] - q[2])
    return orr > 0.95 * mr and oq > 0.95 * mq


@pytest.mark.parametrize("which", ["unique", "repeats"])
def test_dedup_leaves_no_redundant_pair_and_drops_nothing_irredundant(world, rep_world, which):
    d = (world if which == "unique" else rep_world)["dump"]
    n_pairs_checked = n_chains = 0
    for r in range(d.n_reads):
        regs = [(int(d.reg_rb[k]), int(d.reg_re[k]), int(d.reg_qb[k]), int(d.reg_qe[k]), int(d.reg_score[k]), int(d.reg_rid[k]))
                for k in range(int(d.reg_off[r]), int(d.reg_off[r + 1]))]
        keys = [(-g[4], g[0], g[2]) for g in regs]
        assert keys == sorted(keys) and len(set(keys)) == len(keys), r   # (score desc, rb, qb), no identical hits
        for i, p in enumerate(regs):
            for q in regs[:i]:
                if p[5] == q[5]:   # (regions further apart than max_chain_gap are not compared by the call: they do not overlap either)
                    n_pairs_checked += 1
                    assert not redundant(p, q), (r, p, q)
        # every chain that went on to extension is represented: its first seed inside a final region of its contig (a patched merge covers both parts)
        for k in range(int(d.chain_off[r]), int(d.chain_off[r + 1])):
            pos, rid = int(d.chain_pos[k]), int(d.chain_rid[k])
            n_chains += 1
            assert any(g[5] == rid and g[0] <= pos < g[1] for g in regs), ("a kept chain left no region that covers its first seed", r, pos)
    print("region pairs of one contig checked: %d; kept chains with a covering region: %d" % (n_pairs_checked, n_chains))
    assert n_pairs_checked > (10 if which == "unique" else 500) and n_chains > 200


# ------------------------------------------------------------------------------------------------ ksw_u8
def textbook_local_sw(q, t):
    """plain Smith-Waterman-Gotoh over rows of t, columns of q; returns the full H matrix (int64)"""
    n, m = len(t), len(q)
    H = np.zeros((n + 1, m + 1), dtype=np.int64)
    Ecol = np.zeros(m + 1, dtype=np.int64)   # gap that consumes target bases (carried down a column)
    ks = np.arange(m + 1, dtype=np.int64)
    for i in range(1, n + 1):
        s = np.where((q > 3) | (t[i - 1] > 3), -1, np.where(q == t[i - 1], A, -B)).astype(np.int64)
        Ecol = np.maximum(Ecol - E, H[i - 1] - O - E)
        Ecol[0] = 0
        hnf = np.zeros(m + 1, dtype=np.int64)
        hnf[1:] = np.maximum(np.maximum(H[i - 1, :-1] + s, Ecol[1:]), 0)
        # gap along the row: F(k) = max_{j<k} hnf(j) - O - E (k - j) = max-plus prefix scan
        g = hnf - O + E * ks
        pre = np.maximum.accumulate(g)
        F = np.zeros(m + 1, dtype=np.int64)
        F[1:] = pre[:-1] - E * ks[1:]
        H[i] = np.maximum(hnf, F)
        H[i, 0] = 0
        Ecol = np.maximum(Ecol, 0)
    return H[1:, 1:]


def first_max(H):
    mx = int(H.max())
    te = int(np.nonzero((H == mx).any(axis=1))[0][0])
    qe = int(np.nonzero(H[te] == mx)[0][0])
    return mx, te, qe


def test_rescue_smith_waterman_is_the_textbook_local_optimum(oracle):
    oracle.L.lo_ksw_align2.argtypes = [C.c_int32, C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
    rng = np.random.default_rng(5)

    def mutate(x, sub, n_indel):
        y = x.copy()
        m = rng.random(len(y)) < sub
        y[m] = (y[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        for _ in range(n_indel):
            at = int(rng.integers(10, len(y) - 10))
            ln = int(rng.integers(1, 5))
            y = np.concatenate([y[:at], y[at + ln:]]) if rng.random() < 0.5 else np.concatenate([y[:at], rng.integers(0, 4, size=ln).astype(np.uint8), y[at:]])
        return y

    n = n_hit = n_tie = n_gap = 0
    for case in range(120):
        qlen = int(rng.integers(60, 151))
        q = rng.integers(0, 4, size=qlen).astype(np.uint8)
        tlen = int(rng.integers(300, 686))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        kind = case % 6
        if kind <= 2:      # one diverged copy, ungapped or with indels
            c = mutate(q, [0.01, 0.05, 0.12][kind], [0, 1, 2][kind])
            at = int(rng.integers(0, tlen - len(c)))
            t[at:at + len(c)] = c
        elif kind == 3:    # two copies in tandem, equally good: the FIRST row that reaches the score decides
            c = mutate(q, 0.03, 0)
            at = int(rng.integers(0, tlen - 2 * len(c) - 5))
            t[at:at + len(c)] = c
            t[at + len(c) + 3:at + 2 * len(c) + 3] = c
            n_tie += 1
        elif kind == 4:    # a copy whose best path has a gap, beside a shorter ungapped one
            c = mutate(q, 0.02, 1)
            at = int(rng.integers(0, max(1, tlen - len(c) - 90)))
            t[at:at + len(c)] = c
            t[tlen - 80:tlen - 20] = q[:60]
            n_gap += 1
        out = (C.c_int32 * 7)()
        oracle.L.lo_ksw_align2(qlen, q.ctypes.data_as(C.POINTER(C.c_uint8)), tlen, t.ctypes.data_as(C.POINTER(C.c_uint8)), out)
        score, te, qe, tb, qb = int(out[0]), int(out[1]), int(out[2]), int(out[3]), int(out[4])
        H = textbook_local_sw(q, t)
        mx, te0, qe0 = first_max(H)
        assert mx < 250
        assert (score, te, qe) == (mx, te0, qe0), (case, kind, (score, te, qe), (mx, te0, qe0))
        n += 1
        if score >= MINSEED:   # KSW_XSTART: the same question for the reversed prefixes that end at (te, qe), stopped at the first row that reaches the score
            Hr = textbook_local_sw(q[:qe + 1][::-1], t[:te + 1][::-1])
            rows = np.nonzero((Hr >= score).any(axis=1))[0]
            assert len(rows)
            rte = int(rows[0])
            rqe = int(np.nonzero(Hr[rte] == Hr[rte].max())[0][0])
            assert int(Hr[rte].max()) == score
            assert (tb, qb) == (te - rte, qe - rqe), (case, kind, (tb, qb), (te - rte, qe - rqe))
            n_hit += 1
    assert n == 120 and n_hit > 60 and n_tie >= 15 and n_gap >= 15

// k_extend.h — K4: seed extension (BWA mem_chain2aln + ksw_extend2), one wavefront per read.
// Reached in the reference through mem_align1_core (go/src/gobwa/gobwa.go:244,253).
//
// ksw_extend2 is a row sweep with loop-carried f/h1 and a data-dependent [beg,end) window.  It maps onto the
// wavefront WITHOUT changing a single result because, in this kernel, E and F derive from the diagonal term M
// (not from H): inside one row  F(j+1) = max(F(j) - e_ins, max(M(j) - oe_ins, 0))  is a max-plus prefix scan.
// So: lanes = query columns (j = lane + 64*t, the whole eh[] array of upstream lives in registers, 4 per lane),
// one DP row per step, F by a 6-step shuffle scan, row maximum / arg-max (last j among equal maxima) and the
// beg/end trimming by ballots.  Integer VALU work; no MFMA (this is not a contraction).
#pragma once
#include "k_chain.h"

struct ExtRes { int score, qle, tle, gtle, gscore, max_off; };


#define LH_NEG_INF (-0x3fffffff)

// One DP row over slab T (columns 64*T .. 64*T+63).  Uses/updates H##T, E##T, and the running row state.
#define LH_EXT_SLAB(T)                                                                                          \
    if ((T) < NS && 64 * (T) <= end && 64 * (T) + 63 >= beg) {                                                            \
        int j = 64 * (T) + lane;                                                                                \
        int in = j >= beg && j < end;                                                                           \
        int M = 0, e = E##T, tins = 0, enew = E##T;                                                             \
        if (in) {                                                                                               \
            int qv = qb##T;                                                                                     \
            int sc = (tb > 3 || qv > 3) ? -1 : (tb == qv ? a_ : -b_);                                           \
            M = H##T ? H##T + sc : 0;                                                                           \
            int t_ = M - oe_del; t_ = t_ > 0 ? t_ : 0;                                                          \
            enew = e - e_del; enew = enew > t_ ? enew : t_;                                                     \
            tins = M - oe_ins; tins = tins > 0 ? tins : 0;                                                      \
        }                                                                                                       \
        int bj = in ? tins + (j + 1) * e_ins : LH_NEG_INF;                                                      \
        int incl = wave_scan_max_i32(bj, lane);                                                                 \
        int excl = wave_shr1_i32(incl, LH_NEG_INF);                                                             \
        int G = gcarry > excl ? gcarry : excl;                                                                  \
        int f = G - j * e_ins;                                                                                  \
        int h = M > e ? M : e; h = h > f ? h : f;                                                               \
        if (!in) h = 0;                                                                                         \
        int last = wave_readlane(incl, 63);                                                                            \
        gcarry = gcarry > last ? gcarry : last;                                                                 \
        /* row maximum, last j among equal maxima */                                                            \
        int hm = in ? h : -1;                                                                                   \
        int smax = wave_max_i32(hm);                                                                            \
        if (smax >= 0 && smax >= m) {                                                                           \
            u64 bm = __ballot(in && h == smax);                                                                 \
            m = smax; mj = 64 * (T) + 63 - __clzll(bm);                                                         \
        }                                                                                                       \
        /* write back eh[]: eh[j].h = H(i,j-1) for beg<j<=end, eh[beg].h = first-column value, eh[j].e, eh[end].e = 0 */ \
        int hleft = wave_shr1_i32(h, hcarry);                                                                   \
        hcarry = wave_readlane(h, 63);                                                                                 \
        if (j == beg && beg < end) H##T = h1_init;                                                              \
        else if (j > beg && j <= end) H##T = hleft;                                                             \
        if (in) E##T = enew;                                                                                    \
        if (j == end) { E##T = 0; if (beg >= end) H##T = h1_init; }                                             \
        if (in && j == end - 1) hlast_l = h;                                                                    \
    }

#define LH_EXT_NZ(T) (H##T != 0 || E##T != 0)

// ksw_extend2.  Query column j holds qarr[qoff + qstep*j]; target row i is the reference base at tcoord0 + tstep*i.
// NS = number of 64-column slabs instantiated (qlen <= 64*NS): the common short extension pays for one slab only.
template <int NS>
__device__ __forceinline__ ExtRes wave_ksw_extend2(const DIndex& ix, const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qlen, i64 tcoord0, int tstep,
                                                   int tlen, int w, int end_bonus, int zdrop, int h0, int lane, u64* cells) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    // eh[] in registers: column j = lane + 64*t
    int H0, H1, H2, H3, E0 = 0, E1 = 0, E2 = 0, E3 = 0;
    int qb0, qb1, qb2, qb3;
#define LH_EXT_INIT(T)                                                                                 \
    {                                                                                                  \
        int j = 64 * (T) + lane;                                                                       \
        qb##T = j < qlen ? qarr[qoff + qstep * j] : 4;                                                 \
        int v = 0;                                                                                     \
        if (j == 0) v = h0;                                                                            \
        else if (j <= qlen) {                                                                          \
            int vj = h0 - oe_ins - (j - 1) * e_ins;                                                    \
            if (j == 1) v = h0 > oe_ins ? vj : 0;                                                      \
            else v = (vj + e_ins > e_ins && h0 > oe_ins) ? vj : 0;                                     \
        }                                                                                              \
        H##T = v;                                                                                      \
    }
    LH_EXT_INIT(0) LH_EXT_INIT(1) LH_EXT_INIT(2) LH_EXT_INIT(3)
#undef LH_EXT_INIT
    // adjust w
    int maxsc = a_ > 0 ? a_ : 0;   // max entry of mat (a, -b, -1)
    int max_ins = (int)((double)(qlen * maxsc + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * maxsc + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    int tchunk = 4;
    u64 ncell = 0;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {   // next 64 target bases, one per lane
            int ii = i + lane;
            tchunk = ii < tlen ? dev_ref_base(ix, tcoord0 + (i64)tstep * ii) : 4;
        }
        int tb = wave_readlane(tchunk, i & 63);
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1_init;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        else h1_init = 0;
        int m = 0, mj = -1;
        int gcarry = beg * e_ins;   // F(beg) = 0  <=>  G = beg*e_ins
        int hcarry = 0, hlast_l = 0;
        LH_EXT_SLAB(0) LH_EXT_SLAB(1) LH_EXT_SLAB(2) LH_EXT_SLAB(3)
        if (end > beg) ncell += (u64)(end - beg);
        // h1 after the row = H(i,end-1), or the first-column value if the window is empty
        int h1 = h1_init;
        if (beg < end) { int src = (end - 1) & 63; h1 = wave_readlane(hlast_l, src); }
        int j_after = beg < end ? end : beg;
        if (j_after == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        // update beg and end for the next round
        int nbeg = end;
        {
            u64 b0 = __ballot(LH_EXT_NZ(0) && lane >= beg && lane < end);
            u64 b1 = NS > 1 ? __ballot(LH_EXT_NZ(1) && 64 + lane >= beg && 64 + lane < end) : 0;
            u64 b2 = NS > 2 ? __ballot(LH_EXT_NZ(2) && 128 + lane >= beg && 128 + lane < end) : 0;
            u64 b3 = NS > 3 ? __ballot(LH_EXT_NZ(3) && 192 + lane >= beg && 192 + lane < end) : 0;
            if (b0) nbeg = __ffsll((unsigned long long)b0) - 1;
            else if (b1) nbeg = 64 + __ffsll((unsigned long long)b1) - 1;
            else if (b2) nbeg = 128 + __ffsll((unsigned long long)b2) - 1;
            else if (b3) nbeg = 192 + __ffsll((unsigned long long)b3) - 1;
        }
        int nend_j = nbeg - 1;
        {
            u64 b0 = __ballot(LH_EXT_NZ(0) && lane >= nbeg && lane <= end);
            u64 b1 = NS > 1 ? __ballot(LH_EXT_NZ(1) && 64 + lane >= nbeg && 64 + lane <= end) : 0;
            u64 b2 = NS > 2 ? __ballot(LH_EXT_NZ(2) && 128 + lane >= nbeg && 128 + lane <= end) : 0;
            u64 b3 = NS > 3 ? __ballot(LH_EXT_NZ(3) && 192 + lane >= nbeg && 192 + lane <= end) : 0;
            if (b3) nend_j = 192 + 63 - __clzll((unsigned long long)b3);
            else if (b2) nend_j = 128 + 63 - __clzll((unsigned long long)b2);
            else if (b1) nend_j = 64 + 63 - __clzll((unsigned long long)b1);
            else if (b0) nend_j = 63 - __clzll((unsigned long long)b0);
        }
        beg = nbeg;
        end = nend_j + 2 < qlen ? nend_j + 2 : qlen;
    }
    if (cells) *cells += ncell;
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

#ifndef LH_WAVE_NARROW_MAX_LOSS
#define LH_WAVE_NARROW_MAX_LOSS 61   // diagonal loss up to which the wave kernel runs its DP in a provably sufficient band (12 mismatches with the default scoring: band 55)
#endif
// K4, one wavefront per read.  Without a list: grid = n_reads waves.  With a list (k_extend2.h hands over the reads that
// do not suit its lane-per-read kernel): the waves stride over list[range[0] .. range[1]).
__global__ void __launch_bounds__(64) k_extend(DIndex ix, DOpts o, int n_reads, const int32_t* __restrict__ list, const int32_t* __restrict__ range,
                                                const uint8_t* __restrict__ seq, const uint32_t* __restrict__ q4, const i64* __restrict__ seq_off,
                                                const i64* __restrict__ seed_off, const DChain* __restrict__ chains, const DSeed* __restrict__ cseeds,
                                                const int32_t* __restrict__ n_chains, int32_t* __restrict__ sorder, int32_t* __restrict__ sdone,
                                                const i64* __restrict__ reg_off, DReg* __restrict__ regs, int32_t* __restrict__ n_regs,
                                                DCounters* __restrict__ ctr, int count_chains) {
    __shared__ uint8_t q[LH_MAXLEN + 6];
    const int lane = LANE();
    int it_first = list ? range[0] + (int)blockIdx.x : (int)blockIdx.x, it_last = list ? range[1] : n_reads, it_step = list ? (int)gridDim.x : n_reads;
    for (int it = it_first; it < it_last; it += it_step) {
    int r = list ? list[it] : it;
    WAVE_SYNC();   // the previous read's query is no longer in use
    i64 off = seq_off[r];
    int l_query = (int)(seq_off[r + 1] - off);
    if (l_query > LH_MAXLEN) l_query = 0;
    for (int i = lane; i < l_query; i += 64) q[i] = seq[off + i];
    WAVE_SYNC();
    i64 base = seed_off[r];
    DReg* av = regs + reg_off[r];
    int n_av = 0;
    int nch = n_chains[r];
    i64 l_pac = ix.l_pac;
    u64 cells = 0, win = 0;
    for (int ci = 0; ci < nch; ++ci) {
        DChain c = chains[base + ci];
        const DSeed* sd = cseeds + base + c.seed_start;
        int32_t* srt = sorder + base + c.seed_start;   // seed indices by (score, index) ascending
        int32_t* done = sdone + base + c.seed_start;   // 1 = extension performed (upstream: srt[k] != 0)
        int n = c.n;
        if (n == 0) continue;
        // max possible span
        i64 r0 = l_pac << 1, r1 = 0;
        for (int i = lane; i < n; i += 64) {
            DSeed t = sd[i];
            i64 b = t.rbeg - (t.qbeg + dev_cal_max_gap(o, t.qbeg));
            i64 e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + dev_cal_max_gap(o, l_query - t.qbeg - t.len));
            r0 = r0 < b ? r0 : b;
            r1 = r1 > e ? r1 : e;
        }
        i64 rmax0 = wave_min_i64(r0), rmax1 = wave_max_i64(r1);
        rmax0 = rmax0 > 0 ? rmax0 : 0;
        rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
        DSeed s0 = sd[0];
        if (rmax0 < l_pac && l_pac < rmax1) {   // crossing the forward-reverse boundary; then choose one side
            if (s0.rbeg < l_pac) rmax1 = l_pac;
            else rmax0 = l_pac;
        }
        dev_fetch_clamp(ix, &rmax0, s0.rbeg, &rmax1);   // bns_fetch_seq clamps the window to the contig
        win += (u64)(rmax1 - rmax0);
        // order of extension: by seed score (= len) then index, descending (upstream sorts score<<32|i ascending and walks down)
        for (int i = lane; i < n; i += 64) {
            DSeed t = sd[i];
            int rank = 0;
            for (int u = 0; u < n; ++u) { DSeed x = sd[u]; rank += (x.len < t.len) || (x.len == t.len && u < i); }
            srt[rank] = i;
            done[i] = 1;
        }
        WAVE_SYNC();
        for (int k = n - 1; k >= 0; --k) {
            int si = srt[k];
            DSeed s = sd[si];
            // test whether extension has been made before (any earlier region of this read "around" the seed)
            int hit = 0;
            for (int i0 = 0; i0 < n_av; i0 += 64) {
                int i = i0 + lane, f = 0;
                if (i < n_av) {
                    DReg p = av[i];
                    if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) && !(s.len - p.seedlen0 > .1 * l_query)) {
                        int qd = s.qbeg - p.qb; i64 rd = s.rbeg - p.rb;
                        int max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                        int w = max_gap < p.w ? max_gap : p.w;
                        if (qd - rd < w && rd - qd < w) f = 1;
                        else {
                            qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                            max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                            w = max_gap < p.w ? max_gap : p.w;
                            if (qd - rd < w && rd - qd < w) f = 1;
                        }
                    }
                }
                if (__any(f)) { hit = 1; break; }
            }
            if (hit) {   // (almost) contained: extend only if an overlapping, already-extended seed of the chain lies on another diagonal
                int other = 0;
                for (int i0 = k + 1; i0 < n; i0 += 64) {
                    int i = i0 + lane, f = 0;
                    if (i < n) {
                        int ti = srt[i];
                        if (done[ti]) {
                            DSeed t = sd[ti];
                            if (!(t.len < s.len * .95)) {
                                if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) f = 1;
                                if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) f = 1;
                            }
                        }
                    }
                    if (__any(f)) { other = 1; break; }
                }
                if (!other) {
                    if (lane == 0) done[si] = 0;
                    WAVE_SYNC();
                    continue;
                }
            }
            DReg a;
            a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = 0; a.seedcov = 0; a.secondary = 0; a.n_comp = 0; a.is_alt = 0;
            int aw0 = o.w, aw1 = o.w;
            a.w = o.w; a.score = a.truesc = -1; a.rid = c.rid;
            for (int side = 0; side < 2; ++side) {   // one extension site: 0 = left (reversed query prefix vs reversed reference prefix), 1 = right
                int qoff, qstep, qlen, tstep, tlen, bonus, h0, sc0 = a.score, qe = s.qbeg + s.len;
                i64 tc0, re = s.rbeg + s.len;
                if (side == 0) {
                    if (!s.qbeg) { a.score = a.truesc = s.len * o.a; a.qb = 0; a.rb = s.rbeg; continue; }
                    qoff = s.qbeg - 1; qstep = -1; qlen = s.qbeg; tc0 = s.rbeg - 1; tstep = -1; tlen = (int)(s.rbeg - rmax0); bonus = o.pen_clip5; h0 = s.len * o.a;
                } else {
                    if (qe == l_query) { a.qe = l_query; a.re = s.rbeg + s.len; continue; }
                    qoff = qe; qstep = 1; qlen = l_query - qe; tc0 = re; tstep = 1; tlen = (int)(rmax1 - re); bonus = o.pen_clip3; h0 = sc0;
                }
                ExtRes e;
                e.score = -1; e.qle = e.tle = e.gtle = 0; e.gscore = -1; e.max_off = 0;
                int aw = o.w;
                // ksw_extend2 without the DP when it is provably the ungapped extension (the argument is in k_extend2.h, ext_control):
                // every lane walks the diagonal (the same walk: no divergence), eight bases per step against the 4-bit text
                // ... and in a provably sufficient band when the diagonal loses more than one gap's cost but less than zdrop and never drops to
                // zero (same argument, ext_control: with a loss P over the whole query, cells further than B off the diagonal, gap_cost(B + 1) > P,
                // and cells that depend on them stay strictly below their row's diagonal cell: ksw_extend2(w) and ksw_extend2(B) return the same).
                // A read on a repeat family is extended on dozens of copies that differ from it in a handful of bases: the band is ~10 wide, not 100.
                int proven = 0, narrow = 0;
                if (tlen >= qlen) {
                    const int thr = (o.o_ins + o.e_ins) < (o.o_del + o.e_del) ? (o.o_ins + o.e_ins) : (o.o_del + o.e_del);
                    const int p_cap = o.zdrop > 0 && o.zdrop < LH_WAVE_NARROW_MAX_LOSS ? o.zdrop : LH_WAVE_NARROW_MAX_LOSS;
                    const DiagScan ds = dev_diag_scan(ix, o, seq + off, q4, off, qoff, qstep, qlen, tc0, tstep, h0, p_cap > thr ? p_cap : thr);
                    if (ds.done && ds.P < thr) {
                        e.score = ds.mx; e.qle = ds.mxk + 1; e.tle = ds.mxk + 1; e.gscore = ds.sc_run; e.gtle = qlen; e.max_off = 0;
                        a.score = e.score;
                        proven = 1;
                    } else if (ds.done && p_cap > thr) {
                        narrow = 1;
                        while (o.o_ins + o.e_ins * (narrow + 1) <= ds.P || o.o_del + o.e_del * (narrow + 1) <= ds.P) ++narrow;
                    }
                }
                for (int i = 0; i < 2 && !proven; ++i) {   // MAX_BAND_TRY
                    int prev = a.score;
                    aw = o.w << i;
                    const int band = narrow && narrow < aw ? narrow : aw;
                    if (qlen <= 64) e = wave_ksw_extend2<1>(ix, o, q, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &cells);
                    else if (qlen <= 128) e = wave_ksw_extend2<2>(ix, o, q, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &cells);
                    else e = wave_ksw_extend2<4>(ix, o, q, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &cells);
                    a.score = e.score;
                    if (a.score == prev || e.max_off < (aw >> 1) + (aw >> 2)) break;
                }
                if (side == 0) {
                    aw0 = aw;
                    if (e.gscore <= 0 || e.gscore <= a.score - o.pen_clip5) {   // local extension
                        a.qb = s.qbeg - e.qle; a.rb = s.rbeg - e.tle;
                        a.truesc = a.score;
                    } else {   // to-end extension
                        a.qb = 0; a.rb = s.rbeg - e.gtle;
                        a.truesc = e.gscore;
                    }
                } else {
                    aw1 = aw;
                    if (e.gscore <= 0 || e.gscore <= a.score - o.pen_clip3) {   // local extension
                        a.qe = qe + e.qle; a.re = re + e.tle;
                        a.truesc += a.score - sc0;
                    } else {   // to-end extension
                        a.qe = l_query; a.re = re + e.gtle;
                        a.truesc += e.gscore - sc0;
                    }
                }
            }
            // seedcov
            int cov = 0;
            for (int i = lane; i < n; i += 64) {
                DSeed t = sd[i];
                if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
            }
            a.seedcov = wave_sum_i32(cov);
            a.w = aw0 > aw1 ? aw0 : aw1;
            a.seedlen0 = s.len;
            a.frac_rep = c.frac_rep;
            a.is_alt = c.is_alt;
            if (lane == 0) av[n_av] = a;
            n_av++;
            WAVE_SYNC();
        }
    }
    if (lane == 0) {
        n_regs[r] = n_av;
        if (ctr) {
            atomicAdd(&LH_CTR(ctr)->ext_cells, cells);
            if (count_chains) { atomicAdd(&LH_CTR(ctr)->win_bases, win); atomicAdd(&LH_CTR(ctr)->n_chain_ext, (u64)nch); }   // else counted by k_chain_lane
        }
    }
    }
}

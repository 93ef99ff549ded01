// k_global.h — banded global alignment (BWA ksw_global2 / bwa_gen_cigar2) as a wave-parallel row sweep.
// Used by K5 (mem_patch_reg: score only) and K7 (mem_reg2aln: score + traceback -> CIGAR), i.e. the arithmetic behind
// go/src/gobwa/gobwa.go:404 (mem_reg2aln) and the patch step inside mem_align1_core (gobwa.go:244,253).
//
// Same idea as k_extend.h: in ksw_global2 E and F are functions of the diagonal term m, so F inside a row is a
// max-plus prefix scan over the band; lanes own query columns (j = lane + 64*t, upstream's eh[] in registers),
// the per-cell direction byte z[i][j-beg] is written to an HBM slab and walked back by lane 0.
#pragma once
#include "k_extend.h"

#define LH_MINUS_INF (-0x40000000)

#define LH_GLB_SLAB(T)                                                                                     \
    if (64 * (T) <= end && 64 * (T) + 63 >= beg) {                                                         \
        int j = 64 * (T) + lane;                                                                           \
        int in = j >= beg && j < end;                                                                      \
        int mm = LH_MINUS_INF, e = E##T, enew = E##T, tins = LH_MINUS_INF;                                 \
        int d = 0;                                                                                         \
        if (in) {                                                                                          \
            int qv = qb##T;                                                                                \
            int sc = (tb > 3 || qv > 3) ? -1 : (tb == qv ? a_ : -b_);                                      \
            mm = H##T + sc;                                                                                \
            tins = mm - oe_ins;                                                                            \
        }                                                                                                  \
        int bj = in ? tins + (j + 1) * e_ins : -0x7fffffff;                                                \
        int incl = wave_scan_max_i32(bj, lane);                                                            \
        int excl = wave_shr1_i32(incl, -0x7fffffff);                                                       \
        int G = gcarry > excl ? gcarry : excl;                                                             \
        int f = G - j * e_ins;                                                                             \
        int last = wave_readlane(incl, 63);                                                                       \
        gcarry = gcarry > last ? gcarry : last;                                                            \
        int h = 0;                                                                                         \
        if (in) {                                                                                          \
            d = mm >= e ? 0 : 1;                                                                           \
            h = mm >= e ? mm : e;                                                                          \
            d = h >= f ? d : 2;                                                                            \
            h = h >= f ? h : f;                                                                            \
            int t_ = mm - oe_del;                                                                          \
            enew = e - e_del;                                                                              \
            d |= enew > t_ ? 1 << 2 : 0;                                                                   \
            enew = enew > t_ ? enew : t_;                                                                  \
            int fd = f - e_ins;                                                                            \
            d |= fd > tins ? 2 << 4 : 0;                                                                   \
            if (zrow) zrow[j - beg] = (uint8_t)d;                                                          \
        }                                                                                                  \
        int hleft = wave_shr1_i32(h, hcarry);                                                              \
        hcarry = wave_readlane(h, 63);                                                                            \
        if (j == beg && beg < end) H##T = h1_init;                                                         \
        else if (j > beg && j <= end) H##T = hleft;                                                        \
        if (in) E##T = enew;                                                                               \
        if (j == end) { E##T = LH_MINUS_INF; if (beg >= end) H##T = h1_init; }                             \
    }

// ksw_global2; returns the score.  If z != NULL the direction matrix (tlen x n_col bytes) is written for traceback.
__device__ __forceinline__ int wave_ksw_global2(const DIndex& ix, const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qlen, i64 tcoord0, int tstep,
                                                int tlen, int w, uint8_t* z, int lane, u64* cells) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    int H0, H1, H2, H3, E0, E1, E2, E3, qb0, qb1, qb2, qb3;
#define LH_GLB_INIT(T)                                                              \
    {                                                                               \
        int j = 64 * (T) + lane;                                                    \
        qb##T = j < qlen ? qarr[qoff + qstep * j] : 4;                              \
        H##T = j == 0 ? 0 : ((j <= qlen && j <= w) ? -(o_ins + e_ins * j) : LH_MINUS_INF); \
        E##T = LH_MINUS_INF;                                                        \
    }
    LH_GLB_INIT(0) LH_GLB_INIT(1) LH_GLB_INIT(2) LH_GLB_INIT(3)
#undef LH_GLB_INIT
    int tchunk = 4;
    u64 ncell = 0;
    if (tlen > 100000) { o.wd[5] = 1; o.wd[6] = tlen; tlen = 0; }
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {
            int ii = i + lane;
            tchunk = ii < tlen ? dev_ref_base(ix, tcoord0 + (i64)tstep * ii) : 4;
        }
        int tb = wave_readlane(tchunk, i & 63);
        int beg = i > w ? i - w : 0;
        int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : LH_MINUS_INF;
        int gcarry = LH_MINUS_INF + beg * e_ins;
        int hcarry = 0;
        uint8_t* zrow = z ? z + (size_t)i * n_col : (uint8_t*)0;
        LH_GLB_SLAB(0) LH_GLB_SLAB(1) LH_GLB_SLAB(2) LH_GLB_SLAB(3)
        if (end > beg) ncell += (u64)(end - beg);
    }
    if (cells) *cells += ncell;
    // score = eh[qlen].h
    int src = qlen & 63, slab = qlen >> 6;
    int v = slab == 0 ? H0 : slab == 1 ? H1 : slab == 2 ? H2 : H3;
    return wave_readlane(v, src);
}

// The same for a band of at most 64 columns (2w + 1 <= 64: the usual case, a read against a copy that differs in a handful of bases), in BAND
// coordinates: lane d holds column j = i - w + d of row i.  eh[j].h for the next row is the H this lane has just computed (column j + 1 of row
// i + 1 reads H(i, j)): it stays where it is; E(i + 1, j) is needed one lane to the left; F is a scan over the lanes as above.  One pass over 64
// lanes per row whatever the query length, no slab loop.  The direction bytes keep ksw_global2's layout (row i: columns from beg = max(0, i - w)).
#ifdef LH_EMU
__device__ __forceinline__ int wave_shl1_i32(int v, int fill) { int o = __shfl_down(v, 1); return LANE() == 63 ? fill : o; }   // lane i <- lane i+1
#else
__device__ __forceinline__ int wave_shl1_i32(int v, int fill) { return LH_DPP(fill, v, 0x130, 0xF); }   // wave_shl:1
#endif
__device__ __forceinline__ int wave_ksw_global2_band(const DIndex& ix, const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qlen, i64 tcoord0, int tstep,
                                                     int tlen, int w, uint8_t* z, int lane, u64* cells) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const int d = lane;
    int H, E = LH_MINUS_INF;
    {   // eh[] before the first row: lane d = column d - w
        const int j = d - w;
        H = j == 0 ? 0 : ((j >= 1 && j <= qlen && j <= w) ? -(o_ins + e_ins * j) : LH_MINUS_INF);
    }
    int last = qlen <= w ? -(o_ins + e_ins * qlen) : LH_MINUS_INF;   // eh[qlen].h
    int tchunk = 4;
    u64 ncell = 0;
    if (tlen > 100000) { o.wd[5] = 1; o.wd[6] = tlen; tlen = 0; }
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {
            int ii = i + lane;
            tchunk = ii < tlen ? dev_ref_base(ix, tcoord0 + (i64)tstep * ii) : 4;
        }
        const int tb = wave_readlane(tchunk, i & 63);
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int j = i - w + d;
        const int in = j >= beg && j < end;
        int mm = LH_MINUS_INF, tins = LH_MINUS_INF, dir = 0;
        if (in) {
            const int qv = qarr[qoff + qstep * j];
            const int sc = (tb > 3 || qv > 3) ? -1 : (tb == qv ? a_ : -b_);
            mm = (j == 0 ? (i == 0 ? 0 : -(o_del + e_del * i)) : H) + sc;   // eh[0].h of row i is the first-column value row i - 1 left there
            tins = mm - oe_ins;
        }
        const int bj = in ? tins + (j + 1) * e_ins : -0x7fffffff;
        const int incl = wave_scan_max_i32(bj, lane);
        const int excl = wave_shr1_i32(incl, -0x7fffffff);
        const int gc = LH_MINUS_INF + beg * e_ins;
        const int G = gc > excl ? gc : excl;
        const int f = G - j * e_ins;
        int h = LH_MINUS_INF, enew = LH_MINUS_INF;
        if (in) {
            const int e = E;
            dir = mm >= e ? 0 : 1;
            h = mm >= e ? mm : e;
            dir = h >= f ? dir : 2;
            h = h >= f ? h : f;
            const int t_ = mm - oe_del;
            enew = e - e_del;
            dir |= enew > t_ ? 1 << 2 : 0;
            enew = enew > t_ ? enew : t_;
            const int fd = f - e_ins;
            dir |= fd > tins ? 2 << 4 : 0;
            if (z) z[(size_t)i * n_col + (j - beg)] = (uint8_t)dir;
        }
        // the next row: H stays (column j + 1 of row i + 1 reads H(i, j)); E(i + 1, j) moves to the lane that holds column j in row i + 1: d - 1
        H = in ? h : LH_MINUS_INF;
        E = wave_shl1_i32(in ? enew : LH_MINUS_INF, LH_MINUS_INF);
        if (end == qlen && end > beg) last = wave_readlane(H, qlen - 1 - i + w);   // eh[end].h = the row's last H
        if (end > beg) ncell += (u64)(end - beg);
    }
    if (cells) *cells += ncell;
    return last;
}

// ---- the same recurrence for FOUR candidates per wave: a group of 16 lanes per candidate, lane d of the group = band column d (2w + 1 <= 15) ----
// Every cross-lane operation stays inside a DPP row (= the group); the four groups run their own rows side by side, a group whose DP is
// shorter (or that has none: run = 0) idles through the other groups' rows.  qarr / tg / z: the group's LDS areas (tg[i] = reference base of row i).
#ifdef LH_EMU
__device__ __forceinline__ int grp_scan_max_i32(int v, int lane) {
    for (int d = 1; d < 16; d <<= 1) { int o = __shfl_up(v, d); if ((lane & 15) >= d) v = v > o ? v : o; }
    return v;
}
__device__ __forceinline__ int grp_shr1_i32(int v, int fill) { int o = __shfl_up(v, 1); return (LANE() & 15) == 0 ? fill : o; }
__device__ __forceinline__ int grp_shl1_i32(int v, int fill) { int o = __shfl_down(v, 1); return (LANE() & 15) == 15 ? fill : o; }
#else
__device__ __forceinline__ int grp_scan_max_i32(int v, int) {
    const int ID = (int)0x80000000;
    int t;
    t = LH_DPP(ID, v, 0x111, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x112, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x114, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x118, 0xF); v = v > t ? v : t;
    return v;
}
__device__ __forceinline__ int grp_shr1_i32(int v, int fill) { return LH_DPP(fill, v, 0x111, 0xF); }   // row_shr:1
__device__ __forceinline__ int grp_shl1_i32(int v, int fill) { return LH_DPP(fill, v, 0x101, 0xF); }   // row_shl:1
#endif
__device__ __forceinline__ int grp_max_i32(int v) { for (int m = 8; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v > o ? v : o; } return v; }
__device__ __forceinline__ int grp_sum_i32(int v) { for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m); return v; }

// (r05) the same for groups of 32 lanes (two candidates per wave, bands up to 2 * 15 + 1 columns): a DPP row is 16 lanes, so the scan takes the row_bcast:15 step of the
// wave-wide scan and the shifts are the wave's with the group's edge lanes filled
#ifdef LH_EMU
template <int GL> __device__ __forceinline__ int grpN_scan_max_i32(int v, int lane) {
    for (int d = 1; d < GL; d <<= 1) { int o = __shfl_up(v, d); if ((lane & (GL - 1)) >= d) v = v > o ? v : o; }
    return v;
}
template <int GL> __device__ __forceinline__ int grpN_shr1_i32(int v, int fill) { int o = __shfl_up(v, 1); return (LANE() & (GL - 1)) == 0 ? fill : o; }
template <int GL> __device__ __forceinline__ int grpN_shl1_i32(int v, int fill) { int o = __shfl_down(v, 1); return (LANE() & (GL - 1)) == GL - 1 ? fill : o; }
#else
template <int GL> __device__ __forceinline__ int grpN_scan_max_i32(int v, int lane) {
    v = grp_scan_max_i32(v, lane);
    if (GL == 32) { const int t = LH_DPP((int)0x80000000, v, 0x142, 0xA); v = v > t ? v : t; }   // lane 15 of rows 0 and 2 to rows 1 and 3
    return v;
}
template <int GL> __device__ __forceinline__ int grpN_shr1_i32(int v, int fill) {
    if (GL == 16) return grp_shr1_i32(v, fill);
    const int t = LH_DPP(fill, v, 0x138, 0xF);   // wave_shr:1
    return (LANE() & 31) == 0 ? fill : t;
}
template <int GL> __device__ __forceinline__ int grpN_shl1_i32(int v, int fill) {
    if (GL == 16) return grp_shl1_i32(v, fill);
    const int t = LH_DPP(fill, v, 0x130, 0xF);   // wave_shl:1
    return (LANE() & 31) == 31 ? fill : t;
}
#endif
template <int GL> __device__ __forceinline__ int grpN_max_i32(int v) { for (int m = GL / 2; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v > o ? v : o; } return v; }
template <int GL> __device__ __forceinline__ int grpN_sum_i32(int v) { for (int m = GL / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m); return v; }

template <int GL>
__device__ __forceinline__ int grp_ksw_global2_band(const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qlen, const uint8_t* tg, int tlen, int w, uint8_t* z, int lane,
                                                    int run, int tl_max) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const int d = lane & (GL - 1);
    int H, E = LH_MINUS_INF, lastv = LH_MINUS_INF;
    {
        const int j = d - w;
        H = j == 0 ? 0 : ((j >= 1 && j <= qlen && j <= w) ? -(o_ins + e_ins * j) : LH_MINUS_INF);
    }
    for (int i = 0; i < tl_max; ++i) {
        const int ra = run && i < tlen;
        const int tb = ra ? tg[i] : 4;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int j = i - w + d;
        const int in = ra && j >= beg && j < end;
        int mm = LH_MINUS_INF, tins = LH_MINUS_INF, dir = 0;
        if (in) {
            const int qv = qarr[qoff + qstep * j];
            const int sc = (tb > 3 || qv > 3) ? -1 : (tb == qv ? a_ : -b_);
            mm = (j == 0 ? (i == 0 ? 0 : -(o_del + e_del * i)) : H) + sc;
            tins = mm - oe_ins;
        }
        const int bj = in ? tins + (j + 1) * e_ins : -0x7fffffff;
        const int incl = grpN_scan_max_i32<GL>(bj, lane);
        const int excl = grpN_shr1_i32<GL>(incl, -0x7fffffff);
        const int gc = LH_MINUS_INF + beg * e_ins;
        const int G = gc > excl ? gc : excl;
        const int f = G - j * e_ins;
        int h = LH_MINUS_INF, enew = LH_MINUS_INF;
        if (in) {
            const int e = E;
            dir = mm >= e ? 0 : 1;
            h = mm >= e ? mm : e;
            dir = h >= f ? dir : 2;
            h = h >= f ? h : f;
            const int t_ = mm - oe_del;
            enew = e - e_del;
            dir |= enew > t_ ? 1 << 2 : 0;
            enew = enew > t_ ? enew : t_;
            const int fd = f - e_ins;
            dir |= fd > tins ? 2 << 4 : 0;
            z[i * n_col + (j - beg)] = (uint8_t)dir;
            if (j == qlen - 1) lastv = h;
        }
        H = in ? h : LH_MINUS_INF;
        E = grpN_shl1_i32<GL>(in ? enew : LH_MINUS_INF, LH_MINUS_INF);
    }
    // eh[qlen].h after the last row = H(tlen - 1, qlen - 1): the band reaches that cell (callers: w >= |qlen - tlen| + 3), lane qlen - tlen + w held it
    return grpN_max_i32<GL>(run && d == qlen - tlen + w ? lastv : (int)0x80000000);
}

// bwa_gen_cigar2 without traceback: global score of query[qb_..qe_) against the fwd||rev reference interval [rb,re).
// Returns 0 and sets *ok=0 when upstream would reject the interval.
__device__ __forceinline__ int wave_gen_score(const DIndex& ix, const DOpts& o, const uint8_t* q, int qb_, int l_query, int w_, i64 rb, i64 re, int lane, int* ok,
                                              u64* cells) {
    i64 l_pac = ix.l_pac;
    *ok = 0;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
    i64 b2 = rb, e2 = re;   // bns_get_seq clamps to [0, 2*l_pac)
    if (e2 > l_pac << 1) e2 = l_pac << 1;
    if (b2 < 0) b2 = 0;
    if (e2 - b2 != re - rb) return 0;
    int rlen = (int)(re - rb);
    *ok = 1;
    int rev = rb >= l_pac;   // reverse both sequences so that indels are left-aligned on the forward strand
    int qoff = rev ? qb_ + l_query - 1 : qb_, qstep = rev ? -1 : 1;
    i64 t0 = rev ? re - 1 : rb;
    int tstep = rev ? -1 : 1;
    if (l_query == rlen && w_ == 0) {   // no gap; no need to do DP
        int sc = 0;
        for (int i = lane; i < l_query; i += 64) {
            int tb = dev_ref_base(ix, t0 + (i64)tstep * i), qv = q[qoff + qstep * i];
            sc += (tb > 3 || qv > 3) ? -1 : (tb == qv ? o.a : -o.b);
        }
        return wave_sum_i32(sc);
    }
    int max_ins = (int)((double)(((l_query + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
    int max_del = (int)((double)(((l_query + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del;
    max_gap = max_gap > 1 ? max_gap : 1;
    int dl = rlen - l_query; dl = dl < 0 ? -dl : dl;
    int w = (max_gap + dl + 1) >> 1;
    w = w < w_ ? w : w_;
    int min_w = dl + 3;
    w = w > min_w ? w : min_w;
    return wave_ksw_global2(ix, o, q, qoff, qstep, l_query, t0, tstep, rlen, w, (uint8_t*)0, lane, cells);
}

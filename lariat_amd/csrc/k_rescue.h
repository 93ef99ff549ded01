// k_rescue.h — K6: mate rescue (lariat's GoBwaMemMateSW loops, go/src/gobwa/gobwa.go:286-325, around BWA's mem_matesw /
// ksw_align2), one wavefront per read PAIR.
//
// ksw_align2 upstream is Farrar's striped SSE2 kernel (ksw_u8): 16 byte lanes x slen segments, with a lazy-F loop that
// repairs H across stripe boundaries but never refreshes E.  Its results are therefore defined by the striping, not by
// the textbook recurrences.  Restated column-parallel (lane = query position, all 64 lanes busy):
//   hnf(k)    = max(H'(i-1,k-1) + S, E(i,k))                       (saturating u8 arithmetic: floors at 0)
//   F_seg(k)  = max-plus prefix scan of (hnf - oe_ins) RESTARTED at every stripe boundary k = L*slen  -> H_main = max(hnf, F_seg)
//   F_full(k) = the same scan over the whole row                                                    -> H'     = max(hnf, F_full)
//   E(i+1,k)  = max(E - e_del, H_main - oe_del)     row maximum is taken over H_main                (what the SSE code does)
// i.e. one segmented and one plain shuffle scan per row.  The early exit of the lazy-F loop only skips no-op work.
#pragma once
#include "k_dedup.h"

struct KswR { int score, te, qe, tb, qb; };

__device__ __forceinline__ int sat0(int x) { return x < 0 ? 0 : x; }

#define LH_SEG_BIG 8192
#define LH_SEG_OFF 256

#define LH_U8_SLAB(T)                                                                                         \
    if (64 * (T) < ncol) {                                                                                    \
        int k = 64 * (T) + lane;                                                                              \
        int in = k < ncol;                                                                                    \
        int hd = wave_shr1_i32(HP##T, hdcarry);             /* H'(i-1,k-1) */                                  \
        hdcarry = wave_readlane(HP##T, 63);                                                                          \
        int qv = qb##T;                                                                                       \
        int sc = k >= qlen ? 0 : ((tb > 3 || qv > 3) ? -1 : (tb == qv ? a_ : -b_));                           \
        int hnf = sat0(hd + sc);                                                                              \
        int e = E##T;                                                                                         \
        hnf = hnf > e ? hnf : e;                                                                              \
        /* gap-open candidates, G-transformed: cand(k) = hnf(k) - oe_ins + (k+1)*e_ins feeds F(k+1) */        \
        int cand = hnf - oe_ins + (k + 1) * e_ins;                                                            \
        int c1 = in ? cand : -0x3fffffff;                                                                     \
        int c2 = in ? cand + LH_SEG_OFF + LH_SEG_BIG * sg##T : -0x3fffffff;   /* stripe id in the high part */ \
        int i1 = wave_scan_max_i32(c1, lane), i2 = wave_scan_max_i32(c2, lane);                               \
        int x1 = wave_shr1_i32(i1, -0x3fffffff), x2 = wave_shr1_i32(i2, -0x3fffffff);                         \
        x1 = x1 > fcarry ? x1 : fcarry;                                                                       \
        x2 = x2 > scarry ? x2 : scarry;                                                                       \
        int l1_ = wave_readlane(i1, 63), l2_ = wave_readlane(i2, 63);                                                       \
        fcarry = fcarry > l1_ ? fcarry : l1_;                                                                 \
        scarry = scarry > l2_ ? scarry : l2_;                                                                 \
        int ffull = sat0(x1 - k * e_ins);                                                                     \
        int fseg = (x2 >= 0 && x2 / LH_SEG_BIG == sg##T) ? sat0(x2 % LH_SEG_BIG - LH_SEG_OFF - k * e_ins) : 0; \
        int hmain = hnf > fseg ? hnf : fseg;                                                                  \
        int hfull = hnf > ffull ? hnf : ffull;                                                                \
        if (!in) { hmain = 0; hfull = 0; }                                                                    \
        int rm = wave_max_i32(hmain);                                                                         \
        imax = imax > rm ? imax : rm;                                                                         \
        int en = sat0(e - e_del), t_ = sat0(hmain - oe_del);                                                  \
        if (in) { E##T = en > t_ ? en : t_; HN##T = hfull; }                                                  \
    }

// one pass of ksw_u8 over target rows [0,tlen): query column k = qarr[qoff + qstep*k] (complemented if qcomp),
// target row i = base at tcoord0 + tstep*i.  endsc: stop when the running maximum reaches it (KSW_XSTOP) or 0x10000.
__device__ __forceinline__ void wave_ksw_u8(const DIndex& ix, const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qcomp, int qlen, i64 tcoord0,
                                            int tstep, int tlen, int endsc, int lane, int* score_out, int* te_out, int* qe_out, u64* cells) {
    const int a_ = o.a, b_ = o.b, e_del = o.e_del, e_ins = o.e_ins, oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    int slen = (qlen + 15) / 16, ncol = slen * 16;
    int HP0 = 0, HP1 = 0, HP2 = 0, HP3 = 0, HN0 = 0, HN1 = 0, HN2 = 0, HN3 = 0, E0 = 0, E1 = 0, E2 = 0, E3 = 0;
    int HM0 = 0, HM1 = 0, HM2 = 0, HM3 = 0;   // Hmax row
    int qb0, qb1, qb2, qb3, sg0, sg1, sg2, sg3;
#define LH_U8_Q(T)                                                                     \
    {                                                                                  \
        int k = 64 * (T) + lane;                                                       \
        int v = k < qlen ? qarr[qoff + qstep * k] : 4;                                 \
        qb##T = (qcomp && v < 4) ? 3 - v : v;                                          \
        sg##T = k / slen;                                                              \
    }
    LH_U8_Q(0) LH_U8_Q(1) LH_U8_Q(2) LH_U8_Q(3)
#undef LH_U8_Q
    int gmax = 0, te = -1, tchunk = 4;
    u64 ncell = 0;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {
            int ii = i + lane;
            tchunk = ii < tlen ? dev_ref_base(ix, tcoord0 + (i64)tstep * ii) : 4;
        }
        int tb = wave_readlane(tchunk, i & 63);
        int imax = 0, hdcarry = 0, fcarry = -0x3fffffff, scarry = -0x3fffffff;
        LH_U8_SLAB(0) LH_U8_SLAB(1) LH_U8_SLAB(2) LH_U8_SLAB(3)
        ncell += (u64)ncol;
        HP0 = HN0; HP1 = HN1; HP2 = HN2; HP3 = HN3;
        if (imax > gmax) {
            gmax = imax; te = i;
            HM0 = HN0; HM1 = HN1; HM2 = HN2; HM3 = HN3;
            if (gmax >= endsc) break;
        }
    }
    if (cells) *cells += ncell;
    // qe: the smallest query index among the maxima of the saved row
    int best = -1;
    {
        int m0 = wave_max_i32(lane < ncol ? HM0 : -1), m1 = wave_max_i32(64 + lane < ncol ? HM1 : -1);
        int m2 = wave_max_i32(128 + lane < ncol ? HM2 : -1), m3 = wave_max_i32(192 + lane < ncol ? HM3 : -1);
        best = m0 > m1 ? m0 : m1; best = best > m2 ? best : m2; best = best > m3 ? best : m3;
    }
    int qe = -1;
    {
        u64 b0 = __ballot(lane < ncol && HM0 == best), b1 = __ballot(64 + lane < ncol && HM1 == best);
        u64 b2 = __ballot(128 + lane < ncol && HM2 == best), b3 = __ballot(192 + lane < ncol && HM3 == best);
        if (b0) qe = __ffsll((unsigned long long)b0) - 1;
        else if (b1) qe = 64 + __ffsll((unsigned long long)b1) - 1;
        else if (b2) qe = 128 + __ffsll((unsigned long long)b2) - 1;
        else if (b3) qe = 192 + __ffsll((unsigned long long)b3) - 1;
    }
    *score_out = gmax; *te_out = te; *qe_out = qe;
}

// ksw_align2 with KSW_XSUBO|KSW_XSTART|KSW_XBYTE|minsc (mem_matesw's call); the sub-optimal score (score2) is not
// computed: it only feeds mem_alnreg_t.csub, which nothing lariat reads depends on.
__device__ __forceinline__ KswR wave_ksw_align2(const DIndex& ix, const DOpts& o, const uint8_t* qarr, int qoff, int qstep, int qcomp, int qlen, i64 tcoord0,
                                                int tstep, int tlen, int minsc, int lane, u64* cells) {
    KswR r;
    r.tb = r.qb = -1;
    wave_ksw_u8(ix, o, qarr, qoff, qstep, qcomp, qlen, tcoord0, tstep, tlen, 0x10000, lane, &r.score, &r.te, &r.qe, cells);
    if (r.score < minsc) return r;
    // reverse pass over the prefixes ending at (te, qe) to find the start
    int s2, te2, qe2;
    wave_ksw_u8(ix, o, qarr, qoff + qstep * r.qe, -qstep, qcomp, r.qe + 1, tcoord0 + (i64)tstep * r.te, -tstep, r.te + 1, r.score, lane, &s2, &te2, &qe2, cells);
    if (r.score == s2) { r.tb = r.te - te2; r.qb = r.qe - qe2; }
    return r;
}

__device__ __forceinline__ int dev_infer_dir(i64 l_pac, i64 b1, i64 b2, i64* dist) {
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    i64 p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// mem_matesw with lariat's pes (only orientation r=1 enabled, gobwa.go:229-237).  ma/n_ma: the mate's region list.
__device__ __forceinline__ int wave_matesw(const DIndex& ix, const DOpts& o, const DReg& a, const uint8_t* ms, int l_ms, DReg* ma, int n_ma, int32_t* ia,
                                           DReg* tmp, int lane, u64* cells, int* n_sw) {
    i64 l_pac = ix.l_pac;
    int skip1 = 0;
    for (int i0 = 0; i0 < n_ma; i0 += 64) {   // which orientation has been found
        int i = i0 + lane, f = 0;
        if (i < n_ma) {
            i64 dist;
            int r = dev_infer_dir(l_pac, a.rb, ma[i].rb, &dist);
            f = (r == 1 && dist >= o.pes_low && dist <= o.pes_high);
        }
        if (__any(f)) { skip1 = 1; break; }
    }
    if (skip1) return n_ma;   // consistent pair exists; no need to perform SW
    // r = 1: is_rev = 1 (reverse-complement the mate), is_larger = 1
    i64 rb = a.rb + o.pes_low - l_ms, re = a.rb + o.pes_high;
    if (rb < 0) rb = 0;
    if (re > l_pac << 1) re = l_pac << 1;
    int rid = -1;
    if (rb < re) rid = dev_fetch_clamp(ix, &rb, (rb + re) >> 1, &re);
    if (a.rid == rid && re - rb >= o.min_seed_len) {
        (*n_sw)++;
        // query = revcomp(ms): column k = comp(ms[l_ms-1-k])
        KswR aln = wave_ksw_align2(ix, o, ms, l_ms - 1, -1, 1, l_ms, rb, 1, (int)(re - rb), o.min_seed_len * o.a, lane, cells);
        if (aln.score >= o.min_seed_len && aln.qb >= 0) {
            DReg b;
            b.rid = a.rid; b.is_alt = a.is_alt;
            b.qb = l_ms - (aln.qe + 1); b.qe = l_ms - aln.qb;
            b.rb = (l_pac << 1) - (rb + aln.te + 1); b.re = (l_pac << 1) - (rb + aln.tb);
            b.score = aln.score; b.csub = 0; b.secondary = -1;
            b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
            b.truesc = 0; b.sub = 0; b.w = 0; b.seedlen0 = 0; b.n_comp = 0; b.frac_rep = 0;
            // insert so that ma stays sorted by score: before the first element with a smaller score
            int pos = n_ma;
            for (int i0 = 0; i0 < n_ma; i0 += 64) {
                int i = i0 + lane;
                u64 bm = __ballot(i < n_ma && ma[i].score < b.score);
                if (bm) { pos = i0 + __ffsll((unsigned long long)bm) - 1; break; }
            }
            for (int top = n_ma; top > pos; top -= 64) {
                int j = top - 1 - lane;
                DReg v;
                if (j >= pos) v = ma[j];
                WAVE_SYNC();
                if (j >= pos) ma[j + 1] = v;
                WAVE_SYNC();
            }
            if (lane == 0) ma[pos] = b;
            n_ma++;
            WAVE_SYNC();
        }
        n_ma = wave_sort_dedup_patch(ix, o, ms, ma, n_ma, ia, tmp, 0, lane, cells);
    }
    return n_ma;
}

// K6.  grid = n_pairs waves.
__global__ void __launch_bounds__(64) k_rescue(DIndex ix, DOpts o, int n_pairs, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                                const i64* __restrict__ reg_off, DReg* __restrict__ regs, DReg* __restrict__ regs_tmp, int32_t* __restrict__ ia_pool,
                                                int32_t* __restrict__ n_regs, const int32_t* __restrict__ best_score, DCounters* __restrict__ ctr,
                                                const int32_t* __restrict__ list, const int32_t* __restrict__ list_count) {
    __shared__ uint8_t q1[LH_MAXLEN + 6];
    __shared__ uint8_t q2[LH_MAXLEN + 6];
    const int lane = LANE();
    const int n_items = *list_count;   // the pairs k_rescue_filter found a rescue attempt for
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int p = list[item];
    WAVE_SYNC();   // the previous pair's queries are no longer in use
    int r1 = 2 * p, r2 = 2 * p + 1;
    i64 off1 = seq_off[r1], off2 = seq_off[r2];
    int l1 = (int)(off2 - off1), l2 = (int)(seq_off[r2 + 1] - off2);
    if (l1 > LH_MAXLEN) l1 = 0;
    if (l2 > LH_MAXLEN) l2 = 0;
    for (int i = lane; i < l1; i += 64) q1[i] = seq[off1 + i];
    for (int i = lane; i < l2; i += 64) q2[i] = seq[off2 + i];
    WAVE_SYNC();
    i64 ro1 = reg_off[r1], ro2 = reg_off[r2];
    DReg *av1 = regs + ro1, *av2 = regs + ro2;
    int n1 = n_regs[r1], n2 = n_regs[r2];
    int best1 = best_score[r1], best2 = best_score[r2];
    u64 cells = 0;
    int n_sw = 0;
    // rescue read1 from read2's hits (gobwa.go:286-301)
    int num = 0;
    for (int i = 0; i < n2 && num < o.rescue_max_hits && l1 > 0; ++i) {
        DReg a = av2[i];
        if (a.score >= best2 - o.rescue_score_delta) {
            num++;
            n1 = wave_matesw(ix, o, a, q1, l1, av1, n1, ia_pool + ro1 + r1, regs_tmp + ro1, lane, &cells, &n_sw);
        }
    }
    // rescue read2 from read1's (post-rescue) hits; threshold from the pre-rescue best (gobwa.go:309-325)
    int n1_final = n1;
    num = 0;
    for (int i = 0; i < n1_final && num < o.rescue_max_hits && l2 > 0; ++i) {
        DReg a = av1[i];
        if (a.score >= best1 - o.rescue_score_delta) {
            num++;
            n2 = wave_matesw(ix, o, a, q2, l2, av2, n2, ia_pool + ro2 + r2, regs_tmp + ro2, lane, &cells, &n_sw);
        }
    }
    if (lane == 0) {
        n_regs[r1] = n1; n_regs[r2] = n2;
        if (ctr && n_sw) { atomicAdd(&LH_CTR(ctr)->n_rescue, (u64)n_sw); atomicAdd(&LH_CTR(ctr)->rescue_cells, cells); }
    }
    }
}

// One lane per pair: does any of the pair's rescue attempts (gobwa.go:286-325) get past mem_matesw's first test, "a
// consistent pair exists; no need to perform SW"?  Almost no pair does (12 per 10 k reads on the bench data), so the wave
// kernel above only sees the listed ones.  A pair that is not listed is left exactly as it is: without a rescue nothing
// is inserted, so the second loop of k_rescue sees the same regions this filter tests.
__global__ void __launch_bounds__(256) k_rescue_filter(DIndex ix, DOpts o, int n_pairs, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                        const DReg* __restrict__ regs, const int32_t* __restrict__ n_regs, const int32_t* __restrict__ best_score,
                                                        int32_t* __restrict__ list, int32_t* __restrict__ list_count) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int need = 0;
    if (p < n_pairs) {
        const int r1 = 2 * p, r2 = 2 * p + 1;
        const i64 off1 = seq_off[r1], off2 = seq_off[r2];
        int l1 = (int)(off2 - off1), l2 = (int)(seq_off[r2 + 1] - off2);
        if (l1 > LH_MAXLEN) l1 = 0;
        if (l2 > LH_MAXLEN) l2 = 0;
        const DReg *av1 = regs + reg_off[r1], *av2 = regs + reg_off[r2];
        const int n1 = n_regs[r1], n2 = n_regs[r2];
        const int best1 = best_score[r1], best2 = best_score[r2];
        for (int dir = 0; dir < 2 && !need; ++dir) {
            const DReg* from = dir ? av1 : av2;   // the anchors ...
            const DReg* ma = dir ? av2 : av1;     // ... and the mate's own regions
            const int nf = dir ? n1 : n2, nm = dir ? n2 : n1, bestf = dir ? best1 : best2, l_ms = dir ? l2 : l1;
            int num = 0;
            for (int i = 0; i < nf && num < o.rescue_max_hits && l_ms > 0 && !need; ++i) {
                const i64 arb = from[i].rb;
                if (from[i].score < bestf - o.rescue_score_delta) continue;
                num++;
                int skip1 = 0;
                for (int j = 0; j < nm && !skip1; ++j) {
                    i64 dist;
                    int r = dev_infer_dir(ix.l_pac, arb, ma[j].rb, &dist);
                    skip1 = (r == 1 && dist >= o.pes_low && dist <= o.pes_high);
                }
                if (!skip1) need = 1;
            }
        }
    }
    u64 m = __ballot(need);
    if (m) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(list_count, (int32_t)__popcll(m));
        basep = wave_readlane(basep, 0);
        if (need) list[basep + lanes_below(m, lane)] = p;
    }
}

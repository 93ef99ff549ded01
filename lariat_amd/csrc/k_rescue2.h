// k_rescue2.h — K6 when rescue is the bulk of the work (BASELINE configs[4]: reads on segmental duplications and repeat families, where
// GoBwaMemMateSW, gobwa.go:286-325, asks for up to 50 + 50 mem_matesw calls per pair — 100 k DP cells each).
//
// The loop of gobwa.go is sequential per pair (every attempt's first test looks at the regions rescued so far), but the Smith-Waterman of an
// attempt depends on its anchor alone: the window [a.rb + low - l_ms, a.rb + high) and the mate's bases.  So, per direction (read 1 from read 2's
// hits, then read 2 from read 1's post-rescue hits):
//   k_resc_count / k_resc_emit   one lane per pair: the attempts that get past mem_matesw's first test against the regions the mate has NOW
//                                become jobs, bucketed by the striping of their query (slen = ceil(qlen / 16));
//   k_resc_sw<false>             ksw_u8's forward pass over every job, 8 jobs per wave (below);
//   k_resc_sw<true>              the reverse pass (KSW_XSTART) of the jobs that reached min_seed_len, bucketed by THEIR striping;
//   k_resc_apply                 one wave per pair replays the loop in order with the jobs' results at hand: first test against the CURRENT
//                                regions, insert, mem_sort_dedup_patch — an attempt that has become unnecessary drops its job's result, one that
//                                has become necessary (a region that made it unnecessary was merged away) runs k_rescue.h's wave kernel in place.
//
// The DP.  ksw_u8 is Farrar's striped kernel: lane L of 16 owns query columns [L * slen, (L + 1) * slen), F runs along a stripe inside the
// main loop (F_seg: restarted at every stripe) and reaches later stripes only through the lazy-F loop, which repairs H but not E
// (k_rescue.h has the closed form: H_main = max(hnf, F_seg) feeds E and the row maximum, H' = max(hnf, F_full) is what the next row sees).
// Here a job is a SYSTOLIC ARRAY of the same 16 stripes: 16 lanes, lane L works on row s - L at step s, its slen columns one after the
// other in registers — F_seg is a running value, F_full is F_seg or the carry that entered from lane L - 1, decayed by e_ins per column (r05: saturating
// subtraction distributes over max), the carry and the diagonal H enter by one DPP row shift each per step, nothing is scanned and nothing goes through LDS
// but the row's reference base, stored as the byte selector with which one v_perm_b32 per column picks the match score from the query's profile.
// (Measured and dropped, r05: the reverse passes bucketed by score class as well as by striping, so that a wave's eight jobs stop at about the same row: no change.)  All cells are 16-bit halves of packed words
// (v_pk_max_u16, v_pk_sub_u16 clamp = the saturating arithmetic of the SSE code): a lane carries TWO jobs, a wave 8.
// Rows before a job's first and after its last are fed a target base that matches nothing: every cell of such a row is at most the
// value of an earlier cell, so neither "first row that beats the maximum" nor "first row that reaches endsc" can fire there.
#pragma once
#include "k_rescue.h"

#ifdef LH_EMU
__device__ __forceinline__ uint32_t pk_lanes(uint32_t a, uint32_t b, int op) {
    uint32_t r = 0;
    for (int h = 0; h < 2; ++h) {
        const uint32_t x = (a >> (16 * h)) & 0xffffu, y = (b >> (16 * h)) & 0xffffu;
        uint32_t v = op == 0 ? (x > y ? x : y) : op == 1 ? (x < y ? x : y) : op == 2 ? ((x + y) & 0xffffu) : op == 3 ? (x > y ? x - y : 0) : ((x - y) & 0xffffu);
        r |= v << (16 * h);
    }
    return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pk_lanes(a, b, 0); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return pk_lanes(a, b, 1); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_lanes(a, b, 2); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return pk_lanes(a, b, 3); }   // saturating at 0
__device__ __forceinline__ uint32_t pk_subw(uint32_t a, uint32_t b) { return pk_lanes(a, b, 4); }   // wrapping
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) { uint32_t o = __shfl_up(v, 1); return (LANE() & 15) == 0 ? 0u : o; }
// v_perm_b32: byte i of the result = byte sel[i] of {lo: 0..3, hi: 4..7}; selector 0x0c: 0x00 (the selectors 8..11 and >= 13 are not used here)
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) { const uint32_t c = (sel >> (8 * i)) & 0xffu; const uint32_t b = c < 4 ? (lo >> (8 * c)) & 0xffu : c < 8 ? (hi >> (8 * (c - 4))) & 0xffu : 0u; r |= b << (8 * i); }
    return r;
}
#else
typedef unsigned short lh_us2 __attribute__((ext_vector_type(2)));
#define LH_US2(x) __builtin_bit_cast(lh_us2, (uint32_t)(x))
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(LH_US2(a), LH_US2(b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(LH_US2(a), LH_US2(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (lh_us2)(LH_US2(a) + LH_US2(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(LH_US2(a), LH_US2(b))); }
__device__ __forceinline__ uint32_t pk_subw(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (lh_us2)(LH_US2(a) - LH_US2(b))); }
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); }   // row_shr:1, lane 0 of a row: 0
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif
__device__ __forceinline__ int grp16_min(int v) { for (int m = 8; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v < o ? v : o; } return v; }
__device__ __forceinline__ int grp16_max(int v) { for (int m = 8; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v > o ? v : o; } return v; }

// entries of the replay's region list in LDS (k_resc_apply)
#ifndef LH_RA_CAP
#define LH_RA_CAP 320   // measured 192 / 256 / 320 / 384: K6 390 / 323 / 296 / 297 ms on the repeat-family input (the pairs whose lists do not fit run every call from memory)
#endif
#define LH_RJ_TMAX (LH_MAXLEN + 560)        // longest window: pes_high - pes_low + l_ms (lariat: 535 + l_ms)
#define LH_RJ_TSLOT (16 + LH_RJ_TMAX + 16)  // a job's target bases in LDS: 16 non-matching rows in front and behind (the systolic skew)
#define LH_RJ_NB 17                         // buckets: slen 0 .. 16

struct RJob {
    i64 t0;                          // window start (fwd||rev coordinate): target row i = base at t0 + i
    int32_t q0;                      // index in seq[] of the mate's LAST base: query column k = complement of seq[q0 - k]
    int32_t pair;
    int16_t qlen, tlen, anchor;      // anchor: index of the anchoring region in the other mate's list
    int16_t score, te, qe, tb, qb;   // ksw_align2's result; tb = qb = -1: no start
    int16_t rows2;                   // rows the reverse pass executed (its cells: 16 * slen(qe + 1) * rows2)
    int16_t rlo, rn;                 // (r06, k_rescue3.h) the forward pass runs rows [rlo, rlo + rn) of the window: rn = tlen the whole of it, 0 none (no path reaches min_seed_len)
};
struct RMeta {   // per direction, on the device
    int32_t hist[LH_RJ_NB], bstart[LH_RJ_NB + 1], bcur[LH_RJ_NB];       // forward jobs by slen; bstart: bucket starts in the order array, each padded to 8
    int32_t hist2[LH_RJ_NB], bstart2[LH_RJ_NB + 1], bcur2[LH_RJ_NB];    // reverse jobs
    int32_t list_count, long_count;   // listed pairs; those of them whose mate's list is too long for the replay's usual LDS arrays
    int32_t heavy_count;              // pairs whose enumeration is a wave's (k_resc_enum_w)
    int32_t apply_next[2];            // k_resc_apply (the usual instance): the next list entry to look at
};

// which attempts of one pair and direction get past mem_matesw's first test ("a consistent pair exists; no need to perform SW") against the
// mate's regions as they are now, and have a window (bns_fetch_seq's clamp leaves the anchor's contig, at least min_seed_len bases)
struct RescWalk {
    const DReg *from, *ma;
    int nf, nm, bestf, l_ms, r_ms;
    __device__ __forceinline__ void init(int dir, int p, const i64* seq_off, const i64* reg_off, const DReg* regs, const int32_t* n_regs, const int32_t* best_score) {
        const int r1 = 2 * p, r2 = 2 * p + 1;
        int l1 = (int)(seq_off[r2] - seq_off[r1]), l2 = (int)(seq_off[r2 + 1] - seq_off[r2]);
        if (l1 > LH_MAXLEN) l1 = 0;
        if (l2 > LH_MAXLEN) l2 = 0;
        from = regs + reg_off[dir ? r1 : r2]; ma = regs + reg_off[dir ? r2 : r1];
        nf = n_regs[dir ? r1 : r2]; nm = n_regs[dir ? r2 : r1];
        bestf = best_score[dir ? r1 : r2]; l_ms = dir ? l2 : l1; r_ms = dir ? r2 : r1;
    }
};
__device__ __forceinline__ int resc_window(const DIndex& ix, const DOpts& o, const DReg& a, int l_ms, i64* rb_out, i64* re_out) {
    i64 rb = a.rb + o.pes_low - l_ms, re = a.rb + o.pes_high;
    if (rb < 0) rb = 0;
    if (re > ix.l_pac << 1) re = ix.l_pac << 1;
    int ok = 0;
    if (rb < re) {   // bns_fetch_seq clamps the window to the contig that holds its middle: the attempt goes on only if that is the anchor's contig, so its
        // bounds are all that is needed (no search for the contig of the middle: two dependent table reads per attempt)
        int is_rev;
        const i64 pm = dev_depos(ix, (rb + re) >> 1, &is_rev);
        const i64 coff = ix.contig_off[a.rid], cend = coff + ix.contig_len[a.rid];
        if (pm >= coff && pm < cend) {
            i64 far_beg = coff, far_end = cend;
            if (is_rev) { far_beg = (ix.l_pac << 1) - cend; far_end = (ix.l_pac << 1) - coff; }
            rb = rb > far_beg ? rb : far_beg;
            re = re < far_end ? re : far_end;
            ok = 1;
        }
    }
    *rb_out = rb; *re_out = re;
    return ok && re - rb >= o.min_seed_len;
}
// what the packed kernel holds: 8-bit scores next to an 8-bit column key, one-hot bases with a + b <= 16, windows of LH_RJ_TMAX rows, no N in the mate
__device__ __forceinline__ int resc_fast_ok(const DOpts& o, int l_ms, i64 tlen) {
    return l_ms * o.a < 250 && o.a + o.b <= 16 && o.a > 0 && o.b >= 0 && o.o_del + o.e_del < 256 && o.o_ins + o.e_ins < 256 && tlen <= LH_RJ_TMAX && l_ms >= 1;
}

#ifndef LH_RE_HEAVY
#define LH_RE_HEAVY 256   // anchors x mate regions from which on a pair's enumeration is a wave's (k_resc_enum_w)
#endif
// One lane per pair.  EMIT = false: count the pair's jobs (n_jobs[p]), their bucket's histogram, and list the pairs with any attempt to
// replay.  EMIT = true (after the scan of n_jobs and k_resc_offsets): write the jobs and their places in the order array.
template <int DIR, bool EMIT>
__global__ void __launch_bounds__(256) k_resc_enum(DIndex ix, DOpts o, int n_pairs, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                    const DReg* __restrict__ regs, const int32_t* __restrict__ n_regs, const int32_t* __restrict__ best_score, int32_t* __restrict__ n_jobs,
                                                    const i64* __restrict__ job_off, RJob* __restrict__ jobs, int32_t* __restrict__ order, RMeta* __restrict__ meta,
                                                    int32_t* __restrict__ list, int32_t* __restrict__ heavy) {
    __shared__ int32_t sh_hist[LH_RJ_NB];
    const int slot = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    // the counting pass looks at every pair; the emitting pass only at the pairs it listed (a few thousand of two million on unique sequence)
    // (EMIT: the pairs a wave enumerates — bit 29 of their entry — are that kernel's)
    const int p = EMIT ? (slot < meta->list_count && !(list[slot] & (1 << 29)) ? list[slot] & 0x1fffffff : n_pairs) : slot;
    if (!EMIT) { if (threadIdx.x < LH_RJ_NB) sh_hist[threadIdx.x] = 0; __syncthreads(); }
    int nj = 0, need = 0, slen = 0, obase = 0;
    if (EMIT) {   // the pair's places in the order array: one reservation per wave and bucket
        const int want = p < n_pairs ? n_jobs[p] : 0;
        if (want) { const i64 so = seq_off[DIR ? 2 * p + 1 : 2 * p]; int l = (int)(seq_off[(DIR ? 2 * p + 1 : 2 * p) + 1] - so); slen = ((l > LH_MAXLEN ? 0 : l) + 15) / 16; }
        u64 todo = __ballot(want > 0);
        while (todo) {
            const int lead = __ffsll((unsigned long long)todo) - 1;
            const int sl = wave_readlane(slen, lead);
            const int mine = want > 0 && slen == sl;
            const int incl = wave_scan_add_i32(mine ? want : 0);
            const int tot = wave_readlane(incl, 63);
            int base = 0;
            if (lane == 0) base = atomicAdd(&meta->bcur[sl], tot);
            base = wave_readlane(base, 0);
            if (mine) obase = meta->bstart[sl] + base + incl - want;
            todo &= ~__ballot(mine);
        }
    }
    int long_list = 0, is_heavy = 0;
    if (!EMIT) {   // anchors x mate regions beyond what one lane should walk (reads on repeat copies: a hundred of each): listed for k_resc_enum_w
        if (p < n_pairs) is_heavy = (i64)n_regs[DIR ? 2 * p : 2 * p + 1] * (i64)n_regs[DIR ? 2 * p + 1 : 2 * p] > LH_RE_HEAVY;
        const u64 hm = __ballot(is_heavy);
        if (hm) {
            int hb = 0;
            if (lane == 0) hb = atomicAdd(&meta->heavy_count, (int32_t)__popcll(hm));
            hb = wave_readlane(hb, 0);
            if (is_heavy) { heavy[hb + lanes_below(hm, lane)] = p; n_jobs[p] = 0; }
        }
    }
    if (p < n_pairs && !is_heavy) {
        RescWalk w;
        w.init(DIR, p, seq_off, reg_off, regs, n_regs, best_score);
        long_list = w.nm + o.rescue_max_hits > LH_RA_CAP;
        slen = (w.l_ms + 15) / 16;
        int has_n = -1;   // unknown
        i64 jbase = EMIT ? job_off[p] : 0;
        int num = 0;
        for (int i = 0; i < w.nf && num < o.rescue_max_hits && w.l_ms > 0; ++i) {
            const DReg a = w.from[i];
            if (a.score < w.bestf - o.rescue_score_delta) continue;
            num++;
            int skip1 = 0;
            for (int j = 0; j < w.nm && !skip1; ++j) {
                i64 dist;
                int r = dev_infer_dir(ix.l_pac, a.rb, w.ma[j].rb, &dist);
                skip1 = (r == 1 && dist >= o.pes_low && dist <= o.pes_high);
            }
            if (skip1) continue;
            i64 rb, re;
            if (!resc_window(ix, o, a, w.l_ms, &rb, &re)) continue;
            need = 1;
            if (!resc_fast_ok(o, w.l_ms, re - rb)) continue;
            if (has_n < 0) {   // an ambiguous base in the mate scores -1 against everything: left to the wave kernel (k_resc_apply)
                has_n = 0;
                const i64 so = seq_off[w.r_ms];
                for (int k = 0; k < w.l_ms; ++k) has_n |= seq[so + k] > 3;
            }
            if (has_n) continue;
            if (EMIT) {
                RJob jb;
                jb.t0 = rb; jb.q0 = (int32_t)(seq_off[w.r_ms] + w.l_ms - 1); jb.pair = p;
                jb.qlen = (int16_t)w.l_ms; jb.tlen = (int16_t)(re - rb); jb.anchor = (int16_t)i;
                jb.score = 0; jb.te = -1; jb.qe = -1; jb.tb = -1; jb.qb = -1; jb.rows2 = 0; jb.rlo = 0; jb.rn = jb.tlen;
                jobs[jbase + nj] = jb;
                order[obase + nj] = (int32_t)(jbase + nj);
            }
            nj++;
        }
        if (!EMIT) n_jobs[p] = nj;
    }
    if (!EMIT) {
        if (nj) atomicAdd(&sh_hist[slen], nj);
        u64 m = __ballot(need);
        if (m) {
            int basep = 0;
            const u64 ml = __ballot(need && long_list);
            if (lane == 0) { basep = atomicAdd(&meta->list_count, (int32_t)__popcll(m)); if (ml) atomicAdd(&meta->long_count, (int32_t)__popcll(ml)); }
            basep = wave_readlane(basep, 0);
            if (need) list[basep + lanes_below(m, lane)] = p | (long_list ? 1 << 30 : 0);   // bit 30: the mate's list will not fit the replay's usual LDS arrays (k_resc_apply)
        }
        __syncthreads();
        if (threadIdx.x < LH_RJ_NB && sh_hist[threadIdx.x]) atomicAdd(&meta->hist[threadIdx.x], sh_hist[threadIdx.x]);
    }
}

// The same enumeration for a pair with many anchors and many mate regions, by a whole wave (r05): a lane walking 100 x 100 region records alone was what the
// enumeration's 10 ms per launch were.  The mate's region starts in LDS, an anchor's first test against all of them with one ballot per 64.
#define LH_RE_CAP 1024
template <int DIR, bool EMIT>
__global__ void __launch_bounds__(64) k_resc_enum_w(DIndex ix, DOpts o, int n_pairs, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                     const DReg* __restrict__ regs, const int32_t* __restrict__ n_regs, const int32_t* __restrict__ best_score, int32_t* __restrict__ n_jobs,
                                                     const i64* __restrict__ job_off, RJob* __restrict__ jobs, int32_t* __restrict__ order, RMeta* __restrict__ meta,
                                                     int32_t* __restrict__ list, const int32_t* __restrict__ heavy) {
    __shared__ i64 mrb[LH_RE_CAP];
    const int lane = LANE();
    const int n_items = meta->heavy_count;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int p = heavy[item];
        RescWalk w;
        w.init(DIR, p, seq_off, reg_off, regs, n_regs, best_score);
        WAVE_SYNC();   // the previous pair's starts have been read
        const int ncap = w.nm < LH_RE_CAP ? w.nm : LH_RE_CAP;
        for (int j = lane; j < ncap; j += 64) mrb[j] = w.ma[j].rb;
        WAVE_SYNC();
        const int long_list = w.nm + o.rescue_max_hits > LH_RA_CAP;
        const int slen = (w.l_ms + 15) / 16;
        int nj = 0, need = 0, has_n = -1, num = 0, obase = 0;
        const i64 jbase = EMIT ? job_off[p] : 0;
        if (EMIT) {
            const int want = n_jobs[p];
            if (want) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&meta->bcur[slen], want);
                obase = meta->bstart[slen] + wave_readlane(base, 0);
            }
        }
        for (int i0 = 0; i0 < w.nf && num < o.rescue_max_hits && w.l_ms > 0; i0 += 64) {
            int a_sc = 0, a_rid = 0;
            i64 a_rb = 0;
            if (i0 + lane < w.nf) { const DReg& g = w.from[i0 + lane]; a_sc = g.score; a_rb = g.rb; a_rid = g.rid; }
            const int nu = w.nf - i0 < 64 ? w.nf - i0 : 64;
            for (int u = 0; u < nu && num < o.rescue_max_hits; ++u) {
                DReg a;
                a.score = wave_readlane(a_sc, u);
                if (a.score < w.bestf - o.rescue_score_delta) continue;
                num++;
                a.rb = (i64)((u64)(uint32_t)wave_readlane((int)((u64)a_rb >> 32), u) << 32 | (u64)(uint32_t)wave_readlane((int)(uint32_t)(u64)a_rb, u));
                a.rid = wave_readlane(a_rid, u);
                int skip1 = 0;
                for (int j0 = 0; j0 < w.nm && !skip1; j0 += 64) {
                    const int j = j0 + lane;
                    int f = 0;
                    if (j < w.nm) {
                        i64 dist;
                        const int r = dev_infer_dir(ix.l_pac, a.rb, j < ncap ? mrb[j] : w.ma[j].rb, &dist);
                        f = (r == 1 && dist >= o.pes_low && dist <= o.pes_high);
                    }
                    skip1 = __any(f);
                }
                if (skip1) continue;
                i64 rb, re;
                if (!resc_window(ix, o, a, w.l_ms, &rb, &re)) continue;
                need = 1;
                if (!resc_fast_ok(o, w.l_ms, re - rb)) continue;
                if (has_n < 0) {   // an ambiguous base in the mate scores -1 against everything: left to the wave kernel (k_resc_apply)
                    const i64 so = seq_off[w.r_ms];
                    int hn = 0;
                    for (int k = lane; k < w.l_ms; k += 64) hn |= seq[so + k] > 3;
                    has_n = __any(hn);
                }
                if (has_n) continue;
                if (EMIT && lane == 0) {
                    RJob jb;
                    jb.t0 = rb; jb.q0 = (int32_t)(seq_off[w.r_ms] + w.l_ms - 1); jb.pair = p;
                    jb.qlen = (int16_t)w.l_ms; jb.tlen = (int16_t)(re - rb); jb.anchor = (int16_t)(i0 + u);
                    jb.score = 0; jb.te = -1; jb.qe = -1; jb.tb = -1; jb.qb = -1; jb.rows2 = 0; jb.rlo = 0; jb.rn = jb.tlen;
                    jobs[jbase + nj] = jb;
                    order[obase + nj] = (int32_t)(jbase + nj);
                }
                nj++;
            }
        }
        if (!EMIT && lane == 0) {
            n_jobs[p] = nj;
            if (nj) atomicAdd(&meta->hist[slen], nj);
            if (need) {
                const int at = atomicAdd(&meta->list_count, 1);
                if (long_list) atomicAdd(&meta->long_count, 1);
                list[at] = p | (long_list ? 1 << 30 : 0) | 1 << 29;   // bit 29: enumerated by this kernel
            }
        }
    }
}

// bucket starts (each bucket padded to a multiple of 8 jobs: a wave's 8 jobs share their striping); which = 0: forward, 1: reverse
__global__ void k_resc_offsets(RMeta* __restrict__ meta, int which, i64* __restrict__ peek_host, const i64* __restrict__ total_jobs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t* h = which ? meta->hist2 : meta->hist;
    int32_t* bs = which ? meta->bstart2 : meta->bstart;
    int32_t* bc = which ? meta->bcur2 : meta->bcur;
    int acc = 0;
    for (int s = 0; s < LH_RJ_NB; ++s) { bs[s] = acc; bc[s] = 0; acc += (h[s] + 7) & ~7; }
    bs[LH_RJ_NB] = acc;
    if (peek_host) { peek_host[0] = *total_jobs; peek_host[1] = acc; peek_host[2] = meta->list_count; peek_host[3] = meta->long_count; }
}

// the reverse passes (jobs whose forward pass reached minsc) by the striping of THEIR query, the prefix that ends at qe.  A block takes a
// contiguous share of the jobs; SCATTER = false: histogram (one atomic per block and bucket), SCATTER = true (after k_resc_offsets): places.
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_resc_bucket2(i64 n_jobs, const RJob* __restrict__ jobs, int minsc, RMeta* __restrict__ meta, int32_t* __restrict__ order2) {
    __shared__ int32_t sh_n[LH_RJ_NB], sh_base[LH_RJ_NB];
    if (threadIdx.x < LH_RJ_NB) sh_n[threadIdx.x] = 0;
    __syncthreads();
    const i64 per = (n_jobs + gridDim.x - 1) / gridDim.x, j0 = per * blockIdx.x, j1 = j0 + per < n_jobs ? j0 + per : n_jobs;
    for (i64 j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const RJob& jb = jobs[j];
        if (jb.score < minsc || jb.te < 0 || jb.qe < 0 || jb.rlo < 0) continue;   // (rlo < 0: k_resc_cert settled the job, start included)
        atomicAdd(&sh_n[(jb.qe + 1 + 15) / 16], 1);
    }
    __syncthreads();
    if (!SCATTER) {
        if (threadIdx.x < LH_RJ_NB && sh_n[threadIdx.x]) atomicAdd(&meta->hist2[threadIdx.x], sh_n[threadIdx.x]);
        return;
    }
    if (threadIdx.x < LH_RJ_NB) { sh_base[threadIdx.x] = sh_n[threadIdx.x] ? meta->bstart2[threadIdx.x] + atomicAdd(&meta->bcur2[threadIdx.x], sh_n[threadIdx.x]) : 0; sh_n[threadIdx.x] = 0; }
    __syncthreads();
    for (i64 j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const RJob& jb = jobs[j];
        if (jb.score < minsc || jb.te < 0 || jb.qe < 0 || jb.rlo < 0) continue;   // (rlo < 0: k_resc_cert settled the job, start included)
        const int s2 = (jb.qe + 1 + 15) / 16;
        order2[sh_base[s2] + atomicAdd(&sh_n[s2], 1)] = (int32_t)j;
    }
}

// ksw_u8 for the 8 jobs ord[0..7] (-1: none), all of striping SLEN.  REV: the pass of KSW_XSTART over the reversed prefixes that end at
// the forward pass's (te, qe), stopped at the first row that reaches the forward score.
template <int SLEN, bool REV>
__device__ __forceinline__ void resc_sw_run(const DIndex& ix, const DOpts& o, RJob* __restrict__ jobs, const int32_t* __restrict__ ord, const uint8_t* __restrict__ seq, uint8_t* tl,
                                            RMeta* __restrict__ meta, int lane) {
    const int g = lane >> 4, L = lane & 15;
    // ---- the windows' bases into LDS as the BYTE SELECTOR the row's v_perm needs: row i of slot u at tl[u * LH_RJ_TSLOT + 16 + i] = base (the wave's even slots: a lane's
    // low half) or 4 + base (odd slots: its high half); 0x0c — "a zero byte" — outside the window
    for (int i = lane; i < 8 * LH_RJ_TSLOT / 4; i += 64) ((uint32_t*)tl)[i] = 0x0c0c0c0cu;
    WAVE_SYNC();
    int max_t = 0;
    for (int u = 0; u < 8; ++u) {
        const int ju = ord[u];
        if (ju < 0) continue;
        const RJob& J = jobs[ju];
        const int tlen = REV ? J.te + 1 : J.rn;              // (r06) forward: the rows k_resc_cert left (k_rescue3.h)
        const i64 t0 = REV ? J.t0 + J.te : J.t0 + J.rlo;
        for (int i = lane; i < tlen; i += 64) tl[u * LH_RJ_TSLOT + 16 + i] = (uint8_t)((u & 1) * 4 + dev_ref_base(ix, REV ? t0 - i : t0 + i));
        max_t = max_t > tlen ? max_t : tlen;
    }
    WAVE_SYNC();
    // ---- this lane's two jobs (halves of every packed word) and its columns
    const int jA = ord[2 * g], jB = ord[2 * g + 1];
    int qlenA = 0, qlenB = 0, endA = 0x100, endB = 0x100;
    i64 qbA = 0, qbB = 0;
    if (jA >= 0) { const RJob& J = jobs[jA]; qlenA = REV ? J.qe + 1 : J.qlen; qbA = REV ? (i64)J.q0 - J.qe : (i64)J.q0; endA = J.score; }
    if (jB >= 0) { const RJob& J = jobs[jB]; qlenB = REV ? J.qe + 1 : J.qlen; qbB = REV ? (i64)J.q0 - J.qe : (i64)J.q0; endB = J.score; }
    // (r05) the query as a PROFILE: byte t of PA[j] / PB[j] = a + b if column j's base pairs with reference base t (the mate is read as its reverse complement: base v
    // pairs with 3 - v), else 0; one v_perm_b32 per column picks both jobs' bytes for the row's two reference bases (was: one-hot AND, then a packed min)
    uint32_t Hp[SLEN], E[SLEN], PA[SLEN], PB[SLEN], Bc[SLEN];
    const uint32_t ab = (uint32_t)(o.a + o.b);
#pragma unroll
    for (int j = 0; j < SLEN; ++j) {
        const int k = L * SLEN + j;
        uint32_t pa = 0, pb = 0, bc = 0;
        if (k < qlenA) { const int v = seq[REV ? qbA + k : qbA - k]; pa = v < 4 ? ab << (8 * (3 - v)) : 0u; bc |= (uint32_t)(v < 4 ? o.b : 1); }
        if (k < qlenB) { const int v = seq[REV ? qbB + k : qbB - k]; pb = v < 4 ? ab << (8 * (3 - v)) : 0u; bc |= (uint32_t)(v < 4 ? o.b : 1) << 16; }
        Hp[j] = 0; E[j] = 0; PA[j] = pa; PB[j] = pb; Bc[j] = bc;
    }
    const uint32_t c_ed = (uint32_t)o.e_del * 0x10001u, c_oed = (uint32_t)(o.o_del + o.e_del) * 0x10001u;
    const uint32_t c_ei = (uint32_t)o.e_ins * 0x10001u, c_oei = (uint32_t)(o.o_ins + o.e_ins) * 0x10001u;
    uint32_t pubH = 0, pubF = 0, inH = 0, brow = 0, bkey = 0, fnd = 0;
    uint32_t bestv = REV ? ((uint32_t)(endA - 1) << 8 | (uint32_t)(endB - 1) << 24) : 0u;   // REV: "greater than endsc - 1" = "reaches endsc"
    const uint8_t* tA = tl + (2 * g) * LH_RJ_TSLOT + 16 - L;
    const uint8_t* tB = tl + (2 * g + 1) * LH_RJ_TSLOT + 16 - L;
    const int nsteps = max_t + 15;
    for (int s = 0; s < nsteps; ++s) {
        const uint32_t fin = dpp_row_shr1(pubF), hin = dpp_row_shr1(pubH);   // lane L - 1's row s - L: its last column's H', its F_full carry
        uint32_t hd = inH;                                                   // ... and its row s - L - 1: the diagonal of this lane's first column
        inH = hin;
        const uint32_t sel = 0x0c000c00u | (uint32_t)tA[s] | (uint32_t)tB[s] << 16;   // result bytes: PA[tA], 0, PB[tB], 0
        // (r05) the F that carries across stripes is the stripe's own F or what came in from the left, decayed: F_full(j) = max(F_seg(j), fin - j * e_ins) — both chains apply
        // the same max(f - e_ins, u) with saturating subtraction, which distributes over max — so the second chain is one subtraction per column, not a subtraction and a max
        uint32_t fseg = 0, carry = fin, rowkey = 0;
#pragma unroll
        for (int j = 0; j < SLEN; ++j) {
            const uint32_t m = perm_b32(PB[j], PA[j], sel);                  // a + b where the bases match
            uint32_t hnf = pk_subs(pk_add(hd, m), Bc[j]);                    // H'(i-1,k-1) + S, floored at 0 (S = a | -b | -1 | 0 past the query)
            hnf = pk_max(hnf, E[j]);
            const uint32_t hmain = pk_max(hnf, fseg), hfull = pk_max(hmain, carry);
            rowkey = pk_max(rowkey, hmain << 8 | (uint32_t)(255 - j) * 0x10001u);   // the row's maximum and, in its low byte, the smallest column that has it
            E[j] = pk_max(pk_subs(E[j], c_ed), pk_subs(hmain, c_oed));
            fseg = pk_max(pk_subs(fseg, c_ei), pk_subs(hnf, c_oei));
            carry = pk_subs(carry, c_ei);
            hd = Hp[j]; Hp[j] = hfull;
        }
        pubH = Hp[SLEN - 1]; pubF = pk_max(fseg, carry);
        // the stripe's first row that beats its maximum so far (REV: that reaches endsc; then it is locked)
        const uint32_t nb = pk_max(bestv, rowkey & 0xff00ff00u);
        const uint32_t msk = pk_subw(0u, pk_min(nb ^ bestv, 0x10001u));     // 0xffff in the halves that improved
        const uint32_t rowpk = ((uint32_t)(s - L) & 0xffffu) * 0x10001u;
        brow = (brow & ~msk) | (rowpk & msk);
        bkey = (bkey & ~msk) | (rowkey & msk);
        bestv = REV ? (nb | (msk & 0xff00ff00u)) : nb;
        if (REV) {
            fnd |= msk;
            if ((s & 7) == 7) {   // a job is settled once every stripe has passed the earliest row found: s - 15 >= that row
                const int rA = (fnd & 0xffffu) ? (int)(brow & 0xffffu) : 0x7fff, rB = (fnd >> 16) ? (int)(brow >> 16) : 0x7fff;
                const int mA = grp16_min(rA), mB = grp16_min(rB);
                const int doneA = jA < 0 || mA + 15 <= s, doneB = jB < 0 || mB + 15 <= s;
                if (__all(doneA && doneB)) break;
            }
        }
    }
    // ---- the 16 stripes' answers -> the job's
    for (int h = 0; h < 2; ++h) {
        const int jj = h ? jB : jA;
        const int v = (int)((bkey >> (16 * h + 8)) & 0xffu), r = (int)((brow >> (16 * h)) & 0xffffu), kc = (int)((bkey >> (16 * h)) & 0xffu);
        const int hit = REV ? (int)((fnd >> (16 * h)) & 1u) : (v > 0);
        // FWD: te = the first row of the global maximum; REV: the first row that reached endsc, its maximum over the stripes that did so there
        const int vmax = grp16_max(hit ? v : 0);
        const int te = grp16_min(hit && (REV || v == vmax) ? r : 0x7fff);
        const int gmax = REV ? grp16_max(hit && r == te ? v : 0) : vmax;
        const int win = hit && v == gmax && r == te;
        const int wl = grp16_min(win ? L : 16);   // the lowest stripe holds the smallest column
        if (jj >= 0 && L == (wl < 16 ? wl : 0)) {
            RJob& J = jobs[jj];
            const int qe = wl < 16 ? L * SLEN + (255 - kc) : -1;
            if (!REV) {
                J.score = (int16_t)gmax; J.te = (int16_t)(wl < 16 ? te + J.rlo : -1); J.qe = (int16_t)qe;
            } else {
                const int tlen2 = J.te + 1;
                J.rows2 = (int16_t)(wl < 16 ? te + 1 : tlen2);
                if (wl < 16 && gmax == J.score) { J.tb = (int16_t)(J.te - te); J.qb = (int16_t)(J.qe - qe); }
            }
        }
    }
}

template <bool REV>
__global__ void __launch_bounds__(64) k_resc_sw(DIndex ix, DOpts o, RJob* __restrict__ jobs, const int32_t* __restrict__ order, const uint8_t* __restrict__ seq, RMeta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint8_t tl[8 * LH_RJ_TSLOT];
    const int lane = LANE();
    const int32_t* bs = REV ? meta->bstart2 : meta->bstart;
    const int first = blockIdx.x * 8;
    if (first >= bs[LH_RJ_NB]) return;
    int slen = 0;
    for (int s = 1; s < LH_RJ_NB; ++s) if (first >= bs[s]) slen = s;   // the bucket this wave's jobs are in
    const int32_t* ord = order + first;
    switch (slen) {
#define LH_RSW(n) case n: resc_sw_run<n, REV>(ix, o, jobs, ord, seq, tl, meta, lane); break;
        LH_RSW(1) LH_RSW(2) LH_RSW(3) LH_RSW(4) LH_RSW(5) LH_RSW(6) LH_RSW(7) LH_RSW(8) LH_RSW(9) LH_RSW(10) LH_RSW(11) LH_RSW(12) LH_RSW(13) LH_RSW(14) LH_RSW(15) LH_RSW(16)
#undef LH_RSW
        default: break;
    }
}

// ---- k_resc_apply: one wave per listed pair replays mem_matesw's loop for one direction (gobwa.go:286-301 or 309-325) with the results of
// the pair's jobs.
//
// What the loop costs is not the Smith-Waterman any more but what follows every attempt: mem_sort_dedup_patch(opt, 0, 0, 0, n, a) — two
// introsorts of the mate's whole region list and a scan in between, ~30 times per pair over ~100 regions.  After the FIRST of those calls the
// list is in a state in which a further call is a function of the one region that has been added (nothing is patched with bns = 0):
//   * the list L is sorted by (score desc, rb, qb) and no two of its entries are "redundant" (overlap > mask_level_redun of the shorter one on
//     reference and query) within max_chain_gap of each other: every such pair was compared by the scan and one of the two excluded;
//   * so with the new region b in place, the scan (entries by increasing re; each p against the entries to its left while they are within
//     max_chain_gap and of p's contig) only does something in comparisons that involve b.  p = b: going left from b, every redundant q with
//     score <= b's is excluded, until a redundant q with a higher score excludes b (and ends b's loop).  p to the right of b, by increasing
//     re, while b is alive: redundant and p.score < b.score -> p is excluded, else b is;
//   * the final sort puts the survivors back in (score desc, rb, qb) order, which they are in but for b: b is inserted at its place.
// When all re are distinct and no entry has b's (score, rb, qb), the introsorts' handling of equal keys cannot matter, and the above IS the
// call's result: a few reductions over the list, which lives in LDS.  A call after an attempt that added nothing changes nothing.  Anything
// else — equal keys, a list that outgrows the LDS arrays — goes back to wave_sort_dedup_patch on the arrays in memory, for the rest of the pair.
#ifndef LH_RA_CAP_BIG
#define LH_RA_CAP_BIG 1024   // the few pairs whose lists are longer run in a second instance of the kernel with room for them (a pair that ran every call from memory — two
#endif                       // single-lane introsorts of 400 regions, 50 times — took 25-50 ms: the kernel's duration; the sum of all waves' time was 7 ms of the chip)
template <int CAP> struct RescListT {
    i64 rb[CAP], re[CAP];   // (first: wave_sort_dedup_patch's scratch while the list is in memory)
    int32_t qb[CAP], qe[CAP], score[CAP], rid[CAP], src[CAP];   // src: the entry's place in the memory arrays when the list was loaded; -1: rescued since
    uint8_t tied[CAP];      // the entry shares its re with another entry of its contig (a HARMLESS tie, see resc_list_ties: the two are not redundant whichever comes first)
};
typedef RescListT<LH_RA_CAP> RescList;
// "one of the hits is redundant": q = the entry with the smaller re (mem_sort_dedup_patch's a[j]), p = the one with the larger
__device__ __forceinline__ int resc_redundant(const DOpts& o, i64 q_rb, i64 q_re, int q_qb, int q_qe, i64 p_rb, i64 p_re, int p_qb, int p_qe) {
    const i64 orr = q_re - p_rb;
    const i64 oq = q_qb < p_qb ? q_qe - p_qb : p_qe - q_qb;
    const i64 mr = q_re - q_rb < p_re - p_rb ? q_re - q_rb : p_re - p_rb;
    const i64 mq = q_qe - q_qb < p_qe - p_qb ? q_qe - q_qb : p_qe - p_qb;
    return (float)orr > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq;
}

// the list in LDS into the order mem_sort_dedup_patch leaves it in — (score desc, rb, qb), all keys distinct while the list lives in LDS —: every
// entry's rank is the number of entries before it.  (resc_dedup_incremental appends: the order only matters when the list goes back to memory.)
template <int CAP> __device__ __forceinline__ void resc_list_sort(RescListT<CAP>& W, int n, int lane) {
    constexpr int PER = (CAP + 63) / 64;
    i64 e[PER], krb[PER]; int kqb[PER], kqe[PER], ksc[PER], krid[PER], ksrc[PER], rank[PER];
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = t * 64 + lane;
        rank[t] = -1;
        if (k < n) {
            e[t] = W.re[k]; krb[t] = W.rb[k]; kqb[t] = W.qb[k]; kqe[t] = W.qe[k]; ksc[t] = W.score[k]; krid[t] = W.rid[k]; ksrc[t] = W.src[k];
            int rk = 0;
            for (int u = 0; u < n; ++u) {
                const int us = W.score[u];
                rk += us > ksc[t] || (us == ksc[t] && (W.rb[u] < krb[t] || (W.rb[u] == krb[t] && W.qb[u] < kqb[t])));
            }
            rank[t] = rk;
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < PER; ++t)
        if (rank[t] >= 0) { const int d = rank[t]; W.re[d] = e[t]; W.rb[d] = krb[t]; W.qb[d] = kqb[t]; W.qe[d] = kqe[t]; W.score[d] = ksc[t]; W.rid[d] = krid[t]; W.src[d] = ksrc[t]; }   // (W.tied is not carried: the list leaves LDS after this)
    WAVE_SYNC();
}
// the list in LDS back into the memory arrays: entries that were there when it was loaded (src >= 0) keep their other fields, rescued ones get mem_matesw's
template <int CAP> __device__ __forceinline__ void resc_list_store(const DIndex& ix, RescListT<CAP>& W, int n, DReg* ma, DReg* tmp, int lane, int sorted) {
    if (!sorted) resc_list_sort(W, n, lane);
    WAVE_SYNC();
    for (int k = lane; k < n; k += 64) {
        DReg g;
        const int s = W.src[k];
        if (s >= 0) g = ma[s];
        else {
            g.rid = W.rid[k]; g.is_alt = ix.contig_alt ? ix.contig_alt[g.rid] : 0;
            g.csub = 0; g.secondary = -1; g.truesc = 0; g.sub = 0; g.w = 0; g.seedlen0 = 0; g.frac_rep = 0;
        }
        g.rb = W.rb[k]; g.re = W.re[k]; g.qb = W.qb[k]; g.qe = W.qe[k]; g.score = W.score[k];
        if (s < 0) g.seedcov = (int)((g.re - g.rb < g.qe - g.qb ? g.re - g.rb : g.qe - g.qb) >> 1);
        g.n_comp = 1;
        tmp[k] = g;
    }
    WAVE_SYNC();
    for (int k = lane; k < n; k += 64) ma[k] = tmp[k];
    WAVE_SYNC();
}

// mem_sort_dedup_patch(opt, 0, 0, 0, n + 1, list + b) for a CLEAN list in W (see above; in any order) and a new region b, in place: the survivors keep
// their places, b — if it survives — goes to the end (*appended).  Returns the new length, or -1 (W untouched) when equal keys make the introsorts'
// order matter or the list is full: the caller runs the call as written.  One pass finds what decides b's fate (the nearest redundant entry with a
// higher score to the left, the nearest redundant one that is not worse to the right, the nearest entries of another contig: the scans stop there),
// a second one the entries b excludes; the list is only compacted when there are any.
template <int CAP> __device__ __forceinline__ int resc_dedup_incremental(const DOpts& o, RescListT<CAP>& W, int n_ma, const DReg& b, int lane, int* appended) {
    const i64 NINF = -0x7fffffffffffffffll, PINF = 0x7fffffffffffffffll;
    i64 lo_bar = NINF, hi_bar = PINF, r_cand = NINF, s_cand = PINF;
    int tie = 0, same = 0;
    for (int i0 = 0; i0 < n_ma; i0 += 64) {
        const int k = i0 + lane;
        if (k < n_ma) {
            const i64 e = W.re[k], krb = W.rb[k];
            const int ksc = W.score[k], kqb = W.qb[k];
            // (r05) b IS an entry, and one that was rescued earlier in this replay (src < 0: its record is b's, field for field): the two are redundant with equal scores, the
            // scan excludes whichever the first sort puts first and keeps the other — the same record either way — and neither does anything to a third entry that the
            // entry did not do in the call before: the list stays as it is.  (An entry that was there from the start carries its own seedcov, w, ..: which of the two
            // survives is the introsort's to say: declined.)
            const int twin = e == b.re && krb == b.rb && kqb == b.qb && W.qe[k] == b.qe && ksc == b.score && W.rid[k] == b.rid && W.src[k] < 0 && !W.tied[k];
            same |= twin;
            tie |= !twin && (e == b.re || (ksc == b.score && krb == b.rb && kqb == b.qb));
            // an entry that shares its re with another one: which of the two b's scan (or theirs, of b) meets first is the introsort's to say: the call as written
            tie |= W.tied[k] && (e < b.re ? b.rb < e + o.max_chain_gap : krb < b.re + o.max_chain_gap);
            if (W.rid[k] != b.rid) { if (e < b.re && e > lo_bar) lo_bar = e; if (e > b.re && e < hi_bar) hi_bar = e; }
            else if (e < b.re) {
                if (b.rb < e + o.max_chain_gap && b.score < ksc && e > r_cand && resc_redundant(o, krb, e, kqb, W.qe[k], b.rb, b.re, b.qb, b.qe)) r_cand = e;
            } else {
                if (krb < b.re + o.max_chain_gap && !(ksc < b.score) && e < s_cand && resc_redundant(o, b.rb, b.re, b.qb, b.qe, krb, e, kqb, W.qe[k])) s_cand = e;
            }
        }
    }
    if (__any(tie) || n_ma + 1 > CAP) return -1;
    if (__any(same)) { *appended = 0; return n_ma; }
    lo_bar = wave_max_i64(lo_bar); hi_bar = wave_min_i64(hi_bar);
    r_cand = wave_max_i64(r_cand); s_cand = wave_min_i64(s_cand);
    const i64 r_star = r_cand > lo_bar ? r_cand : NINF;   // (beyond an entry of another contig the scan to the left never gets)
    const int b_dead_a = r_star != NINF;
    const i64 s_star = !b_dead_a && s_cand < hi_bar ? s_cand : PINF;
    const int b_alive = !b_dead_a && s_star == PINF;
    int n_new = 0, compact = 0;
    for (int i0 = 0; i0 < n_ma; i0 += 64) {   // the entries b excludes: to its left while it goes left (down to r_star), to its right while it is alive (up to s_star)
        const int k = i0 + lane;
        int dead = 0;
        i64 e = 0, krb = 0; int kqb = 0, kqe = 0, ksc = 0, krid = 0, ksrc = 0, ktie = 0;
        if (k < n_ma) {
            e = W.re[k]; krb = W.rb[k]; kqb = W.qb[k]; kqe = W.qe[k]; ksc = W.score[k]; krid = W.rid[k];
            if (e < b.re && e > lo_bar && e > r_star && krid == b.rid && b.rb < e + o.max_chain_gap && !(b.score < ksc))
                dead = resc_redundant(o, krb, e, kqb, kqe, b.rb, b.re, b.qb, b.qe);
            else if (!b_dead_a && e > b.re && e < hi_bar && e < s_star && krid == b.rid && krb < b.re + o.max_chain_gap && ksc < b.score)
                dead = resc_redundant(o, b.rb, b.re, b.qb, b.qe, krb, e, kqb, kqe);
        }
        const u64 mk = __ballot(k < n_ma && !dead);
        compact |= __any(dead);
        if (compact) {   // (wave-uniform) from the first chunk with an excluded entry on, the survivors move up
            if (k < n_ma) { ksrc = W.src[k]; ktie = W.tied[k]; }
            WAVE_SYNC();   // every lane holds its entry before any is moved
            if (k < n_ma && !dead) { const int d = n_new + lanes_below(mk, lane); W.re[d] = e; W.rb[d] = krb; W.qb[d] = kqb; W.qe[d] = kqe; W.score[d] = ksc; W.rid[d] = krid; W.src[d] = ksrc; W.tied[d] = (uint8_t)ktie; }
            WAVE_SYNC();
        }
        n_new += __popcll(mk);
    }
    n_ma = n_new;
    *appended = b_alive;
    if (b_alive) {
        if (lane == 0) { W.re[n_ma] = b.re; W.rb[n_ma] = b.rb; W.qb[n_ma] = b.qb; W.qe[n_ma] = b.qe; W.score[n_ma] = b.score; W.rid[n_ma] = b.rid; W.src[n_ma] = -1; W.tied[n_ma] = 0; }
        n_ma++;
        WAVE_SYNC();
    }
    return n_ma;
}

// Equal end positions in a list just loaded into W (n entries, any order).  mem_sort_dedup_patch sorts by re with an unstable sort: which of two entries
// with the same re its scan takes as "p" and which as "q" is the introsort's to say, call by call.  If the two — of one contig — are not redundant WHICHEVER comes first
// (the rule's two overlap tests fail both ways) no call ever does anything to either on the other's account: the tie is harmless
// and the entries are only marked (W.tied: a later region that comes near one of them sends its call to the code as written).  Returns 1 if the list
// holds a tie that is not harmless (the calls run as written from then on), else 0.
template <int CAP> __device__ __forceinline__ int resc_list_ties(const DOpts& o, RescListT<CAP>& W, int n, int lane) {
    int harmful = 0;
    for (int k = lane; k < n; k += 64) {
        const i64 pre = W.re[k], prb = W.rb[k];
        const int pqb = W.qb[k], pqe = W.qe[k], prid = W.rid[k];
        int t = 0;
        for (int u = 0; u < n; ++u) {
            if (u == k || W.re[u] != pre) continue;
            t = 1;
            // (an equal re on ANOTHER contig cannot happen with real coordinates — a position belongs to one contig — but if it did, the entry would end
            // the scans of its neighbours' contig or not, depending on where the sort puts it: pairs never compared before would be: not harmless)
            if (W.rid[u] != prid) harmful = 1;
            else harmful |= resc_redundant(o, W.rb[u], pre, W.qb[u], W.qe[u], prb, pre, pqb, pqe) || resc_redundant(o, prb, pre, pqb, pqe, W.rb[u], pre, W.qb[u], W.qe[u]);
        }
        W.tied[k] = (uint8_t)t;
    }
    return __any(harmful);
}

// ---- (r05) the call AS WRITTEN on the list in LDS.  A call that resc_dedup_incremental declines (the added region shares its end position, or its whole sort key, with
// an entry: typically it IS an entry — the window of a later anchor holds a region the mate already has, just outside the insert-size bounds of the first test) used to
// send the list to memory for wave_sort_dedup_patch and back: a serial introsort and a scan whose every step reads two 72-byte records from memory.  A pair in which one
// attempt does that, does it in most of its fifty: 0.1 % of the pairs took 5 .. 40 ms each while the others take 0.15 ms, and k_resc_apply lasted as long as its slowest
// pair (profiles/r05_resc_apply_pair_times.log).  Here the same call — same two sorts (ranked when all keys differ, else klib's introsort on the same packed keys by the whole wave, lh_sort.h,
// so equal keys end up where klib leaves them), same scan, same removal of identical hits — reads and writes LDS only, and the scan's walk to the left looks at 64 entries at
// once: for a fixed p no step depends on an earlier step of the same walk (without patching p never changes, and excluding q only concerns q), so the walk ends at the nearest
// entry that is off p's contig / beyond max_chain_gap / redundant with a higher score, and every redundant live entry nearer than that is excluded.
template <int CAP> struct RescScratchT { i64 lk[CAP]; uint16_t ord[CAP], ord2[CAP]; };
// lk[0, n): keys with their index in the low ib bits; all different above them: ord[rank] = index (one sorted order whatever the algorithm), returns 1; else returns 0
__device__ __forceinline__ int resc_rank_keys(const i64* lk, int n, int ib, int lane, uint16_t* ord) {
    int tie = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        if (e < n) {
            const i64 key = lk[e], k = key >> ib;
            int rk = 0;
            for (int j = 0; j < n; ++j) { const i64 kj = lk[j] >> ib; rk += kj < k; tie |= (kj == k) & (j != e); }
            ord[rk] = (uint16_t)(key & (((i64)1 << ib) - 1));
        }
    }
    WAVE_SYNC();
    return !__any(tie);
}
// can the second sort's packed key — (largest score - score) | rb (33 bits) | qb (8 bits) | index — hold every entry of the list and b?  (else: the call on the records in memory)
template <int CAP> __device__ __forceinline__ int resc_lds_keys_ok(const RescListT<CAP>& W, int n, const DReg* b, int lane, int* smax_out) {
    int smax = 0, bad = 0;
    if (b) { smax = b->score; bad = b->score < 0 || b->rb < 0 || b->rb >= (1ll << 33) || b->qb < 0 || b->qb > 255; }
    for (int k = lane; k < n; k += 64) {
        const int sc = W.score[k];
        smax = smax > sc ? smax : sc;
        bad |= sc < 0 || W.rb[k] < 0 || W.rb[k] >= (1ll << 33) || W.qb[k] < 0 || W.qb[k] > 255;
    }
    smax = wave_max_i32(smax);
    const int ib = n + 1 <= 512 ? 9 : 11;
    *smax_out = smax;
    return !__any(bad) && (32 - __clz(smax | 1)) + 33 + 8 + ib <= 63 && n + 1 <= CAP && n + 1 <= 2048;
}
// b into the list in its final order (score desc, rb, qb), before the first entry with a smaller score (mem_matesw); returns the new length
template <int CAP> __device__ __forceinline__ int resc_list_insert(RescListT<CAP>& W, int n, const DReg& b, int lane) {
    constexpr int PER = (CAP + 63) / 64;
    int pos = n;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int k = i0 + lane;
        const u64 bm = __ballot(k < n && W.score[k] < b.score);
        if (bm) { pos = i0 + __ffsll((unsigned long long)bm) - 1; break; }
    }
    i64 e[PER], krb[PER]; int kqb[PER], kqe[PER], ksc[PER], krid[PER], ksrc[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = t * 64 + lane;
        if (k >= pos && k < n) { e[t] = W.re[k]; krb[t] = W.rb[k]; kqb[t] = W.qb[k]; kqe[t] = W.qe[k]; ksc[t] = W.score[k]; krid[t] = W.rid[k]; ksrc[t] = W.src[k]; }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = t * 64 + lane;
        if (k >= pos && k < n) { const int d = k + 1; W.re[d] = e[t]; W.rb[d] = krb[t]; W.qb[d] = kqb[t]; W.qe[d] = kqe[t]; W.score[d] = ksc[t]; W.rid[d] = krid[t]; W.src[d] = ksrc[t]; }
    }
    if (lane == 0) { W.re[pos] = b.re; W.rb[pos] = b.rb; W.qb[pos] = b.qb; W.qe[pos] = b.qe; W.score[pos] = b.score; W.rid[pos] = b.rid; W.src[pos] = -1; }
    WAVE_SYNC();
    return n + 1;
}
// mem_sort_dedup_patch(opt, 0, 0, 0, n, list) on W[0, n) in the order the call finds it (resc_lds_keys_ok said yes; smax from there); W is left in the call's final order
// and W.tied / *harmful_out as resc_list_ties would set them (an entry whose twin the second sort's removal of identical hits takes away may keep its mark: a mark only
// ever sends a call here).  Returns the new length.
template <int CAP> __device__ __forceinline__ int resc_dedup_lds(const DOpts& o, RescListT<CAP>& W, RescScratchT<CAP>& S, int n, int smax, int lane, int known_tie, int* harmful_out) {
    *harmful_out = 0;
    if (n <= 1) { if (n == 1 && lane == 0) W.tied[0] = 0; WAVE_SYNC(); return n; }
    constexpr int PER = (CAP + 63) / 64;
    const int ib = n <= 512 ? 9 : 11;
    const i64 imask = ((i64)1 << ib) - 1;
    WAVE_SYNC();
    for (int k = lane; k < n; k += 64) S.lk[k] = W.re[k] << ib | (i64)k;
    WAVE_SYNC();
    // sort by the END position (known_tie: the caller has seen two equal end positions — no point in trying to rank)
    if (known_tie || !resc_rank_keys(S.lk, n, ib, lane, S.ord)) {
        wave_introsort_i64<PER>(n, S.lk, ib, lane, S.ord, S.ord2, o.wd);   // (klib's introsort move for move, by the wave: lh_sort.h)
        WAVE_SYNC();
        for (int k = lane; k < n; k += 64) S.ord[k] = (uint16_t)(S.lk[k] & imask);
        WAVE_SYNC();
    }
    // the scan
    for (int c0 = 0; c0 < n; c0 += 64) {
        u64 act;
        {
            const int i = c0 + lane;
            int a_ = 0;
            if (i >= 1 && i < n) { const int e = S.ord[i], em = S.ord[i - 1]; a_ = !(W.rid[e] != W.rid[em] || W.rb[e] >= W.re[em] + o.max_chain_gap); }
            act = __ballot(a_);
        }
        while (act) {
            const int i = c0 + __ffsll((unsigned long long)act) - 1;
            act &= act - 1;
            const int pe = S.ord[i];
            const i64 p_rb = W.rb[pe], p_re = W.re[pe];
            const int p_qb = W.qb[pe], p_qe = W.qe[pe], p_sc = W.score[pe], p_rid = W.rid[pe];
            for (int j0 = i - 1; j0 >= 0; j0 -= 64) {
                const int j = j0 - lane;
                int stop = 1, kp = 0, kq = 0, qx = 0, q_qb = 0;
                if (j >= 0) {
                    qx = S.ord[j];
                    const i64 q_re = W.re[qx];
                    if (p_rid == W.rid[qx] && p_rb < q_re + o.max_chain_gap) {
                        stop = 0;
                        q_qb = W.qb[qx];
                        const int q_qe = W.qe[qx];
                        if (q_qe != q_qb && resc_redundant(o, W.rb[qx], q_re, q_qb, q_qe, p_rb, p_re, p_qb, p_qe)) { if (p_sc < W.score[qx]) kp = 1; else kq = 1; }
                    }
                }
                const u64 sm = __ballot(stop || kp);
                const int first = sm ? __ffsll((unsigned long long)sm) - 1 : 64;
                if (kq && lane < first) W.qe[qx] = q_qb;          // a[j] is excluded
                if (kp && lane == first) W.qe[pe] = p_qb;         // p is, by a better entry: the end of its walk
                if (sm) break;
            }
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
    // exclude what the scan marked (in the order of the first sort: the second sort's input), and the second sort's keys
    int m = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        int e = 0, alive = 0;
        if (i < n) { e = S.ord[i]; alive = W.qe[e] > W.qb[e]; }
        const u64 mk = __ballot(alive);
        if (alive) {
            const int d = m + lanes_below(mk, lane);
            S.lk[d] = (i64)(smax - W.score[e]) << (41 + ib) | W.rb[e] << (8 + ib) | (i64)W.qb[e] << ib | (i64)e;
            S.ord2[d] = (uint16_t)e;
        }
        m += (int)__popcll(mk);
    }
    WAVE_SYNC();
    // equal end positions among the survivors (resc_list_ties' marks and verdict, read off the sorted order: equal ends are neighbours)
    {
        int harmful = 0;
        for (int d = lane; d < m; d += 64) {
            const int e = S.ord2[d];
            const i64 pre = W.re[e], prb = W.rb[e];
            const int pqb = W.qb[e], pqe = W.qe[e], prid = W.rid[e];
            int t = d > 0 && W.re[S.ord2[d - 1]] == pre;
            for (int u = d + 1; u < m; ++u) {
                const int x = S.ord2[u];
                if (W.re[x] != pre) break;
                t = 1;
                if (W.rid[x] != prid) harmful = 1;
                else harmful |= resc_redundant(o, W.rb[x], pre, W.qb[x], W.qe[x], prb, pre, pqb, pqe) || resc_redundant(o, prb, pre, pqb, pqe, W.rb[x], pre, W.qb[x], W.qe[x]);
            }
            W.tied[e] = (uint8_t)t;
        }
        *harmful_out = __any(harmful);
    }
    WAVE_SYNC();
    if (!resc_rank_keys(S.lk, m, ib, lane, S.ord)) {
        wave_introsort_i64<PER>(m, S.lk, ib, lane, S.ord, S.ord2, o.wd);
        WAVE_SYNC();
        // identical hits (score, rb, qb: the key above the index): all but the first of a run go
        int m2 = 0;
        for (int c0 = 0; c0 < m; c0 += 64) {
            const int i = c0 + lane;
            int keep = 0;
            i64 key = 0;
            if (i < m) { key = S.lk[i]; keep = i == 0 || (S.lk[i - 1] >> ib) != (key >> ib); }
            const u64 mk = __ballot(keep);
            WAVE_SYNC();
            if (keep) S.ord[m2 + lanes_below(mk, lane)] = (uint16_t)(key & imask);
            m2 += (int)__popcll(mk);
        }
        m = m2;
        WAVE_SYNC();
    }
    // the survivors into their places
    i64 e_[PER], rb_[PER]; int qb_[PER], qe_[PER], sc_[PER], rid_[PER], src_[PER], tie_[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = t * 64 + lane;
        if (k < m) { const int x = S.ord[k]; e_[t] = W.re[x]; rb_[t] = W.rb[x]; qb_[t] = W.qb[x]; qe_[t] = W.qe[x]; sc_[t] = W.score[x]; rid_[t] = W.rid[x]; src_[t] = W.src[x]; tie_[t] = W.tied[x]; }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = t * 64 + lane;
        if (k < m) { W.re[k] = e_[t]; W.rb[k] = rb_[t]; W.qb[k] = qb_[t]; W.qe[k] = qe_[t]; W.score[k] = sc_[t]; W.rid[k] = rid_[t]; W.src[k] = src_[t]; W.tied[k] = (uint8_t)tie_[t]; }
    }
    WAVE_SYNC();
    return m;
}

// Contigs that INTERLEAVE in the order of end positions — an entry of another contig between two entries of one — cannot happen with real coordinates (a contig is an
// interval of the concatenated reference, forward or reverse), and the incremental form relies on it: the scan's walk stops at an entry of another contig, so excluding such an
// entry would let two entries meet that no call has compared.  Checked when a list is loaded (a list that fails has every call run as written); a rescued region lies in its
// anchor's contig, so a list stays as it was found.  Returns 1 if some entry has an entry of another contig and, beyond that but within max_chain_gap, one of its own to its left.
template <int CAP> __device__ __forceinline__ int resc_list_interleaved(const DOpts& o, const RescListT<CAP>& W, int n, int lane) {
    int bad = 0;
    for (int k = lane; k < n; k += 64) {
        const i64 pre = W.re[k], prb = W.rb[k];
        const int prid = W.rid[k];
        i64 lo = -0x7fffffffffffffffll;   // the nearest entry of another contig to the left
        for (int u = 0; u < n; ++u) { const i64 e = W.re[u]; if (W.rid[u] != prid && e <= pre && e > lo) lo = e; }
        // (within the scan's reach only: a contig's forward and reverse strands are two intervals with other contigs in between, and far apart)
        for (int u = 0; u < n; ++u) { const i64 e = W.re[u]; bad |= W.rid[u] == prid && e <= lo && prb < e + o.max_chain_gap; }
    }
    return __any(bad);
}

// development aid (-DLH_RFA_PROF): shader clocks per part of the replay, and how its calls were decided
#ifdef LH_RFA_PROF
__device__ unsigned long long lh_resc_prof[24];
#define RA_PROF(k_) { const unsigned long long now_ = (unsigned long long)clock64(); if (lane == 0) atomicAdd(&lh_resc_prof[k_], now_ - prof_t_); prof_t_ = (unsigned long long)clock64(); }
#define RA_COUNT(k_) { if (lane == 0) atomicAdd(&lh_resc_prof[k_], 1ull); }
#else
#define RA_PROF(k_) {}
#define RA_COUNT(k_) {}
#endif
#ifdef LH_RA_HIST   // development aid: pairs by log2 of their replay's duration in 10 ns ticks (wall_clock64): [0, 32) pairs whose calls were all decided incrementally, [32, 64) pairs with a call run as written; [64], [65]: the longest of each kind
__device__ unsigned long long lh_resc_hist[2 * 32 + 2];
#endif
template <int DIR, int CAP>
__global__ void __launch_bounds__(64) k_resc_apply(DIndex ix, DOpts o, int n_pairs, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                    DReg* __restrict__ regs, DReg* __restrict__ regs_tmp, int32_t* __restrict__ ia_pool, int32_t* __restrict__ n_regs,
                                                    const int32_t* __restrict__ best_score, const uint8_t* __restrict__ clean, DCounters* __restrict__ ctr, const int32_t* __restrict__ list, RMeta* __restrict__ meta,
                                                    const int32_t* __restrict__ n_jobs, const i64* __restrict__ job_off, const RJob* __restrict__ jobs) {
    __shared__ uint8_t qm[LH_MAXLEN + 6];
    __shared__ RescListT<CAP> W;
    __shared__ RescScratchT<CAP> S;
    const int lane = LANE();
    const int n_items = meta->list_count;
    // (r05) the entries are handed out one at a time: with a fixed share per wave (18 entries) the kernel's duration was that of its unluckiest wave — the SQ
    // counters showed a third of the resident waves on average (profiles/r05_pmc_repeats.json) — and a pair's cost spans two orders of magnitude
    // (the long-list instance — a handful of pairs — keeps its fixed stride: its waves would each walk the whole list through the counter)
    for (int turn = 0;; ++turn) {
        int item = (int)blockIdx.x + turn * (int)gridDim.x;
        if (CAP == LH_RA_CAP) {
            if (lane == 0) item = atomicAdd(&meta->apply_next[0], 1);
            item = wave_readlane(item, 0);
        }
        if (item >= n_items) break;
        // two instances of this kernel share the list: the one with the usual LDS arrays takes the pairs whose mate's list fits them with everything that
        // may come, the one with LH_RA_CAP_BIG entries the others (k_resc_enum decided, bit 30 of the entry: every pair is replayed exactly once)
        if (((list[item] >> 30) & 1) != (CAP != LH_RA_CAP)) continue;
        const int p = list[item] & 0x1fffffff;
        const int r1 = 2 * p, r2 = 2 * p + 1;
        const int r_ms = DIR ? r2 : r1, r_from = DIR ? r1 : r2;
        WAVE_SYNC();   // the previous pair's query and list are no longer in use
        const i64 off_ms = seq_off[r_ms];
        int l_ms = (int)(seq_off[r_ms + 1] - off_ms);
        if (l_ms > LH_MAXLEN) l_ms = 0;
        for (int i = lane; i < l_ms; i += 64) qm[i] = seq[off_ms + i];
        WAVE_SYNC();
        const i64 ro_ms = reg_off[r_ms];
        DReg* ma = regs + ro_ms;
        const DReg* from = regs + reg_off[r_from];
        int n_ma = n_regs[r_ms];
        const int nf = n_regs[r_from], bestf = best_score[r_from];
        const RJob* pj = jobs + job_off[p];
        const int npj = n_jobs[p];
        // the pair's jobs and, 64 at a time, its anchors live in the lanes' registers (lane k: job k / anchor i0 + k) and are read with v_readlane: a dependent
        // read from memory per iteration — anchor record, job anchor, job record — was most of an iteration's latency
        const bool jreg = npj <= 64;
        RJob myjob;
        myjob.anchor = 0x7fff; myjob.score = 0; myjob.te = myjob.qe = myjob.tb = myjob.qb = 0; myjob.tlen = 0; myjob.rows2 = 0;
        if (jreg && lane < npj) myjob = pj[lane];
        int a_sc = 0, a_rid = 0, a_alt = 0;
        i64 a_rb = 0;
        u64 amask = 0;
        int jp = 0;
        u64 cells = 0, cells_x = 0;
        int n_sw = 0, num = 0;
        int w_sorted = 1;   // the list in LDS is in the call's final order (until a rescued region is appended)
        int mode = 0;   // 0: the list is in memory, no dedup call yet; 1: in LDS (W), clean; 2: in memory for good (too long); 3: in memory for this call; 4: in LDS with a tie that is not harmless (every call as written, resc_dedup_lds)
        // (r05) the first test below reads every mate region's rb, per anchor — fifty anchors a pair, most of them skipped by that very test, so that many
        // pairs never get as far as loading the list (mode 0 for good): a gather of 72-byte records from memory per anchor and 64 regions was the replay's
        // largest part.  The rb alone are in LDS from the start; W.rb is the as-written call's scratch, so they are loaded again after every such call.
        int inter = 0;   // the list in LDS has contigs that interleave (see resc_list_interleaved)
        int rb_lds = 0;
        if (n_ma <= CAP) { for (int k = lane; k < n_ma; k += 64) W.rb[k] = ma[k].rb; rb_lds = 1; }
        WAVE_SYNC();
#ifdef LH_RFA_PROF
        unsigned long long prof_t_ = (unsigned long long)clock64();
        RA_COUNT(8)
#endif
#ifdef LH_RA_HIST
        const unsigned long long pair_t0_ = (unsigned long long)wall_clock64();
        int n_asw_ = 0;
#endif
        for (int i = 0; i < nf && num < o.rescue_max_hits && l_ms > 0; ++i) {
            if ((i & 63) == 0) {
                const int k = i + lane;
                if (k < nf) { const DReg& g = from[k]; a_sc = g.score; a_rb = g.rb; a_rid = g.rid; a_alt = g.is_alt; }
                amask = __ballot(k < nf && a_sc >= bestf - o.rescue_score_delta);   // the chunk's anchors that pass gobwa.go's score test
            }
            {   // (r05) straight to the next anchor that passes: a list of several hundred regions has its fifty best in front
                const u64 rest = amask >> (i & 63);
                if (!(rest & 1)) { if (!rest) i |= 63; else i += __ffsll((unsigned long long)rest) - 2; continue; }
            }
            DReg a;
            a.score = wave_readlane(a_sc, i & 63);
            a.rb = (i64)((u64)(uint32_t)wave_readlane((int)((u64)a_rb >> 32), i & 63) << 32 | (u64)(uint32_t)wave_readlane((int)(uint32_t)(u64)a_rb, i & 63));
            a.rid = wave_readlane(a_rid, i & 63); a.is_alt = wave_readlane(a_alt, i & 63);
            num++;
            if (jreg) jp = (int)__popcll(__ballot(lane < npj && myjob.anchor < i));   // (jobs of attempts that have become unnecessary are passed over)
            else while (jp < npj && pj[jp].anchor < i) ++jp;
            RA_PROF(16)
            int skip1 = 0;
            for (int i0 = 0; i0 < n_ma; i0 += 64) {                 // which orientation has been found
                int k = i0 + lane, f = 0;
                if (k < n_ma) {
                    i64 dist;
                    int r = dev_infer_dir(ix.l_pac, a.rb, (mode == 1 || mode == 4 || rb_lds) ? W.rb[k] : ma[k].rb, &dist);
                    f = (r == 1 && dist >= o.pes_low && dist <= o.pes_high);
                }
                if (__any(f)) { skip1 = 1; break; }
            }
            RA_PROF(0)
            if (skip1) continue;
            i64 rb, re;
            if (!resc_window(ix, o, a, l_ms, &rb, &re)) continue;
            n_sw++;
            RA_COUNT(9)
            KswR aln;
            if (jp < npj && (jreg ? wave_readlane((int)myjob.anchor, jp) : (int)pj[jp].anchor) == i) {
                RJob J;
                if (jreg) {
                    J.score = (int16_t)wave_readlane((int)myjob.score, jp); J.te = (int16_t)wave_readlane((int)myjob.te, jp); J.qe = (int16_t)wave_readlane((int)myjob.qe, jp);
                    J.tb = (int16_t)wave_readlane((int)myjob.tb, jp); J.qb = (int16_t)wave_readlane((int)myjob.qb, jp); J.tlen = (int16_t)wave_readlane((int)myjob.tlen, jp);
                    J.rows2 = (int16_t)wave_readlane((int)myjob.rows2, jp);
                } else J = pj[jp];
                aln.score = J.score; aln.te = J.te; aln.qe = J.qe; aln.tb = J.tb; aln.qb = J.qb;
                cells += (u64)(16 * ((l_ms + 15) / 16)) * (u64)J.tlen;
                if (J.score >= o.min_seed_len * o.a) cells += (u64)(16 * ((J.qe + 1 + 15) / 16)) * (u64)J.rows2;
            } else {   // no job: an attempt the enumeration saw as unnecessary (or left to this kernel): k_rescue.h's wave-wide kernel
                RA_PROF(1)
                const u64 cells_before = cells;
                aln = wave_ksw_align2(ix, o, qm, l_ms - 1, -1, 1, l_ms, rb, 1, (int)(re - rb), o.min_seed_len * o.a, lane, &cells);
                cells_x += cells - cells_before;   // (this kernel runs every cell it accounts for)
                RA_PROF(17)
                RA_COUNT(18)
            }
            const int hit = aln.score >= o.min_seed_len && aln.qb >= 0;
            RA_PROF(1)
            DReg b;
            if (hit) {
                b.rid = a.rid; b.is_alt = a.is_alt;
                b.qb = l_ms - (aln.qe + 1); b.qe = l_ms - aln.qb;
                b.rb = (ix.l_pac << 1) - (rb + aln.te + 1); b.re = (ix.l_pac << 1) - (rb + aln.tb);
                b.score = aln.score; b.csub = 0; b.secondary = -1;
                b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                b.truesc = 0; b.sub = 0; b.w = 0; b.seedlen0 = 0; b.n_comp = 0; b.frac_rep = 0;
            }
            if (mode == 0 && n_ma + o.rescue_max_hits <= CAP) {
                // the first attempt that gets here: the list as k_dedup left it (sorted, no identical hits).  If it also holds no redundant pair and no
                // two equal re — one pass over its pairs — the call that follows this attempt is already a function of the added region alone
                for (int k = lane; k < n_ma; k += 64) {
                    const DReg& g = ma[k];
                    W.rb[k] = g.rb; W.re[k] = g.re; W.qb[k] = g.qb; W.qe[k] = g.qe; W.score[k] = g.score; W.rid[k] = g.rid; W.src[k] = k;
                }
                WAVE_SYNC();
                // (r05) K5 says when there is nothing to look for: no region of the list was merged and no two end where another does (k_dedup.h: clean) — the pass below was a
                // third of the replay of a pair whose calls are all incremental
                const int known = clean[r_ms];
                int dirty = 0;
                for (int k0 = 0; k0 < n_ma && !known; k0 += 64) {
                    const int k = k0 + lane;
                    if (k < n_ma) {
                        const i64 pre = W.re[k], prb = W.rb[k];
                        const int pqb = W.qb[k], pqe = W.qe[k], prid = W.rid[k], psc = W.score[k];
                        for (int u = 0; u < n_ma; ++u) {
                            const i64 qre = W.re[u];
                            if (qre < pre && W.rid[u] == prid && prb < qre + o.max_chain_gap && resc_redundant(o, W.rb[u], qre, W.qb[u], W.qe[u], prb, pre, pqb, pqe)) dirty = 1;
                            if (u < k && (W.score[u] < psc || (W.score[u] == psc && (W.rb[u] > prb || (W.rb[u] == prb && W.qb[u] >= pqb))))) dirty = 1;   // (not in the call's final order)
                        }
                    }
                }
                mode = __any(dirty) ? 0 : 1;
                if (mode == 1 && resc_list_interleaved(o, W, n_ma, lane)) mode = 0;   // (not with real coordinates; the reload below keeps such a list's calls as written)
                if (mode == 1 && known) { for (int k = lane; k < n_ma; k += 64) W.tied[k] = 0; }
                else if (mode == 1 && resc_list_ties(o, W, n_ma, lane)) mode = 0;   // (two entries with one re that are redundant one way round: the first call as written)
                WAVE_SYNC();
                RA_PROF(2)
                if (mode == 0) RA_COUNT(10)
            }
            if (mode == 1 || mode == 4) {   // the list is in LDS
                if (mode == 1) {
                    if (!hit) continue;   // a clean list and nothing new: the call changes nothing
                    // the call as a function of b
                    int app = 0;
                    const int n_inc = resc_dedup_incremental(o, W, n_ma, b, lane, &app);
                    RA_PROF(3)
                    if (n_inc >= 0) { n_ma = n_inc; if (app) w_sorted = 0; RA_COUNT(11) continue; }
                    RA_COUNT(12)
                }
                // equal keys: this call as written — in LDS if the packed sort keys hold the list (resc_dedup_lds), else on the list in memory (mode 3: it comes back)
                int smax = 0;
                if (resc_lds_keys_ok(W, n_ma, hit ? &b : nullptr, lane, &smax)) {
                    if (!w_sorted) { resc_list_sort(W, n_ma, lane); w_sorted = 1; }   // the order the previous call left: what this one finds
                    if (hit) n_ma = resc_list_insert(W, n_ma, b, lane);
                    int harmful = 0;
                    n_ma = resc_dedup_lds(o, W, S, n_ma, smax, lane, mode == 1 && hit, &harmful);   // (mode 1: the incremental form declined: equal keys, nearly always equal ends)
                    mode = (harmful | inter) ? 4 : 1;   // (4: a tie that is not harmless stays: every further call as written)
                    WAVE_SYNC();
                    RA_PROF(5)
                    RA_COUNT(13)
#ifdef LH_RA_HIST
                    ++n_asw_;
#endif
                    continue;
                }
                resc_list_store(ix, W, n_ma, ma, regs_tmp + ro_ms, lane, w_sorted);
                mode = 3;
            }
            // ---- the list in memory: insert, then mem_sort_dedup_patch as written
            if (hit) {
                int pos = n_ma;   // before the first element with a smaller score
                for (int i0 = 0; i0 < n_ma; i0 += 64) {
                    int k = i0 + lane;
                    u64 bm = __ballot(k < n_ma && ma[k].score < b.score);
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
            RA_PROF(4)
            RA_COUNT(13)
#ifdef LH_RA_HIST
            ++n_asw_;
#endif
            n_ma = wave_sort_dedup_patch(ix, o, qm, ma, n_ma, ia_pool + ro_ms + r_ms, regs_tmp + ro_ms, 0, lane, &cells, (i64*)&W, 2 * CAP);   // (the list is in memory during the call: W's two 64-bit arrays are the sorts' scratch)
            RA_PROF(5)
            if (mode != 2 && n_ma + o.rescue_max_hits <= CAP) {   // (back) into LDS if the list fits with everything that may still come and no two re are equal
                for (int k = lane; k < n_ma; k += 64) {
                    const DReg& g = ma[k];
                    W.rb[k] = g.rb; W.re[k] = g.re; W.qb[k] = g.qb; W.qe[k] = g.qe; W.score[k] = g.score; W.rid[k] = g.rid; W.src[k] = k;
                }
                WAVE_SYNC();
                w_sorted = 1;
                // equal end positions left behind: harmless ones are marked, one that is not keeps the list in memory for good
                inter = resc_list_interleaved(o, W, n_ma, lane);
                mode = (resc_list_ties(o, W, n_ma, lane) | inter) ? 4 : 1;   // (4: the list stays in LDS, every call as written)
                WAVE_SYNC();
                rb_lds = 1;
                if (mode == 4) RA_COUNT(14)
            } else {
                mode = 2; RA_COUNT(15)
                rb_lds = n_ma <= CAP;
                if (rb_lds) for (int k = lane; k < n_ma; k += 64) W.rb[k] = ma[k].rb;
                WAVE_SYNC();
            }
            RA_PROF(6)
        }
        if (mode == 1 || mode == 4) resc_list_store(ix, W, n_ma, ma, regs_tmp + ro_ms, lane, w_sorted);
        RA_PROF(7)
#ifdef LH_RA_HIST
        if (lane == 0) {
            const unsigned long long dt_ = (unsigned long long)wall_clock64() - pair_t0_;
            int b_ = 0;
            while (b_ < 31 && (dt_ >> (b_ + 1))) ++b_;
            atomicAdd(&lh_resc_hist[(n_asw_ != 0) * 32 + b_], 1ull);
            atomicMax(&lh_resc_hist[64 + (n_asw_ != 0)], dt_);
        }
#endif
        if (lane == 0) {
            n_regs[r_ms] = n_ma;
            if (ctr && n_sw) { atomicAdd(&LH_CTR(ctr)->n_rescue, (u64)n_sw); atomicAdd(&LH_CTR(ctr)->rescue_cells, cells); if (cells_x) atomicAdd(&LH_CTR(ctr)->rescue_cells_exec, cells_x); }
        }
    }
}

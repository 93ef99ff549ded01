// k_dedup.h — K5: mem_sort_dedup_patch (+ mem_patch_reg), one wavefront per read.
// Reached through mem_align1_core (go/src/gobwa/gobwa.go:244,253) with patching, and through mem_matesw
// (gobwa.go:291,315) without (bns == 0).
//
// The two unstable introsorts and the order-dependent pairwise exclusion run on lane 0 over an index array; the one
// heavy step — mem_patch_reg's global re-alignment score — is a wave-parallel DP (k_global.h), so the pairwise loop is
// wave-uniform and every lane follows it.
#pragma once
#include "k_global.h"

#define LH_PATCH_MAX_R_BW 0.05f
#define LH_PATCH_MIN_SC_RATIO 0.90f

// mem_patch_reg; a = earlier region (q), b = later region (p).  Returns the merged score (>0) or 0.
__device__ __forceinline__ int wave_patch_reg(const DIndex& ix, const DOpts& o, const uint8_t* q, const DReg& a, const DReg& b, int* w_out, int lane, u64* cells) {
    if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;                 // on different strands
    if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;        // not colinear
    int w = (int)((a.re - b.rb) - (a.qe - b.qb));                      // required bandwidth
    w = w > 0 ? w : -w;
    double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);   // relative bandwidth
    r = r > 0. ? r : -r;
    if (a.re < b.rb || a.qe < b.qb) {   // no overlap on query or on ref
        if (w > o.w << 1 || r >= LH_PATCH_MAX_R_BW) return 0;
    } else if (w > o.w << 2 || r >= LH_PATCH_MAX_R_BW * 2) return 0;
    w += a.w + b.w;
    w = w < o.w << 2 ? w : o.w << 2;
    int ok;
    int score = wave_gen_score(ix, o, q, a.qb, b.qe - a.qb, w, a.rb, b.re, lane, &ok, cells);
    if (!ok) return 0;
    int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);   // predicted score from query
    int r_s = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);   // predicted score from ref
    if ((double)score / (q_s > r_s ? q_s : r_s) < LH_PATCH_MIN_SC_RATIO) return 0;
    *w_out = w;
    return score;
}

// mem_sort_dedup_patch over av[0..n).  ia: int scratch [n]; tmp: DReg scratch [n].  Returns the new count (uniform).
// lk / lk_cap: LDS scratch for the two sorts (may be null).  With it, what the introsorts move is one 64-bit word per region — the sort key with the
// region's index in its low 9 bits, compared without them — instead of an index whose every comparison reads two region records from memory: the
// same comparisons, the same moves, the same order for equal keys.
// lk[0, n): 64-bit sort keys with the element's index in their low `ib` bits.  When no two keys are equal above those bits there is only one sorted order,
// whatever the algorithm: every lane ranks its elements against all of them (n / 64 LDS sweeps) and writes the element's index to its place in ia[] — what the
// callers read off the sorted keys.  Returns 1 if that was done; 0 when two keys are equal (the introsort's order of equal keys is part of the result: the
// caller runs it on lk, which is untouched, and fills ia from it).
__device__ __forceinline__ int wave_rank_sort_keys(const i64* lk, int n, int ib, int lane, int32_t* ia) {
    int tie = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        if (e < n) {
            const i64 key = lk[e], k = key >> ib;
            int rk = 0;
            for (int j = 0; j < n; ++j) { const i64 kj = lk[j] >> ib; rk += kj < k; tie |= (kj == k) & (j != e); }
            ia[rk] = (int)(key & (((i64)1 << ib) - 1));
        }
    }
    WAVE_SYNC();
    return !__any(tie);
}

__device__ __forceinline__ int wave_sort_dedup_patch(const DIndex& ix, const DOpts& o, const uint8_t* q, DReg* av, int n, int32_t* ia, DReg* tmp,
                                                     int do_patch, int lane, u64* cells, i64* lk = nullptr, int lk_cap = 0, int* clean_out = nullptr) {
    // *clean_out (r05): 1 if the list that comes out is known to be one that a further call without patching leaves alone AND holds no two equal end positions — no region
    // was merged (a merged region has grown past entries its walk had already compared it with) and the first sort found all end positions different; K6's replay then
    // skips its own look at the list (k_rescue2.h).  0 says nothing.
    if (clean_out) *clean_out = 1;
    if (n <= 1) return n;
    int merged = 0;
    const bool packed = lk && n <= lk_cap && n <= 2048;
    const int ib = n <= 512 ? 9 : 11;   // bits of the region index below the key
    for (int i = lane; i < n; i += 64) { ia[i] = i; av[i].n_comp = 1; if (packed) lk[i] = av[i].re << ib | (i64)i; }
    WAVE_SYNC();
    // sort by the END position, not START!  (r05: all end positions different — the usual list — is one sorted order: ranked by the whole wave)
    const int ranked1 = packed && wave_rank_sort_keys(lk, n, ib, lane, ia);
    if (!ranked1 && lane == 0) {
        if (packed) dev_introsort(n, lk, [&](i64 x, i64 y) { return (x >> ib) < (y >> ib); }, o.wd);
        else dev_introsort(n, ia, [&](int x, int y) { return av[x].re < av[y].re; }, o.wd);
    }
    WAVE_SYNC();
    if (packed && !ranked1) {
        for (int i = lane; i < n; i += 64) ia[i] = (int)(lk[i] & ((1 << ib) - 1));
        WAVE_SYNC();
    }
    int wd = 1000000;
    // (r05) the scan only does something at an entry whose predecessor lies on its contig within max_chain_gap — decided for 64 entries at once from fields that
    // no earlier step of the scan changes (an entry's own rb before its turn, its predecessor's re and rid): on repeat families most regions stand alone
    for (int c0 = 0; c0 < n; c0 += 64) {
    u64 act;
    {
        const int i = c0 + lane;
        int a_ = 0;
        if (i >= 1 && i < n) { const DReg& P = av[ia[i]]; const DReg& PM = av[ia[i - 1]]; a_ = !(P.rid != PM.rid || P.rb >= PM.re + o.max_chain_gap); }
        act = __ballot(a_);
    }
    while (act) {
        const int i = c0 + __ffsll((unsigned long long)act) - 1;
        act &= act - 1;
        DReg p = av[ia[i]];
        for (int j = i - 1; j >= 0; --j) {
            DReg qq = av[ia[j]];
            LH_WATCH_D(o.wd, wd, 4, break)
            WAVE_SYNC();   // every lane has its copy before lane 0 may overwrite the entry
            if (!(p.rid == qq.rid && p.rb < qq.re + o.max_chain_gap)) break;
            if (qq.qe == qq.qb) continue;   // a[j] has been excluded
            i64 orr = qq.re - p.rb;   // overlap length on the reference
            i64 oq = qq.qb < p.qb ? qq.qe - p.qb : p.qe - qq.qb;   // overlap length on the query
            i64 mr = qq.re - qq.rb < p.re - p.rb ? qq.re - qq.rb : p.re - p.rb;   // min ref len in alignment
            i64 mq = qq.qe - qq.qb < p.qe - p.qb ? qq.qe - qq.qb : p.qe - p.qb;   // min qry len in alignment
            int score = 0, w = 0;
            if ((float)orr > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {   // one of the hits is redundant
                if (p.score < qq.score) {
                    p.qe = p.qb;
                    if (lane == 0) av[ia[i]].qe = p.qb;
                    break;
                } else {
                    if (lane == 0) av[ia[j]].qe = qq.qb;
                }
            } else if (do_patch && qq.rb < p.rb && (score = wave_patch_reg(ix, o, q, qq, p, &w, lane, cells)) > 0) {   // then merge q into p
                merged = 1;
                p.n_comp += qq.n_comp + 1;
                p.seedcov = p.seedcov > qq.seedcov ? p.seedcov : qq.seedcov;
                p.sub = p.sub > qq.sub ? p.sub : qq.sub;
                p.csub = p.csub > qq.csub ? p.csub : qq.csub;
                p.qb = qq.qb; p.rb = qq.rb;
                p.truesc = p.score = score;
                p.w = w;
                if (lane == 0) { av[ia[i]] = p; av[ia[j]].qb = qq.qe; }
            }
            WAVE_SYNC();
        }
        WAVE_SYNC();
    }
    }
    WAVE_SYNC();   // all lanes are done reading ia[]/av[] before lane 0 compacts and re-sorts them
    // (r05) what the scan excluded leaves the order array — 64 entries at a time, the survivors' places from a ballot; it was one lane reading every record in turn, as was the
    // pass over identical hits below: two or three dependent reads from memory per region, a third of K5 on repeat families
    int m = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        int id = 0, alive = 0;
        if (i < n) { id = ia[i]; alive = av[id].qe > av[id].qb; }
        const u64 mk = __ballot(alive);
        WAVE_SYNC();   // every lane holds its entry before any is moved (to a place at or before its own)
        if (alive) ia[m + lanes_below(mk, lane)] = id;
        m += (int)__popcll(mk);
    }
    WAVE_SYNC();
    int pk2 = packed;
    if (packed) {   // (score desc, rb, qb) as one ascending key: the bits of the list's largest score, 33 and 8 above the index; a region outside those ranges: the comparator on the records
        int smax = 0;
        for (int i = lane; i < m; i += 64) { const int sc = av[ia[i]].score; smax = smax > sc ? smax : sc; }
        smax = wave_max_i32(smax);
        const int sb = 32 - __clz(smax | 1);
        int bad = sb + 33 + 8 + ib > 63;
        for (int i = lane; i < m; i += 64) {
            const DReg& g = av[ia[i]];
            bad |= g.score < 0 || g.rb < 0 || g.rb >= (1ll << 33) || g.qb < 0 || g.qb > 255;
            lk[i] = (i64)(smax - g.score) << (41 + ib) | g.rb << (8 + ib) | (i64)g.qb << ib | (i64)ia[i];
        }
        pk2 = !__any(bad);
    }
    WAVE_SYNC();
    if (pk2) {
        // (the keys hold ia[i]: every lane has read its own before the ranking overwrites ia — the WAVE_SYNC above)
        const int ranked2 = wave_rank_sort_keys(lk, m, ib, lane, ia);
        if (!ranked2) {
            if (lane == 0) dev_introsort(m, lk, [&](i64 x, i64 y) { return (x >> ib) < (y >> ib); }, o.wd);
            WAVE_SYNC();
            for (int i = lane; i < m; i += 64) ia[i] = (int)(lk[i] & ((1 << ib) - 1));
        }
        WAVE_SYNC();
    }
    if (!pk2) {
        if (lane == 0) dev_introsort(m, ia, [&](int x, int y) {
            const DReg &A = av[x], &B = av[y];
            return A.score > B.score || (A.score == B.score && (A.rb < B.rb || (A.rb == B.rb && A.qb < B.qb)));
        }, o.wd);
        WAVE_SYNC();
    }
    {   // identical hits: an entry with its predecessor's (score, rb, qb) is marked and left out (the first entry stays; the comparison reads fields no mark changes)
        int m2 = 0, carry = 0;
        for (int c0 = 0; c0 < m; c0 += 64) {
            const int i = c0 + lane;
            const int id = i < m ? ia[i] : 0;
            const int pid = wave_shr1_i32(id, carry);   // the entry before: the previous chunk's last for lane 0 (its place may have been written by now)
            carry = wave_readlane(id, 63);
            int keep = 0;
            if (i < m) {
                int dup = 0;
                if (i > 0) { const DReg &A = av[id], &B = av[pid]; dup = A.score == B.score && A.rb == B.rb && A.qb == B.qb; }
                if (dup) av[id].qe = av[id].qb;
                keep = !dup;
            }
            const u64 mk = __ballot(keep);
            WAVE_SYNC();
            if (keep) ia[m2 + lanes_below(mk, lane)] = id;
            m2 += (int)__popcll(mk);
        }
        m = m2;
    }
    WAVE_SYNC();
    for (int i = lane; i < m; i += 64) tmp[i] = av[ia[i]];
    WAVE_SYNC();
    for (int i = lane; i < m; i += 64) av[i] = tmp[i];
    WAVE_SYNC();
    if (clean_out) *clean_out = ranked1 && !merged;
    return m;
}

// K5, lane per read: a read with at most one region has nothing to sort, exclude or patch (mem_sort_dedup_patch returns at
// once): only its best score is recorded.  A read with TWO regions — at human-genome scale every third read: its alignment
// and the extension of a chance-match chain somewhere else — is finished here too, mem_sort_dedup_patch unrolled for n = 2,
// unless the two regions get as far as mem_patch_reg's re-alignment (a DP: the wave kernel's).  The rest is listed.
__device__ __forceinline__ int dev_patch_needs_dp(const DIndex& ix, const DOpts& o, const DReg& a, const DReg& b) {   // mem_patch_reg up to its DP
    if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
    if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
    int w = (int)((a.re - b.rb) - (a.qe - b.qb));
    w = w > 0 ? w : -w;
    double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);
    r = r > 0. ? r : -r;
    if (a.re < b.rb || a.qe < b.qb) {
        if (w > o.w << 1 || r >= LH_PATCH_MAX_R_BW) return 0;
    } else if (w > o.w << 2 || r >= LH_PATCH_MAX_R_BW * 2) return 0;
    return 1;
}
__global__ void __launch_bounds__(256) k_dedup_fast(DIndex ix, DOpts o, int n_reads, const i64* __restrict__ reg_off, DReg* __restrict__ regs, int32_t* __restrict__ n_regs,
                                                     int32_t* __restrict__ best_score, int32_t* __restrict__ list, int32_t* __restrict__ list_count, uint8_t* __restrict__ clean) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int need = 0;
    if (r < n_reads) {
        int n = n_regs[r];
        clean[r] = 1;   // (nothing to compare, or two regions that were — unless they end at the same position, below; a listed read: k_dedup's word)
        if (n <= 1) best_score[r] = n == 1 ? regs[reg_off[r]].score : 0;
        else if (n == 2) {
            // (the decisions read six fields of each record; the records themselves move at most once, as 8-byte words: whole-record
            // selects made the compiler keep them in 376 B of scratch)
            DReg* av = regs + reg_off[r];
            const i64 rb0 = av[0].rb, re0 = av[0].re, rb1 = av[1].rb, re1 = av[1].re;
            const int qb0 = av[0].qb, qe0 = av[0].qe, sc0 = av[0].score, rid0 = av[0].rid, qb1 = av[1].qb, qe1 = av[1].qe, sc1 = av[1].score, rid1 = av[1].rid;
            // sorted by re (klib's introsort on two elements: swapped only when strictly out of order): qq = the first, p = the second
            const int s = re1 < re0;
            const i64 q_rb = s ? rb1 : rb0, q_re = s ? re1 : re0, p_rb = s ? rb0 : rb1, p_re = s ? re0 : re1;
            const int q_qb = s ? qb1 : qb0, q_qe = s ? qe1 : qe0, q_sc = s ? sc1 : sc0, q_rid = s ? rid1 : rid0;
            const int p_qb = s ? qb0 : qb1, p_qe = s ? qe0 : qe1, p_sc = s ? sc0 : sc1, p_rid = s ? rid0 : rid1;
            int kp = p_qe > p_qb, kq = q_qe > q_qb;   // still there
            if (p_rid == q_rid && p_rb < q_re + o.max_chain_gap) {
                const i64 orr = q_re - p_rb;
                const i64 oq = q_qb < p_qb ? q_qe - p_qb : p_qe - q_qb;
                const i64 mr = q_re - q_rb < p_re - p_rb ? q_re - q_rb : p_re - p_rb;
                const i64 mq = q_qe - q_qb < p_qe - p_qb ? q_qe - q_qb : p_qe - p_qb;
                if ((float)orr > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {
                    if (p_sc < q_sc) kp = 0;
                    else kq = 0;
                } else if (q_rb < p_rb) {
                    DReg a, b;   // (mem_patch_reg's tests up to its DP read the spans only)
                    a.rb = q_rb; a.re = q_re; a.qb = q_qb; a.qe = q_qe; b.rb = p_rb; b.re = p_re; b.qb = p_qb; b.qe = p_qe;
                    need = dev_patch_needs_dp(ix, o, a, b);
                }
            }
            if (!need) {
                // exclude, sort by (score desc, rb, qb), drop an identical hit
                int m = 0, first = 0;   // first: the record (0 / 1) that ends up in front
                if (kq && kp) {   // order after the first sort: qq, p
                    const bool p_first = p_sc > q_sc || (p_sc == q_sc && (p_rb < q_rb || (p_rb == q_rb && p_qb < q_qb)));
                    first = p_first ? 1 - s : s;
                    m = (p_sc == q_sc && p_rb == q_rb && p_qb == q_qb) ? 1 : 2;
                } else if (kq) { first = s; m = 1; }
                else if (kp) { first = 1 - s; m = 1; }
                if (m >= 1 && first == 1) {   // the records change places (or the second one moves to the front)
                    constexpr int NW = (int)(sizeof(DReg) / 8);
                    static_assert(sizeof(DReg) % 8 == 0, "a region record is a whole number of 8-byte words");
                    u64* w0 = (u64*)&av[0];
                    u64* w1 = (u64*)&av[1];
                    u64 a[NW], b[NW];
#pragma unroll
                    for (int k = 0; k < NW; ++k) { a[k] = w0[k]; b[k] = w1[k]; }
#pragma unroll
                    for (int k = 0; k < NW; ++k) w0[k] = b[k];
                    if (m == 2) {
#pragma unroll
                        for (int k = 0; k < NW; ++k) w1[k] = a[k];
                    }
                }
                if (m >= 1) av[0].n_comp = 1;
                if (m == 2) av[1].n_comp = 1;
                n_regs[r] = m;
                if (m == 2 && re0 == re1) clean[r] = 0;
                const int f_sc = first ? sc1 : sc0, o_sc = first ? sc0 : sc1;
                best_score[r] = m == 0 ? 0 : (m == 2 && o_sc > f_sc ? o_sc : f_sc);
            }
        } else need = 1;
    }
    u64 m = __ballot(need);
    if (m) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(list_count, (int32_t)__popcll(m));
        basep = wave_readlane(basep, 0);
        if (need) list[basep + lanes_below(m, lane)] = r;
    }
}

// K5, wave per listed read (list == null: every read).  Also records the best pre-rescue score of the read (gobwa.go:264-283).
__global__ void __launch_bounds__(64) k_dedup(DIndex ix, DOpts o, int n_reads, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                               const i64* __restrict__ reg_off, DReg* __restrict__ regs, DReg* __restrict__ regs_tmp, int32_t* __restrict__ ia_pool,
                                               int32_t* __restrict__ n_regs, int32_t* __restrict__ best_score, DCounters* __restrict__ ctr,
                                               const int32_t* __restrict__ list, const int32_t* __restrict__ list_count, uint8_t* __restrict__ clean) {
    __shared__ uint8_t q[LH_MAXLEN + 6];
    __shared__ i64 lk[512];
    const int lane = LANE();
    const int n_items = list ? *list_count : n_reads;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int r = list ? list[item] : item;
    WAVE_SYNC();   // the previous read's query is no longer in use
    i64 off = seq_off[r];
    int l_query = (int)(seq_off[r + 1] - off);
    if (l_query > LH_MAXLEN) l_query = 0;
    for (int i = lane; i < l_query; i += 64) q[i] = seq[off + i];
    WAVE_SYNC();
    i64 ro = reg_off[r];
    DReg* av = regs + ro;
    int n = n_regs[r];
    u64 cells = 0;
    int is_clean = 0;
    n = wave_sort_dedup_patch(ix, o, q, av, n, ia_pool + ro + r, regs_tmp + ro, 1, lane, &cells, lk, 512, &is_clean);
    int best = 0;
    for (int i = lane; i < n; i += 64) { int s = av[i].score; best = best > s ? best : s; }
    best = wave_max_i32(best);
    if (lane == 0) {
        n_regs[r] = n; best_score[r] = best; clean[r] = (uint8_t)is_clean;
        if (ctr && cells) atomicAdd(&LH_CTR(ctr)->glob_cells, cells);
    }
    }
}

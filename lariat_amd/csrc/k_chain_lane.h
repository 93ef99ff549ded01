// k_chain_lane.h — K3 for the common read (one lane per read), with round 0 of K4 at its end.
// Replaces BWA's mem_chain (chaining half), mem_chain_weight, mem_chain_flt and the start of mem_chain2aln, reached through
// mem_align1_core (go/src/gobwa/gobwa.go:244,253).
#pragma once
#include "k_extend2.h"

// K3 for the common read: one LANE per read when it has at most LH_CHAIN_LANE_MAX seeds and LH_CHAIN_LANE_MAXC chains (mem_chain is
// a short sequential program over a handful of seeds; a wave per read spent its time on launch and single-lane latency).  Same
// arithmetic and the same order of B-tree / sort / filter operations as the wave kernel above.  Reads with more are listed for it.
//
// Measured (profiles/r02_pmc_bench.json): with the chains, the ordered index, the sort array and the seed links of a read in
// per-read slices of HBM arrays, every field access of every lane was its own 64-B sector — 16.8 GB of traffic per 2 M pairs for
// ~2 GB of seeds and chains, and a chain of dependent round trips per seed.  Here a read's chains live in LDS (eight words per
// chain, word w of lane L at cl[w * 64 + L]: conflict-free), the ordered index / sort array / kept list are nibbles of a register,
// a seed's chain is a nibble of a 64-bit register, and mem_chain_weight is accumulated as seeds are appended (its two loops visit
// the seeds in the order they were appended).  HBM sees the seeds once on the way in (twice for kept chains) and the results.
#ifndef LH_CHAIN_LANE_WAVES
#define LH_CHAIN_LANE_WAVES 2   // 16 KB of LDS per wave: 10 waves per CU
#endif
#define LH_CHAIN_LANE_MAXC 8
// chain record: 0,1 pos | 2 last seed's rbeg - pos (later: seeds flattened so far) | 3 first_qbeg, last_qbeg << 8, last_len << 16 | 4 rid, is_alt << 30 |
// 5 n, weight over the query << 8 (later: n, kept << 8, first + 1 << 16) | 6 query end, weight over the reference << 8 (later: w) | 7 reference end - pos (later: seed_start)
#define CLW(ci_, f_) cl[(((ci_) << 3) + (f_)) * 64 + lane]
#define NIB(x_, i_) ((int)(((x_) >> ((i_) << 2)) & 0xf))
#define NIB_SET(x_, i_, v_) ((x_) = ((x_) & ~(0xfu << ((i_) << 2))) | ((uint32_t)(v_) << ((i_) << 2)))
// klib's ks_introsort for n <= 16 (see dev_introsort_small) over the nibbles of a register
template <class Lt> __device__ __forceinline__ void dev_introsort_small_nib(int n, uint32_t& a, Lt lt, int32_t* wdp) {
    if (n < 1) return;
    if (n == 2) {
        if (lt(NIB(a, 1), NIB(a, 0))) { int x = NIB(a, 0), y = NIB(a, 1); NIB_SET(a, 0, y); NIB_SET(a, 1, x); }
        return;
    }
    if (n > 1) {
        int t = n - 1, i = 0, j = t, k = i + ((j - i) >> 1) + 1;
        if (lt(NIB(a, k), NIB(a, i))) {
            if (lt(NIB(a, k), NIB(a, j))) k = j;
        } else k = lt(NIB(a, j), NIB(a, i)) ? i : j;
        const int rp = NIB(a, k);
        if (k != t) { int y = NIB(a, t); NIB_SET(a, k, y); NIB_SET(a, t, rp); }
        int wd = 4 * LH_CHAIN_LANE_MAX + 8;
        for (;;) {
            do { ++i; LH_WATCH(wdp, wd, 4, return) } while (lt(NIB(a, i), rp));
            do { --j; LH_WATCH(wdp, wd, 5, return) } while (i <= j && lt(rp, NIB(a, j)));
            if (j <= i) break;
            int x = NIB(a, i), y = NIB(a, j); NIB_SET(a, i, y); NIB_SET(a, j, x);
        }
        { int x = NIB(a, i), y = NIB(a, t); NIB_SET(a, i, y); NIB_SET(a, t, x); }
    }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && lt(NIB(a, j), NIB(a, j - 1)); --j) { int x = NIB(a, j), y = NIB(a, j - 1); NIB_SET(a, j, y); NIB_SET(a, j - 1, x); }
}
// (Measured and dropped in r04: the reads handed to the lanes in the order of their seed counts — a counting sort of the reads, most seeds first, so that a wave's
// 64 reads last about equally long: 4.40 ms against 3.71.  What the lanes gain in step they lose in locality: neighbouring reads' seeds and chains are neighbours in memory.)
// fuse = 1: the read's mem_chain2aln starts right here (ext_control<false>, k_extend2.h: round 0 of K4) — it finishes (n_regs), queues its
// first ksw_extend2 call for round 1 (next_*), or is left to the wave-per-read extension kernel (defer_*).
__global__ void __launch_bounds__(64, LH_CHAIN_LANE_WAVES) k_chain_lane(DIndex ix, DOpts o, int n_reads, i64 pool_cap, const DSeed* __restrict__ seeds, const int32_t* __restrict__ s_rid,
                                                    const int32_t* __restrict__ l_rep, DChain* __restrict__ chains, DSeed* __restrict__ cseeds, int32_t* __restrict__ n_chains,
                                                    int32_t* __restrict__ status, int32_t* __restrict__ list, int32_t* __restrict__ list_count, int32_t* __restrict__ srt,
                                                    i64* __restrict__ chain_rmax, DCounters* __restrict__ ctr, ExtArgs A, int fuse,
                                                    int32_t* __restrict__ next_count, int32_t* __restrict__ next_list, int32_t* __restrict__ next_key,
                                                    int32_t* __restrict__ defer_count, int32_t* __restrict__ defer_list) {
    const i64* __restrict__ seq_off = A.seq_off; const i64* __restrict__ seed_off = A.seed_off; int32_t* __restrict__ sdone = A.sdone;
    __shared__ uint32_t cl[LH_CHAIN_LANE_MAXC * 8 * 64];
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int heavy = 0, ext_out = 0, ext_key = 0;
    u64 win = 0, cells = 0;
    int nch_done = 0;
    if (r < n_reads) {
        const i64 base = seed_off[r];
        const int S = (int)(seed_off[r + 1] - base);
        if (seed_off[r + 1] > pool_cap) { n_chains[r] = 0; A.n_regs[r] = 0; atomicOr(&status[r], LH_ST_POOL_OVERFLOW); }
        else if (S > LH_CHAIN_LANE_MAX) heavy = 1;
        else {
            const int len = (int)(seq_off[r + 1] - seq_off[r]);
            const DSeed* sd_ = seeds + base;
            const int32_t* rid_ = s_rid + base;
            int nch = 0;
            uint32_t od = 0;    // chain ids in B-tree (position) order: insertion after equals
            u64 smap0 = ~0ull, smap1 = ~0ull;   // nibble s (seeds 0..15 / 16..31): the chain seed s was appended to (0xf: none — contained in its chain, or bridging contigs)
#define SMAP_SET(s_, v_) { if ((s_) < 16) smap0 = (smap0 & ~(0xfull << ((s_) << 2))) | ((u64)(v_) << ((s_) << 2)); else smap1 = (smap1 & ~(0xfull << (((s_) - 16) << 2))) | ((u64)(v_) << (((s_) - 16) << 2)); }
#define SMAP_GET(s_) ((int)((((s_) < 16 ? smap0 : smap1) >> (((s_) & 15) << 2)) & 0xf))
            DSeed pn; pn.rbeg = 0; pn.qbeg = 0; pn.len = 0;
            int ridn = -1;
            if (S > 0) { pn = sd_[0]; ridn = rid_[0]; }
            for (int s = 0; s < S && !heavy; ++s) {
                const DSeed p = pn;
                const int rid = ridn;
                if (s + 1 < S) { pn = sd_[s + 1]; ridn = rid_[s + 1]; }   // in flight while this seed is chained
                if (rid < 0) continue;   // bridging contigs / the forward-reverse boundary
                int to_add = 1, lo = 0;
                for (int k = 0; k < nch; ++k) {   // the chains at or before the seed (pos never changes once a chain exists)
                    const int ci = NIB(od, k);
                    const i64 pos = (i64)((u64)CLW(ci, 0) | (u64)CLW(ci, 1) << 32);
                    if (pos <= p.rbeg) lo = k + 1; else break;
                }
                if (lo > 0) {
                    const int ci = NIB(od, lo - 1);
                    const i64 c_pos = (i64)((u64)CLW(ci, 0) | (u64)CLW(ci, 1) << 32);
                    const uint32_t qp = CLW(ci, 3), ridw = CLW(ci, 4);
                    const i64 c_last_rbeg = c_pos + (int32_t)CLW(ci, 2);
                    const int c_first_qbeg = qp & 0xff, c_last_qbeg = (qp >> 8) & 0xff, c_last_len = (qp >> 16) & 0xff, c_rid = (int)(ridw & 0x3fffffffu);
                    const i64 qend = c_last_qbeg + c_last_len, rend = c_last_rbeg + c_last_len;   // test_and_merge
                    int res = 0;   // 0: new chain, 1: contained, 2: appended
                    if (rid != c_rid) res = 0;
                    else if (p.qbeg >= c_first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= c_pos && p.rbeg + p.len <= rend) res = 1;
                    else if ((c_last_rbeg < ix.l_pac || c_pos < ix.l_pac) && p.rbeg >= ix.l_pac) res = 0;
                    else {
                        i64 x = p.qbeg - c_last_qbeg, y = p.rbeg - c_last_rbeg;
                        if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c_last_len < o.max_chain_gap && y - c_last_len < o.max_chain_gap) res = 2;
                    }
                    if (res == 2) {
                        const int rr = (int)(p.rbeg - c_pos);
                        CLW(ci, 2) = (uint32_t)rr;
                        CLW(ci, 3) = (uint32_t)c_first_qbeg | (uint32_t)p.qbeg << 8 | (uint32_t)p.len << 16;
                        // mem_chain_weight, one seed further: over the query, over the reference
                        uint32_t w5 = CLW(ci, 5), w6 = CLW(ci, 6);
                        int n = w5 & 0xff, wq = (int)(w5 >> 8), endq = w6 & 0xff, wr = (int)(w6 >> 8), endr = (int32_t)CLW(ci, 7);
                        if (p.qbeg >= endq) wq += p.len; else if (p.qbeg + p.len > endq) wq += p.qbeg + p.len - endq;
                        endq = endq > p.qbeg + p.len ? endq : p.qbeg + p.len;
                        if (rr >= endr) wr += p.len; else if (rr + p.len > endr) wr += rr + p.len - endr;
                        endr = endr > rr + p.len ? endr : rr + p.len;
                        CLW(ci, 5) = (uint32_t)(n + 1) | (uint32_t)wq << 8; CLW(ci, 6) = (uint32_t)endq | (uint32_t)wr << 8; CLW(ci, 7) = (uint32_t)endr;
                        SMAP_SET(s, ci)
                    }
                    to_add = (res == 0);
                }
                if (to_add) {
                    if (nch == LH_CHAIN_LANE_MAXC) { heavy = 1; break; }
                    const uint32_t lowm = lo ? (0xffffffffu >> (32 - (lo << 2))) : 0u;
                    od = (od & lowm) | ((uint32_t)nch << (lo << 2)) | ((lo < 7) ? ((od & ~lowm) << 4) : 0u);
                    CLW(nch, 0) = (uint32_t)(u64)p.rbeg; CLW(nch, 1) = (uint32_t)((u64)p.rbeg >> 32); CLW(nch, 2) = 0;
                    CLW(nch, 3) = (uint32_t)p.qbeg | (uint32_t)p.qbeg << 8 | (uint32_t)p.len << 16;
                    CLW(nch, 4) = (uint32_t)rid | ((ix.contig_alt && ix.contig_alt[rid]) ? 1u << 30 : 0u);
                    CLW(nch, 5) = 1u | (uint32_t)p.len << 8; CLW(nch, 6) = (uint32_t)(p.qbeg + p.len) | (uint32_t)p.len << 8; CLW(nch, 7) = (uint32_t)p.len;
                    SMAP_SET(s, nch)
                    nch++;
                }
            }
            if (!heavy) {
            // mem_chain_weight's result; kept = 0, first = -1
            for (int k = 0; k < nch; ++k) {
                const uint32_t w5 = CLW(k, 5), w6 = CLW(k, 6);
                int wq = (int)(w5 >> 8), wr = (int)(w6 >> 8);
                int w = wr < wq ? wr : wq;
                CLW(k, 6) = (uint32_t)(w < 1 << 30 ? w : (1 << 30) - 1);
                CLW(k, 5) = w5 & 0xff;
            }
#define C_W(id_) ((int)CLW(id_, 6))
#define C_BEG(id_) ((int)(CLW(id_, 3) & 0xff))
#define C_END(id_) ((int)(((CLW(id_, 3) >> 8) & 0xff) + ((CLW(id_, 3) >> 16) & 0xff)))
#define C_KEPT(id_) ((int)((CLW(id_, 5) >> 8) & 0xff))
#define C_SET_KEPT(id_, v_) (CLW(id_, 5) = (CLW(id_, 5) & ~0xff00u) | ((uint32_t)(v_) << 8))
#define C_FIRST(id_) ((int)((CLW(id_, 5) >> 16) & 0xff) - 1)
#define C_SET_FIRST(id_, v_) (CLW(id_, 5) = (CLW(id_, 5) & ~0xff0000u) | ((uint32_t)((v_) + 1) << 16))
            // mem_chain_flt
            uint32_t st = 0;
            int n = 0;
            for (int k = 0; k < nch; ++k) {   // chains in position order (B-tree traversal), dropping light ones
                const int id = NIB(od, k);
                if (C_W(id) < o.min_chain_weight) continue;
                NIB_SET(st, n, id); n++;
            }
            if (n > 0) {
                dev_introsort_small_nib(n, st, [&](int x, int y) { return C_W(x) > C_W(y); }, o.wd);
                uint32_t kl = 0;   // the kept chains (indices into st)
                int nk = 0;
                C_SET_KEPT(NIB(st, 0), 3);
                NIB_SET(kl, nk, 0); nk++;
                for (int i = 1; i < n; ++i) {
                    int large_ovlp = 0, k;
                    const int idi = NIB(st, i);
                    const int ai_beg = C_BEG(idi), ai_end = C_END(idi), ai_w = C_W(idi), ai_alt = (int)(CLW(idi, 4) >> 30) & 1;
                    for (k = 0; k < nk; ++k) {
                        const int idj = NIB(st, NIB(kl, k));
                        const int aj_beg = C_BEG(idj), aj_end = C_END(idj), aj_w = C_W(idj), aj_alt = (int)(CLW(idj, 4) >> 30) & 1;
                        int b_max = aj_beg > ai_beg ? aj_beg : ai_beg;
                        int e_min = aj_end < ai_end ? aj_end : ai_end;
                        if (e_min > b_max && (!aj_alt || ai_alt)) {   // have overlap; don't consider ovlp where the kept chain is ALT while the current chain is primary
                            int li = ai_end - ai_beg, lj = aj_end - aj_beg;
                            int min_l = li < lj ? li : lj;
                            if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
                                large_ovlp = 1;
                                if (C_FIRST(idj) < 0) C_SET_FIRST(idj, i);
                                if (ai_w < aj_w * o.drop_ratio && aj_w - ai_w >= o.min_seed_len << 1) break;
                            }
                        }
                    }
                    if (k == nk) { NIB_SET(kl, nk, i); nk++; C_SET_KEPT(idi, large_ovlp ? 2 : 3); }
                }
                for (int i = 0; i < nk; ++i) {
                    int f = C_FIRST(NIB(st, NIB(kl, i)));
                    if (f >= 0) C_SET_KEPT(NIB(st, f), 1);
                }
                int i, k;
                for (i = k = 0; i < n; ++i) {   // don't extend more than max_chain_extend .kept=1/2 chains
                    int kp = C_KEPT(NIB(st, i));
                    if (kp == 0 || kp == 3) continue;
                    if (++k >= o.max_chain_extend) break;
                }
                for (; i < n; ++i)
                    if (C_KEPT(NIB(st, i)) < 3) C_SET_KEPT(NIB(st, i), 0);
            }
            int m = 0, sstart = 0;   // emit kept chains in sorted order
            const float frac_rep = (float)l_rep[r] / len;
            for (int i = 0; i < n; ++i) {
                const int id = NIB(st, i);
                const int kp = C_KEPT(id);
                if (kp == 0) { CLW(id, 7) = 0xffffffffu; continue; }
                DChain oc;
                oc.pos = (i64)((u64)CLW(id, 0) | (u64)CLW(id, 1) << 32); oc.rid = (int)(CLW(id, 4) & 0x3fffffffu); oc.n = (int)(CLW(id, 5) & 0xff); oc.seed_start = sstart;
                oc.w = C_W(id); oc.kept = kp; oc.is_alt = (int)(CLW(id, 4) >> 30) & 1;
                oc.frac_rep = frac_rep; oc.pad = id;
                chains[base + m] = oc;
                CLW(id, 7) = (uint32_t)sstart; CLW(id, 2) = 0;   // where its seeds go; how many are there
                sstart += oc.n;
                m++;
            }
            for (int k = 0; k < nch; ++k) if (C_W(k) < o.min_chain_weight) CLW(k, 7) = 0xffffffffu;   // (never entered st)
            for (int s = 0; s < S; ++s) {   // every kept chain's seed list, flattened, in the order the seeds were appended
                const int id = SMAP_GET(s);
                if (id == 0xf) continue;
                const uint32_t ss = CLW(id, 7);
                if (ss == 0xffffffffu) continue;
                const uint32_t t = CLW(id, 2);
                cseeds[base + ss + t] = sd_[s];
                CLW(id, 2) = t + 1;
            }
            n_chains[r] = m;
#undef SMAP_SET
#undef SMAP_GET
#undef C_W
#undef C_BEG
#undef C_END
#undef C_KEPT
#undef C_SET_KEPT
#undef C_FIRST
#undef C_SET_FIRST
            // K4's pre-pass for this read (k_extend computes its own): per kept chain the reference window of mem_chain2aln and the
            // order in which its seeds are extended (srt[], "extended" flags in sdone[])
            {
                int l_query = len > LH_MAXLEN ? 0 : len;
                const i64 l_pac = ix.l_pac;
                for (int ci = 0; ci < m; ++ci) {
                    DChain c = chains[base + ci];
                    const DSeed* sd = cseeds + base + c.seed_start;
                    int32_t* so = srt + base + c.seed_start;   // seed indices by (score, index) ascending
                    int32_t* done = sdone + base + c.seed_start;
                    const int n = c.n;
                    if (n == 0) continue;
                    i64 r0 = l_pac << 1, r1 = 0;
                    for (int i = 0; i < n; ++i) {
                        DSeed t = sd[i];
                        i64 b = t.rbeg - (t.qbeg + dev_cal_max_gap(o, t.qbeg));
                        i64 e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + dev_cal_max_gap(o, l_query - t.qbeg - t.len));
                        r0 = r0 < b ? r0 : b;
                        r1 = r1 > e ? r1 : e;
                    }
                    i64 rmax0 = r0 > 0 ? r0 : 0, rmax1 = r1 < l_pac << 1 ? r1 : l_pac << 1;
                    DSeed s0 = sd[0];
                    if (rmax0 < l_pac && l_pac < rmax1) {   // crossing the forward-reverse boundary; then choose one side
                        if (s0.rbeg < l_pac) rmax1 = l_pac;
                        else rmax0 = l_pac;
                    }
                    dev_fetch_clamp(ix, &rmax0, s0.rbeg, &rmax1);
                    win += (u64)(rmax1 - rmax0);
                    chain_rmax[2 * (base + ci)] = rmax0; chain_rmax[2 * (base + ci) + 1] = rmax1;
                    for (int i = 0; i < n; ++i) {   // by seed score (= len) then index, ascending
                        DSeed t = sd[i];
                        int rank = 0;
                        for (int u = 0; u < n; ++u) { DSeed x = sd[u]; rank += (x.len < t.len) || (x.len == t.len && u < i); }
                        so[rank] = i;
                        done[i] = 1;
                    }
                }
                nch_done = m;
            }
            if (fuse) ext_out = ext_control<false>(ix, o, A, r, nullptr, lane, &ext_key, &cells);
            }
        }
    }
    ext_append(ext_out, r, ext_key, lane, next_count, next_list, next_key, defer_count, defer_list);
    if (ctr) {
        unsigned w32 = (unsigned)win;   // < 2^32 window bases per read
        u64 wtot = (u64)(uint32_t)wave_sum_i32((int)(w32 >> 16)) << 16;
        wtot += (u64)(uint32_t)wave_sum_i32((int)(w32 & 0xffff));
        int ctot = wave_sum_i32(nch_done);
        if (lane == 0 && (wtot || ctot)) { atomicAdd(&LH_CTR(ctr)->win_bases, wtot); atomicAdd(&LH_CTR(ctr)->n_chain_ext, (u64)ctot); }
    }
    u64 hm = __ballot(heavy);
    if (hm) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(list_count, (int32_t)__popcll(hm));
        basep = wave_readlane(basep, 0);
        if (heavy) list[basep + lanes_below(hm, lane)] = r;
    }
}

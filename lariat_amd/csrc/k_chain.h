// k_chain.h — K3: seed chaining + chain filtering, one wavefront per read.
// Replaces the chaining half of BWA's mem_chain (kbtree insert / test_and_merge), mem_chain_weight and mem_chain_flt,
// reached through mem_align1_core (go/src/gobwa/gobwa.go:244,253).  mem_flt_chained_seeds is a no-op for reads this
// short.
//
// The ordered set of chains (upstream: a B-tree keyed by the first seed's position) is a position-sorted index array in
// HBM scratch: predecessor search is a wave-uniform binary search, insertion is a lane-parallel shift.  Seeds of a chain
// form a linked list (append only), flattened at the end.  The unstable introsort by weight and the greedy pairwise
// filter are order-dependent and run on lane 0.
#pragma once
#include "k_seed.h"
#include "lh_sort.h"

struct DChainTmp {
    i64 pos, last_rbeg;
    int32_t first_qbeg, last_qbeg, last_len, rid, n, head, tail, w, kept, first, beg, end;
};
struct DChain {
    i64 pos;
    int32_t rid, n, seed_start, w, kept, is_alt;
    float frac_rep;
    int32_t pad;
};

__global__ void __launch_bounds__(64) k_chain(DIndex ix, DOpts o, int n_reads, const i64* __restrict__ seq_off, const i64* __restrict__ seed_off,
                                               i64 pool_cap, const DSeed* __restrict__ seeds, const int32_t* __restrict__ s_rid,
                                               const int32_t* __restrict__ l_rep, int32_t* __restrict__ s_next, DChainTmp* __restrict__ ct,
                                               int32_t* __restrict__ ord, int32_t* __restrict__ srt, DChain* __restrict__ chains,
                                               DSeed* __restrict__ cseeds, int32_t* __restrict__ n_chains, int32_t* __restrict__ status,
                                               const int32_t* __restrict__ list, const int32_t* __restrict__ list_count) {
    const int lane = LANE();
    const int n_items = list ? *list_count : n_reads;   // with a list: the reads k_chain_lane left to this kernel
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int r = list ? list[item] : item;
    WAVE_SYNC();
    i64 base = seed_off[r];
    int S = (int)(seed_off[r + 1] - base);
    if (seed_off[r + 1] > pool_cap) {   // pool overflow: flag, produce nothing (host retries with a smaller batch)
        if (lane == 0) { n_chains[r] = 0; atomicOr(&status[r], LH_ST_POOL_OVERFLOW); }
        continue;
    }
    int len = (int)(seq_off[r + 1] - seq_off[r]);
    const DSeed* sd_ = seeds + base;
    const int32_t* rid_ = s_rid + base;
    int32_t* nx = s_next + base;
    DChainTmp* C = ct + base;
    int32_t* od = ord + base;
    int32_t* st = srt + base;
    int nch = 0;
    for (int s = 0; s < S; ++s) {
        int rid = rid_[s];
        if (rid < 0) continue;   // bridging contigs / the forward-reverse boundary
        DSeed p = sd_[s];
        int to_add = 1, lo = 0;
        if (nch > 0) {
            int hi = nch;
            while (lo < hi) { int m = (lo + hi) >> 1; if (C[od[m]].pos <= p.rbeg) lo = m + 1; else hi = m; }
            if (lo > 0) {
                int ci = od[lo - 1];
                DChainTmp c = C[ci];
                WAVE_SYNC();   // every lane has its copy before lane 0 updates the chain
                // test_and_merge
                i64 qend = c.last_qbeg + c.last_len, rend = c.last_rbeg + c.last_len;
                int res = 0;   // 0: new chain, 1: contained, 2: appended
                if (rid != c.rid) res = 0;
                else if (p.qbeg >= c.first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= c.pos && p.rbeg + p.len <= rend) res = 1;
                else if ((c.last_rbeg < ix.l_pac || c.pos < ix.l_pac) && p.rbeg >= ix.l_pac) res = 0;
                else {
                    i64 x = p.qbeg - c.last_qbeg, y = p.rbeg - c.last_rbeg;
                    if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.last_len < o.max_chain_gap && y - c.last_len < o.max_chain_gap) res = 2;
                }
                if (res == 2 && lane == 0) {
                    nx[c.tail] = s; nx[s] = -1;
                    C[ci].tail = s; C[ci].n = c.n + 1; C[ci].last_rbeg = p.rbeg; C[ci].last_qbeg = p.qbeg; C[ci].last_len = p.len;
                }
                to_add = (res == 0);
            }
        }
        if (to_add) {
            // insert chain id nch at sorted position lo: shift od[lo..nch) right by one, lane-parallel, from the top
            for (int top = nch; top > lo; top -= 64) {
                int j = top - 1 - lane;
                int v = j >= lo ? od[j] : 0;
                WAVE_SYNC();
                if (j >= lo) od[j + 1] = v;
                WAVE_SYNC();
            }
            if (lane == 0) {
                od[lo] = nch;
                DChainTmp c;
                c.pos = p.rbeg; c.last_rbeg = p.rbeg; c.first_qbeg = p.qbeg; c.last_qbeg = p.qbeg; c.last_len = p.len; c.rid = rid;
                c.n = 1; c.head = s; c.tail = s; c.w = 0; c.kept = 0; c.first = -1; c.beg = 0; c.end = 0;
                C[nch] = c;
                nx[s] = -1;
            }
            nch++;
        }
        WAVE_SYNC();
    }
    // mem_chain_weight for every chain (lane per chain), chn_beg / chn_end
    for (int k = lane; k < nch; k += 64) {
        DChainTmp c = C[k];
        int w = 0;
        i64 end = 0;
        for (int s = c.head; s >= 0; s = nx[s]) {
            DSeed t = sd_[s];
            if (t.qbeg >= end) w += t.len;
            else if (t.qbeg + t.len > end) w += t.qbeg + t.len - (int)end;
            end = end > t.qbeg + t.len ? end : t.qbeg + t.len;
        }
        int tmp = w;
        w = 0; end = 0;
        for (int s = c.head; s >= 0; s = nx[s]) {
            DSeed t = sd_[s];
            if (t.rbeg >= end) w += t.len;
            else if (t.rbeg + t.len > end) w += (int)(t.rbeg + t.len - end);
            end = end > t.rbeg + t.len ? end : t.rbeg + t.len;
        }
        w = w < tmp ? w : tmp;
        C[k].w = w < 1 << 30 ? w : (1 << 30) - 1;
        C[k].beg = c.first_qbeg;
        C[k].end = c.last_qbeg + c.last_len;
    }
    WAVE_SYNC();
    // mem_chain_flt (order dependent): lane 0
    if (lane == 0) {
        int n = 0;
        for (int k = 0; k < nch; ++k) {   // chains in position order (B-tree traversal), dropping light ones
            int id = od[k];
            C[id].first = -1; C[id].kept = 0;
            if (C[id].w < o.min_chain_weight) continue;
            st[n++] = id;
        }
        int nk = 0;   // kept-chain list reuses od[]
        if (n > 0) {
            dev_introsort(n, st, [&](int x, int y) { return C[x].w > C[y].w; });
            C[st[0]].kept = 3;
            od[nk++] = 0;
            for (int i = 1; i < n; ++i) {
                int large_ovlp = 0, k;
                DChainTmp ai = C[st[i]];
                for (k = 0; k < nk; ++k) {
                    int j = od[k];
                    DChainTmp aj = C[st[j]];
                    int b_max = aj.beg > ai.beg ? aj.beg : ai.beg;
                    int e_min = aj.end < ai.end ? aj.end : ai.end;
                    if (e_min > b_max) {   // have overlap (no ALT contigs: is_alt == 0)
                        int li = ai.end - ai.beg, lj = aj.end - aj.beg;
                        int min_l = li < lj ? li : lj;
                        if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
                            large_ovlp = 1;
                            if (aj.first < 0) C[st[j]].first = i;
                            if (ai.w < aj.w * o.drop_ratio && aj.w - ai.w >= o.min_seed_len << 1) break;
                        }
                    }
                }
                if (k == nk) { od[nk++] = i; C[st[i]].kept = large_ovlp ? 2 : 3; }
            }
            for (int i = 0; i < nk; ++i) {
                int f = C[st[od[i]]].first;
                if (f >= 0) C[st[f]].kept = 1;
            }
            int i, k;
            for (i = k = 0; i < n; ++i) {   // don't extend more than max_chain_extend .kept=1/2 chains
                int kp = C[st[i]].kept;
                if (kp == 0 || kp == 3) continue;
                if (++k >= o.max_chain_extend) break;
            }
            for (; i < n; ++i)
                if (C[st[i]].kept < 3) C[st[i]].kept = 0;
        }
        // emit kept chains in sorted order
        int m = 0, sstart = 0;
        for (int i = 0; i < n; ++i) {
            DChainTmp c = C[st[i]];
            if (c.kept == 0) continue;
            DChain oc;
            oc.pos = c.pos; oc.rid = c.rid; oc.n = c.n; oc.seed_start = sstart; oc.w = c.w; oc.kept = c.kept; oc.is_alt = 0;
            oc.frac_rep = (float)l_rep[r] / len; oc.pad = st[i];   // pad carries the tmp id for the flatten step
            chains[base + m] = oc;
            sstart += c.n;
            m++;
        }
        n_chains[r] = m;
    }
    WAVE_SYNC();
    int m = n_chains[r];
    for (int k = lane; k < m; k += 64) {   // flatten each chain's seed list
        DChain oc = chains[base + k];
        int s = C[oc.pad].head;
        for (int t = 0; s >= 0; s = nx[s], ++t) cseeds[base + oc.seed_start + t] = sd_[s];
    }
    }
}

// klib's ks_introsort for n <= 16: one median-of-3 partition step, then insertion sort (ranges of <= 16 elements are
// never pushed on its stack).  Per-lane control flow: callable from lane-per-read kernels.
#define LH_CHAIN_LANE_MAX 16
#define LH_EXT_COMPLEX_SEEDS 6   // K4 buckets (k_extend2.h): reads with more seeds go to the wave-per-read extension kernel
#define LH_EXT_SUB 8             // sub-buckets per primary bucket
// primary bucket of a read: 0 = many seeds (wave kernel); 1..16 first extension >= 128 columns (longest first), 17..24 64..127,
// 25 = no DP expected but >= 64 columns (LDS class 128 in case a later extension needs one); 26..33 < 64 columns; 34 = no DP
// expected, < 64 columns
#define LH_EXT_PRIMARY 35
#define LH_EXT_HEAVY_COLS 32     // a full-band extension of this many query columns or more is "heavy" for the lane kernel
#ifndef LH_NARROW_MAX_LOSS
#define LH_NARROW_MAX_LOSS 16    // diagonal loss up to which the lane DP runs in a narrow band (3 mismatches with the default scoring); a long side that loses more goes to the wave-per-read kernel: one such DP is ~2,500 cells = 0.2 ms on a single lane, the tail of a whole round
#endif
__device__ __forceinline__ int lh_ext_bucket(int nseeds, int longest, int cheap) {
    if (nseeds > LH_EXT_COMPLEX_SEEDS) return 0;
    int L = longest >> 3 < 31 ? longest >> 3 : 31;
    if (cheap) return L >= 8 ? 25 : 34;
    return L >= 8 ? 32 - L : 33 - L;
}
template <class T, class Lt> __device__ inline void dev_introsort_small(int n, T* a, Lt lt) {
    T rp, swap_tmp;
    if (n < 1) return;
    if (n == 2) {
        if (lt(a[1], a[0])) { swap_tmp = a[0]; a[0] = a[1]; a[1] = swap_tmp; }
        return;
    }
    if (n > 1) {
        T *s = a, *t = a + (n - 1), *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
        if (lt(*k, *i)) {
            if (lt(*k, *j)) k = j;
        } else k = lt(*j, *i) ? i : j;
        rp = *k;
        if (k != t) { swap_tmp = *k; *k = *t; *t = swap_tmp; }
        int wd = 4 * LH_CHAIN_LANE_MAX + 8;
        for (;;) {
            do { ++i; LH_WATCH(wd, 4, return) } while (lt(*i, rp));
            do { --j; LH_WATCH(wd, 5, return) } while (i <= j && lt(rp, *j));
            if (j <= i) break;
            swap_tmp = *i; *i = *j; *j = swap_tmp;
        }
        swap_tmp = *i; *i = *t; *t = swap_tmp;
    }
    for (T* i = a + 1; i < a + n; ++i)
        for (T* j = i; j > a && lt(*j, *(j - 1)); --j) { T tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}

// K3 for the common read: one LANE per read when it has at most LH_CHAIN_LANE_MAX seeds (mem_chain is a short sequential
// program over a handful of seeds; a wave per read spent its time on launch and single-lane latency: 5.4 ms per 2 M
// reads).  Same arithmetic and the same order of B-tree / sort / filter operations as the wave kernel above, whose
// single-lane sections appear here inline.  Reads with more seeds are listed for the wave kernel.
#ifndef LH_CHAIN_LANE_WAVES
#define LH_CHAIN_LANE_WAVES 4
#endif
__global__ void __launch_bounds__(64, LH_CHAIN_LANE_WAVES) k_chain_lane(DIndex ix, DOpts o, int n_reads, const i64* __restrict__ seq_off, const i64* __restrict__ seed_off,
                                                    i64 pool_cap, const DSeed* __restrict__ seeds, const int32_t* __restrict__ s_rid,
                                                    const int32_t* __restrict__ l_rep, int32_t* __restrict__ s_next, DChainTmp* __restrict__ ct,
                                                    int32_t* __restrict__ ord, int32_t* __restrict__ srt, DChain* __restrict__ chains,
                                                    DSeed* __restrict__ cseeds, int32_t* __restrict__ n_chains, int32_t* __restrict__ status,
                                                    int32_t* __restrict__ list, int32_t* __restrict__ list_count,
                                                    int32_t* __restrict__ sdone, i64* __restrict__ chain_rmax, int32_t* __restrict__ ext_key, DCounters* __restrict__ ctr,
                                                    const uint8_t* __restrict__ seq) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int heavy = 0;
    u64 win = 0;
    int nch_done = 0;
    if (r < n_reads) {
        const i64 base = seed_off[r];
        const int S = (int)(seed_off[r + 1] - base);
        if (seed_off[r + 1] > pool_cap) { n_chains[r] = 0; ext_key[r] = 34 * LH_EXT_SUB; atomicOr(&status[r], LH_ST_POOL_OVERFLOW); }
        else if (S > LH_CHAIN_LANE_MAX) heavy = 1;
        else {
            const int len = (int)(seq_off[r + 1] - seq_off[r]);
            const DSeed* sd_ = seeds + base;
            const int32_t* rid_ = s_rid + base;
            int32_t* nx = s_next + base;
            DChainTmp* C = ct + base;
            int32_t* od = ord + base;
            int32_t* st = srt + base;
            int nch = 0;
            // the chain the previous seed was tested against stays in registers (cc, index cci; dirty = it differs from C[cci]): consecutive
            // seeds of a read mostly meet the same chain, and every field update used to be a store of its own
            DChainTmp cc;
            cc.pos = 0; cc.last_rbeg = 0; cc.first_qbeg = cc.last_qbeg = cc.last_len = cc.rid = cc.n = cc.head = cc.tail = cc.w = cc.kept = cc.first = cc.beg = cc.end = 0;
            int cci = -1, dirty = 0;
            for (int s = 0; s < S; ++s) {
                int rid = rid_[s];
                if (rid < 0) continue;   // bridging contigs / the forward-reverse boundary
                DSeed p = sd_[s];
                int to_add = 1, lo = 0;
                if (nch > 0) {
                    int hi = nch;
                    while (lo < hi) { int m = (lo + hi) >> 1; if (C[od[m]].pos <= p.rbeg) lo = m + 1; else hi = m; }   // (pos never changes once a chain exists)
                    if (lo > 0) {
                        int ci = od[lo - 1];
                        if (ci != cci) {
                            if (dirty) C[cci] = cc;
                            cc = C[ci]; cci = ci; dirty = 0;
                        }
                        const DChainTmp c = cc;
                        i64 qend = c.last_qbeg + c.last_len, rend = c.last_rbeg + c.last_len;   // test_and_merge
                        int res = 0;   // 0: new chain, 1: contained, 2: appended
                        if (rid != c.rid) res = 0;
                        else if (p.qbeg >= c.first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= c.pos && p.rbeg + p.len <= rend) res = 1;
                        else if ((c.last_rbeg < ix.l_pac || c.pos < ix.l_pac) && p.rbeg >= ix.l_pac) res = 0;
                        else {
                            i64 x = p.qbeg - c.last_qbeg, y = p.rbeg - c.last_rbeg;
                            if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.last_len < o.max_chain_gap && y - c.last_len < o.max_chain_gap) res = 2;
                        }
                        if (res == 2) {
                            nx[c.tail] = s; nx[s] = -1;
                            cc.tail = s; cc.n = c.n + 1; cc.last_rbeg = p.rbeg; cc.last_qbeg = p.qbeg; cc.last_len = p.len;
                            dirty = 1;
                        }
                        to_add = (res == 0);
                    }
                }
                if (to_add) {
                    for (int j = nch; j > lo; --j) od[j] = od[j - 1];
                    od[lo] = nch;
                    DChainTmp c;
                    c.pos = p.rbeg; c.last_rbeg = p.rbeg; c.first_qbeg = p.qbeg; c.last_qbeg = p.qbeg; c.last_len = p.len; c.rid = rid;
                    c.n = 1; c.head = s; c.tail = s; c.w = 0; c.kept = 0; c.first = -1; c.beg = 0; c.end = 0;
                    C[nch] = c;
                    nx[s] = -1;
                    nch++;
                }
            }
            if (dirty) C[cci] = cc;
            for (int k = 0; k < nch; ++k) {   // mem_chain_weight, chn_beg / chn_end
                DChainTmp c = C[k];
                int w = 0;
                i64 end = 0;
                for (int s = c.head; s >= 0; s = nx[s]) {
                    DSeed t = sd_[s];
                    if (t.qbeg >= end) w += t.len;
                    else if (t.qbeg + t.len > end) w += t.qbeg + t.len - (int)end;
                    end = end > t.qbeg + t.len ? end : t.qbeg + t.len;
                }
                int tmp = w;
                w = 0; end = 0;
                for (int s = c.head; s >= 0; s = nx[s]) {
                    DSeed t = sd_[s];
                    if (t.rbeg >= end) w += t.len;
                    else if (t.rbeg + t.len > end) w += (int)(t.rbeg + t.len - end);
                    end = end > t.rbeg + t.len ? end : t.rbeg + t.len;
                }
                w = w < tmp ? w : tmp;
                C[k].w = w < 1 << 30 ? w : (1 << 30) - 1;
                C[k].beg = c.first_qbeg;
                C[k].end = c.last_qbeg + c.last_len;
            }
            // mem_chain_flt
            int n = 0;
            for (int k = 0; k < nch; ++k) {   // chains in position order (B-tree traversal), dropping light ones
                int id = od[k];
                C[id].first = -1; C[id].kept = 0;
                if (C[id].w < o.min_chain_weight) continue;
                st[n++] = id;
            }
            int nk = 0;   // kept-chain list reuses od[]
            if (n > 0) {
                dev_introsort_small(n, st, [&](int x, int y) { return C[x].w > C[y].w; });
                C[st[0]].kept = 3;
                od[nk++] = 0;
                for (int i = 1; i < n; ++i) {
                    int large_ovlp = 0, k;
                    DChainTmp ai = C[st[i]];
                    for (k = 0; k < nk; ++k) {
                        int j = od[k];
                        DChainTmp aj = C[st[j]];
                        int b_max = aj.beg > ai.beg ? aj.beg : ai.beg;
                        int e_min = aj.end < ai.end ? aj.end : ai.end;
                        if (e_min > b_max) {   // have overlap (no ALT contigs: is_alt == 0)
                            int li = ai.end - ai.beg, lj = aj.end - aj.beg;
                            int min_l = li < lj ? li : lj;
                            if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
                                large_ovlp = 1;
                                if (aj.first < 0) C[st[j]].first = i;
                                if (ai.w < aj.w * o.drop_ratio && aj.w - ai.w >= o.min_seed_len << 1) break;
                            }
                        }
                    }
                    if (k == nk) { od[nk++] = i; C[st[i]].kept = large_ovlp ? 2 : 3; }
                }
                for (int i = 0; i < nk; ++i) {
                    int f = C[st[od[i]]].first;
                    if (f >= 0) C[st[f]].kept = 1;
                }
                int i, k;
                for (i = k = 0; i < n; ++i) {   // don't extend more than max_chain_extend .kept=1/2 chains
                    int kp = C[st[i]].kept;
                    if (kp == 0 || kp == 3) continue;
                    if (++k >= o.max_chain_extend) break;
                }
                for (; i < n; ++i)
                    if (C[st[i]].kept < 3) C[st[i]].kept = 0;
            }
            int m = 0, sstart = 0;   // emit kept chains in sorted order, each with its seed list flattened
            for (int i = 0; i < n; ++i) {
                DChainTmp c = C[st[i]];
                if (c.kept == 0) continue;
                DChain oc;
                oc.pos = c.pos; oc.rid = c.rid; oc.n = c.n; oc.seed_start = sstart; oc.w = c.w; oc.kept = c.kept; oc.is_alt = 0;
                oc.frac_rep = (float)l_rep[r] / len; oc.pad = st[i];
                chains[base + m] = oc;
                int t = 0;
                for (int s = c.head; s >= 0; s = nx[s], ++t) cseeds[base + sstart + t] = sd_[s];
                sstart += c.n;
                m++;
            }
            n_chains[r] = m;
            // K4's pre-pass for this read (k_ext_prep does it for the reads of the wave kernel): per kept chain the reference
            // window of mem_chain2aln and the order in which its seeds are extended; per read the bucket of k_extend_lane.
            // ord[] / srt[] are free again: srt[] receives the seed order, sdone[] the "extended" flags.
            {
                int l_query = len > LH_MAXLEN ? 0 : len;
                const i64 l_pac = ix.l_pac;
                int longest = 0, shorter = 0, nseeds = 0, have_top = 0;
                DSeed top;
                i64 top_r0 = 0, top_r1 = 0;
                top.rbeg = 0; top.qbeg = 0; top.len = 0;
                for (int ci = 0; ci < m; ++ci) {
                    DChain c = chains[base + ci];
                    const DSeed* sd = cseeds + base + c.seed_start;
                    int32_t* so = srt + base + c.seed_start;
                    int32_t* done = sdone + base + c.seed_start;
                    const int n = c.n;
                    nseeds += n;
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
                        if (ci == 0 && rank == n - 1) {
                            int lt = t.qbeg, rt = l_query - t.qbeg - t.len;
                            longest = lt > rt ? lt : rt; shorter = lt > rt ? rt : lt;
                            top = t; top_r0 = rmax0; top_r1 = rmax1; have_top = 1;
                        }
                    }
                }
                // Is the first seed's extension provably ungapped on both sides (k_extend2.h: diagonal penalties below one gap's
                // cost)?  Such reads usually need no DP at all; they get their own bucket so that whole waves skip it.
                // And is one of them long and outside what k_extend_lane's narrow band covers (diagonal loss of LH_NARROW_MAX_LOSS or
                // more, e.g. behind an indel)?  One full-band DP keeps a whole wave of the lane kernel busy for ~1 ms: those reads go
                // to the wave-per-read kernel with the complex ones.  Routing only: K4 decides again from the same scan.
                int cheap = 0, heavy = 0;
                if (have_top && nseeds <= LH_EXT_COMPLEX_SEEDS) {
                    const int thr = (o.o_ins + o.e_ins) < (o.o_del + o.e_del) ? (o.o_ins + o.e_ins) : (o.o_del + o.e_del);
                    const uint8_t* q = seq + seq_off[r];
                    cheap = 1;
                    for (int side = 0; side < 2 && cheap; ++side) {
                        int qlen = side ? l_query - top.qbeg - top.len : top.qbeg;
                        i64 tlen = side ? top_r1 - (top.rbeg + top.len) : top.rbeg - top_r0;
                        if (qlen == 0) continue;
                        if (tlen < qlen) { cheap = 0; heavy |= qlen >= LH_EXT_HEAVY_COLS; continue; }
                        LaneTgt tg;
                        tg.init(ix, side ? top.rbeg + top.len : top.rbeg - 1, side ? 1 : -1);
                        int P = 0, run = top.len * o.a;
                        for (int k = 0; k < qlen; ++k) {
                            int qv = q[side ? top.qbeg + top.len + k : top.qbeg - 1 - k], tb = tg.base(k);
                            int loss = qv > 3 ? o.a + 1 : (tb == qv ? 0 : o.a + o.b);
                            P += loss; run += o.a - loss;
                            if (P >= thr) cheap = 0;
                            if (P >= LH_NARROW_MAX_LOSS || run <= 0) { heavy |= qlen >= LH_EXT_HEAVY_COLS; break; }
                            if (!cheap && qlen < LH_EXT_HEAVY_COLS) break;   // nothing left to learn from a short side
                        }
                    }
                }
                // a second chain (at human-genome scale every third read has one: a chance match kept as the first shadowed chain)
                // needs a DP even when the first chain does not: the read goes to an LDS class, where that DP is a few dozen cells
                int nz_ = 0;
                for (int ci = 0; ci < m; ++ci) nz_ += chains[base + ci].n > 0;
                const int cheap_top = cheap;
                if (nz_ >= 2) cheap = 0;
                int prim = lh_ext_bucket(heavy ? LH_EXT_COMPLEX_SEEDS + 1 : nseeds, longest, cheap);
                // sub-bucket: reads of one wave should do the same kind of work at the same time (the lanes run their chains one
                // after the other): a second chain or not, a DP for the first chain or not, then the length of the shorter side
                int sub = cheap ? (shorter >> 4 < LH_EXT_SUB - 1 ? shorter >> 4 : LH_EXT_SUB - 1) : (nz_ >= 2 ? 4 : 0) + (cheap_top ? 2 : 0) + (shorter >= 32 ? 1 : 0);
                ext_key[r] = prim * LH_EXT_SUB + sub;
                nch_done = m;
            }
        }
    }
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

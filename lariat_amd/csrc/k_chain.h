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
#ifndef LH_CHAIN_LDS
#define LH_CHAIN_LDS 256   // seeds (and so chains) of a read whose tables k_chain keeps in LDS
#endif
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
    // a read's chain table, its two index arrays and its seeds live in LDS while they fit (reads on repeat families have dozens of chains and a
    // hundred seeds: every step of the loops below is a dependent access, ~1 us from HBM); larger reads use the HBM scratch as before
    __shared__ DChainTmp Cs[LH_CHAIN_LDS];
    __shared__ i64 okk[LH_CHAIN_LDS + 1];   // while the read's seeds are being chained: one key per chain, (pos << 16 | chain); afterwards the two index arrays
    int32_t* const ods = (int32_t*)okk;
    int32_t* const sts = ods + LH_CHAIN_LDS + 1;
    __shared__ int32_t nxs[LH_CHAIN_LDS], rids[LH_CHAIN_LDS];
    static_assert(LH_CHAIN_LDS < 512, "chain ids in 9 bits of the packed sort keys");
    __shared__ DSeed sds[LH_CHAIN_LDS];
    __shared__ int32_t sh_n;
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
    const bool in_lds = S <= LH_CHAIN_LDS;
    const DSeed* sd_ = in_lds ? sds : seeds + base;
    const int32_t* rid_ = in_lds ? rids : s_rid + base;
    int32_t* nx = in_lds ? nxs : s_next + base;
    DChainTmp* C = in_lds ? Cs : ct + base;
    int32_t* od = in_lds ? ods : ord + base;
    int32_t* st = in_lds ? sts : srt + base;
    if (in_lds) {
        for (int s = lane; s < S; s += 64) { sds[s] = seeds[base + s]; rids[s] = s_rid[base + s]; }
        WAVE_SYNC();
    }
    int nch = 0;
    if (in_lds) {
        // (r04) BWA keeps the chains in a B-tree by position and asks it for the chain before the seed.  Here the chains' keys (pos << 16 | chain:
        // chains of equal pos in the order they were made, as the tree holds them) sit in LDS in NO order and the chain before the seed is a
        // maximum over them — one read and one reduction per seed instead of a binary search of dependent reads and a shifted insertion; the
        // tree's traversal order, which mem_chain_flt starts from, is made once at the end, by counting.
        for (int s = 0; s < S; ++s) {
            const int rid = rids[s];
            if (rid < 0) continue;   // bridging contigs / the forward-reverse boundary
            const DSeed p = sds[s];
            i64 bestk = -1;
            for (int k = lane; k < nch; k += 64) { const i64 v = okk[k]; if ((v >> 16) <= p.rbeg && v > bestk) bestk = v; }
            bestk = wave_max_i64(bestk);
            int to_add = 1;
            if (bestk >= 0) {
                const int ci = (int)(bestk & 0xffff);
                const DChainTmp c = Cs[ci];
                WAVE_SYNC();   // every lane has its copy before lane 0 updates the chain
                const i64 qend = c.last_qbeg + c.last_len, rend = c.last_rbeg + c.last_len;   // test_and_merge
                int res = 0;   // 0: new chain, 1: contained, 2: appended
                if (rid != c.rid) res = 0;
                else if (p.qbeg >= c.first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= c.pos && p.rbeg + p.len <= rend) res = 1;
                else if ((c.last_rbeg < ix.l_pac || c.pos < ix.l_pac) && p.rbeg >= ix.l_pac) res = 0;
                else {
                    const i64 x = p.qbeg - c.last_qbeg, y = p.rbeg - c.last_rbeg;
                    if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.last_len < o.max_chain_gap && y - c.last_len < o.max_chain_gap) res = 2;
                }
                if (res == 2 && lane == 0) {
                    nxs[c.tail] = s; nxs[s] = -1;
                    Cs[ci].tail = s; Cs[ci].n = c.n + 1; Cs[ci].last_rbeg = p.rbeg; Cs[ci].last_qbeg = p.qbeg; Cs[ci].last_len = p.len;
                }
                to_add = (res == 0);
            }
            if (to_add) {
                if (lane == 0) {
                    okk[nch] = (i64)p.rbeg << 16 | (i64)nch;
                    DChainTmp c;
                    c.pos = p.rbeg; c.last_rbeg = p.rbeg; c.first_qbeg = p.qbeg; c.last_qbeg = p.qbeg; c.last_len = p.len; c.rid = rid;
                    c.n = 1; c.head = s; c.tail = s; c.w = 0; c.kept = 0; c.first = -1; c.beg = 0; c.end = 0;
                    Cs[nch] = c;
                    nxs[s] = -1;
                }
                nch++;
            }
            WAVE_SYNC();
        }
        {   // od[] = the chains by (pos, order of creation)
            constexpr int PER = (LH_CHAIN_LDS + 63) / 64;
            int rk[PER];
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int k = t * 64 + lane;
                rk[t] = -1;
                if (k < nch) {
                    const i64 v = okk[k];
                    int c_ = 0;
                    for (int u = 0; u < nch; ++u) c_ += okk[u] < v;
                    rk[t] = c_;
                }
            }
            WAVE_SYNC();
#pragma unroll
            for (int t = 0; t < PER; ++t) if (rk[t] >= 0) ods[rk[t]] = t * 64 + lane;
            WAVE_SYNC();
        }
    } else
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
    // mem_chain_flt (order dependent): the unstable introsort by weight on lane 0; the greedy scan chain by chain, every chain against the
    // kept ones so far by the whole wave (the reference's inner loop stops at the first kept chain that shadows it: the lowest set bit)
    int n = 0;
    if (lane == 0) {
        for (int k = 0; k < nch; ++k) {   // chains in position order (B-tree traversal), dropping light ones
            int id = od[k];
            C[id].first = -1; C[id].kept = 0;
            if (C[id].w < o.min_chain_weight) continue;
            st[n++] = id;
        }
        if (n > 0) {
            if (in_lds) {   // the same sort on (weight << 9 | chain): what the comparisons see is the weight, what moves is one word — no second, dependent read per comparison
                for (int k = 0; k < n; ++k) st[k] = C[st[k]].w << 9 | st[k];   // (a chain's weight is at most the read's length)
                dev_introsort(n, st, [&](int x, int y) { return (x >> 9) > (y >> 9); }, o.wd);
                for (int k = 0; k < n; ++k) st[k] &= 511;
            } else dev_introsort(n, st, [&](int x, int y) { return C[x].w > C[y].w; }, o.wd);
            C[st[0]].kept = 3;
            od[0] = 0;
        }
        sh_n = n;
    }
    WAVE_SYNC();
    n = sh_n;
    if (n > 0) {
        int nk = 1;   // kept-chain list reuses od[]
        for (int i = 1; i < n; ++i) {
            const DChainTmp ai = C[st[i]];
            const int ai_alt = ix.contig_alt && ix.contig_alt[ai.rid];
            int large_ovlp = 0, shadowed = 0;
            for (int k0 = 0; k0 < nk && !shadowed; k0 += 64) {
                const int k = k0 + lane;
                int ov = 0, brk = 0, j = 0;
                if (k < nk) {
                    j = od[k];
                    const DChainTmp aj = C[st[j]];
                    int b_max = aj.beg > ai.beg ? aj.beg : ai.beg;
                    int e_min = aj.end < ai.end ? aj.end : ai.end;
                    if (e_min > b_max && (!(ix.contig_alt && ix.contig_alt[aj.rid]) || ai_alt)) {   // have overlap; don't consider ovlp where the kept chain is ALT while the current chain is primary
                        int li = ai.end - ai.beg, lj = aj.end - aj.beg;
                        int min_l = li < lj ? li : lj;
                        if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
                            ov = 1;
                            brk = ai.w < aj.w * o.drop_ratio && aj.w - ai.w >= o.min_seed_len << 1;
                        }
                    }
                }
                const u64 mbrk = __ballot(brk);
                const int kb = mbrk ? __ffsll((unsigned long long)mbrk) - 1 : 64;   // the scan ends AT the first shadowing chain
                if (ov && lane <= kb && C[st[j]].first < 0) C[st[j]].first = i;
                large_ovlp |= __any(ov && lane <= kb);
                shadowed = mbrk != 0;
            }
            WAVE_SYNC();
            if (!shadowed) {
                if (lane == 0) { od[nk] = i; C[st[i]].kept = large_ovlp ? 2 : 3; }
                nk++;
                WAVE_SYNC();
            }
        }
        if (lane == 0) {
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
    }
    WAVE_SYNC();
    int m = 0;
    {   // emit kept chains in sorted order
        int sstart = 0;
        const float frac_rep = (float)l_rep[r] / len;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            int kept = 0, cn = 0;
            DChainTmp c;
            if (i < n) { c = C[st[i]]; kept = c.kept != 0; cn = kept ? c.n : 0; }
            const u64 km = __ballot(kept);
            const int incl = wave_scan_add_i32(cn);
            if (kept) {
                DChain oc;
                oc.pos = c.pos; oc.rid = c.rid; oc.n = c.n; oc.seed_start = sstart + incl - cn; oc.w = c.w; oc.kept = c.kept; oc.is_alt = (ix.contig_alt && ix.contig_alt[c.rid]) ? 1 : 0;
                oc.frac_rep = frac_rep; oc.pad = st[i];   // pad carries the tmp id for the flatten step
                chains[base + m + lanes_below(km, lane)] = oc;
            }
            m += (int)__popcll(km);
            sstart += wave_readlane(incl, 63);
        }
        if (lane == 0) n_chains[r] = m;
    }
    WAVE_SYNC();
    for (int k = lane; k < m; k += 64) {   // flatten each chain's seed list
        DChain oc = chains[base + k];
        int s = C[oc.pad].head;
        for (int t = 0; s >= 0; s = nx[s], ++t) cseeds[base + oc.seed_start + t] = sd_[s];
    }
    }
}

#ifndef LH_CHAIN_LANE_MAX
#define LH_CHAIN_LANE_MAX 32     // seeds a lane chains (k_chain_lane.h: one nibble per seed in two registers); reads with more are chained by the wave kernel above
#endif
#ifndef LH_EXT_HEAVY_COLS
#define LH_EXT_HEAVY_COLS 64     // a full-band extension of this many query columns or more is "heavy" for a lane (k_extend2.h); measured 32 / 48 / 64 / 80 / 96: K4 8.87 / 8.68 / 8.33 / 8.79 / 8.99 ms (the deferred reads run in the wave kernel beside the rounds just as well: no difference)
#endif
#ifndef LH_NARROW_MAX_LOSS
#define LH_NARROW_MAX_LOSS 36    // diagonal loss up to which the lane DP runs in a narrow band: 7 mismatches with the default scoring, band 29 — the widest the circular 64-word window holds
                                 // (LH_EXT_CIRC_MAX_W).  r05: 26 -> 36; between two copies of a repeat family 6 or 7 mismatches in a 100-base side are common, and at 26 those sides ran the full
                                 // band in a live-interval window that outgrows a lane (1.32 M -> 0.96 M calls for k_ext_wround per 400 k pairs of configs[4], K4 124 -> 108 ms)
#endif

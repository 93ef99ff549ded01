// k_chain_cl.h — K3 for the reads of repeat families (r05): one wavefront per read, its seeds chained CLUSTER BY CLUSTER.
// Replaces, like k_chain.h, the chaining half of BWA's mem_chain (kbtree insert / test_and_merge), mem_chain_weight and mem_chain_flt,
// reached through mem_align1_core (go/src/gobwa/gobwa.go:244,253).
//
// A read on a copy of a repeat family has a hundred seeds (p90 245, max > 1,000) that fall into 60-odd position clusters of one to
// five seeds — one per copy — and end up as 50 chains of 1.5 seeds.  mem_chain walks the seeds in order and asks its B-tree for the
// chain with the greatest position <= the seed's: k_chain does that with one wave-wide maximum per seed, 45 k instructions per read,
// most of them one lane's.  But the walk only ever couples seeds that lie close together:
//   * test_and_merge(chain c, seed p) can answer "contained" or "append" only if p.rbeg - c.last_rbeg < len + w (contained: p ends
//     before c's last seed does; append: y - x <= w with x < len), so a seed further than GAP = len + w above every seed of a chain
//     starts a chain of its own whatever that chain is, exactly as if the tree had held no chain below it;
//   * with the seeds sorted by position and cut wherever two neighbours are more than GAP apart, every chain's seeds lie in ONE
//     cluster (induction over the walk), the predecessor the tree returns for a seed is either a chain of the seed's own cluster —
//     then it is the one a walk over that cluster alone would find, ties among equal positions going to the chain made last, as the
//     tree keeps them — or a chain of a lower cluster, which by the first point changes nothing.
// So the walk over the read = independent walks over its clusters, each in the read's seed order: ONE LANE PER CLUSTER.  The order
// the tree is traversed in (mem_chain_flt's starting order, which its unstable sort's result depends on) is cluster by cluster,
// inside a cluster by (position, order of creation).  mem_chain_weight is kept incrementally per chain.  mem_chain_flt: the
// introsort by weight runs on keys held in REGISTERS, read and written by lane index (v_readlane, compare + select: no memory latency in
// its dependent chain; the same comparisons and moves as ks_introsort, lh_sort.h); the greedy scan keeps every chain's (beg, end,
// weight, alt) as one word per lane and decides a chain against all kept ones with one ballot.
// Reads with a cluster of more than LH_CHAIN_CL_MAX seeds (tandem repeats, low complexity: one lane would walk hundreds) or more
// seeds than the largest instance holds go to k_chain, which takes any read.
#pragma once
#include "k_chain.h"

#ifndef LH_CHAIN_CL_MAX
#define LH_CHAIN_CL_MAX 32       // seeds of one cluster a lane walks; a read with a larger cluster goes to k_chain
#endif
#ifndef LH_CHAIN_CL_CAP_A
#define LH_CHAIN_CL_CAP_A 256    // seeds of a read the first instance holds (16 KB of LDS per wave: 92 % of the listed reads on repeat families) ...
#define LH_CHAIN_CL_CAP_B 512    // ... the second (8 %) ...
#define LH_CHAIN_CL_CAP_C 1024   // ... and the third (0.5 %)
#endif

// development aid (tools/prof_rfa.sh: -DLH_RFA_PROF): shader clocks per phase, summed over the waves
#ifdef LH_RFA_PROF
__device__ unsigned long long lh_chain_prof[16];
#define CH_PROF(k_) { const unsigned long long now_ = (unsigned long long)clock64(); if (lane == 0) atomicAdd(&lh_chain_prof[k_], now_ - prof_t_); prof_t_ = (unsigned long long)clock64(); }
#else
#define CH_PROF(k_)
#endif
struct ChSlot { uint16_t head, tail, n; uint8_t wq, endq; int32_t wr, endr_rel; };   // a chain while it is being built: first / last seed, seeds, mem_chain_weight's two running sums
static_assert(sizeof(ChSlot) == 16, "ChSlot");

// ---- ks_introsort on an index space (the algorithm of dev_introsort, lh_sort.h, with get / set instead of pointers): the same comparisons
// and moves in the same order, so the same order of equal keys.  All branches on values that get() returns: wave-uniform when get() is. ----
template <class G, class St, class Lt> __device__ __forceinline__ void ix_insertsort(int s, int t, G& get, St& set, Lt& lt) {
    for (int i = s + 1; i < t; ++i) {   // (the element on its way down is held in a register: the array after every outer step is the one the swaps leave)
        const auto x = get(i);
        int j = i;
        for (; j > s; --j) {
            const auto y = get(j - 1);
            if (!lt(x, y)) break;
            set(j, y);
        }
        if (j != i) set(j, x);
    }
}
template <class G, class St, class Lt> __device__ __forceinline__ void ix_combsort(int n, int a, G& get, St& set, Lt& lt) {
    const double shrink_factor = 1.2473309501039786540366528676643;
    int do_swap, gap = n;
    do {
        if (gap > 2) {
            gap = (int)(gap / shrink_factor);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (int i = a; i < a + n - gap; ++i) {
            const int j = i + gap;
            const auto x = get(j), y = get(i);
            if (lt(x, y)) { set(i, x); set(j, y); do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) ix_insertsort(a, a + n, get, set, lt);
}
template <class G, class St, class Lt> __device__ __forceinline__ void dev_introsort_ix(int n, G get, St set, Lt lt, LhIsortStk* stack, int32_t* wdp) {
    if (n < 1) return;
    if (n == 2) {
        const auto x = get(1), y = get(0);
        if (lt(x, y)) { set(0, x); set(1, y); }
        return;
    }
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
    int top = 0, s = 0, t = n - 1;
    d <<= 1;
    int wd = 100000 + 64 * n;
    while (1) {
        LH_WATCH_S(wdp, wd, 1, return)
        if (s < t) {
            if (--d == 0) {
                ix_combsort(t - s + 1, s, get, set, lt);
                t = s;
                continue;
            }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            {
                const auto vk = get(k), vi = get(i), vj = get(j);
                if (lt(vk, vi)) {
                    if (lt(vk, vj)) k = j;
                } else k = lt(vj, vi) ? i : j;
            }
            const auto rp = get(k);
            if (k != t) { const auto vt = get(t); set(k, vt); set(t, rp); }
            auto vi = rp, vj = rp;
            for (;;) {   // (a[t] holds the pivot throughout: j never comes back to t)
                do { ++i; vi = get(i); LH_WATCH_S(wdp, wd, 2, return) } while (lt(vi, rp));
                for (;;) { --j; LH_WATCH_S(wdp, wd, 3, return) if (!(i <= j)) break; vj = get(j); if (!lt(rp, vj)) break; }
                if (j <= i) break;
                set(i, vj); set(j, vi);
            }
            set(i, rp); set(t, vi);
            if (i - s > t - i) {
                if (i - s > 16) { if (top >= LH_ISORT_STK) { wdp[1] = 1; return; } stack[top].left = s; stack[top].right = i - 1; stack[top].depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { if (top >= LH_ISORT_STK) { wdp[1] = 1; return; } stack[top].left = i + 1; stack[top].right = t; stack[top].depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) {
                ix_insertsort(0, n, get, set, lt);
                return;
            } else { --top; s = stack[top].left; t = stack[top].right; d = stack[top].depth; }
        }
    }
}

// an array of up to NV * 64 words in NV registers of the wave, element i = lane i & 63 of register i >> 6; i must be wave-uniform.  All lanes call.
#ifndef LH_EMU
template <int NV> struct WaveRegs {
    int v[NV];
    __device__ __forceinline__ int get(int i) const {
        const int l = __builtin_amdgcn_readfirstlane(i & 63), q = __builtin_amdgcn_readfirstlane(i >> 6);
        int r = __builtin_amdgcn_readlane(v[0], l);
#pragma unroll
        for (int t = 1; t < NV; ++t) { const int x = __builtin_amdgcn_readlane(v[t], l); r = q == t ? x : r; }
        return r;
    }
    __device__ __forceinline__ void set(int i, int x) {   // (a compare and a select: v_writelane_b32 wants its lane select in M0 on gfx9, which inline assembly may not clobber)
        const int l = __builtin_amdgcn_readfirstlane(i & 63), q = __builtin_amdgcn_readfirstlane(i >> 6);
        const int me = LANE();
#pragma unroll
        for (int t = 0; t < NV; ++t) v[t] = (q == t && me == l) ? x : v[t];
    }
};
#endif

#ifndef LH_EMU
template <int NV, class Lt> __device__ __forceinline__ void sort_in_regs(int n, int32_t* st, int lane, Lt lt, LhIsortStk* stk, int32_t* wdp) {
    WaveRegs<NV> w;
#pragma unroll
    for (int t = 0; t < NV; ++t) w.v[t] = t * 64 + lane < n ? st[t * 64 + lane] : 0;
    dev_introsort_ix(n, [&](int i) { return w.get(i); }, [&](int i, int x) { w.set(i, x); }, lt, stk, wdp);
#pragma unroll
    for (int t = 0; t < NV; ++t) if (t * 64 + lane < n) st[t * 64 + lane] = w.v[t];
}
#endif

// mem_chain_flt's test of chain i against kept chain j, both as packed words (beg | end << 8 | weight << 16 | is_alt << 25): *ov = significant overlap,
// returns whether j shadows i
__device__ __forceinline__ int flt_pair(const DOpts& o, uint32_t di, uint32_t dj, int* ov) {
    const int ib = (int)(di & 255), ie = (int)(di >> 8 & 255), iw = (int)(di >> 16 & 511), ialt = (int)(di >> 25 & 1);
    const int jb = (int)(dj & 255), je = (int)(dj >> 8 & 255), jw = (int)(dj >> 16 & 511), jalt = (int)(dj >> 25 & 1);
    const int b_max = jb > ib ? jb : ib, e_min = je < ie ? je : ie;
    *ov = 0;
    if (e_min > b_max && (!jalt || ialt)) {   // have overlap; don't consider ovlp where the kept chain is ALT while the current chain is primary
        const int li = ie - ib, lj = je - jb;
        const int min_l = li < lj ? li : lj;
        if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
            *ov = 1;
            return iw < jw * o.drop_ratio && jw - iw >= o.min_seed_len << 1;
        }
    }
    return 0;
}

template <int CAP> __global__ void __launch_bounds__(64) k_chain_cl(DIndex ix, DOpts o, const i64* __restrict__ seq_off, const i64* __restrict__ seed_off, i64 pool_cap,
                                                                     const DSeed* __restrict__ seeds, const int32_t* __restrict__ s_rid, const int32_t* __restrict__ l_rep,
                                                                     DChain* __restrict__ chains, DSeed* __restrict__ cseeds, int32_t* __restrict__ n_chains, int32_t* __restrict__ status,
                                                                     const int32_t* __restrict__ list, const int32_t* __restrict__ list_count, int s_lo, int last,
                                                                     int32_t* __restrict__ fb_list, int32_t* __restrict__ fb_count) {
    static_assert(CAP >= 64 && CAP <= 2048 && (CAP & (CAP - 1)) == 0, "seed ids in 11 bits of the sort keys");
    // LDS (60 bytes per seed slot): what lives to the end, then two regions that change hands between the phases
    __shared__ DSeed sd[CAP];
    __shared__ int32_t rids[CAP];
    __shared__ ChSlot sl[CAP];           // chain slots: a cluster's chains sit at its first sorted position onwards (a cluster of k seeds makes at most k chains)
    __shared__ int16_t chain_of[CAP];    // seed -> chain slot, -1: in none (bridging seed, contained seed)
    __shared__ uint16_t rank_[CAP];      // seed -> its place among its chain's seeds
    __shared__ int32_t sstart[CAP];      // chain slot -> first place of its seeds in the output, -1: not kept
    __shared__ u64 skey[CAP];            // the seeds' (rbeg << 11 | seed) sorted; once the clusters are chained: od[] (chains in the tree's order) | st[] (weight << 11 | chain);
    int32_t* const od = (int32_t*)skey;  // once st[] is made, od's half holds klist[] (the kept chains, in order) and keptv[]
    int32_t* const st = od + CAP;
    uint16_t* const klist = (uint16_t*)od;
    uint8_t* const keptv = (uint8_t*)(klist + CAP);
    __shared__ __attribute__((aligned(16))) uint16_t wreg[4 * CAP + 4];   // while the clusters are chained: cstart | cnc | cid | corder; in mem_chain_flt: Dd[] (a chain as one word) | reachv[]
    uint16_t* const cstart = wreg;           // [CAP + 1]
    uint16_t* const cnc = wreg + CAP + 2;
    uint16_t* const cid = cnc + CAP;
    uint16_t* const corder = cid + CAP;
    uint32_t* const Dd = (uint32_t*)wreg;
    uint16_t* const reachv = (uint16_t*)((uint32_t*)wreg + CAP);   // chain -> the kept chain that shadows it (0xffff: none)
    static_assert(sizeof(uint16_t) * (4 * CAP + 4) >= 6 * CAP, "Dd | reachv over the cluster tables");
    const int lane = LANE();
    const int n_items = *list_count;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int r = list[item];
        const i64 base = seed_off[r];
        const int S = (int)(seed_off[r + 1] - base);
        if (S <= s_lo || (S > CAP && !last)) continue;   // another instance's read
        if (seed_off[r + 1] > pool_cap) {   // pool overflow: flag, produce nothing (host retries with a smaller batch)
            if (lane == 0) { n_chains[r] = 0; atomicOr(&status[r], LH_ST_POOL_OVERFLOW); }
            continue;
        }
        if (S > CAP) {
            if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = r;
            continue;
        }
        const int len = (int)(seq_off[r + 1] - seq_off[r]);
        WAVE_SYNC();   // the previous read's tables have been read
#ifdef LH_RFA_PROF
        unsigned long long prof_t_ = (unsigned long long)clock64();
        if (lane == 0) atomicAdd(&lh_chain_prof[15], 1ull);
#endif
        // ---- the read's seeds and their sort keys ----
        int NP = 64;
        while (NP < S) NP <<= 1;
        int V = 0;
        for (int s0 = 0; s0 < NP; s0 += 64) {
            const int s = s0 + lane;
            u64 key = ~0ull;
            if (s < S) {
                const DSeed p = seeds[base + s];
                const int rid = s_rid[base + s];
                sd[s] = p; rids[s] = rid; chain_of[s] = -1;
                if (rid >= 0) key = (u64)p.rbeg << 11 | (u64)s;   // (rid < 0: bridging contigs / the forward-reverse boundary: mem_chain skips it)
            }
            skey[s] = key;
            V += (int)__popcll(__ballot(key != ~0ull));
        }
        CH_PROF(0)
        // bitonic sort of skey[0, NP) (the padding sorts last)
        for (int k = 2; k <= NP; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                WAVE_SYNC();
                for (int t = lane; t < NP / 2; t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const u64 a = skey[i], b = skey[l];
                    if ((a > b) == ((i & k) == 0)) { skey[i] = b; skey[l] = a; }
                }
            }
        WAVE_SYNC();
        CH_PROF(1)
        // ---- clusters: a gap of more than len + w between neighbours ----
        const i64 GAP = (i64)len + o.w;
        int ncl = 0;
        for (int p0 = 0; p0 < V; p0 += 64) {
            const int p = p0 + lane;
            int stt = 0;
            if (p < V) stt = p == 0 || (i64)(skey[p] >> 11) - (i64)(skey[p - 1] >> 11) > GAP;
            const u64 m = __ballot(stt);
            if (stt) cstart[ncl + lanes_below(m, lane)] = (uint16_t)p;
            ncl += (int)__popcll(m);
        }
        if (lane == 0) cstart[ncl] = (uint16_t)V;
        WAVE_SYNC();
        {
            int big = 0;
            for (int c = lane; c < ncl; c += 64) big |= cstart[c + 1] - cstart[c] > LH_CHAIN_CL_MAX;
            if (__any(big)) {   // a cluster one lane should not walk alone: the wave-per-seed kernel
                if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = r;
                continue;
            }
        }
        CH_PROF(2)
        // ---- one lane per cluster: mem_chain's walk over the cluster's seeds in the read's seed order ----
        for (int c = lane; c < ncl; c += 64) {
            const int p0 = cstart[c], k = cstart[c + 1] - p0;
            for (int j = 0; j < k; ++j) {   // the cluster's seed ids, ascending
                const int id = (int)(skey[p0 + j] & 2047);
                int q = j;
                while (q > 0 && cid[p0 + q - 1] > id) { cid[p0 + q] = cid[p0 + q - 1]; --q; }
                cid[p0 + q] = (uint16_t)id;
            }
            int nc = 0;
            for (int j = 0; j < k; ++j) {
                const int s = cid[p0 + j];
                const DSeed p = sd[s];
                const int rid = rids[s];
                int bt = -1;
                i64 bpos = -1;
                for (int t = 0; t < nc; ++t) {   // the chain with the greatest position <= p.rbeg; among equal positions the one made last
                    const i64 cp = sd[sl[p0 + t].head].rbeg;
                    if (cp <= p.rbeg && cp >= bpos) { bpos = cp; bt = t; }
                }
                int res = 0;   // 0: new chain, 1: contained, 2: appended
                if (bt >= 0) {   // test_and_merge
                    const ChSlot ch = sl[p0 + bt];
                    const DSeed hd = sd[ch.head], tl = sd[ch.tail];
                    const i64 qend = tl.qbeg + tl.len, rend = tl.rbeg + tl.len;
                    if (rid != rids[ch.head]) res = 0;
                    else if (p.qbeg >= hd.qbeg && p.qbeg + p.len <= qend && p.rbeg >= hd.rbeg && p.rbeg + p.len <= rend) res = 1;
                    else if ((tl.rbeg < ix.l_pac || hd.rbeg < ix.l_pac) && p.rbeg >= ix.l_pac) res = 0;
                    else {
                        const i64 x = p.qbeg - tl.qbeg, y = p.rbeg - tl.rbeg;
                        if (y >= 0 && x - y <= o.w && y - x <= o.w && x - tl.len < o.max_chain_gap && y - tl.len < o.max_chain_gap) res = 2;
                    }
                    if (res == 2) {
                        ChSlot u = ch;
                        rank_[s] = u.n; chain_of[s] = (int16_t)(p0 + bt);
                        u.tail = (uint16_t)s; u.n = (uint16_t)(u.n + 1);
                        // mem_chain_weight, one seed further: covered query bases, covered reference bases
                        const int qe = p.qbeg + p.len;
                        if (p.qbeg >= u.endq) u.wq = (uint8_t)(u.wq + p.len);
                        else if (qe > u.endq) u.wq = (uint8_t)(u.wq + qe - u.endq);
                        u.endq = (uint8_t)(u.endq > qe ? u.endq : qe);
                        const i64 rel = p.rbeg - hd.rbeg, re_ = rel + p.len;   // relative to the chain's position
                        if (rel >= u.endr_rel) u.wr += p.len;
                        else if (re_ > u.endr_rel) u.wr += (int)(re_ - u.endr_rel);
                        u.endr_rel = (int32_t)(u.endr_rel > re_ ? u.endr_rel : re_);
                        sl[p0 + bt] = u;
                    }
                }
                if (res == 0) {
                    ChSlot u;
                    u.head = (uint16_t)s; u.tail = (uint16_t)s; u.n = 1; u.wq = (uint8_t)p.len; u.endq = (uint8_t)(p.qbeg + p.len); u.wr = p.len; u.endr_rel = p.len;
                    sl[p0 + nc] = u;
                    rank_[s] = 0; chain_of[s] = (int16_t)(p0 + nc);
                    ++nc;
                }
            }
            cnc[c] = (uint16_t)nc;
            for (int t = 0; t < nc; ++t) {   // the cluster's chains by (position, order of creation)
                const i64 cp = sd[sl[p0 + t].head].rbeg;
                int q = t;
                while (q > 0 && sd[sl[p0 + corder[p0 + q - 1]].head].rbeg > cp) { corder[p0 + q] = corder[p0 + q - 1]; --q; }
                corder[p0 + q] = (uint16_t)t;
            }
        }
        WAVE_SYNC();
        CH_PROF(3)
        // ---- the tree's traversal order, the chains' weights: mem_chain_flt's input ----
        int nall = 0;
        for (int c0 = 0; c0 < ncl; c0 += 64) {
            const int c = c0 + lane;
            const int mync = c < ncl ? cnc[c] : 0;
            const int inc = wave_scan_add_i32(mync);
            if (mync) {
                const int p0 = cstart[c], at = nall + inc - mync;
                for (int t = 0; t < mync; ++t) od[at + t] = p0 + corder[p0 + t];
            }
            nall += wave_readlane(inc, 63);
        }
        WAVE_SYNC();
        int n = 0;
        for (int i0 = 0; i0 < nall; i0 += 64) {
            const int i = i0 + lane;
            int ok = 0, key = 0;
            if (i < nall) {
                const int slot = od[i];
                const ChSlot u = sl[slot];
                const int w = u.wq < u.wr ? u.wq : u.wr;
                sstart[slot] = -1;
                ok = w >= o.min_chain_weight;
                key = w << 11 | slot;
            }
            const u64 m = __ballot(ok);
            if (ok) st[n + lanes_below(m, lane)] = key;
            n += (int)__popcll(m);
        }
        WAVE_SYNC();
        if (n == 0) {
            if (lane == 0) n_chains[r] = 0;
            continue;
        }
        CH_PROF(4)
        // ---- mem_chain_flt: ks_introsort by weight, descending ----
        {
            auto lt = [](int x, int y) { return (x >> 11) > (y >> 11); };
            LhIsortStk* const stk = lh_isort_stack_ptr();
#ifdef LH_EMU
            if (lane == 0) dev_introsort_ix(n, [&](int i) { return st[i]; }, [&](int i, int x) { st[i] = x; }, lt, stk, o.wd);
#else
            // keys in one, two or four registers of the wave, read and written by lane index: no memory in the sort's dependent chain
            if (n <= 64) sort_in_regs<1>(n, st, lane, lt, stk, o.wd);
            else if (n <= 128) sort_in_regs<2>(n, st, lane, lt, stk, o.wd);
            else if (n <= 256) sort_in_regs<4>(n, st, lane, lt, stk, o.wd);
            else if (lane == 0) dev_introsort_ix(n, [&](int i) { return st[i]; }, [&](int i, int x) { st[i] = x; }, lt, stk, o.wd);
#endif
        }
        WAVE_SYNC();
        CH_PROF(5)
        // ---- mem_chain_flt: the greedy scan, in ROUNDS.  The reference takes the chains in order of weight and tests each against the kept ones so far, up to
        // the first that shadows it (weight below drop_ratio x that chain's).  A chain can therefore only be shadowed by one more than 1 / drop_ratio times
        // as heavy: the sorted order falls into rounds [s, e) with w[e - 1] >= drop_ratio x w[s] — a handful: weights lie between min_seed_len and the
        // read's length — inside which no chain shadows another.  A round's chains are decided side by side, a lane each: against the kept chains of the
        // earlier rounds (in order, up to the first that shadows), then — those that stay — against the round's own kept chains before them.  chain_first of
        // a kept chain is the FIRST chain that reaches it in its scan with a significant overlap: found last, a lane per kept chain.  Chain i as one word:
        // beg | end << 8 | weight << 16 | is_alt << 25 (a chain's weight is at most the read's length) ----
        for (int i = lane; i < n; i += 64) {
            const int key = st[i];
            const ChSlot u = sl[key & 2047];
            const DSeed hd = sd[u.head], tl = sd[u.tail];
            const int alt = ix.contig_alt && ix.contig_alt[rids[u.head]];
            Dd[i] = (uint32_t)hd.qbeg | (uint32_t)(tl.qbeg + tl.len) << 8 | (uint32_t)(key >> 11) << 16 | (uint32_t)alt << 25;
            keptv[i] = 0;
        }
        WAVE_SYNC();
        int nk = 0;
        for (int s0 = 0; s0 < n;) {
            // the round: up to the first chain lighter than drop_ratio x the round's heaviest (the same float comparison as the rule's)
            const int ws = (int)(Dd[s0] >> 16 & 511);
            int e0 = n;
            for (int i0 = s0 + 1; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const u64 m = __ballot(i < n && (int)(Dd[i] >> 16 & 511) < ws * o.drop_ratio);
                if (m) { e0 = i0 + __ffsll((unsigned long long)m) - 1; break; }
            }
            const int nk_prev = nk;
            for (int i0 = s0; i0 < e0; i0 += 64) {   // against the kept chains of the earlier rounds, in order, up to the first that shadows
                const int i = i0 + lane;
                int shadowed = 0, large = 0;
                if (i < e0) {
                    const uint32_t di = Dd[i];
                    int reach = 0xffff;   // how far the chain's scan of the kept list gets: to the end, or to the chain that shadows it
                    for (int k = 0; k < nk_prev; ++k) {
                        const int j = klist[k];
                        int ov;
                        const int brk = flt_pair(o, di, Dd[j], &ov);
                        large |= ov;
                        if (brk) { shadowed = 1; reach = j; break; }
                    }
                    reachv[i] = (uint16_t)reach;
                }
                const u64 mk = __ballot(i < e0 && !shadowed);
                if (i < e0 && !shadowed) { klist[nk + lanes_below(mk, lane)] = (uint16_t)i; keptv[i] = large ? 2 : 3; }
                nk += (int)__popcll(mk);
            }
            WAVE_SYNC();
            for (int k0 = nk_prev; k0 < nk; k0 += 64) {   // the round's kept chains: a significant overlap with one of the round's kept chains before them?
                const int k = k0 + lane;
                if (k < nk) {
                    const int i = klist[k];
                    if (keptv[i] == 3) {
                        const uint32_t di = Dd[i];
                        for (int k2 = k - 1; k2 >= nk_prev; --k2) {
                            int ov;
                            flt_pair(o, di, Dd[klist[k2]], &ov);
                            if (ov) { keptv[i] = 2; break; }
                        }
                    }
                }
            }
            WAVE_SYNC();
            s0 = e0;
        }
        // chain_first of kept chain j = the first chain after it whose scan reaches j with a significant overlap; that chain's kept becomes 1
        for (int k0 = 0; k0 < nk; k0 += 64) {
            const int k = k0 + lane;
            int f = -1;
            if (k < nk) {
                const int j = klist[k];
                const uint32_t dj = Dd[j];
                for (int i = j + 1; i < n; ++i) {
                    if ((int)reachv[i] < j) continue;   // shadowed by a kept chain before j: its scan ended there
                    int ov;
                    flt_pair(o, Dd[i], dj, &ov);
                    if (ov) { f = i; break; }
                }
            }
            if (f >= 0) keptv[f] = 1;   // (several lanes may name the same chain: the same value)
        }
        WAVE_SYNC();
        if (n >= o.max_chain_extend) {   // don't extend more than max_chain_extend .kept=1/2 chains (mem_opt_init: 1 << 30)
            if (lane == 0) {
                int i, k;
                for (i = k = 0; i < n; ++i) {
                    const int kp = keptv[i];
                    if (kp == 0 || kp == 3) continue;
                    if (++k >= o.max_chain_extend) break;
                }
                for (; i < n; ++i)
                    if (keptv[i] < 3) keptv[i] = 0;
            }
            WAVE_SYNC();
        }
        CH_PROF(6)
        // ---- emit the kept chains in sorted order, then every seed to its chain's place ----
        int m = 0;
        {
            int sacc = 0;
            const float frac_rep = (float)l_rep[r] / len;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                int kept = 0, cn = 0, slot = 0, key = 0;
                ChSlot u;
                if (i < n) { key = st[i]; slot = key & 2047; u = sl[slot]; kept = keptv[i] != 0; cn = kept ? u.n : 0; }
                const u64 kmk = __ballot(kept);
                const int incl = wave_scan_add_i32(cn);
                if (kept) {
                    const int rid = rids[u.head];
                    DChain oc;
                    oc.pos = sd[u.head].rbeg; oc.rid = rid; oc.n = u.n; oc.seed_start = sacc + incl - cn; oc.w = key >> 11; oc.kept = keptv[i];
                    oc.is_alt = (ix.contig_alt && ix.contig_alt[rid]) ? 1 : 0;
                    oc.frac_rep = frac_rep; oc.pad = 0;
                    chains[base + m + lanes_below(kmk, lane)] = oc;
                    sstart[slot] = oc.seed_start;
                }
                m += (int)__popcll(kmk);
                sacc += wave_readlane(incl, 63);
            }
            if (lane == 0) n_chains[r] = m;
        }
        WAVE_SYNC();
        for (int s = lane; s < S; s += 64) {
            const int c = chain_of[s];
            if (c >= 0 && sstart[c] >= 0) cseeds[base + sstart[c] + rank_[s]] = sd[s];
        }
        CH_PROF(7)
    }
}

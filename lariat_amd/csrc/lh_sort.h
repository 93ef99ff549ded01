// lh_sort.h — sequential (single-lane) device sorts whose tie order is part of the result contract.
//   dev_introsort : the unstable introsort BWA applies (via klib's KSORT_INIT) to chains (by weight), alignment
//                   regions (by end, then by score) and the per-chain seed order — reached through
//                   mem_align1_core / mem_matesw (go/src/gobwa/gobwa.go:244,253,291,315).
//   dev_gosort    : Go 1.9's sort.Sort as used by lariat at go/src/inference/lariat.go:1546 (ByPosition) and
//                   go/src/inference/split.go:108 (SortSplitScoring).
// Equal keys keep whatever order the comparison/swap sequence of these exact algorithms produces, so both are
// implemented as the same algorithms (on small index arrays, by one lane).
#pragma once
#include "lh_dev.h"

template <class T, class Lt> __device__ inline void dev_insertsort(T* s, T* t, Lt lt) {
    for (T* i = s + 1; LH_UNI(i < t); ++i)
        for (T* j = i; LH_UNI(j > s && lt(*j, *(j - 1))); --j) { T tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}

template <class T, class Lt> __device__ inline void dev_combsort(int n, T* a, Lt lt) {
    const double shrink_factor = 1.2473309501039786540366528676643;
    int do_swap;
    int gap = n;
    do {
        if (gap > 2) {
            gap = (int)(gap / shrink_factor);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (T* i = a; LH_UNI(i < a + n - gap); ++i) {
            T* j = i + gap;
            if (LH_UNI(lt(*j, *i))) { T tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (LH_UNI(do_swap || gap > 2));
    if (gap != 1) dev_insertsort(a, a + n, lt);
}

// (its stack lives in LDS: ONE lane of a SINGLE-WAVE block runs it — k_chain, k_dedup, k_resc_apply, k_diag_resc_dedup: __launch_bounds__(64), lane 0;
// checked at entry — a private array would be 1.7 KB of scratch in every kernel that sorts; dev_introsort_ix, k_chain_cl.h, takes its stack as an argument)
#define LH_ISORT_STK 64
struct LhIsortStk { int32_t left, right, depth; };
__device__ __forceinline__ LhIsortStk* lh_isort_stack_ptr() {   // ONE array per kernel, whatever the number of element types and orders it sorts by (a template's own __shared__ array exists once per instantiation)
    __shared__ LhIsortStk lh_isort_stack[LH_ISORT_STK];
    return lh_isort_stack;
}
template <class T, class Lt> __device__ inline void dev_introsort(int n, T* a, Lt lt, int32_t* wdp) {
    typedef LhIsortStk Stk;
    Stk* const stack = lh_isort_stack_ptr();
#ifndef LH_EMU
    if (blockDim.x != 64) { wdp[1] = 1; return; }   // the stack is ONE shared array per kernel: a single-wave block, one lane (a wider block's waves would share it: refused, the watchdog slot says so)
#endif
    int d;
    T rp, swap_tmp;
    T *s, *t, *i, *j, *k;
    if (n < 1) return;
    else if (n == 2) {
        if (lt(a[1], a[0])) { swap_tmp = a[0]; a[0] = a[1]; a[1] = swap_tmp; }
        return;
    }
    for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
    Stk* top = stack;
    s = a; t = a + (n - 1); d <<= 1;
    int wd = 100000 + 64 * n;
    while (1) {
        LH_WATCH_S(wdp, wd, 1, return)
        if (LH_UNI(s < t)) {
            if (LH_UNI(--d == 0)) {
                dev_combsort((int)(t - s + 1), s, lt);
                t = s;
                continue;
            }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (LH_UNI(lt(*k, *i))) {
                if (LH_UNI(lt(*k, *j))) k = j;
            } else k = LH_UNI(lt(*j, *i)) ? i : j;
            rp = *k;
            if (k != t) { swap_tmp = *k; *k = *t; *t = swap_tmp; }
            for (;;) {
                do { ++i; LH_WATCH_S(wdp, wd, 2, return) } while (LH_UNI(lt(*i, rp)));
                do { --j; LH_WATCH_S(wdp, wd, 3, return) } while (LH_UNI(i <= j && lt(rp, *j)));
                if (LH_UNI(j <= i)) break;
                swap_tmp = *i; *i = *j; *j = swap_tmp;
            }
            swap_tmp = *i; *i = *t; *t = swap_tmp;
            if (LH_UNI(i - s > t - i)) {
                if (i - s > 16) { if (top - stack >= LH_ISORT_STK) { wdp[1] = 1; return; } top->left = (int32_t)(s - a); top->right = (int32_t)(i - 1 - a); top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { if (top - stack >= LH_ISORT_STK) { wdp[1] = 1; return; } top->left = (int32_t)(i + 1 - a); top->right = (int32_t)(t - a); top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (LH_UNI(top == stack)) {
                dev_insertsort(a, a + n, lt);
                return;
            } else { --top; s = a + top->left; t = a + top->right; d = top->depth; }
        }
    }
}

// ---- (r05) klib's introsort by a whole wave, move for move.  a[0, n): 64-bit words in LDS compared above their low `ib` bits (the sort keys of k_dedup.h / k_rescue2.h: the
// element's index sits below the key, so equal keys are told apart afterwards — and where the introsort leaves them is part of the result).  What is serial in
// ks_introsort is the sequence of partitions, not a partition: its two scans only ever look at positions neither has passed (a swap writes behind them), so on the
// array as the partition finds it the i-scan stops, in turn, at the positions of (s, t] whose element is not below the pivot — ascending: I_1 < I_2 < .. — and the j-scan at the
// positions of [s, t) whose element is not above it — descending: J_1 > J_2 > .. —; the r-th swap is a[I_r] <-> a[J_r], they go on while J_r > I_r (a condition that, once
// false, stays false), and the scan ends with i on the next I (or on J_R, where the last swap put an element that stops it, if that comes first).  Both lists come from ballots, the swaps are independent, the pivot rule and the stack are ks_introsort's.
// The closing insertion sort is stable: its result is the array ranked by (key, place before it).  The depth limit's comb sort (never seen on region lists) stays one lane's.
// pa, pb: LDS scratch, n 16-bit entries each.  One single-wave block (the stack is dev_introsort's).
template <int PER> __device__ inline void wave_introsort_i64(int n, i64* a, int ib, int lane, uint16_t* pa, uint16_t* pb, int32_t* wdp) {
    typedef LhIsortStk Stk;
    Stk* const stack = lh_isort_stack_ptr();
#ifndef LH_EMU
    if (blockDim.x != 64) { wdp[1] = 1; return; }
#endif
    if (n < 2) return;
    WAVE_SYNC();
    if (n == 2) {
        if (lane == 0 && (a[1] >> ib) < (a[0] >> ib)) { const i64 x = a[0]; a[0] = a[1]; a[1] = x; }
        WAVE_SYNC();
        return;
    }
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
    int top = 0, s = 0, t = n - 1;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) {
                if (lane == 0) dev_combsort(t - s + 1, a + s, [&](i64 x, i64 y) { return (x >> ib) < (y >> ib); });
                WAVE_SYNC();
                t = s;
                continue;
            }
            int k = s + ((t - s) >> 1) + 1;
            {
                const i64 vs = a[s] >> ib, vk = a[k] >> ib, vt = a[t] >> ib;
                if (vk < vs) { if (vk < vt) k = t; }
                else k = vt < vs ? s : t;
            }
            const i64 rpw = a[k], rp = rpw >> ib;
            WAVE_SYNC();
            if (k != t && lane == 0) { a[k] = a[t]; a[t] = rpw; }
            WAVE_SYNC();
            // the i-scan's stops, ascending (a[t] is the pivot: the last of them), and the j-scan's, descending
            int ni = 0, nj = 0;
            for (int x0 = s + 1; x0 <= t; x0 += 64) {
                const int x = x0 + lane;
                const int isI = x <= t && !((a[x] >> ib) < rp);
                const u64 m = __ballot(isI);
                if (isI) pa[ni + lanes_below(m, lane)] = (uint16_t)x;
                ni += (int)__popcll(m);
            }
            for (int x0 = t - 1; x0 >= s; x0 -= 64) {
                const int x = x0 - lane;
                const int isJ = x >= s && !(rp < (a[x] >> ib));
                const u64 m = __ballot(isJ);
                if (isJ) pb[nj + lanes_below(m, lane)] = (uint16_t)x;
                nj += (int)__popcll(m);
            }
            WAVE_SYNC();
            const int np = ni < nj ? ni : nj;
            int R = 0;
            for (int r0 = 0; r0 < np; r0 += 64) {
                const int r = r0 + lane;
                const int sw = r < np && pb[r] > pa[r];
                const u64 m = __ballot(sw);
                if (sw) { const int xi = pa[r], xj = pb[r]; const i64 u = a[xi]; a[xi] = a[xj]; a[xj] = u; }
                R += (int)__popcll(m);
                if (m != ~0ull) break;
            }
            WAVE_SYNC();
            // where the i-scan ends: at the next of its stops — or before that on the place of the last swap's j, which now holds an element that is not below the pivot
            // (R < ni: the swaps stop at the latest when the i-scan is at t, where no j is above it)
            int i = pa[R];
            if (R > 0 && pb[R - 1] < i) i = pb[R - 1];
            WAVE_SYNC();
            if (lane == 0) { const i64 u = a[i]; a[i] = a[t]; a[t] = u; }
            WAVE_SYNC();
            if (i - s > t - i) {
                if (i - s > 16) { if (top >= LH_ISORT_STK) { wdp[1] = 1; return; } if (lane == 0) { stack[top].left = s; stack[top].right = i - 1; stack[top].depth = d; } ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { if (top >= LH_ISORT_STK) { wdp[1] = 1; return; } if (lane == 0) { stack[top].left = i + 1; stack[top].right = t; stack[top].depth = d; } ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) break;
            WAVE_SYNC();
            --top; s = stack[top].left; t = stack[top].right; d = stack[top].depth;
        }
    }
    // the insertion sort over the whole array
    WAVE_SYNC();
    i64 v[PER]; int rk[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int x = u * 64 + lane;
        rk[u] = -1;
        if (x < n) {
            v[u] = a[x];
            const i64 kx = v[u] >> ib;
            int c = 0;   // (the whole array: klib's median rule can leave a[s] above the pivot on the pivot's left — the partitions do not bound how far an element still has to go)
            for (int y = 0; y < n; ++y) { const i64 ky = a[y] >> ib; c += ky < kx || (ky == kx && y < x); }
            rk[u] = c;
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < PER; ++u) if (rk[u] >= 0) a[rk[u]] = v[u];
    WAVE_SYNC();
}

// ---- Go 1.9 sort.Sort over an index space [0,n): less(i,j), swp(i,j) ----
// NOTE: called lane-parallel (one independent sort per lane) by k_rfa, so its branches must stay per-lane: no LH_UNI here.
template <class L, class S> __device__ inline void gs_insertion(L& less, S& swp, int a, int b) {
    for (int i = a + 1; i < b; i++)
        for (int j = i; (j > a && less(j, j - 1)); j--) swp(j, j - 1);
}
template <class L, class S> __device__ inline void gs_siftdown(L& less, S& swp, int lo, int hi, int first) {
    int root = lo;
    for (;;) {
        int child = 2 * root + 1;
        if ((child >= hi)) return;
        if ((child + 1 < hi && less(first + child, first + child + 1))) child++;
        if ((!less(first + root, first + child))) return;
        swp(first + root, first + child);
        root = child;
    }
}
template <class L, class S> __device__ inline void gs_heapsort(L& less, S& swp, int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) gs_siftdown(less, swp, i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swp(first, first + i); gs_siftdown(less, swp, lo, i, first); }
}
template <class L, class S> __device__ inline void gs_median3(L& less, S& swp, int m1, int m0, int m2) {
    if ((less(m1, m0))) swp(m1, m0);
    if ((less(m2, m1))) { swp(m2, m1); if ((less(m1, m0))) swp(m1, m0); }
}
template <class L, class S> __device__ inline void gs_pivot(L& less, S& swp, int lo, int hi, int* midlo, int* midhi) {
    int m = (int)((unsigned)(lo + hi) >> 1);
    if (hi - lo > 40) {
        int s = (hi - lo) / 8;
        gs_median3(less, swp, lo, lo + s, lo + 2 * s);
        gs_median3(less, swp, m, m - s, m + s);
        gs_median3(less, swp, hi - 1, hi - 1 - s, hi - 1 - 2 * s);
    }
    gs_median3(less, swp, lo, m, hi - 1);
    int pivot = lo;
    int a = lo + 1, c = hi - 1;
    for (; (a < c && less(a, pivot)); a++) {}
    int b = a;
    for (;;) {
        for (; (b < c && !less(pivot, b)); b++) {}
        for (; (b < c && less(pivot, c - 1)); c--) {}
        if ((b >= c)) break;
        swp(b, c - 1);
        b++; c--;
    }
    bool protect = hi - c < 5;
    if (!protect && hi - c < (hi - lo) / 4) {
        int dups = 0;
        if ((!less(pivot, hi - 1))) { swp(c, hi - 1); c++; dups++; }
        if ((!less(b - 1, pivot))) { b--; dups++; }
        if ((!less(m, pivot))) { swp(m, b - 1); b--; dups++; }
        protect = dups > 1;
    }
    if (protect) {
        for (;;) {
            for (; (a < b && !less(b - 1, pivot)); b--) {}
            for (; (a < b && less(a, pivot)); a++) {}
            if ((a >= b)) break;
            swp(a, b - 1);
            a++; b--;
        }
    }
    swp(pivot, b - 1);
    *midlo = b - 1; *midhi = c;
}
// stk: LH_GOSORT_STK frames of three words for THIS lane, word w of frame f at stk[(3 * f + w) * stride] (the caller's memory: a private array
// would be scratch; the smaller side is finished first, so at most log2(n) + 1 frames are ever waiting)
#define LH_GOSORT_STK 40
template <class L, class S> __device__ inline void dev_gosort(int n, L less, S swp, int32_t* stk, int stride) {
    // quickSort(data, 0, n, maxDepth(n)) with the recursion on the smaller side turned into an explicit stack
    int sp = 0;
    int depth = 0;
    for (int i = n; i > 0; i >>= 1) depth++;
#define GS_PUSH(a_, b_, d_) { if (sp < LH_GOSORT_STK) { stk[(3 * sp) * stride] = (a_); stk[(3 * sp + 1) * stride] = (b_); stk[(3 * sp + 2) * stride] = (d_); } ++sp; }
    GS_PUSH(0, n, depth * 2)
    while ((sp > 0)) {
        --sp;
        if (sp >= LH_GOSORT_STK) return;   // (cannot happen for n < 2^39; the result would be unsorted rather than memory overwritten)
        int a = stk[(3 * sp) * stride], b = stk[(3 * sp + 1) * stride], maxDepth = stk[(3 * sp + 2) * stride];
        bool done = false;
        while ((b - a > 12)) {
            if ((maxDepth == 0)) { gs_heapsort(less, swp, a, b); done = true; break; }
            maxDepth--;
            int mlo, mhi;
            gs_pivot(less, swp, a, b, &mlo, &mhi);
            // Go recurses into the smaller side FIRST and then loops on the larger one.  The two sides are disjoint
            // index ranges, so finishing the larger side later (explicit stack) issues the same Less/Swap calls per range.
            if (mlo - a < b - mhi) { GS_PUSH(mhi, b, maxDepth) b = mlo; }
            else { GS_PUSH(a, mlo, maxDepth) a = mhi; }
        }
        if (done) continue;
        if (b - a > 1) {
            for (int i = a + 6; i < b; i++)
                if ((less(i, i - 6))) swp(i, i - 6);
            gs_insertion(less, swp, a, b);
        }
    }
}

#undef GS_PUSH

// ---- the same sort by a whole wave ----
// quickSort's two sides after a doPivot are disjoint index ranges, and so are the index spaces of separate sorts: whichever lane works
// on a range, and whenever, the Less / Swap calls inside that range are those of the serial algorithm, so the result (tie order
// included) is the same.  Ranges wait in a queue (qa / qb / qd: first, end, remaining depth; at most one entry per doPivot call, i.e.
// fewer than the number of elements); each turn the lanes take one range each and make ONE step of quickSort on it: a doPivot (the
// two sides are queued, or finished at once if they are short), or the heap sort of a range whose depth is used up.
// Called by all lanes of the wave; `nsort` sorts over [first[k], first[k + 1]) with Go's depth limit for their sizes.  With sa / sb (room for half the longest
// range each) the ranges longer than LH_GOSORT_WAVE_MIN are partitioned by the whole wave (wave_go_pivot) before the lanes take one range each.
// (r06) ONE doPivot by the whole wave, for a long range: the partition loops of gs_pivot move two pointers towards each other, the left one stopping at an element of
// the class that belongs right, the right one at an element that belongs left, and swap what they stop at.  Whatever the data, the k-th stop of the left pointer
// meets the k-th stop of the right one, until they cross at the boundary bd = first + (elements that belong left): the swaps are those of the misplaced elements
// left of bd, in ascending order, with the misplaced ones right of it, in descending order — disjoint pairs, all known after one counting pass, done at once.  The
// pointers both end at bd.  Less and Swap calls per range are those of the serial loop (in another order, on disjoint pairs), so the result is the same array.
// right(p): the element at p belongs right of the boundary.  sa / sb: room for (c - a) / 2 places each.  Returns bd; called by all lanes with a, c uniform.
template <class P, class S> __device__ inline int wave_go_partition(int a, int c, P right, S& swp, int32_t* sa, int32_t* sb) {
    const int lane = LANE();
    int n_left = 0;
    for (int base = a; base < c; base += 64) {
        const int p = base + lane;
        n_left += __popcll(__ballot(p < c && !right(p)));
    }
    const int bd = a + n_left;
    int na = 0, nb = 0;   // misplaced elements left of bd so far (ascending places), right of it (ascending as well: read backwards below)
    for (int base = a; base < c; base += 64) {
        const int p = base + lane;
        const int r = p < c && right(p);
        const int fa = r && p < bd, fb = p < c && !r && p >= bd;
        const u64 ba = __ballot(fa), bb = __ballot(fb);
        if (fa) sa[na + lanes_below(ba, lane)] = p;
        if (fb) sb[nb + lanes_below(bb, lane)] = p;
        na += __popcll(ba); nb += __popcll(bb);
    }
    WAVE_SYNC();
    for (int k = lane; k < na; k += 64) swp(sa[k], sb[nb - 1 - k]);   // (na == nb)
    WAVE_SYNC();
    return bd;
}
template <class L, class S> __device__ inline void wave_go_pivot(L& less, S& swp, int lo, int hi, int* midlo, int* midhi, int32_t* sa, int32_t* sb) {
    const int lane = LANE();
    const int m = (int)((unsigned)(lo + hi) >> 1);
    if (lane == 0) {
        if (hi - lo > 40) {
            const int s = (hi - lo) / 8;
            gs_median3(less, swp, lo, lo + s, lo + 2 * s);
            gs_median3(less, swp, m, m - s, m + s);
            gs_median3(less, swp, hi - 1, hi - 1 - s, hi - 1 - 2 * s);
        }
        gs_median3(less, swp, lo, m, hi - 1);
    }
    WAVE_SYNC();
    const int pivot = lo;
    int a = hi - 1, c = hi - 1;
    for (int base = lo + 1; base < c; base += 64) {   // for (; a < c && less(a, pivot); a++)
        const int p = base + lane;
        const u64 stop = __ballot(p < c && !less(p, pivot));
        if (stop) { a = base + __ffsll((long long)stop) - 1; break; }
    }
    int b = wave_go_partition(a, c, [&](int p) { return less(pivot, p); }, swp, sa, sb);
    c = b;
    int protect = hi - c < 5;
    if (!protect && hi - c < (hi - lo) / 4) {
        int dups = 0;
        if (lane == 0) {
            if (!less(pivot, hi - 1)) { swp(c, hi - 1); c++; dups++; }
            if (!less(b - 1, pivot)) { b--; dups++; }
            if (!less(m, pivot)) { swp(m, b - 1); b--; dups++; }
        }
        b = wave_readlane(b, 0); c = wave_readlane(c, 0); dups = wave_readlane(dups, 0);
        WAVE_SYNC();
        protect = dups > 1;
    }
    if (protect) b = wave_go_partition(a, b, [&](int p) { return !less(p, pivot); }, swp, sa, sb);   // (smaller than the pivot | equal to it)
    if (lane == 0) swp(pivot, b - 1);
    WAVE_SYNC();
    *midlo = b - 1; *midhi = c;
}
#ifndef LH_GOSORT_WAVE_MIN
#define LH_GOSORT_WAVE_MIN 96   // ranges longer than this are partitioned by the whole wave (one lane's doPivot: a round trip to LDS or memory per element)
#endif
// (r06, late) the first part alone, for a list too long for LDS: quickSort(a0, b0) with Go's depth limit for its size, the ranges longer than `limit` partitioned by
// the whole wave until none is left (a range whose depth is used up stays, whatever its length).  Returns the number of ranges now waiting in qa / qb / qd (first,
// end, remaining depth); the caller sorts each of them with that depth (wave_gosort with depth0), e.g. from LDS.
template <class L, class S> __device__ inline int wave_gosort_split(int a0, int b0, int limit, L less, S swp, int32_t* qa, int32_t* qb, int32_t* qd, int32_t* sa, int32_t* sb) {
    const int lane = LANE();
    int tail = 1;
    if (lane == 0) {
        int depth = 0;
        for (int i = b0 - a0; i > 0; i >>= 1) depth++;
        qa[0] = a0; qb[0] = b0; qd[0] = depth * 2;
    }
    WAVE_SYNC();
    for (int q = 0; q < tail;) {
        const int a = qa[q], b = qb[q], maxDepth = qd[q];
        if (b - a <= limit || maxDepth == 0) { ++q; continue; }
        int mlo, mhi;
        WAVE_SYNC();
        wave_go_pivot(less, swp, a, b, &mlo, &mhi, sa, sb);
        if (lane == 0) {
            qb[q] = mlo; qd[q] = maxDepth - 1;
            qa[tail] = mhi; qb[tail] = b; qd[tail] = maxDepth - 1;
        }
        ++tail;
        WAVE_SYNC();
    }
    return tail;
}
template <class L, class S> __device__ inline void wave_gosort(int nsort, const int32_t* first, L less, S swp, int32_t* qa, int32_t* qb, int32_t* qd, int32_t* sa = nullptr, int32_t* sb = nullptr, int depth0 = -1) {
    const int lane = LANE();
    int head = 0, tail = 0;
    for (int base = 0; base < nsort; base += 64) {
        const int k = base + lane;
        int a = 0, b = 0;
        if (k < nsort) { a = first[k]; b = first[k + 1]; }
        const int want = b - a > 1;
        const int at = tail + wave_scan_add_i32(want) - want;
        if (want) {
            int depth = 0;
            for (int i = b - a; i > 0; i >>= 1) depth++;
            qa[at] = a; qb[at] = b; qd[at] = depth0 >= 0 ? depth0 : depth * 2;   // (depth0: a range of a longer sort, with what that sort has left it)
        }
        tail += __popcll(__ballot(want));
    }
    WAVE_SYNC();
    auto finish_short = [&](int a, int b) {   // the tail of quickSort for a range of at most 12 elements
        if (b - a > 1) {
            for (int i = a + 6; i < b; i++)
                if (less(i, i - 6)) swp(i, i - 6);
            gs_insertion(less, swp, a, b);
        }
    };
    if (sa) {   // (r06) the long ranges first, one at a time by the whole wave; what is left waits in the queue as before (an entry with b - a < 2 is skipped there)
        for (int q = 0; q < tail;) {
            const int a = qa[q], b = qb[q], maxDepth = qd[q];
            if (b - a <= LH_GOSORT_WAVE_MIN || maxDepth == 0) { ++q; continue; }
            int mlo, mhi;
            WAVE_SYNC();
            wave_go_pivot(less, swp, a, b, &mlo, &mhi, sa, sb);
            if (lane == 0) {
                qb[q] = mlo; qd[q] = maxDepth - 1;
                qa[tail] = mhi; qb[tail] = b; qd[tail] = maxDepth - 1;
            }
            ++tail;
            WAVE_SYNC();
        }
    }
    while (head < tail) {
        const int take = tail - head < 64 ? tail - head : 64;
        int c0a = 0, c0b = 0, c1a = 0, c1b = 0, cd = 0, n0 = 0, n1 = 0;
        if (lane < take) {
            const int a = qa[head + lane], b = qb[head + lane];
            int maxDepth = qd[head + lane];
            if (b - a > 12) {
                if (maxDepth == 0) gs_heapsort(less, swp, a, b);
                else {
                    maxDepth--;
                    int mlo, mhi;
                    gs_pivot(less, swp, a, b, &mlo, &mhi);
                    cd = maxDepth;
                    if (mlo - a > 12) { c0a = a; c0b = mlo; n0 = 1; } else finish_short(a, mlo);
                    if (b - mhi > 12) { c1a = mhi; c1b = b; n1 = 1; } else finish_short(mhi, b);
                }
            } else finish_short(a, b);
        }
        head += take;
        const int cnt = n0 + n1;
        const int at = tail + wave_scan_add_i32(cnt) - cnt;
        if (n0) { qa[at] = c0a; qb[at] = c0b; qd[at] = cd; }
        if (n1) { qa[at + n0] = c1a; qb[at + n0] = c1b; qd[at + n0] = cd; }
        tail += wave_sum_i32(cnt);
        WAVE_SYNC();
    }
}

// ---- (r06) a sorting network by the whole wave, for lists whose keys are all different (then there is only one sorted order and no contract about equal keys):
// the bitonic merge sort with every comparator pointing the same way (the first step of a merge mirrors the second half), so that the list needs no padding —
// places from n on stand for +infinity and a comparator that reaches one does nothing.  a[0 .. n) in LDS or in memory, ascending; n log^2 n / 128 steps per lane.
// one pass of comparators: U of a lane's comparators at a time, all their reads issued before the first write (the comparators of a pass touch disjoint places; a read
// per comparator waiting for the previous one's write was a round trip each)
template <int U, class IDX> __device__ __forceinline__ void wave_ce_pass(u64* a, int ncmp, int n, int lane, IDX idx) {
    for (int c0 = lane; c0 < ncmp; c0 += 64 * U) {
        u64 x[U], y[U];
        int lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u;
            lo[u] = -1; hi[u] = 0; x[u] = 0; y[u] = 0;
            if (c < ncmp) {
                int l, h;
                idx(c, l, h);
                if (h < n) { lo[u] = l; hi[u] = h; x[u] = a[l]; y[u] = a[h]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (lo[u] >= 0 && x[u] > y[u]) { a[lo[u]] = y[u]; a[hi[u]] = x[u]; }
    }
}
// (distances and block sizes are powers of two: lk / lj are their logarithms, so that a comparator's places come from shifts, not divisions)
#define LH_NET_FLIP(lk_) [&](int c, int& lo, int& hi) { const int blk = c >> ((lk_) - 1), p = c & ((1 << ((lk_) - 1)) - 1); lo = (blk << (lk_)) + p; hi = (blk << (lk_)) + (1 << (lk_)) - 1 - p; }
#define LH_NET_STEP(lj_) [&](int c, int& lo, int& hi) { lo = ((c >> (lj_)) << ((lj_) + 1)) + (c & ((1 << (lj_)) - 1)); hi = lo + (1 << (lj_)); }
__device__ __forceinline__ void wave_bitonic_u64(u64* a, int n, int lane) {
    int lnp = 6;
    while ((1 << lnp) < n) ++lnp;
    const int ncmp = 1 << (lnp - 1);
    for (int lk = 1; lk <= lnp; ++lk) {
        // the flip: place p of the first half of a block of k = 2^lk against place k - 1 - p of the block
        wave_ce_pass<4>(a, ncmp, n, lane, LH_NET_FLIP(lk));
        WAVE_SYNC();
        for (int lj = lk - 2; lj >= 0; --lj) {
            wave_ce_pass<4>(a, ncmp, n, lane, LH_NET_STEP(lj));
            WAVE_SYNC();
        }
    }
}

// (r06, late) the same network for a list longer than the LDS buffer: BL places (a power of two) at a time.  Comparators of the steps with k <= BL, and those of the later
// steps at distances below BL, stay inside BL-aligned blocks: a block is loaded once, taken through all of them in LDS and stored — only the flips of the steps k > BL
// and their distances >= BL run on the list in memory (one pass for up to 2 BL places, three for 4 BL; a pass per step in memory was 66 round trips for 2,048 places,
// a third of K8 on 400-pair barcodes).  g[0 .. n) in memory, lds: BL words.
template <int BL> __device__ __forceinline__ void wave_bitonic_u64_blocks(u64* g, int n, u64* lds, int lane) {
    int lbl = 0;
    while ((1 << lbl) < BL) ++lbl;
    int lnp = lbl;
    while ((1 << lnp) < n) ++lnp;
    const int ncmp = 1 << (lnp - 1);
    for (int b0 = 0; b0 < n; b0 += BL) {   // every block in order (the steps k = 2 .. BL)
        const int cnt = n - b0 < BL ? n - b0 : BL;
        for (int i = lane; i < cnt; i += 64) lds[i] = g[b0 + i];
        WAVE_SYNC();
        wave_bitonic_u64(lds, cnt, lane);
        for (int i = lane; i < cnt; i += 64) g[b0 + i] = lds[i];
        WAVE_SYNC();
    }
    for (int lk = lbl + 1; lk <= lnp; ++lk) {
        wave_ce_pass<8>(g, ncmp, n, lane, LH_NET_FLIP(lk));
        WAVE_SYNC();
        for (int lj = lk - 2; lj >= lbl; --lj) {
            wave_ce_pass<8>(g, ncmp, n, lane, LH_NET_STEP(lj));
            WAVE_SYNC();
        }
        for (int b0 = 0; b0 < n; b0 += BL) {   // distances BL / 2 .. 1, block by block
            const int cnt = n - b0 < BL ? n - b0 : BL;
            for (int i = lane; i < cnt; i += 64) lds[i] = g[b0 + i];
            WAVE_SYNC();
            for (int lj = lbl - 1; lj >= 0; --lj) {
                wave_ce_pass<4>(lds, BL >> 1, cnt, lane, LH_NET_STEP(lj));
                WAVE_SYNC();
            }
            for (int i = lane; i < cnt; i += 64) g[b0 + i] = lds[i];
            WAVE_SYNC();
        }
    }
}
#undef LH_NET_FLIP
#undef LH_NET_STEP

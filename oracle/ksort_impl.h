// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// Restatement of klib ksort.h's ks_introsort / ks_combsort / insertion sort
// (used by BWA for mem_intv, mem_flt, mem_ars, mem_ars2 and the uint64 seed sort).
// The algorithm is unstable, so equal-key order is reproduced only by running
// exactly this sequence of comparisons and swaps.
#pragma once
#include <cstddef>
#include <vector>

namespace orc {

template <class T, class Lt> static inline void ks_insertsort(T* s, T* t, Lt lt) {
    for (T* i = s + 1; i < t; ++i)
        for (T* j = i; j > s && lt(*j, *(j - 1)); --j) { T tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}

template <class T, class Lt> static inline void ks_combsort(size_t n, T* a, Lt lt) {
    const double shrink_factor = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    do {
        if (gap > 2) {
            gap = (size_t)(gap / shrink_factor);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (T* i = a; i < a + n - gap; ++i) {
            T* j = i + gap;
            if (lt(*j, *i)) { T tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) ks_insertsort(a, a + n, lt);
}

template <class T, class Lt> void ks_introsort(size_t n, T* a, Lt lt) {
    struct Stack { T *left, *right; int depth; };
    int d;
    T rp, swap_tmp;
    T *s, *t, *i, *j, *k;
    if (n < 1) return;
    else if (n == 2) {
        if (lt(a[1], a[0])) { swap_tmp = a[0]; a[0] = a[1]; a[1] = swap_tmp; }
        return;
    }
    for (d = 2; 1ul << d < n; ++d) {}
    std::vector<Stack> stack(sizeof(size_t) * d + 2);
    Stack* top = stack.data();
    s = a; t = a + (n - 1); d <<= 1;
    while (1) {
        if (s < t) {
            if (--d == 0) {
                ks_combsort((size_t)(t - s + 1), s, lt);
                t = s;
                continue;
            }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (lt(*k, *i)) {
                if (lt(*k, *j)) k = j;
            } else k = lt(*j, *i) ? i : j;
            rp = *k;
            if (k != t) { swap_tmp = *k; *k = *t; *t = swap_tmp; }
            for (;;) {
                do ++i; while (lt(*i, rp));
                do --j; while (i <= j && lt(rp, *j));
                if (j <= i) break;
                swap_tmp = *i; *i = *j; *j = swap_tmp;
            }
            swap_tmp = *i; *i = *t; *t = swap_tmp;
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack.data()) {
                ks_insertsort(a, a + n, lt);
                return;
            } else { --top; s = top->left; t = top->right; d = top->depth; }
        }
    }
}

}  // namespace orc

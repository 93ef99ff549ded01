// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// Restatement of Go 1.9's sort.Sort (src/sort/sort.go: insertionSort, siftDown, heapSort, medianOfThree,
// doPivot, quickSort, maxDepth).  lariat (README.md:14 "Go 1.9.2") calls it at
// go/src/inference/lariat.go:1546 (ByPosition) and go/src/inference/split.go:108 (SortSplitScoring).
// It is unstable; the equal-key order is reproduced by issuing the same Less/Swap sequence.
#pragma once

namespace orc {
namespace gosort {

template <class L, class S> void insertionSort(L& less, S& swp, int a, int b) {
    for (int i = a + 1; i < b; i++)
        for (int j = i; j > a && less(j, j - 1); j--) swp(j, j - 1);
}
template <class L, class S> void siftDown(L& less, S& swp, int lo, int hi, int first) {
    int root = lo;
    for (;;) {
        int child = 2 * root + 1;
        if (child >= hi) return;
        if (child + 1 < hi && less(first + child, first + child + 1)) child++;
        if (!less(first + root, first + child)) return;
        swp(first + root, first + child);
        root = child;
    }
}
template <class L, class S> void heapSort(L& less, S& swp, int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) siftDown(less, swp, i, hi, first);
    for (int i = hi - 1; i >= 0; i--) {
        swp(first, first + i);
        siftDown(less, swp, lo, i, first);
    }
}
template <class L, class S> void medianOfThree(L& less, S& swp, int m1, int m0, int m2) {
    if (less(m1, m0)) swp(m1, m0);
    if (less(m2, m1)) {
        swp(m2, m1);
        if (less(m1, m0)) swp(m1, m0);
    }
}
template <class L, class S> void doPivot(L& less, S& swp, int lo, int hi, int* midlo, int* midhi) {
    int m = (int)((unsigned)(lo + hi) >> 1);
    if (hi - lo > 40) {   // Tukey's "Ninther"
        int s = (hi - lo) / 8;
        medianOfThree(less, swp, lo, lo + s, lo + 2 * s);
        medianOfThree(less, swp, m, m - s, m + s);
        medianOfThree(less, swp, hi - 1, hi - 1 - s, hi - 1 - 2 * s);
    }
    medianOfThree(less, swp, lo, m, hi - 1);
    int pivot = lo;
    int a = lo + 1, c = hi - 1;
    for (; a < c && less(a, pivot); a++) {}
    int b = a;
    for (;;) {
        for (; b < c && !less(pivot, b); b++) {}      // data[b] <= pivot
        for (; b < c && less(pivot, c - 1); c--) {}   // data[c-1] > pivot
        if (b >= c) break;
        swp(b, c - 1);
        b++;
        c--;
    }
    bool protect = hi - c < 5;
    if (!protect && hi - c < (hi - lo) / 4) {
        int dups = 0;
        if (!less(pivot, hi - 1)) {   // data[hi-1] = pivot
            swp(c, hi - 1);
            c++;
            dups++;
        }
        if (!less(b - 1, pivot)) {   // data[b-1] = pivot
            b--;
            dups++;
        }
        if (!less(m, pivot)) {   // data[m] = pivot
            swp(m, b - 1);
            b--;
            dups++;
        }
        protect = dups > 1;
    }
    if (protect) {
        for (;;) {
            for (; a < b && !less(b - 1, pivot); b--) {}   // data[b] == pivot
            for (; a < b && less(a, pivot); a++) {}        // data[a] < pivot
            if (a >= b) break;
            swp(a, b - 1);
            a++;
            b--;
        }
    }
    swp(pivot, b - 1);
    *midlo = b - 1;
    *midhi = c;
}
template <class L, class S> void quickSort(L& less, S& swp, int a, int b, int maxDepth) {
    while (b - a > 12) {
        if (maxDepth == 0) { heapSort(less, swp, a, b); return; }
        maxDepth--;
        int mlo, mhi;
        doPivot(less, swp, a, b, &mlo, &mhi);
        if (mlo - a < b - mhi) {
            quickSort(less, swp, a, mlo, maxDepth);
            a = mhi;
        } else {
            quickSort(less, swp, mhi, b, maxDepth);
            b = mlo;
        }
    }
    if (b - a > 1) {
        for (int i = a + 6; i < b; i++)
            if (less(i, i - 6)) swp(i, i - 6);
        insertionSort(less, swp, a, b);
    }
}
}  // namespace gosort

template <class Less, class Swap> void go19_sort(int n, Less less, Swap swp) {
    int depth = 0;
    for (int i = n; i > 0; i >>= 1) depth++;
    gosort::quickSort(less, swp, 0, n, depth * 2);
}

}  // namespace orc

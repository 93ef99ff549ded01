// k_index_build.h — FM-index construction ON THE DEVICE (lh_index_build_device).
//
// What `bwa index` does on a host in an hour (the reference consumes its files: go/src/gobwa/gobwa.go:130; formats pinned by
// the fixture go/src/test/inputs/phix/PhiX.fa.*) sized for 288 GB of HBM: the suffixes of the text fwd || revcomp are
//   1. counted by their first LH_IB_PFX symbols (LDS-private histograms),
//   2. gathered chunk by chunk (a chunk = a range of prefixes holding at most 2^build_chunk_log2 suffixes) together with
//      their first 32 symbols as one 64-bit key, and radix-sorted by that key,
//   3. finished where keys tie: each group of equal keys is sorted in place by direct comparison through the packed text
//      (one lane per group; on a genome groups are short: repeats longer than 32 bases),
// and the BWT in the `.bwt` file's layout (occurrence counts interleaved every 128 symbols) is derived from the full suffix
// array by a blocked count scan.  Everything downstream (occurrence records, dense SA, ISA, filters) is lh_host.inc's.
// The sentinel is smallest: a suffix that is a prefix of another sorts first (zero padding + length comparison).
#pragma once
#include "lh_dev.h"

#define LH_IB_PFX 6                       // symbols of the chunking prefix
#define LH_IB_BINS (1 << (2 * LH_IB_PFX))

// 32 symbols starting at text position i (zero padded past the end: W has two spare zero words)
__device__ __forceinline__ u64 ib_key(const u64* __restrict__ W, u64 i) {
    u64 wi = i >> 5;
    int sh = (int)(i & 31) << 1;
    u64 a = W[wi];
    return sh ? (a << sh) | (W[wi + 1] >> (64 - sh)) : a;
}
__device__ __forceinline__ int ib_sym(const u64* __restrict__ W, u64 p) { return (int)(W[p >> 5] >> (62 - ((int)(p & 31) << 1))) & 3; }

// suffix a < suffix b, given that their first `off0` symbols are equal (as padded keys)
__device__ __forceinline__ bool ib_suf_less(const u64* __restrict__ W, u64 n, u64 a, u64 b, u64 off0) {
    u64 la = n - a, lb = n - b, l = la < lb ? la : lb;
    for (u64 off = off0; off < l; off += 32) {
        u64 wa = ib_key(W, a + off), wb = ib_key(W, b + off);
        if (wa != wb) {
            int lead = __clzll((long long)(wa ^ wb)) >> 1;   // first differing symbol
            if (off + (u64)lead >= l) break;                  // the difference lies in the padding
            return wa < wb;
        }
    }
    return la < lb;   // one is a prefix of the other: the shorter one (sentinel first) is smaller
}

// the text fwd || revcomp, 32 symbols per u64, MSB first
__global__ void __launch_bounds__(256) k_ib_text(const uint8_t* __restrict__ pac, i64 l_pac, u64 n_words, u64* __restrict__ W) {
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (u64)gridDim.x * blockDim.x) {
        u64 v = 0;
        for (int k = 0; k < 32; ++k) {
            i64 p = (i64)(w << 5) + k;
            if (p >= 2 * l_pac) break;
            u64 b = p < l_pac ? (u64)dev_pac(pac, p) : (u64)(3 - dev_pac(pac, 2 * l_pac - 1 - p));
            v |= b << (62 - 2 * k);
        }
        W[w] = v;
    }
}

// histogram of the LH_IB_PFX-symbol prefixes of all n suffixes; one thread per text word (32 suffixes)
__global__ void __launch_bounds__(256) k_ib_hist(const u64* __restrict__ W, u64 n, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[LH_IB_BINS];
    for (int b = threadIdx.x; b < LH_IB_BINS; b += 256) h[b] = 0;
    __syncthreads();
    u64 n_words = (n + 31) >> 5;
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (u64)gridDim.x * blockDim.x) {
        u64 a = W[w], b = W[w + 1];
        for (int k = 0; k < 32; ++k) {
            if ((w << 5) + k >= n) break;
            u64 key = k ? (a << (2 * k)) | (b >> (64 - 2 * k)) : a;
            atomicAdd(&h[key >> (64 - 2 * LH_IB_PFX)], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < LH_IB_BINS; b += 256) if (h[b]) atomicAdd(&hist[b], (unsigned long long)h[b]);
}

// (key, position) of every suffix whose prefix bin lies in [bin_lo, bin_hi), appended in no particular order
__global__ void __launch_bounds__(256) k_ib_gather(const u64* __restrict__ W, u64 n, uint32_t bin_lo, uint32_t bin_hi, unsigned long long* __restrict__ cursor,
                                                   u64* __restrict__ keys, u64* __restrict__ vals) {
    __shared__ uint32_t s_cnt[256];
    __shared__ unsigned long long s_base;
    u64 n_words = (n + 31) >> 5;
    u64 n_iter = (n_words + (u64)gridDim.x * 256 - 1) / ((u64)gridDim.x * 256);
    for (u64 it = 0; it < n_iter; ++it) {
        u64 w = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        u64 a = 0, b = 0;
        uint32_t mine = 0;
        if (w < n_words) {
            a = W[w]; b = W[w + 1];
            for (int k = 0; k < 32; ++k) {
                if ((w << 5) + k >= n) break;
                u64 key = k ? (a << (2 * k)) | (b >> (64 - 2 * k)) : a;
                uint32_t bin = (uint32_t)(key >> (64 - 2 * LH_IB_PFX));
                mine += bin >= bin_lo && bin < bin_hi;
            }
        }
        s_cnt[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int t = 0; t < 256; ++t) { uint32_t c = s_cnt[t]; s_cnt[t] = tot; tot += c; }
            s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (mine) {
            u64 o = s_base + s_cnt[threadIdx.x];
            for (int k = 0; k < 32; ++k) {
                if ((w << 5) + k >= n) break;
                u64 key = k ? (a << (2 * k)) | (b >> (64 - 2 * k)) : a;
                uint32_t bin = (uint32_t)(key >> (64 - 2 * LH_IB_PFX));
                if (bin >= bin_lo && bin < bin_hi) { keys[o] = key; vals[o] = (w << 5) + k; ++o; }
            }
        }
        __syncthreads();
    }
}

// number of chunk slots j > 0 whose key equals its predecessor's
__global__ void __launch_bounds__(256) k_ib_count_ties(const u64* __restrict__ keys, u64 cnt, unsigned long long* __restrict__ n_ties) {
    unsigned long long mine = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1; j < cnt; j += (u64)gridDim.x * blockDim.x) mine += keys[j] == keys[j - 1];
    if (mine) atomicAdd(n_ties, mine);
}

// every group of equal keys (suffixes that share their first 32 symbols), sorted by direct comparison of the text from symbol 32 on.
// Groups of up to LH_IB_LANE_GROUP suffixes are heap-sorted in place by their head's lane.  Larger ones — a real reference has
// satellite arrays, long exact duplications and fwd / revcomp palindromes whose tie groups hold 10^5 .. 10^6 suffixes sharing
// kilobases — are only LISTED here and sorted by k_ib_sort_big, one 256-thread block per group (a lane's heap sort of such a group
// is m log m comparisons of L / 32 dependent reads each, all on one lane).
#define LH_IB_LANE_GROUP 64
__global__ void __launch_bounds__(256) k_ib_sort_groups(const u64* __restrict__ W, u64 n, const u64* __restrict__ keys, u64* __restrict__ vals, u64 cnt,
                                                        u64* __restrict__ big_list, unsigned long long* __restrict__ big_count) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j + 1 < cnt; j += (u64)gridDim.x * blockDim.x) {
        u64 k = keys[j];
        if ((j > 0 && keys[j - 1] == k) || keys[j + 1] != k) continue;   // not the head of a group of >= 2
        u64 e = j + 2;
        while (e < cnt && e - j <= LH_IB_LANE_GROUP && keys[e] == k) ++e;
        u64* v = vals + j;
        u64 m = e - j;
        if (m > LH_IB_LANE_GROUP) {   // (its length is found by the block that sorts it)
            const u64 slot = atomicAdd(big_count, 1ull);
            big_list[slot] = j;
            continue;
        }
        if (m == 2) {
            if (ib_suf_less(W, n, v[1], v[0], 32)) { u64 t = v[0]; v[0] = v[1]; v[1] = t; }
            continue;
        }
        // heap sort (max-heap, in place)
        for (u64 start = m / 2; start-- > 0;) {
            u64 root = start;
            for (;;) {
                u64 child = 2 * root + 1;
                if (child >= m) break;
                if (child + 1 < m && ib_suf_less(W, n, v[child], v[child + 1], 32)) ++child;
                if (!ib_suf_less(W, n, v[root], v[child], 32)) break;
                u64 t = v[root]; v[root] = v[child]; v[child] = t;
                root = child;
            }
        }
        for (u64 end = m - 1; end > 0; --end) {
            u64 t = v[0]; v[0] = v[end]; v[end] = t;
            u64 root = 0;
            for (;;) {
                u64 child = 2 * root + 1;
                if (child >= end) break;
                if (child + 1 < end && ib_suf_less(W, n, v[child], v[child + 1], 32)) ++child;
                if (!ib_suf_less(W, n, v[root], v[child], 32)) break;
                u64 t2 = v[root]; v[root] = v[child]; v[child] = t2;
                root = child;
            }
        }
    }
}
// a listed group, one block per group: runs of 16 by insertion sort (one thread each), then merge passes between vals and tmp (the
// chunk's spare value buffer) in which every thread produces segments of 32 outputs from their merge-path split.  Suffixes are
// distinct, so the order is total: no stability question.
__global__ void __launch_bounds__(256) k_ib_sort_big(const u64* __restrict__ W, u64 n, const u64* __restrict__ keys, u64* __restrict__ vals, u64* __restrict__ tmp, u64 cnt,
                                                     const u64* __restrict__ big_list, const unsigned long long* __restrict__ big_count) {
    const u64 n_big = *big_count;
    for (u64 g = blockIdx.x; g < n_big; g += gridDim.x) {
        const u64 j = big_list[g], k = keys[j];
        __shared__ u64 s_m;
        if (threadIdx.x == 0) { u64 e = j + 1; while (e < cnt && keys[e] == k) ++e; s_m = e - j; }
        __syncthreads();
        const u64 m = s_m;
        u64* src = vals + j;
        u64* dst = tmp + j;
        for (u64 r0 = (u64)threadIdx.x * 16; r0 < m; r0 += 256 * 16) {   // runs of 16
            const u64 r1 = r0 + 16 < m ? r0 + 16 : m;
            for (u64 i = r0 + 1; i < r1; ++i) {
                const u64 x = src[i];
                u64 q = i;
                while (q > r0 && ib_suf_less(W, n, x, src[q - 1], 32)) { src[q] = src[q - 1]; --q; }
                src[q] = x;
            }
        }
        __syncthreads();
        for (u64 R = 16; R < m; R <<= 1) {
            for (u64 o0 = (u64)threadIdx.x * 32; o0 < m; o0 += 256 * 32) {   // outputs [o0, o0 + 32) of this pass
                const u64 pb = o0 / (2 * R) * (2 * R);
                const u64 a0 = pb, a1 = pb + R < m ? pb + R : m, b0 = a1, b1 = pb + 2 * R < m ? pb + 2 * R : m;
                const u64 na = a1 - a0, nb = b1 - b0, d = o0 - pb;
                u64 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;   // how many of the first d outputs come from A
                while (lo < hi) {
                    const u64 ai = (lo + hi) >> 1;   // A[ai] vs B[d - ai - 1]: A[ai] goes before iff it is smaller
                    if (ib_suf_less(W, n, src[b0 + (d - ai - 1)], src[a0 + ai], 32)) hi = ai; else lo = ai + 1;
                }
                u64 ai = lo, bi = d - lo;
                const u64 o1 = o0 + 32 < pb + na + nb ? o0 + 32 : pb + na + nb;
                for (u64 oo = o0; oo < o1; ++oo) {
                    bool take_a;
                    if (ai >= na) take_a = false;
                    else if (bi >= nb) take_a = true;
                    else take_a = !ib_suf_less(W, n, src[b0 + bi], src[a0 + ai], 32);
                    dst[oo] = take_a ? src[a0 + ai++] : src[b0 + bi++];
                }
            }
            __syncthreads();
            u64* t = src; src = dst; dst = t;
        }
        if (src != vals + j) for (u64 i = threadIdx.x; i < m; i += 256) vals[j + i] = src[i];
        __syncthreads();
    }
}

// bit (row0 + j) of rep <- suffixes j and j + 1 of the chunk share their first 19 bases, both complete (lh_dev.h LH_BLOOM_K);
// chunks of different prefixes never do.  The sweep filter's "occurs at least twice" set is built from these bits.
__global__ void __launch_bounds__(256) k_ib_rep_bits(const u64* __restrict__ keys, const u64* __restrict__ vals, u64 cnt, u64 row0, u64 n, u64* __restrict__ rep,
                                                     unsigned long long* __restrict__ n_rep) {
    unsigned long long mine = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j + 1 < cnt; j += (u64)gridDim.x * blockDim.x) {
        if ((keys[j] >> (64 - 2 * LH_BLOOM_K)) != (keys[j + 1] >> (64 - 2 * LH_BLOOM_K))) continue;
        if (vals[j] + LH_BLOOM_K > n || vals[j + 1] + LH_BLOOM_K > n) continue;
        u64 row = row0 + j;
        atomicOr((unsigned long long*)&rep[row >> 6], 1ull << (row & 63));
        ++mine;
    }
    if (mine) atomicAdd(n_rep, mine);
}

// LCP array and k-mer tree bounds from the sorted chunk (sequential reads of the keys; lh_index.inc has the generic versions that
// read the text through the suffix array, for indexes loaded from files).  lcp[row] = symbols shared with the previous row's suffix
// (capped at 255): from the 32-symbol keys, by direct comparison when they are equal.  raw[2e], raw[2e+1] = first row / last row + 1
// of the group of rows whose suffix starts with the string of tree entry e (level L at (4^L - 4) / 3, code with symbol t at bits 2t).
__device__ __forceinline__ uint32_t ib_code_lsb(u64 key) {   // the first 16 symbols of an MSB-first key, symbol t at bits 2t..2t+1
    u64 r = __brevll(key);
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    return (uint32_t)r;
}
__global__ void __launch_bounds__(256) k_ib_lcp_ktree(const u64* __restrict__ W, u64 n, const u64* __restrict__ keys, const u64* __restrict__ vals, u64 cnt, u64 row0,
                                                      const u64* __restrict__ sa, uint8_t* __restrict__ lcp, int levels, u64* __restrict__ raw) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += (u64)gridDim.x * blockDim.x) {
        const u64 row = row0 + j, v = vals[j], k = keys[j];
        u64 vp = 0, kp = 0;
        bool have = true;
        if (j > 0) { vp = vals[j - 1]; kp = keys[j - 1]; }
        else if (row0 > 1) { vp = sa[row0 - 1]; kp = ib_key(W, vp); }
        else have = false;
        u64 la = n - v, lb = have ? n - vp : 0, lim = la < lb ? la : lb;
        u64 l = 0;
        if (have) {
            u64 x = k ^ kp;
            l = x ? (u64)(__clzll((long long)x) >> 1) : 32;
            if (l >= 32)
                for (; l < 256 && l < lim; l += 32) {
                    u64 wa = ib_key(W, v + l), wb = ib_key(W, vp + l);
                    if (wa != wb) { l += (u64)(__clzll((long long)(wa ^ wb)) >> 1); break; }
                }
            l = l < lim ? l : lim;
        }
        if (lcp) lcp[row] = (uint8_t)(l > 255 ? 255 : l);
        if (raw) {
            const uint32_t c = ib_code_lsb(k), cp = ib_code_lsb(kp);
            for (int L = (int)(l < 32 ? l : 32) + 1; L <= levels; ++L) {
                const uint32_t m = (1u << (2 * L)) - 1u;
                const u64 off = ((1ull << (2 * L)) - 4) / 3;
                if ((u64)L <= la) raw[2 * (off + (c & m))] = row;
                if (have && (u64)L <= lb) raw[2 * (off + (cp & m)) + 1] = row;
            }
        }
    }
}
// the last row's groups end behind the last row
__global__ void k_ib_ktree_close(const u64* __restrict__ W, u64 n, const u64* __restrict__ sa, int levels, u64* __restrict__ raw) {
    if (threadIdx.x || blockIdx.x) return;
    const u64 v = sa[n], la = n - v;
    const uint32_t c = ib_code_lsb(ib_key(W, v));
    for (int L = 1; L <= levels; ++L)
        if ((u64)L <= la) raw[2 * ((((1ull << (2 * L)) - 4) / 3) + (c & ((1u << (2 * L)) - 1u))) + 1] = n + 1;
}

// chunk -> its rows of the full suffix array (row 0 is the sentinel's); the row of suffix 0 is `primary`
__global__ void __launch_bounds__(256) k_ib_write_sa(const u64* __restrict__ vals, u64 cnt, u64 row0, u64* __restrict__ sa, unsigned long long* __restrict__ primary) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += (u64)gridDim.x * blockDim.x) {
        u64 v = vals[j];
        sa[row0 + j] = v;
        if (v == 0) *primary = row0 + j;
    }
}

// ---- BWT in the .bwt file's layout (bwt.c bwt_bwtupdate_core): per 128 symbols [4 x u64 counts before the block][8 x u32 of
// 2-bit symbols, first symbol in the top bits]; one more count record after the last block.  The '$' row is left out.
// symbol of stored position j: row = j + (j >= primary); preceded by T[sa[row] - 1] (row 0: the sentinel suffix, preceded by T[n-1])
__device__ __forceinline__ uint32_t ib_bwt_sym(const u64* __restrict__ W, const u64* __restrict__ sa, u64 n, u64 primary, u64 j) {
    u64 row = j + (j >= primary);
    u64 s = row == 0 ? n : sa[row];
    return (uint32_t)ib_sym(W, s - 1);
}
// pass 1: the symbol words of every block and the block's own counts (4 x u32)
__global__ void __launch_bounds__(256) k_ib_bwt_blocks(const u64* __restrict__ W, const u64* __restrict__ sa, u64 n, u64 primary, u64 n_blk, uint32_t* __restrict__ bwa,
                                                       u64 bwa_words, uint32_t* __restrict__ blk_cnt) {
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blk; b += (u64)gridDim.x * blockDim.x) {
        uint32_t c[4] = {0, 0, 0, 0};
        for (int t = 0; t < 8; ++t) {
            u64 j0 = (b << 7) + ((u64)t << 4);
            if (j0 >= n) break;
            uint32_t word = 0;
            for (int s = 0; s < 16 && j0 + s < n; ++s) {
                uint32_t sym = ib_bwt_sym(W, sa, n, primary, j0 + s);
                word |= sym << ((15 - s) << 1);
                ++c[sym];
            }
            u64 wo = (b << 4) + 8 + t;
            if (wo < bwa_words) bwa[wo] = word;
        }
        blk_cnt[4 * b] = c[0]; blk_cnt[4 * b + 1] = c[1]; blk_cnt[4 * b + 2] = c[2]; blk_cnt[4 * b + 3] = c[3];
    }
}
// pass 2: per tile of LH_IB_TILE blocks the tile's totals
#define LH_IB_TILE 1024
__global__ void __launch_bounds__(256) k_ib_tile_sums(const uint32_t* __restrict__ blk_cnt, u64 n_blk, u64* __restrict__ tile_sum) {
    __shared__ u64 s[4][256];
    u64 b0 = (u64)blockIdx.x * LH_IB_TILE;
    u64 c[4] = {0, 0, 0, 0};
    for (int u = 0; u < LH_IB_TILE / 256; ++u) {
        u64 b = b0 + (u64)threadIdx.x * (LH_IB_TILE / 256) + u;
        if (b < n_blk) for (int q = 0; q < 4; ++q) c[q] += blk_cnt[4 * b + q];
    }
    for (int q = 0; q < 4; ++q) s[q][threadIdx.x] = c[q];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) for (int q = 0; q < 4; ++q) s[q][threadIdx.x] += s[q][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int q = 0; q < 4; ++q) tile_sum[4 * (u64)blockIdx.x + q] = s[q][0];
}
// pass 3: exclusive scan of the tile totals (one workgroup; thread t owns a contiguous slice)
__global__ void __launch_bounds__(256) k_ib_scan_tiles(u64 n_tiles, u64* __restrict__ tile_sum) {
    __shared__ u64 s[4][256];
    u64 per = (n_tiles + 255) / 256;
    u64 t0 = (u64)threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    u64 c[4] = {0, 0, 0, 0};
    for (u64 t = t0; t < t1; ++t) for (int q = 0; q < 4; ++q) c[q] += tile_sum[4 * t + q];
    for (int q = 0; q < 4; ++q) s[q][threadIdx.x] = c[q];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int q = 0; q < 4; ++q) { u64 acc = 0; for (int t = 0; t < 256; ++t) { u64 v = s[q][t]; s[q][t] = acc; acc += v; } }
    __syncthreads();
    for (int q = 0; q < 4; ++q) c[q] = s[q][threadIdx.x];
    for (u64 t = t0; t < t1; ++t) for (int q = 0; q < 4; ++q) { u64 v = tile_sum[4 * t + q]; tile_sum[4 * t + q] = c[q]; c[q] += v; }
}
// pass 4: the count records (counts BEFORE each block), and the closing record after the last block
__global__ void __launch_bounds__(256) k_ib_bwa_counts(const uint32_t* __restrict__ blk_cnt, u64 n_blk, const u64* __restrict__ tile_base, uint32_t* __restrict__ bwa,
                                                       u64 bwa_words) {
    __shared__ u64 s[4][256];
    const int PER = LH_IB_TILE / 256;
    u64 b0 = (u64)blockIdx.x * LH_IB_TILE + (u64)threadIdx.x * PER;
    u64 c[4] = {0, 0, 0, 0};
    for (int u = 0; u < PER; ++u) if (b0 + u < n_blk) for (int q = 0; q < 4; ++q) c[q] += blk_cnt[4 * (b0 + u) + q];
    for (int q = 0; q < 4; ++q) s[q][threadIdx.x] = c[q];
    __syncthreads();
    if (threadIdx.x < 4) { int q = threadIdx.x; u64 acc = tile_base[4 * (u64)blockIdx.x + q]; for (int t = 0; t < 256; ++t) { u64 v = s[q][t]; s[q][t] = acc; acc += v; } }
    __syncthreads();
    for (int q = 0; q < 4; ++q) c[q] = s[q][threadIdx.x];
    for (int u = 0; u < PER; ++u) {
        u64 b = b0 + u;
        if (b > n_blk) break;
        u64 wo = b < n_blk ? b << 4 : bwa_words - 8;   // b == n_blk: the closing record
        for (int q = 0; q < 4; ++q) { bwa[wo + 2 * q] = (uint32_t)c[q]; bwa[wo + 2 * q + 1] = (uint32_t)(c[q] >> 32); }
        if (b < n_blk) for (int q = 0; q < 4; ++q) c[q] += blk_cnt[4 * b + q];
    }
}

// every v-th row of a suffix array sampled every `have`-th row (v a multiple of have)
__global__ void __launch_bounds__(256) k_ib_sa_subsample(const u64* __restrict__ sa, u64 step, u64 n_new, u64* __restrict__ out) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += (u64)gridDim.x * blockDim.x) out[j] = sa[j * step];
}

// the inverse of lh_host.inc's k_occ_relayout: occurrence records -> the .bwt file's layout (lh_index_export)
__global__ void __launch_bounds__(256) k_ib_occ_to_bwa(const uint4* __restrict__ occ, const u64* __restrict__ sb, int sb_shift, u64 n, u64 n_blk, uint32_t* __restrict__ bwa,
                                                       u64 bwa_words) {
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b <= n_blk; b += (u64)gridDim.x * blockDim.x) {
        u64 wo = b < n_blk ? b << 4 : bwa_words - 8;
        u64 c[4];
        if (b < n_blk) {
            uint4 h = occ[4 * b];
            c[0] = h.x; c[1] = h.y; c[2] = h.z; c[3] = h.w;
            if (sb) { const u64* q = sb + (((b << 7) >> sb_shift) << 2); c[0] += q[0]; c[1] += q[1]; c[2] += q[2]; c[3] += q[3]; }
        } else {   // totals: the last record's counts plus its symbols
            u64 last = n - 1, r = last >> 6;
            uint4 h = occ[2 * r], d = occ[2 * r + 1];
            int m = (int)(last & 63) + 1;
            u64 mask = ~0ull >> (64 - m);
            u64 hi = ((u64)d.y << 32 | d.x) & mask, lo = ((u64)d.w << 32 | d.z) & mask;
            u64 i3 = hi & lo, i2 = hi ^ i3, i1 = lo ^ i3;
            int n3 = __popcll(i3), n2 = __popcll(i2), n1 = __popcll(i1);
            c[0] = (u64)h.x + (u64)(m - n1 - n2 - n3); c[1] = (u64)h.y + n1; c[2] = (u64)h.z + n2; c[3] = (u64)h.w + n3;
            if (sb) { const u64* q = sb + ((last >> sb_shift) << 2); c[0] += q[0]; c[1] += q[1]; c[2] += q[2]; c[3] += q[3]; }
        }
        for (int q = 0; q < 4; ++q) { bwa[wo + 2 * q] = (uint32_t)c[q]; bwa[wo + 2 * q + 1] = (uint32_t)(c[q] >> 32); }
        if (b == n_blk) continue;
        for (int half = 0; half < 2; ++half) {
            u64 r = 2 * b + half;
            if ((r << 6) >= n) break;
            uint4 d = occ[2 * r + 1];
            u64 hi = (u64)d.y << 32 | d.x, lo = (u64)d.w << 32 | d.z;
            for (int t = 0; t < 4; ++t) {
                u64 j0 = (r << 6) + ((u64)t << 4);
                if (j0 >= n) break;
                uint32_t word = 0;
                for (int s = 0; s < 16; ++s) {
                    int bit = t * 16 + s;
                    uint32_t sym = (uint32_t)((hi >> bit & 1) << 1 | (lo >> bit & 1));
                    if (j0 + s < n) word |= sym << ((15 - s) << 1);
                }
                bwa[wo + 8 + 4 * half + t] = word;
            }
        }
    }
}

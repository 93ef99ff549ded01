// k_seed.h — K2: SA lookups (bwt_sa) for every sampled occurrence of every SMEM interval, one LANE per seed,
// plus the exclusive scan that sizes the seed pool.  Replaces the `bwt_sa` / `bns_intv2rid` part of BWA's mem_chain
// (reached through mem_align1_core, go/src/gobwa/gobwa.go:244,253).
// Each lane chases ~sa_intv/2 dependent LF steps, every step one random 64-B occurrence block: pure HBM latency,
// hidden by running one independent chain per lane.
#pragma once
#include "lh_dev.h"

struct DSeed { i64 rbeg; int32_t qbeg, len; };

// exclusive scan out[i] = sum_{j<i} max(in[j] + add, at_least), out[n] = total, in three launches:
//   k_scan_partial (per 2048-element tile: tile sums) -> k_scan_tiles (one workgroup scans the tile sums) -> k_scan_final.
#define LH_SCAN_TILE 2048
__device__ __forceinline__ i64 scan_val(const int32_t* in, int i, int n, int add, int at_least) {
    if (i >= n) return 0;
    i64 v = (i64)in[i] + add;
    return v < at_least ? at_least : v;
}
__global__ void __launch_bounds__(256) k_scan_partial(int n, const int32_t* __restrict__ in, int add, int at_least, i64* __restrict__ tile_sum) {
    __shared__ i64 part[256];
    int t = threadIdx.x, base = blockIdx.x * LH_SCAN_TILE + t * 8;
    i64 s = 0;
    for (int u = 0; u < 8; ++u) s += scan_val(in, base + u, n, add, at_least);
    part[t] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) part[t] += part[t + d];
        __syncthreads();
    }
    if (t == 0) tile_sum[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(256) k_scan_tiles(int n_tiles, i64* __restrict__ tile_sum) {   // in place: exclusive scan, total at [n_tiles]
    __shared__ i64 part[256];
    __shared__ i64 carry_s;
    int t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 256) {
        int i = base + t;
        i64 v = i < n_tiles ? tile_sum[i] : 0;
        part[t] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            i64 o = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += o;
            __syncthreads();
        }
        i64 excl = part[t] - v + carry_s;
        __syncthreads();
        if (i < n_tiles) tile_sum[i] = excl;
        if (t == 255) carry_s += part[255];
        __syncthreads();
    }
    if (t == 0) tile_sum[n_tiles] = carry_s;
}
__global__ void __launch_bounds__(256) k_scan_final(int n, const int32_t* __restrict__ in, int add, int at_least, const i64* __restrict__ tile_sum, int n_tiles,
                                                    i64* __restrict__ out) {
    __shared__ i64 part[256];
    int t = threadIdx.x, base = blockIdx.x * LH_SCAN_TILE + t * 8;
    i64 loc[8], s = 0;
    for (int u = 0; u < 8; ++u) { loc[u] = s; s += scan_val(in, base + u, n, add, at_least); }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        i64 o = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    i64 excl = part[t] - s + tile_sum[blockIdx.x];
    for (int u = 0; u < 8; ++u) if (base + u < n) out[base + u] = excl + loc[u];
    if (blockIdx.x == 0 && t == 0) out[n] = tile_sum[n_tiles];
}

// owner[g] = the read seed slot g belongs to (one lane per read writes its few slots): spares K2 a 21-step binary search per seed
__global__ void __launch_bounds__(256) k_seed_owner(int n_reads, const i64* __restrict__ seed_off, i64 pool_cap, int32_t* __restrict__ owner) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
        i64 e = seed_off[r + 1] < pool_cap ? seed_off[r + 1] : pool_cap;
        for (i64 g = seed_off[r]; g < e; ++g) owner[g] = r;
    }
}

// K2.  one lane per seed; grid-stride over the pool.
__global__ void __launch_bounds__(256) k_seed(DIndex ix, DOpts o, int n_reads, const i64* __restrict__ seed_off, i64 pool_cap,
                                              const DIntv* __restrict__ intv, const int32_t* __restrict__ n_intv, DSeed* __restrict__ seeds,
                                              int32_t* __restrict__ s_rid, DCounters* __restrict__ ctr, const int32_t* __restrict__ owner,
                                              const int32_t* __restrict__ big_slot, const DIntv* __restrict__ big_slab) {
    i64 total = seed_off[n_reads];
    if (total > pool_cap) total = pool_cap;
    i64 stride = (i64)gridDim.x * blockDim.x;
    i64 nrounds = (total + stride - 1) / stride;
    for (i64 rd = 0; rd < nrounds; ++rd) {
        i64 g = rd * stride + (i64)blockIdx.x * blockDim.x + threadIdx.x;
        int nlf = 0, nsa = 0;
        if (g < total) {
            int r = owner[g];
            i64 u = g - seed_off[r];
            const int bs = big_slot ? big_slot[r] : -1;   // (a read with more than LH_MAX_INTV intervals: the sorted half of its big-slab slot, k_smem4.h)
            const DIntv* iv = bs < 0 ? intv + (size_t)r * LH_MAX_INTV : big_slab + ((size_t)bs * 2 + 1) * LH_BIG_INTV;
            int n = n_intv[r];
            u64 x0 = 0, step = 1, info = 0;
            for (int t = 0; t < n; ++t) {
                u64 s = iv[t].x2;
                u64 st = 1, c = s;   // (the usual interval has at most max_occ occurrences: every one is a seed; the 64-bit divisions are for the others)
                if (s > (u64)o.max_occ) {
                    st = s / (u64)o.max_occ;
                    c = (s + st - 1) / st;
                    if (c > (u64)o.max_occ) c = (u64)o.max_occ;
                }
                if ((u64)u < c) { x0 = iv[t].x0; step = st; info = iv[t].info; break; }
                u -= (i64)c;
            }
            i64 rbeg;
            if (x0 >> 62 & 1) rbeg = (i64)(x0 & ~(1ull << 62));   // K1 stored the interval's one occurrence by its text position (LH_POSF, k_smem4.h)
            else rbeg = (i64)dev_sa(ix, x0 + (u64)u * step, &nlf);
            nsa = 1;   // (n_sa counts the reference's bwt_sa calls: one per seed)
            int qbeg = (int)(info >> 32), slen = (int)(uint32_t)info - qbeg;
            DSeed sd;
            sd.rbeg = rbeg; sd.qbeg = qbeg; sd.len = slen;
            seeds[g] = sd;
            s_rid[g] = dev_intv2rid(ix, rbeg, rbeg + slen);
        }
        if (ctr) {
            int tl = wave_sum_i32(nlf), ts = wave_sum_i32(nsa);
            if (LANE() == 0 && ts) { atomicAdd(&LH_CTR(ctr)->n_lf, (u64)tl); atomicAdd(&LH_CTR(ctr)->n_sa, (u64)ts); }
        }
    }
}

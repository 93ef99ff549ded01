// k_smem3.h — K1 v3: SMEM seeding with one LANE per read (64 independent FM-index walks per wavefront).
//
// Measured on MI355X (profiles/r01_*): giving a read a whole wave (the first K1, since removed) makes the kernel VALU-issue bound — the
// forward extension is one dependent chain, so 64 lanes spend ~120 instructions per bwt_extend.  Here every lane runs
// BWA's mem_collect_intv (bwt_smem1a passes 1+2, bwt_seed_strategy1 pass 3; reached from go/src/gobwa/gobwa.go:244,253)
// for its own read: one wave instruction advances up to 64 walks, each lane decodes its own 64-B occurrence blocks
// (4 x 16-B loads), and the memory system sees 64 independent requests in flight per wave.  Divergence is limited to
// loop trip counts (all lanes run the same loop nest).  The per-walk interval lists (bwt_smem1a's prev/curr) live in an
// HBM scratch slab, interleaved by thread so that equal list slots of neighbouring lanes are adjacent.
//
// Output: unsorted intervals + count per read; k_smem_fin sorts by `info` and derives seed counts / l_rep.
#pragma once
#include "lh_dev.h"

struct TList {   // one thread's interval list inside the interleaved slab: field f of entry e at base[(e*4+f)*stride]
    u64* base;
    size_t stride;
    __device__ __forceinline__ DIntv get(int e) const {
        DIntv v;
        const u64* p = base + (size_t)e * 4 * stride;
        v.x0 = p[0]; v.x1 = p[stride]; v.x2 = p[2 * stride]; v.info = p[3 * stride];
        return v;
    }
    __device__ __forceinline__ void put(int e, const DIntv& v) const {
        u64* p = base + (size_t)e * 4 * stride;
        p[0] = v.x0; p[stride] = v.x1; p[2 * stride] = v.x2; p[3 * stride] = v.info;
    }
};

struct S3Out { DIntv* out; int n, ovf; unsigned n_ext; };

__device__ __forceinline__ void s3_emit(S3Out& so, const DIntv& m, int min_seed_len) {
    int slen = (int)(uint32_t)m.info - (int)(m.info >> 32);
    if (slen < min_seed_len) return;
    if (so.n >= LH_MAX_INTV) { so.ovf = 1; return; }
    so.out[so.n++] = m;
}

// bwt_smem1a(bwt, len, q, x, min_intv, max_intv = 0, ...) for one lane
__device__ __forceinline__ int lane_smem1(const DIndex& ix, int len, const uint8_t* q, int x, int min_intv, TList A, TList B, S3Out& so, int min_seed_len) {
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    TList curr = A, prev = B;
    DIntv ik = dev_set_intv(ix, q[x]);
    ik.info = (u64)(x + 1);
    int ncurr = 0, i;
    for (i = x + 1; i < len; ++i) {   // forward search
        int b = q[i];
        if (b < 4) {
            DIntv ok = dev_extend_c(ix, ik, 3 - b, 0);
            so.n_ext++;
            if (ok.x2 != ik.x2) {
                curr.put(ncurr++, ik);
                if (ok.x2 < (u64)min_intv) break;
            }
            ik = ok; ik.info = (u64)(i + 1);
        } else {
            curr.put(ncurr++, ik);
            break;
        }
    }
    if (i == len) curr.put(ncurr++, ik);
    int ret = (int)ik.info;   // end of the longest forward match (the last interval pushed)
    // the backward sweep visits curr in reverse (longest match first): index it backwards instead of copying
    int nprev = ncurr, rev = 1;
    { TList t = curr; curr = prev; prev = t; }
    int have_mem = 0, last_mem_start = 0;
    for (i = x - 1; i >= -1; --i) {
        int c = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1);
        ncurr = 0;
        u64 last_size = 0;
        for (int j = 0; j < nprev; ++j) {
            DIntv p = prev.get(rev ? nprev - 1 - j : j);
            DIntv ok = p;
            int fail = 1;
            if (c >= 0) { ok = dev_extend_c(ix, p, c, 1); so.n_ext++; fail = ok.x2 < (u64)min_intv; }
            if (fail) {
                if (ncurr == 0 && (!have_mem || i + 1 < last_mem_start)) {   // no longer match survived, not contained in the previous MEM
                    DIntv m = p;
                    m.info |= (u64)(i + 1) << 32;
                    s3_emit(so, m, min_seed_len);
                    have_mem = 1; last_mem_start = i + 1;
                }
            } else if (ncurr == 0 || ok.x2 != last_size) {
                ok.info = p.info;
                curr.put(ncurr++, ok);
                last_size = ok.x2;
            }
        }
        if (ncurr == 0) break;
        { TList t = curr; curr = prev; prev = t; }
        nprev = ncurr; rev = 0;
    }
    return ret;
}

// grid-stride: thread t handles reads t, t + T, ...
__global__ void __launch_bounds__(256) k_smem3(DIndex ix, DOpts o, int n_reads, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                                DIntv* __restrict__ intv_out, int32_t* __restrict__ n_intv, int32_t* __restrict__ status, u64* __restrict__ slab,
                                                DCounters* __restrict__ ctr) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    TList A, B;
    A.base = slab + t; A.stride = T;
    B.base = slab + (size_t)(LH_MAXLEN + 2) * 4 * T + t; B.stride = T;
    unsigned n_ext_total = 0;
    const int split_len = (int)(o.min_seed_len * o.split_factor + .499);
    for (size_t r = t; r < (size_t)n_reads; r += T) {
        i64 off = seq_off[r];
        int len = (int)(seq_off[r + 1] - off);
        int st = 0;
        if (len > LH_MAXLEN) { st |= LH_ST_TOO_LONG; len = 0; }
        const uint8_t* q = seq + off;
        S3Out so;
        so.out = intv_out + r * LH_MAX_INTV; so.n = 0; so.ovf = 0; so.n_ext = 0;
        if (len >= o.min_seed_len) {
            int x = 0;
            while (x < len) {   // first pass: all SMEMs
                if (q[x] < 4) x = lane_smem1(ix, len, q, x, 1, A, B, so, o.min_seed_len);
                else ++x;
            }
            int old_n = so.n;   // second pass: re-seed inside long, rare SMEMs
            for (int k = 0; k < old_n; ++k) {
                DIntv p = so.out[k];
                int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
                if (end - start < split_len || p.x2 > (u64)o.split_width) continue;
                lane_smem1(ix, len, q, (start + end) >> 1, (int)p.x2 + 1, A, B, so, o.min_seed_len);
            }
            if (o.max_mem_intv > 0) {   // third pass: LAST-like forward-only seeds (bwt_seed_strategy1)
                x = 0;
                while (x < len) {
                    if (q[x] > 3) { ++x; continue; }
                    DIntv ik = dev_set_intv(ix, q[x]);
                    int i, nx = len;
                    for (i = x + 1; i < len; ++i) {
                        int b = q[i];
                        if (b > 3) { nx = i + 1; break; }
                        DIntv ok = dev_extend_c(ix, ik, 3 - b, 0);
                        so.n_ext++;
                        if (ok.x2 < (u64)o.max_mem_intv && i - x >= o.min_seed_len) {
                            ok.info = (u64)x << 32 | (u64)(i + 1);
                            if (ok.x2 > 0) { if (so.n >= LH_MAX_INTV) so.ovf = 1; else so.out[so.n++] = ok; }
                            nx = i + 1;
                            break;
                        }
                        ik = ok;
                    }
                    x = nx;
                }
            }
        }
        if (so.ovf) st |= LH_ST_INTV_OVERFLOW;
        n_intv[r] = so.n; status[r] = st;
        n_ext_total += so.n_ext;
    }
    if (ctr) {
        unsigned tot = (unsigned)wave_sum_i32((int)n_ext_total);
        if (LANE() == 0 && tot) atomicAdd(&LH_CTR(ctr)->n_ext, (u64)tot);
    }
}

// sort each read's intervals by info (rank sort; equal keys are identical intervals), seed counts, l_rep.  16 lanes per read.
__global__ void __launch_bounds__(256) k_smem_fin(DOpts o, int n_reads, DIntv* __restrict__ intv, const int32_t* __restrict__ n_intv, int32_t* __restrict__ seed_cnt,
                                                   int32_t* __restrict__ l_rep_out) {
    int gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    int r = gid < n_reads ? gid : n_reads - 1;
    int live = gid < n_reads;
    DIntv* a = intv + (size_t)r * LH_MAX_INTV;
    int n = n_intv[r];
    DIntv mine[4];
    int rank[4];
    for (int t = 0; t < 4; ++t) {
        int e = sub + 16 * t;
        mine[t].x0 = mine[t].x1 = mine[t].x2 = 0; mine[t].info = ~0ull;
        if (e < n) mine[t] = a[e];
        rank[t] = 0;
    }
    for (int u = 0; u < n; ++u) {
        u64 oi = a[u].info;
        for (int t = 0; t < 4; ++t) { int e = sub + 16 * t; rank[t] += (oi < mine[t].info) || (oi == mine[t].info && u < e); }
    }
    __syncthreads();   // every lane holds its entries before any is overwritten
    int cnt = 0;
    for (int t = 0; t < 4; ++t) {
        int e = sub + 16 * t;
        if (e < n && live) {
            a[rank[t]] = mine[t];
            u64 s = mine[t].x2;
            u64 step = s > (u64)o.max_occ ? s / (u64)o.max_occ : 1;
            u64 c = (s + step - 1) / step;
            cnt += (int)(c < (u64)o.max_occ ? c : (u64)o.max_occ);
        }
    }
    cnt += (int)dpp_xor1((uint32_t)cnt); cnt += (int)dpp_xor2((uint32_t)cnt); cnt += (int)dpp_half_mirror((uint32_t)cnt); cnt += (int)dpp_ror8((uint32_t)cnt);
    __syncthreads();
    if (sub == 0 && live) {
        int b = 0, e = 0, l_rep = 0;
        for (int u = 0; u < n; ++u) {
            DIntv p = a[u];
            if (p.x2 <= (u64)o.max_occ) continue;
            int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
            if (sb > e) { l_rep += e - b; b = sb; e = se; }
            else e = e > se ? e : se;
        }
        l_rep += e - b;
        seed_cnt[r] = cnt; l_rep_out[r] = l_rep;
    }
}

// k_smem2.h — K1 v2: SMEM seeding as a step-synchronous state machine, FOUR reads per wavefront.
//
// v1 (k_smem.h) gives one read the whole wave; its forward extension is a single dependent chain, so 48 of 64 lanes
// repeat the same instructions and the kernel is VALU-issue bound (measured: ~450 issue cycles per bwt_extend).  Here each
// 16-lane row of the wave owns a read and runs BWA's mem_collect_intv (bwt_smem1a passes 1+2, bwt_seed_strategy1 pass 3;
// reached from go/src/gobwa/gobwa.go:244,253) as a small state machine.  Every iteration of the main loop performs at
// most ONE bwt_extend per row, and all four rows share the same instruction stream for it (coop_extend: lanes 0-7 decode
// the occurrence block of k, lanes 8-15 the block of k+size; DPP row reductions).  Bookkeeping between extends is
// row-divergent but short.  Rows pull their next read independently, so a slow read does not idle the other three.
//
// Output: unsorted intervals + count per read; k_smem_fin sorts by `info` and derives seed counts / l_rep (a kernel
// boundary instead of an in-kernel read-back of global stores).
#pragma once
#include "k_smem.h"

#define LH_S2_LIST 32   // interval list entries kept in LDS per row and list; deeper lists spill to the row's global slab

// Rows communicate through LDS written by one lane and read by the row's other lanes.  On the GPU the LDS queue of a wave
// is in order, so no wait is needed; the CPU emulator runs lanes one after another between rendezvous points, so there
// the row sits out one iteration (all lanes meet in coop_extend) before the data is consumed.
#ifdef LH_EMU
#define S2_SYNC_POINT() (waited = 1)
#else
#define S2_SYNC_POINT() ((void)0)
#endif

enum {
    S2_NEXT = 0, S2_WAITQ, S2_P1_HEAD, S2_FWD_CHECK, S2_FWD_EXT, S2_FWD_DONE, S2_BWD_BEGIN, S2_BWD_J, S2_BWD_EXT, S2_BWD_END, S2_SMEM_DONE,
    S2_P2_HEAD, S2_P3_HEAD, S2_P3_CHECK, S2_P3_EXT, S2_FINISH, S2_IDLE
};

__device__ __forceinline__ DIntv s2_get(const DIntv* lds, const volatile DIntv* gl, int j) {
    if (j < LH_S2_LIST) return lds[j];
    DIntv v;
    v.x0 = gl[j].x0; v.x1 = gl[j].x1; v.x2 = gl[j].x2; v.info = gl[j].info;
    return v;
}
__device__ __forceinline__ void s2_put(DIntv* lds, volatile DIntv* gl, int j, const DIntv& v) {
    if (j < LH_S2_LIST) lds[j] = v;
    else { gl[j].x0 = v.x0; gl[j].x1 = v.x1; gl[j].x2 = v.x2; gl[j].info = v.info; }
}

// grid = resident waves; every 16-lane row strides over the reads
__global__ void __launch_bounds__(64, 4) k_smem2(DIndex ix, DOpts o, int n_reads, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                                  DIntv* __restrict__ intv_out, int32_t* __restrict__ n_intv, int32_t* __restrict__ status,
                                                  DIntv* __restrict__ spill, DCounters* __restrict__ ctr) {
    __shared__ DIntv lds_a[4][LH_S2_LIST];
    __shared__ DIntv lds_b[4][LH_S2_LIST];
    __shared__ uint8_t qs[4][LH_MAXLEN + 6];
    __shared__ uint32_t p1_info[4][LH_MAX_INTV];   // pass-1 SMEMs for re-seeding: start<<16 | end
    __shared__ uint8_t p1_size[4][LH_MAX_INTV];    // min(size, 255)
    const int lane = LANE(), g = lane >> 4, sub = lane & 15;
    uint8_t* q = qs[g];
    DIntv* la = lds_a[g];
    DIntv* lb = lds_b[g];
    volatile DIntv* ga = spill + ((size_t)blockIdx.x * 4 + g) * 2 * (LH_MAXLEN + 2);
    volatile DIntv* gbl = ga + (LH_MAXLEN + 2);
    // row state (uniform inside the row)
    int state = S2_NEXT;
    int r_next = blockIdx.x * 4 + g, r_stride = gridDim.x * 4, r_cur = -1;
    int len = 0, nout = 0, ovf = 0, st = 0;
    int pass = 0, x = 0, i = 0, min_intv = 1, ncurr = 0, nprev = 0, j = 0, cc = 0, ret = 0, have_mem = 0, last_mem_start = 0, sx = 0;
    int old_n = 0, k2 = 0, swapped = 0;
    u64 last_size = 0;
    DIntv ik, pcur;
    ik.x0 = ik.x1 = ik.x2 = ik.info = 0; pcur = ik;
    DIntv* out = intv_out;
    unsigned n_ext = 0;
    const int split_len = (int)(o.min_seed_len * o.split_factor + .499);

#define S2_CURR_LDS (swapped ? lb : la)
#define S2_CURR_GL (swapped ? gbl : ga)
#define S2_PREV_LDS (swapped ? la : lb)
#define S2_PREV_GL (swapped ? ga : gbl)
#define S2_EMIT(m_)                                                                                        \
    {                                                                                                      \
        int slen_ = (int)(uint32_t)(m_).info - (int)((m_).info >> 32);                                     \
        if (slen_ >= o.min_seed_len) {                                                                     \
            if (nout >= LH_MAX_INTV) ovf = 1;                                                              \
            else {                                                                                         \
                if (sub == 0) {                                                                            \
                    out[nout] = (m_);                                                                      \
                    if (pass == 1) {                                                                       \
                        p1_info[g][nout] = (uint32_t)((m_).info >> 32) << 16 | (uint32_t)((m_).info & 0xffff); \
                        p1_size[g][nout] = (uint8_t)((m_).x2 > 255 ? 255 : (m_).x2);                        \
                    }                                                                                      \
                }                                                                                          \
                nout++;                                                                                    \
            }                                                                                              \
        }                                                                                                  \
    }

    for (;;) {
        // ---- advance every row through its extend-free transitions ----
        int waited = 0;
        while (state != S2_IDLE && state != S2_FWD_EXT && state != S2_BWD_EXT && state != S2_P3_EXT && !waited) {
            switch (state) {
                case S2_NEXT: {
                    if (r_next >= n_reads) { state = S2_IDLE; break; }
                    r_cur = r_next; r_next += r_stride;
                    i64 off = seq_off[r_cur];
                    len = (int)(seq_off[r_cur + 1] - off);
                    st = 0;
                    if (len > LH_MAXLEN) { st |= LH_ST_TOO_LONG; len = 0; }
                    for (int t = sub; t < len; t += 16) q[t] = seq[off + t];
                    out = intv_out + (size_t)r_cur * LH_MAX_INTV;
                    nout = 0; ovf = 0;
                    state = S2_WAITQ; S2_SYNC_POINT();   // the row's other lanes fill q
                    break;
                }
                case S2_WAITQ:
                    if (len < o.min_seed_len) state = S2_FINISH;
                    else { pass = 1; x = 0; state = S2_P1_HEAD; }
                    break;
                case S2_P1_HEAD:   // first pass: all SMEMs
                    while (x < len && q[x] > 3) ++x;
                    if (x >= len) { pass = 2; old_n = nout; k2 = 0; state = S2_P2_HEAD; }
                    else { sx = x; min_intv = 1; ik = dev_set_intv(ix, q[x]); ik.info = (u64)(x + 1); i = x + 1; ncurr = 0; swapped = 0; state = S2_FWD_CHECK; }
                    break;
                case S2_P2_HEAD: {   // second pass: re-seed inside long, rare SMEMs
                    int found = 0;
                    while (k2 < old_n) {
                        uint32_t inf = p1_info[g][k2];
                        int start = (int)(inf >> 16), end = (int)(inf & 0xffff), sz = p1_size[g][k2];
                        k2++;
                        if (end - start < split_len || sz > o.split_width) continue;
                        sx = (start + end) >> 1; min_intv = sz + 1; found = 1;
                        break;
                    }
                    if (!found) { pass = 3; x = 0; state = o.max_mem_intv > 0 ? S2_P3_HEAD : S2_FINISH; break; }
                    if (q[sx] > 3) break;   // bwt_smem1a returns immediately on an ambiguous base; stay in P2_HEAD
                    ik = dev_set_intv(ix, q[sx]); ik.info = (u64)(sx + 1); i = sx + 1; ncurr = 0; swapped = 0; state = S2_FWD_CHECK;
                    break;
                }
                case S2_FWD_CHECK:   // head of the forward loop: for (i = x+1; i < len; ++i)
                    if (i >= len) { if (sub == 0) s2_put(S2_CURR_LDS, S2_CURR_GL, ncurr, ik); ncurr++; state = S2_FWD_DONE; }
                    else if (q[i] > 3) { if (sub == 0) s2_put(S2_CURR_LDS, S2_CURR_GL, ncurr, ik); ncurr++; state = S2_FWD_DONE; }
                    else { cc = 3 - q[i]; state = S2_FWD_EXT; }
                    break;
                case S2_FWD_DONE: {   // reverse curr into prev (longest match first), then start the backward sweep
                    nprev = ncurr;
                    for (int t = sub; t < nprev; t += 16) s2_put(S2_PREV_LDS, S2_PREV_GL, t, s2_get(S2_CURR_LDS, S2_CURR_GL, nprev - 1 - t));
                    // ret = end of the longest forward match = info of the LAST pushed interval (known to every lane)
                    ret = (int)ik.info;
                    i = sx - 1; have_mem = 0; last_mem_start = 0;
                    state = S2_BWD_BEGIN; S2_SYNC_POINT();   // other lanes of the row finish the reversal before prev[] is read
                    break;
                }
                case S2_BWD_BEGIN:
                    cc = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1);
                    j = 0; ncurr = 0;
                    state = S2_BWD_J;
                    break;
                case S2_BWD_J:
                    if (j >= nprev) { state = S2_BWD_END; break; }
                    pcur = s2_get(S2_PREV_LDS, S2_PREV_GL, j);
                    if (cc >= 0) { state = S2_BWD_EXT; break; }
                    // cannot extend (start of read / ambiguous base): keep the hit if nothing longer survived this round
                    if (ncurr == 0 && (!have_mem || i + 1 < last_mem_start)) {
                        DIntv m = pcur;
                        m.info |= (u64)(i + 1) << 32;
                        S2_EMIT(m)
                        have_mem = 1; last_mem_start = i + 1;
                    }
                    j++;
                    break;
                case S2_BWD_END:
                    if (ncurr == 0) { state = S2_SMEM_DONE; break; }
                    swapped ^= 1; nprev = ncurr;
                    i--;
                    if (i < -1) { state = S2_SMEM_DONE; break; }
                    state = S2_BWD_BEGIN; S2_SYNC_POINT();   // curr[] (written by lane 0 of the row) becomes prev[] for all lanes
                    break;
                case S2_SMEM_DONE:
                    if (pass == 1) { x = ret; state = S2_P1_HEAD; }
                    else state = S2_P2_HEAD;
                    break;
                case S2_P3_HEAD:   // third pass: LAST-like forward-only seeds (bwt_seed_strategy1)
                    while (x < len && q[x] > 3) ++x;
                    if (x >= len) { state = S2_FINISH; break; }
                    ik = dev_set_intv(ix, q[x]); i = x + 1;
                    state = S2_P3_CHECK;
                    break;
                case S2_P3_CHECK:
                    if (i >= len) { x = len; state = S2_P3_HEAD; }
                    else if (q[i] > 3) { x = i + 1; state = S2_P3_HEAD; }
                    else { cc = 3 - q[i]; state = S2_P3_EXT; }
                    break;
                case S2_FINISH:
                    if (ovf) st |= LH_ST_INTV_OVERFLOW;
                    if (sub == 0) { n_intv[r_cur] = nout; status[r_cur] = st; }
                    state = S2_NEXT;
                    break;
                default: break;
            }
        }
        if (!__any(state != S2_IDLE)) break;
        // ---- one cooperative bwt_extend for every row that wants one ----
        int want = state == S2_FWD_EXT || state == S2_BWD_EXT || state == S2_P3_EXT;
        int back = state == S2_BWD_EXT;
        DIntv arg = back ? pcur : ik;
        if (!want) { arg.x0 = 1; arg.x1 = 1; arg.x2 = 1; }   // harmless in-range dummy for rows without a request
        DIntv ok = coop_extend(ix, arg, want ? cc : 0, back, lane);
        // ---- consume ----
        if (state == S2_FWD_EXT) {
            n_ext++;
            int done = 0;
            if (ok.x2 != ik.x2) {
                if (sub == 0) s2_put(S2_CURR_LDS, S2_CURR_GL, ncurr, ik);
                ncurr++;
                if (ok.x2 < (u64)min_intv) done = 1;   // too small to be extended further: ik stays the last pushed interval
            }
            if (done) state = S2_FWD_DONE;
            else { ik = ok; ik.info = (u64)(i + 1); i++; state = S2_FWD_CHECK; }
        } else if (state == S2_BWD_EXT) {
            n_ext++;
            if (ok.x2 < (u64)min_intv) {
                if (ncurr == 0 && (!have_mem || i + 1 < last_mem_start)) {
                    DIntv m = pcur;
                    m.info |= (u64)(i + 1) << 32;
                    S2_EMIT(m)
                    have_mem = 1; last_mem_start = i + 1;
                }
            } else if (ncurr == 0 || ok.x2 != last_size) {
                ok.info = pcur.info;
                if (sub == 0) s2_put(S2_CURR_LDS, S2_CURR_GL, ncurr, ok);
                ncurr++; last_size = ok.x2;
            }
            j++;
            state = S2_BWD_J;
        } else if (state == S2_P3_EXT) {
            n_ext++;
            if (ok.x2 < (u64)o.max_mem_intv && i - x >= o.min_seed_len) {
                DIntv m = ok;
                m.info = (u64)x << 32 | (u64)(i + 1);
                if (m.x2 > 0) {
                    int save = pass; pass = 3;
                    if (nout >= LH_MAX_INTV) ovf = 1;
                    else { if (sub == 0) out[nout] = m; nout++; }
                    pass = save;
                }
                x = i + 1; state = S2_P3_HEAD;
            } else { ik = ok; i++; state = S2_P3_CHECK; }
        }
    }
    if (ctr) {
        unsigned tot = (sub == 0) ? n_ext : 0;
        tot = (unsigned)wave_sum_i32((int)tot);
        if (lane == 0) atomicAdd(&ctr->n_ext, (u64)tot);
    }
#undef S2_CURR_LDS
#undef S2_CURR_GL
#undef S2_PREV_LDS
#undef S2_PREV_GL
#undef S2_EMIT
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
    cnt += dpp_xor1(cnt); cnt += dpp_xor2(cnt); cnt += dpp_half_mirror(cnt); cnt += dpp_ror8(cnt);
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

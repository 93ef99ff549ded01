// k_smem.h — K1: SMEM seeding, one wavefront per read.
//
// Replaces the seeding half of BWA's mem_chain (mem_collect_intv -> bwt_smem1a / bwt_seed_strategy1 / bwt_extend /
// bwt_2occ4) that lariat reaches through mem_align1_core (go/src/gobwa/gobwa.go:244,253).
//
// Mapping to the wavefront:
//   * forward extension is an inherently sequential chain of dependent occurrence-block reads; all lanes
//     follow it redundantly (same addresses -> one 64-B request per block);
//   * backward extension extends EVERY surviving interval by the same base: lane j owns interval j, so one
//     memory round trip advances up to 64 intervals (vs. one per interval on a CPU);
//   * the "keep if size differs from the last kept / emit MEM if nothing longer survived" rules of bwt_smem1a
//     become a ballot + prefix-popcount compaction, because interval sizes are monotone in j;
//   * interval lists live in LDS (64 entries per list); deeper lists (homopolymers) spill to a global slab.
#pragma once
#include "lh_dev.h"

#define LH_SMEM_LDS_ENTRIES 64

struct SmemWs {   // per-wave working storage
    DIntv* lds_a; DIntv* lds_b;      // LDS lists
    DIntv* gl_a; DIntv* gl_b;        // global overflow slabs (LH_MAXLEN+1 entries each)
};

__device__ __forceinline__ DIntv* smem_slot(DIntv* lds, DIntv* gl, int j) { return j < LH_SMEM_LDS_ENTRIES ? lds + j : gl + j; }

// Cooperative bwt_extend: 16 lanes per interval.  Lanes 0-7 of the group read the occurrence block of k = x[!is_back]-1,
// lanes 8-15 the block of k + size: lane w takes BWT word w (4 B) and, for w < 4, the running count of base w (8 B) —
// two coalesced loads per 64-B block instead of every lane decoding both blocks.  Packed per-word counts are summed with
// 3 xor-shuffles; the four sizes and ok[c] are then shared inside the group.  `ik` and `c` must be uniform within the
// 16-lane group; the result is too.  (~45 instructions per step instead of ~350 when every lane decoded two blocks.)
__device__ __forceinline__ DIntv coop_extend(const DIndex& ix, const DIntv& ik, int c, int is_back, int lane) {
    u64 xa = is_back ? ik.x0 : ik.x1;   // x[!is_back]
    u64 xb = is_back ? ik.x1 : ik.x0;   // x[is_back]
    int half = (lane >> 3) & 1, w = lane & 7, gb = lane & ~15;
    u64 k = half ? xa - 1 + ik.x2 : xa - 1;
    int none = (k == (u64)-1);
    u64 kk = none ? 0 : k - (k >= ix.primary);
    // straight-line code: both loads of the row are issued back to back (no branch around either)
    const uint32_t* blk = ix.bwt + ((kk >> 7) << 4);
    uint32_t word = blk[8 + w];
    u64 cntv = ((const u64*)blk)[w & 3];
    int nfull = (int)((kk & 127) >> 4);
    uint32_t pm = 0x55555555u & ~((1u << ((~(uint32_t)kk & 15) << 1)) - 1);
    uint32_t msk = w < nfull ? 0x55555555u : (w == nfull ? pm : 0u);
    uint32_t x = occ_word(word, msk);
    if (none) { x = 0; cntv = 0; }
    if (w >= 4) cntv = 0;
    x += dpp_xor1(x); x += dpp_xor2(x); x += dpp_half_mirror(x);   // sum over the 8 lanes of the half-row
    u64 cnt = cntv + ((x >> ((w & 3) << 3)) & 0xff);   // lanes w < 4: occ of base w up to k
    u64 other = dpp_ror8_u64(cnt);   // lane ^ 8 within the 16-lane row
    u64 tk = half ? other : cnt, tl = half ? cnt : other;
    u64 size = tl - tk;                                  // lanes w < 4: ok[w].x[2]
    // lanes 0-3 of the row are one quad holding tk[w], size[w]: share the four sizes by DPP, let lane c assemble ok[c],
    // then one round of (independent) ds_bpermute broadcasts it to the row
    u64 s1 = dpp_quad_bcast_u64<1>(size), s2 = dpp_quad_bcast_u64<2>(size), s3 = dpp_quad_bcast_u64<3>(size);
    u64 acc = xb + ((xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) ? 1 : 0);   // ok[3].x[is_back]
    u64 nbw = acc + (w < 3 ? s3 : 0) + (w < 2 ? s2 : 0) + (w < 1 ? s1 : 0);        // ok[w].x[is_back]
    u64 l2w = w == 0 ? ix.L2[0] : w == 1 ? ix.L2[1] : w == 2 ? ix.L2[2] : ix.L2[3];
    u64 naw = l2w + 1 + tk;                                                        // ok[w].x[!is_back]
    int src = gb + c;
    u64 na = shfl_u64(naw, src), nb = shfl_u64(nbw, src);
    DIntv o;
    o.x2 = shfl_u64(size, src);
    if (is_back) { o.x0 = na; o.x1 = nb; } else { o.x1 = na; o.x0 = nb; }
    o.info = 0;
    return o;
}

// state of the output list of one read
struct SmemOut {
    DIntv* out;     // global [LH_MAX_INTV]
    int n;          // intervals written (wave-uniform)
    int overflow;
    int n_ext;      // bwt_extend count (telemetry; uniform)
};

__device__ __forceinline__ void smem_emit(SmemOut& so, const DIntv& m, int min_seed_len, int lane) {
    int slen = (int)(uint32_t)m.info - (int)(m.info >> 32);
    if (slen < min_seed_len) return;
    if (so.n >= LH_MAX_INTV) { so.overflow = 1; return; }
    if (lane == 0) so.out[so.n] = m;
    so.n++;
}

// bwt_smem1a(bwt, len, q, x, min_intv, max_intv=0, mem, tmpvec).  Emits qualifying MEMs straight into `so`
// (their relative order is irrelevant: mem_collect_intv sorts by info and equal keys are identical intervals).
// Optionally records the emitted MEMs of THIS call into rec[] (pass 1 needs them for re-seeding); returns ret.
__device__ __forceinline__ int wave_smem1(const DIndex& ix, int len, const uint8_t* q, int x, int min_intv, SmemWs& ws, SmemOut& so,
                                          int min_seed_len, int lane) {
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    DIntv* clds = ws.lds_a; DIntv* cgl = ws.gl_a;   // curr
    DIntv* plds = ws.lds_b; DIntv* pgl = ws.gl_b;   // prev
    DIntv ik = dev_set_intv(ix, q[x]);
    ik.info = (u64)(x + 1);
    int ncurr = 0, i;
    // ---- forward search (uniform) ----
    for (i = x + 1; i < len; ++i) {
        if (q[i] < 4) {
            int c = 3 - q[i];
            DIntv ok = coop_extend(ix, ik, c, 0, lane);
            so.n_ext++;
            if (ok.x2 != ik.x2) {
                if (lane == 0) *smem_slot(clds, cgl, ncurr) = ik;
                ncurr++;
                if (ok.x2 < (u64)min_intv) break;
            }
            ik = ok; ik.info = (u64)(i + 1);
        } else {
            if (lane == 0) *smem_slot(clds, cgl, ncurr) = ik;
            ncurr++;
            break;
        }
    }
    if (i == len) { if (lane == 0) *smem_slot(clds, cgl, ncurr) = ik; ncurr++; }
    WAVE_SYNC();
    // reverse curr into prev (longest match first)
    int nprev = ncurr;
    for (int base = 0; base < nprev; base += 64) {
        int j = base + lane;
        if (j < nprev) *smem_slot(plds, pgl, j) = *smem_slot(clds, cgl, nprev - 1 - j);
    }
    WAVE_SYNC();
    int ret = (int)smem_slot(plds, pgl, 0)->info;
    // ---- backward search for MEMs: lane j owns prev[j] ----
    int have_mem = 0, last_mem_start = 0;
    for (i = x - 1; i >= -1; --i) {
        int c = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1);
        int nkept = 0;
        u64 carry_size = 0;     // ok.x2 of the last entry of the previous 64-block (uniform)
        int any_before = 0;     // an entry of an earlier block already survived
        int first_fails = 0;
        for (int base = 0; base < nprev; base += 4) {   // 4 intervals per round, 16 lanes each
            int j = base + (lane >> 4);
            int valid = j < nprev;
            DIntv p, ok;
            p.x0 = p.x1 = p.x2 = p.info = 0;
            if (valid) p = *smem_slot(plds, pgl, j);
            ok = p;
            int fail = 1;
            if (c >= 0) {   // wave-uniform: every group extends by the same base
                ok = coop_extend(ix, p, c, 1, lane);
                fail = !valid || ok.x2 < (u64)min_intv;
                so.n_ext += (nprev - base) < 4 ? (nprev - base) : 4;
            }
            if (base == 0) first_fails = __shfl(fail, 0);
            // survivors: !fail and (first survivor overall, or size differs from the immediate predecessor's)
            u64 prev_size = shfl_u64(ok.x2, (lane - 16) & 63);
            int prev_fail = __shfl(fail, (lane - 16) & 63);
            if (lane < 16) { prev_size = carry_size; prev_fail = any_before ? 0 : 1; }
            int keep = valid && !fail && (prev_fail || ok.x2 != prev_size);
            int rep = (lane & 15) == 0;
            u64 km = __ballot(keep && rep);
            if (keep && rep) {
                ok.info = p.info;
                *smem_slot(clds, cgl, nkept + lanes_below(km, lane)) = ok;
            }
            nkept += __popcll(km);
            u64 sm = __ballot(valid && !fail && rep);
            if (sm) any_before = 1;
            carry_size = shfl_u64(ok.x2, 48);
        }
        // entries that cannot be extended form a prefix; only entry 0 can be emitted (curr is still empty when it is visited)
        if (first_fails) {
            if (!have_mem || i + 1 < last_mem_start) {
                DIntv m = *smem_slot(plds, pgl, 0);
                m.info |= (u64)(i + 1) << 32;
                smem_emit(so, m, min_seed_len, lane);
                have_mem = 1; last_mem_start = i + 1;
            }
        }
        WAVE_SYNC();
        if (nkept == 0) break;
        DIntv* t;
        t = clds; clds = plds; plds = t;
        t = cgl; cgl = pgl; pgl = t;
        nprev = nkept;
    }
    return ret;
}

// bwt_seed_strategy1 (forward only; uniform)
__device__ __forceinline__ int wave_seed_strategy1(const DIndex& ix, int len, const uint8_t* q, int x, int min_len, int max_intv, DIntv* mem, SmemOut& so, int lane) {
    mem->x0 = mem->x1 = mem->x2 = mem->info = 0;
    if (q[x] > 3) return x + 1;
    DIntv ik = dev_set_intv(ix, q[x]);
    for (int i = x + 1; i < len; ++i) {
        if (q[i] < 4) {
            int c = 3 - q[i];
            DIntv ok = coop_extend(ix, ik, c, 0, lane);
            so.n_ext++;
            if (ok.x2 < (u64)max_intv && i - x >= min_len) {
                *mem = ok;
                mem->info = (u64)x << 32 | (u64)(i + 1);
                return i + 1;
            }
            ik = ok;
        } else return i + 1;
    }
    return len;
}

// K1.  grid = min(n_reads, resident waves); each wave strides over reads.
__global__ void __launch_bounds__(64, 8) k_smem(DIndex ix, DOpts o, int n_reads, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                              DIntv* __restrict__ intv_out, int32_t* __restrict__ n_intv, int32_t* __restrict__ seed_cnt,
                                              int32_t* __restrict__ l_rep_out, int32_t* __restrict__ status, DIntv* __restrict__ spill,
                                              DCounters* __restrict__ ctr) {
    __shared__ DIntv lds_a[LH_SMEM_LDS_ENTRIES];
    __shared__ DIntv lds_b[LH_SMEM_LDS_ENTRIES];
    __shared__ uint8_t q[LH_MAXLEN + 6];
    DIntv* srt = lds_a;   // the interval lists are idle while the output is sorted
    int lane = LANE();
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {   // persistent waves: spill slabs are per resident wave
    i64 off = seq_off[r];
    int len = (int)(seq_off[r + 1] - off);
    DIntv* out = intv_out + (size_t)r * LH_MAX_INTV;
    int st = 0;
    if (len > LH_MAXLEN) { st |= LH_ST_TOO_LONG; len = 0; }
    for (int i = lane; i < len; i += 64) q[i] = seq[off + i];
    WAVE_SYNC();
    SmemWs ws;
    ws.lds_a = lds_a; ws.lds_b = lds_b;
    ws.gl_a = spill + (size_t)blockIdx.x * 2 * (LH_MAXLEN + 2);
    ws.gl_b = ws.gl_a + (LH_MAXLEN + 2);
    SmemOut so;
    so.out = out; so.n = 0; so.overflow = 0; so.n_ext = 0;
    if (len >= o.min_seed_len) {
        // first pass: all SMEMs (mem_collect_intv)
        int x = 0;
        while (x < len) {
            if (q[x] < 4) x = wave_smem1(ix, len, q, x, 1, ws, so, o.min_seed_len, lane);
            else ++x;
        }
        // second pass: re-seed inside long, rare SMEMs
        int split_len = (int)(o.min_seed_len * o.split_factor + .499);
        int old_n = so.n;
        WAVE_SYNC();
        for (int k = 0; k < old_n; ++k) {
            DIntv p = out[k];
            int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
            if (end - start < split_len || p.x2 > (u64)o.split_width) continue;
            wave_smem1(ix, len, q, (start + end) >> 1, (int)p.x2 + 1, ws, so, o.min_seed_len, lane);
            WAVE_SYNC();
        }
        // third pass: LAST-like forward-only seeds
        if (o.max_mem_intv > 0) {
            x = 0;
            while (x < len) {
                if (q[x] < 4) {
                    DIntv m;
                    x = wave_seed_strategy1(ix, len, q, x, o.min_seed_len, o.max_mem_intv, &m, so, lane);
                    if (m.x2 > 0) smem_emit(so, m, 0, lane);
                } else ++x;
            }
        }
    }
    WAVE_SYNC();
    // sort by info (rank sort; equal keys are identical intervals so their order is irrelevant)
    int n = so.n;
    DIntv mine;
    mine.x0 = mine.x1 = mine.x2 = 0; mine.info = ~0ull;
    if (lane < n) mine = out[lane];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
        u64 oi = shfl_u64(mine.info, u);
        rank += (oi < mine.info) || (oi == mine.info && u < lane);
    }
    if (lane < n) srt[rank] = mine;
    WAVE_SYNC();
    if (lane < n) out[lane] = srt[lane];
    // seed counts (mem_chain's occurrence sampling) and l_rep (uniform over the sorted list)
    int cnt = 0;
    if (lane < n) {
        u64 s = srt[lane].x2;
        u64 step = s > (u64)o.max_occ ? s / (u64)o.max_occ : 1;
        u64 c = (s + step - 1) / step;
        cnt = (int)(c < (u64)o.max_occ ? c : (u64)o.max_occ);
    }
    int total = wave_sum_i32(cnt);
    int b = 0, e = 0, l_rep = 0;
    for (int u = 0; u < n; ++u) {
        DIntv p = srt[u];
        int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
        if (p.x2 <= (u64)o.max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    if (so.overflow) st |= LH_ST_INTV_OVERFLOW;
    if (lane == 0) {
        n_intv[r] = n; seed_cnt[r] = total; l_rep_out[r] = l_rep; status[r] = st;
        if (ctr) atomicAdd(&LH_CTR(ctr)->n_ext, (u64)so.n_ext);
    }
    WAVE_SYNC();
    }
}

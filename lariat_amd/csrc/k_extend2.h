// k_extend2.h — K4 v3: seed extension (BWA mem_chain2aln + ksw_extend2), a read's control flow on ONE LANE, its DPs in sorted rounds.
// Reached in the reference through mem_align1_core (go/src/gobwa/gobwa.go:244,253).
//
// Measured on MI355X (profiles/r01_*): the wave-per-read kernel (k_extend.h: lanes = query columns, one DP row per step)
// is VALU-issue bound at ~120 instructions per row for extensions that are ~35 columns wide.  ksw_extend2 is a short sequential
// program, so here every lane runs it for its own read exactly as written (row-major cells, f/h1 carried in registers):
//   - eh[] lives in LDS, one 32-bit word per query column: h (16 bits) | e (13 bits) | the column's query base (3 bits),
//     word j of lane L at ehl[j*64+L] (conflict-free for any per-lane j); one LDS read + one write per cell;
//   - three exact shortcuts keep the DP small (see ext_control): no DP for a provably ungapped extension, a provably
//     sufficient narrow band otherwise (in a circular 64-word window of eh[], whatever the query length), and the full band
//     in a window that follows the live interval ksw_extend2 maintains;
//   - the DPs of different reads run side by side only when they have the same shape: see "K4 as ROUNDS" below.
#pragma once
#include "k_extend.h"

#define EH_H(v) ((int)((v) & 0xffffu))
#define EH_E(v) ((int)(((v) >> 16) & 0x1fffu))
#define EH_Q(v) ((int)((v) >> 29))
#define EH_PACK(q, e, h) ((uint32_t)(q) << 29 | (uint32_t)(e) << 16 | (uint32_t)(h))

// ksw_extend2 for one lane.  Column j's query base is q[qoff + qstep*j]; row i's target base is tg.base(i).
// circ_mask = 63: the band is narrow (w <= LH_EXT_CIRC_MAX_W), row i only touches columns i-w .. i+w+2, so eh[] lives in a
// circular window of 64 words per lane whatever the query length; a column's first-row value (a closed form) and its query base
// are written when the window reaches it — exactly the value the full array would still hold there.  circ_mask = -1: eh[] as is.
#define LH_EXT_CIRC_MAX_W 30
__device__ __forceinline__ ExtRes lane_ksw_extend2(const DOpts& o, const uint8_t* q, uint32_t* ehl, int lane, int qoff, int qstep, int qlen, LaneTgt& tg, int tlen, int w,
                                                   int end_bonus, int zdrop, int h0, u64* cells, int circ_mask) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
#define EHW(j_) ehl[((j_) & circ_mask) * 64 + lane]
    int maxsc = a_ > 0 ? a_ : 0;   // max entry of mat (a, -b, -1)
    int max_ins = (int)((double)(qlen * maxsc + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * maxsc + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    // the first row: column j holds max(0, h0 - oe_ins - (j-1)*e_ins) (j >= 1), h0 (j = 0), and the read's nt4 byte
    int n_mat = 0;   // columns [0, n_mat) of eh[] have been written
#define EH_MATERIALIZE(upto_)                                                                                \
    for (int lim_ = (upto_) < qlen ? (upto_) : qlen; n_mat <= lim_; ++n_mat) {                               \
        int v_ = n_mat == 0 ? h0 : h0 - oe_ins - (n_mat - 1) * e_ins;                                        \
        v_ = v_ > 0 ? v_ : 0;                                                                                \
        int qv_ = n_mat < qlen ? (int)q[qoff + qstep * n_mat] : 4;                                           \
        EHW(n_mat) = EH_PACK(qv_ > 4 ? 4 : qv_, 0, v_);                                                      \
    }
    EH_MATERIALIZE(circ_mask >= 0 ? w + 2 : qlen)
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    u64 ncell = 0;
    for (int i = 0; i < tlen; ++i) {
        int f = 0, h1, m = 0, mj = -1;
        const uint32_t tbs = (uint32_t)tg.base(i) << 29;
        if (circ_mask >= 0) EH_MATERIALIZE(i + w + 2)
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        int j;
        uint32_t p = EHW(beg);
        // the cells of eh[beg .. end] that are not zero after this row (ksw_extend2 scans for them to shrink the interval): noted while they are written
        int fnz = end, lnz = -1;
        for (j = beg; j < end; ++j) {
            uint32_t pnext = EHW(j + 1);   // j + 1 <= qlen: fetched while this cell is computed
            int M = EH_H(p), e = EH_E(p);
            int sc = (int32_t)p < 0 ? -1 : ((p ^ tbs) < 0x20000000u ? a_ : -b_);   // the column's base sits in the word's top three bits: compared in place
            M = M ? M + sc : 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            int t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            EHW(j) = (p & 0xE0000000u) | (uint32_t)e << 16 | (uint32_t)h1;
            if (e | h1) { fnz = fnz < j ? fnz : j; lnz = j; }
            h1 = h;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
            p = pnext;
        }
        { uint32_t pe = EHW(end); EHW(end) = EH_PACK(EH_Q(pe), 0, h1); if (h1) lnz = end; }
        if (end > beg) ncell += (u64)(end - beg);
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        beg = fnz;                        // for (j = beg; j < end && eh[j] is zero; ++j); beg = j
        j = lnz >= beg ? lnz : beg - 1;   // for (j = end; j >= beg && eh[j] is zero; --j)
        end = j + 2 < qlen ? j + 2 : qlen;
    }
#undef EHW
#undef EH_MATERIALIZE
    if (cells) *cells += ncell;
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// ksw_extend2 with the FULL band in a circular 64-word window of eh[]: what a row really touches is the live interval
// [beg, end] that ksw_extend2 itself maintains (it shrinks both ends past zero cells after every row), not the band.  A seed
// that does not belong to an alignment — at human-genome scale every third read carries one: a 19..22-base chance match that
// mem_chain_flt keeps as the first shadowed chain — dies within a few rows and a few dozen columns, however long the query
// side is.  Exactness: (1) row 0 nominally sweeps min(qlen, w+1) columns, but once the first-row values have run out
// (column value 0, so M = 0 and e = 0), and the insertion score f carried along the row and the cell itself are 0, every
// further cell of the row is identically 0 and is left out; mj only moves on a cell >= the row maximum, which is > 0 or the
// row ends the extension; the `end == qlen` bookkeeping uses the nominal end.  (2) After row 0 every column outside the
// stored window holds h = e = 0 in the full array (beyond the first row's non-zero prefix; or shrunk past, which only
// happens over zero cells; a column is never written again once beg has passed it), so a column entering the window at the
// advancing end is materialised as zero + its query base.  (3) Column c and c - 64 share a slot: the live interval must stay
// narrower than 64 columns, else *overflow is set and the caller defers the read to the wave-per-read kernel.
__device__ __forceinline__ ExtRes lane_ksw_extend2_dyn(const DOpts& o, const uint8_t* q, uint32_t* ehl, int lane, int qoff, int qstep, int qlen, LaneTgt& tg, int tlen,
                                                       int w, int end_bonus, int zdrop, int h0, u64* cells, int* overflow) {
    const int a_ = o.a, b_ = o.b, o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
#define EHW(j_) ehl[((j_) & 63) * 64 + lane]
    int maxsc = a_ > 0 ? a_ : 0;
    int max_ins = (int)((double)(qlen * maxsc + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * maxsc + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    int n_mat = 0;   // columns [0, n_mat) have a slot holding their value; columns >= n_mat are zero (after row 0) and not stored yet
#define EH_MAT(upto_, row0_)                                                                                  \
    for (; n_mat <= (upto_); ++n_mat) {                                                                       \
        int v_ = 0;                                                                                           \
        if (row0_) { v_ = n_mat == 0 ? h0 : h0 - oe_ins - (n_mat - 1) * e_ins; v_ = v_ > 0 ? v_ : 0; }        \
        int qv_ = n_mat < qlen ? (int)q[qoff + qstep * n_mat] : 4;                                            \
        EHW(n_mat) = EH_PACK(qv_ > 4 ? 4 : qv_, 0, v_);                                                       \
    }
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    u64 ncell = 0;
    int over = 0;
    for (int i = 0; i < tlen; ++i) {
        int f = 0, h1, m = 0, mj = -1;
        const uint32_t tbs = (uint32_t)tg.base(i) << 29;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        int j, cut = 0;
        if (i == 0) { EH_MAT(1 < qlen ? 1 : qlen, 1) }
        else {
            if (end - beg > 63) { over = 1; break; }
            EH_MAT(end, 0)
        }
        uint32_t p = EHW(beg);
        int fnz = end, lnz = -1;   // the cells of eh[beg .. end] that are not zero after this row, noted while they are written
        for (j = beg; j < end; ++j) {
            if (i == 0 && n_mat <= j + 1) {   // row 0 materialises its columns as it goes (j + 1 <= qlen)
                if (j + 1 > 63) { over = 1; break; }
                EH_MAT(j + 1, 1)
            }
            uint32_t pnext = EHW(j + 1);
            int M = EH_H(p), e = EH_E(p);
            int sc = (int32_t)p < 0 ? -1 : ((p ^ tbs) < 0x20000000u ? a_ : -b_);
            const int m_in = M;
            M = M ? M + sc : 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            int t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            EHW(j) = (p & 0xE0000000u) | (uint32_t)e << 16 | (uint32_t)h1;
            if (e | h1) { fnz = fnz < j ? fnz : j; lnz = j; }
            h1 = h;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
            p = pnext;
            ++ncell;
            if (i == 0 && m_in == 0 && e == 0 && f == 0 && h == 0 && j >= 1) { cut = 1; ++j; break; }   // the rest of row 0 is identically zero
        }
        if (over) break;
        if (cut) {
            // cells [j, end) and eh[end] are zero; stored slots among them (at most column j, just materialised with a zero first-row value) already are
            h1 = 0;
        } else { uint32_t pe = EHW(end); EHW(end) = EH_PACK(EH_Q(pe), 0, h1); if (h1) lnz = end; }
        if (end == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        beg = fnz;                        // the scans of ksw_extend2 (see lane_ksw_extend2); cells a cut row 0 left out are zero
        j = lnz >= beg ? lnz : beg - 1;
        end = j + 2 < qlen ? j + 2 : qlen;
    }
#undef EHW
#undef EH_MAT
    if (cells) *cells += ncell;
    *overflow = over;
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// ---- K4 as ROUNDS over a job queue -----------------------------------------------------------------------------------------
// Measured (profiles/r02_*): with one lane running a read's whole mem_chain2aln, the lanes of a wave sit in different DPs of
// different widths — 14 of 64 lanes active, 74 % of the instructions in the DP cell loop.  The control flow of mem_chain2aln
// (chains -> seeds by score -> left / right extension -> band retry) is cheap; the DPs are what has to run side by side with
// DPs of the same shape.  So a read's program is RESUMABLE: it runs until it needs a ksw_extend2 call that is not provably the
// diagonal, saves where it is (ExtSt, 16 B; the region under construction in its output slot), and queues the call as a JOB with
// a key (kind of DP, band width, query columns).  The jobs of a round are counting-sorted by key; in the next round lane g runs
// job g's DP — beside 63 DPs of the same kind, width and length — and then the read's control flow up to its next job.  A read
// needs 0 .. 4 rounds on unique sequence; what is left after LH_EXT_ROUNDS, and live-interval windows that outgrow their 64
// columns, goes to the wave-per-read kernel as before.  Same calls, same arguments, same order per read as mem_chain2aln.
#ifndef LH_EXT_ROUNDS
#define LH_EXT_ROUNDS 6   // every round ends with the latency of its slowest DP (~0.2 ms): what is left after six goes to the wave kernel (tests also build with 2)
#endif
#define LH_EXT_JOB_BINS 1536
struct ExtSt { int32_t w0, narrow, sc0, w1; };   // w0: seed rank | side << 24 | band try << 25 | (left side used the doubled band) << 26; w1: chain | regions so far << 16 (a read in the rounds has fewer than 32,768 seeds)
struct DExtJobs {
    int32_t hist[LH_EXT_JOB_BINS], cursor[LH_EXT_JOB_BINS];
    int32_t count[LH_EXT_ROUNDS + 2];    // jobs queued for round k (k = 1 ..)
    int32_t range[2 * (LH_EXT_ROUNDS + 2)];   // [2k, 2k+1] = the sorted order's slice round k runs
    int32_t next[LH_EXT_ROUNDS + 2];          // round k's slice counter: its waves take slices of 64 jobs in the sorted order (heaviest bins first)
    int32_t kinds[3 * (LH_EXT_ROUNDS + 2)];   // diagnostics: narrow / live-interval / short full-band jobs of round k
    int32_t wnext;                            // (the long queue) k_ext_wround's next call
    int32_t wave_range[2], heavy_range[2], defer_range[2];   // [0, n): the reads the wave kernels chain and extend (k_chain_lane's list); the reads round 0 / the later rounds left to the wave extension kernel
};
// job key: bins 0..959 = narrow band in the circular window (same w: same cells per row), 960..1215 = full band in the live-interval window, 1216..1471 = full band, fewer than 64 columns
// bins in the order they should START (a round's tail is its last waves: the cheap ones): wide bands and long query sides first
__device__ __forceinline__ int lh_ext_job_key(int narrow, int qlen, int h0) {
    const int ql = 31 - (qlen >> 3 < 31 ? qlen >> 3 : 31), hh = 7 - (h0 < 16 ? 0 : h0 < 40 ? (h0 - 16) >> 2 : 6 + (h0 >= 80));   // h0 sets the width of a chance match's band: classes 16..19, 20..23, ..
    if (narrow && narrow <= LH_EXT_CIRC_MAX_W) return (LH_EXT_CIRC_MAX_W - narrow) * 32 + ql;
    if (qlen < 64) return 1216 + (31 - (qlen >> 1)) * 8 + hh;
    return 960 + ql * 8 + hh;
}
__global__ void __launch_bounds__(256) k_extj_count(const int32_t* __restrict__ n_jobs, const int32_t* __restrict__ key, DExtJobs* __restrict__ jb) {
    __shared__ int32_t hist[LH_EXT_JOB_BINS];
    const int n = *n_jobs;
    if ((int)blockIdx.x * 1024 >= n) return;
    for (int b = threadIdx.x; b < LH_EXT_JOB_BINS; b += 256) hist[b] = 0;
    __syncthreads();
    for (int blk = blockIdx.x; blk * 1024 < n; blk += gridDim.x)
        for (int u = 0; u < 4; ++u) {
            int i = (blk * 4 + u) * 256 + threadIdx.x;
            if (i < n) atomicAdd(&hist[key[i]], 1);
        }
    __syncthreads();
    for (int b = threadIdx.x; b < LH_EXT_JOB_BINS; b += 256) if (hist[b]) atomicAdd(&jb->hist[b], hist[b]);
}
__global__ void __launch_bounds__(256) k_extj_offsets(DExtJobs* __restrict__ jb, int round, int reuse) {   // exclusive scan of the histogram; clears it for the next round.  reuse >= 0 (the long queue alternates between two slots): slot `reuse` collects the next round's jobs, this round's slice counter starts at 0
    __shared__ int32_t part[256];
    const int t = threadIdx.x, per = LH_EXT_JOB_BINS / 256;
    int loc[per], s = 0;
    for (int u = 0; u < per; ++u) { loc[u] = s; s += jb->hist[t * per + u]; }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int o = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    const int excl = part[t] - s;
    for (int u = 0; u < per; ++u) { const int b = t * per + u, c = jb->hist[b]; if (c) atomicAdd(&jb->kinds[3 * round + (b < 960 ? 0 : b < 1216 ? 1 : 2)], c); }
    for (int u = 0; u < per; ++u) { jb->cursor[t * per + u] = excl + loc[u]; jb->hist[t * per + u] = 0; }
    if (t == 255) {
        jb->range[2 * round] = 0; jb->range[2 * round + 1] = part[255];
        if (reuse >= 0) { jb->count[reuse] = 0; jb->next[round] = 0; jb->count[3] = 0; jb->wnext = 0; }   // (count[3]: the calls listed for k_ext_wround, consumed by now)
    }
}
__global__ void __launch_bounds__(256) k_extj_scatter(const int32_t* __restrict__ n_jobs, const int32_t* __restrict__ key, const int32_t* __restrict__ list,
                                                       DExtJobs* __restrict__ jb, int32_t* __restrict__ order) {
    __shared__ int32_t hist[LH_EXT_JOB_BINS], basep[LH_EXT_JOB_BINS];
    const int n = *n_jobs;
    for (int blk = blockIdx.x; blk * 1024 < n; blk += gridDim.x) {
        __syncthreads();
        for (int b = threadIdx.x; b < LH_EXT_JOB_BINS; b += 256) hist[b] = 0;
        __syncthreads();
        int k[4], rank[4];
        for (int u = 0; u < 4; ++u) {
            int i = (blk * 4 + u) * 256 + threadIdx.x;
            k[u] = i < n ? key[i] : -1;
            rank[u] = k[u] >= 0 ? atomicAdd(&hist[k[u]], 1) : 0;
        }
        __syncthreads();
        for (int b = threadIdx.x; b < LH_EXT_JOB_BINS; b += 256) if (hist[b]) basep[b] = atomicAdd(&jb->cursor[b], hist[b]);
        __syncthreads();
        for (int u = 0; u < 4; ++u) {
            int i = (blk * 4 + u) * 256 + threadIdx.x;
            if (k[u] >= 0) order[basep[k[u]] + rank[u]] = list[i];
        }
    }
}

struct ExtArgs {   // what a read's control flow reads and writes
    const uint8_t* seq; const uint32_t* q4; const i64* seq_off; const i64* seed_off; const DChain* chains; const DSeed* cseeds; const int32_t* n_chains;
    const int32_t* sorder; int32_t* sdone; const i64* chain_rmax; const i64* reg_off; DReg* regs; int32_t* n_regs; ExtSt* est;
    // the long queue's units (one chain of a read each, indexed like the chain: seed_off[read] + chain): owner read, saved state, regions found; per read: "extend me from scratch"
    const int32_t* u_read; ExtSt* est_u; int32_t* nreg_u; int32_t* rflag;
};
// mem_chain2aln for read r, from the start (DP = false) or from its queued ksw_extend2 call (DP = true: the call is made here, in
// the lane's LDS window ehl), up to the next call that needs a DP.  Returns 0: the read is finished (n_regs written); 1: a call was
// queued (state saved, *key = its bin); 2: the read is left to the wave-per-read kernel (a live-interval window outgrew its 64 columns,
// or — first seed of the read only, as before — a long side whose diagonal loses LH_NARROW_MAX_LOSS or more, e.g. behind an indel:
// one full-band DP of that size keeps a lane busy for most of a millisecond).
// UNIT (the long queue, below): id names ONE chain of a read — the program runs that chain alone, its regions go to the chain's own slots
// (region slot = seed slot: a chain yields at most one region per seed), and the test against earlier regions only sees the chain's own.
// WAVE (UNIT only; k_ext_wround): all 64 lanes run the unit's program in step, and its pending call — one whose live interval outgrew a lane's
// 64-column window — is made by the whole wave (wave_ksw_extend2, k_extend.h: ehl = the read's bytes in LDS); the unit then goes back to the lanes' queue.
template <bool DP, bool UNIT = false, bool WAVE = false>
__device__ __forceinline__ int ext_control(const DIndex& ix, const DOpts& o, const ExtArgs& A, const int id, uint32_t* ehl, const int lane, int* key_out, u64* cells_out) {
    const int r = UNIT ? A.u_read[id] : id;
    const uint8_t* seq = A.seq; const uint32_t* q4 = A.q4; const i64* seq_off = A.seq_off; const i64* seed_off = A.seed_off; const DChain* chains = A.chains;
    const DSeed* cseeds = A.cseeds; const int32_t* n_chains = A.n_chains; const int32_t* sorder = A.sorder; int32_t* sdone = A.sdone; const i64* chain_rmax = A.chain_rmax;
    const i64* reg_off = A.reg_off; DReg* regs = A.regs; int32_t* n_regs = A.n_regs; ExtSt* est = A.est;
    int out = 0, key = 0;
    u64 cells = 0;
    {
        const i64 off = seq_off[r];
        int l_query = (int)(seq_off[r + 1] - off);
        if (l_query > LH_MAXLEN) l_query = 0;
        const uint8_t* q = seq + off;
        const i64 base = seed_off[r];
        const int ci0 = UNIT ? (int)(id - base) : 0;
        DReg* av = regs + reg_off[r] + (UNIT ? chains[id].seed_start : 0);
        const int nch = UNIT ? ci0 + 1 : n_chains[r];
        int ci = ci0, k = 0, n_av = 0, side = 0, tri = 0, narrow = 0, sc0 = 0, aw0 = o.w;
        bool resume = false;
        DReg a;
        a.rb = a.re = 0; a.qb = a.qe = 0; a.rid = 0; a.score = a.truesc = -1; a.sub = a.csub = 0; a.w = o.w; a.seedcov = 0; a.secondary = 0; a.seedlen0 = 0; a.n_comp = 0; a.is_alt = 0; a.frac_rep = 0;
        if (DP) {
            const ExtSt st = UNIT ? A.est_u[id] : est[r];
            k = st.w0 & 0xffff; ci = UNIT ? ci0 : st.w1 & 0xffff; n_av = (st.w1 >> 16) & 0xffff; side = (st.w0 >> 24) & 1; tri = (st.w0 >> 25) & 1; aw0 = o.w << ((st.w0 >> 26) & 1);
            narrow = st.narrow; sc0 = st.sc0;
            a = av[n_av];
            resume = true;
        }
        while (ci < nch && !out) {
            const DChain c = chains[base + ci];
            const DSeed* sd = cseeds + base + c.seed_start;
            const int32_t* srt = sorder + base + c.seed_start;
            int32_t* done = sdone + base + c.seed_start;
            const int n = c.n;
            if (n == 0) { ++ci; continue; }
            const i64 rmax0 = chain_rmax[2 * (base + ci)], rmax1 = chain_rmax[2 * (base + ci) + 1];
            if (!resume) k = n - 1;
            while (k >= 0 && !out) {
                const int si = srt[k];
                const DSeed s = sd[si];
                if (!resume) {
                    // test whether extension has been made before (any earlier region of this read "around" the seed)
                    int hit = 0;
                    for (int i = 0; i < n_av && !hit; ++i) {
                        DReg p = av[i];
                        if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;   // not fully contained
                        if (s.len - p.seedlen0 > .1 * l_query) continue;   // this seed may give a better alignment
                        int qd = s.qbeg - p.qb; i64 rd = s.rbeg - p.rb;
                        int max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                        int w = max_gap < p.w ? max_gap : p.w;
                        if (qd - rd < w && rd - qd < w) { hit = 1; break; }
                        qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                        max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                        w = max_gap < p.w ? max_gap : p.w;
                        if (qd - rd < w && rd - qd < w) { hit = 1; break; }
                    }
                    if (hit) {   // (almost) contained: extend only if an overlapping, already-extended seed of the chain lies on another diagonal
                        int other = 0;
                        for (int i = k + 1; i < n && !other; ++i) {
                            int ti = srt[i];
                            if (!done[ti]) continue;
                            DSeed t = sd[ti];
                            if (t.len < s.len * .95) continue;
                            if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) other = 1;
                            if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) other = 1;
                        }
                        if (!other) { done[si] = 0; --k; continue; }
                    }
                    a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = 0; a.seedcov = 0; a.secondary = 0; a.n_comp = 0; a.is_alt = 0;
                    a.w = o.w; a.score = a.truesc = -1; a.rid = c.rid;
                    aw0 = o.w; side = 0;
                }
                const bool first_seed = !DP && !UNIT && ci == 0 && n_av == 0 && k == n - 1;
                int aw1 = o.w;
                while (side < 2 && !out) {   // 0 = left (reversed query prefix vs reversed reference prefix), 1 = right
                    int qoff, qstep, qlen, tstep, tlen, bonus, h0;
                    const int qe = s.qbeg + s.len;
                    i64 tc0;
                    const i64 re = s.rbeg + s.len;
                    if (!resume) sc0 = a.score;
                    if (side == 0) {
                        if (!s.qbeg) { a.score = a.truesc = s.len * o.a; a.qb = 0; a.rb = s.rbeg; side = 1; continue; }
                        qoff = s.qbeg - 1; qstep = -1; qlen = s.qbeg; tc0 = s.rbeg - 1; tstep = -1; tlen = (int)(s.rbeg - rmax0); bonus = o.pen_clip5; h0 = s.len * o.a;
                    } else {
                        if (qe == l_query) { a.qe = l_query; a.re = s.rbeg + s.len; side = 2; continue; }
                        qoff = qe; qstep = 1; qlen = l_query - qe; tc0 = re; tstep = 1; tlen = (int)(rmax1 - re); bonus = o.pen_clip3; h0 = sc0;
                    }
                    ExtRes e;
                    e.score = -1; e.qle = e.tle = e.gtle = 0; e.gscore = -1; e.max_off = 0;
                    int aw = o.w;
                    if (!resume) {
                        // ksw_extend2 without the DP when it is provably the ungapped extension.  A cell off the diagonal is reached
                        // through at least one gap and at most min(i,j)+1 aligned pairs: H(i,j) <= h0 + (min(i,j)+1)*a - min(oe_ins,
                        // oe_del).  If the penalties lost along the diagonal (a+b per mismatch, a+1 per ambiguous base) stay below
                        // that gap cost over the whole query, every row's maximum is its diagonal cell, strictly: max / qle / tle
                        // come from the diagonal's running score (first maximum), max_off is 0, the band is not retried, and
                        // gscore = the diagonal's last value at gtle = qlen (all other cells of the last column are smaller).
                        // With the default scoring that is any extension with at most one mismatch: most of them.
                        // When the diagonal loses more than that, the DP is still run, but in a band that is provably wide enough:
                        // a cell more than B off the diagonal is reached through a gap of more than B bases, so it stays below
                        // h0 + (min(i,j)+1)*a - gap_cost(B+1); a cell inside the band whose value depends on such a cell went
                        // through a second gap as well.  If the whole diagonal loses less than gap_cost(B+1) and never drops to 0,
                        // all those cells are strictly below their row's diagonal cell (and below the diagonal's end in the last
                        // column): row maxima, their columns, gscore / gtle and max_off of ksw_extend2(w) and of ksw_extend2(B)
                        // are the same, zdrop (> the loss) cannot fire, the band is not retried.  The region keeps w = opt->w.
                        int proven = 0;
                        narrow = 0;
                        if (tlen >= qlen) {
                            const int thr = (o.o_ins + o.e_ins) < (o.o_del + o.e_del) ? (o.o_ins + o.e_ins) : (o.o_del + o.e_del);
                            const int p_cap = o.zdrop > 0 && o.zdrop < LH_NARROW_MAX_LOSS ? o.zdrop : LH_NARROW_MAX_LOSS;
                            const DiagScan ds = dev_diag_scan(ix, o, q, q4, off, qoff, qstep, qlen, tc0, tstep, h0, p_cap);
                            if (ds.done && ds.P < thr) {
                                e.score = ds.mx; e.qle = ds.mxk + 1; e.tle = ds.mxk + 1; e.gscore = ds.sc_run; e.gtle = qlen; e.max_off = 0;
                                a.score = e.score;
                                proven = 1;
                            } else if (ds.done) {
                                narrow = 1;
                                while (o.o_ins + o.e_ins * (narrow + 1) <= ds.P || o.o_del + o.e_del * (narrow + 1) <= ds.P) ++narrow;
                            } else if (first_seed && qlen >= LH_EXT_HEAVY_COLS) { out = 2; break; }
                        } else if (first_seed && qlen >= LH_EXT_HEAVY_COLS) { out = 2; break; }
                        if (!proven) {   // queue the call: the first try of MAX_BAND_TRY
                            tri = 0;
                            key = lh_ext_job_key(narrow, qlen, h0);
                            out = 1;
                            break;
                        }
                    } else {
                        resume = false;
                        if (DP) {
                            aw = o.w << tri;
                            const int circ = narrow && narrow <= LH_EXT_CIRC_MAX_W;   // a narrow band needs 64 words whatever qlen is
                            const int dyn = !circ && qlen >= 64;                       // full band, query side longer than the LDS window: live-interval window
                            LaneTgt tg;
                            tg.init(ix, tc0, tstep);
                            if (WAVE) {
                                const uint8_t* ql = (const uint8_t*)ehl;
                                const int band = narrow && narrow < aw ? narrow : aw;
                                u64 wc = 0;
                                if (qlen <= 64) e = wave_ksw_extend2<1>(ix, o, ql, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &wc);
                                else if (qlen <= 128) e = wave_ksw_extend2<2>(ix, o, ql, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &wc);
                                else e = wave_ksw_extend2<4>(ix, o, ql, qoff, qstep, qlen, tc0, tstep, tlen, band, bonus, o.zdrop, h0, lane, &wc);
                                cells += wc;
                            } else if (dyn) {
                                int over = 0;
                                e = lane_ksw_extend2_dyn(o, q, ehl, lane, qoff, qstep, qlen, tg, tlen, aw, bonus, o.zdrop, h0, &cells, &over);
                                if (over) { out = 2; break; }
                            } else
                                e = lane_ksw_extend2(o, q, ehl, lane, qoff, qstep, qlen, tg, tlen, narrow && narrow < aw ? narrow : aw, bonus, o.zdrop, h0, &cells, circ ? 63 : -1);
                            const int prev = a.score;   // try 0: the score the side started from; try 1: the first try's
                            a.score = e.score;
                            if (tri == 0 && !(a.score == prev || e.max_off < (aw >> 1) + (aw >> 2))) {   // MAX_BAND_TRY: once more with the doubled band
                                tri = 1;
                                key = lh_ext_job_key(0, qlen, h0);
                                out = 1;
                                break;
                            }
                        }
                    }
                    if (side == 0) {
                        aw0 = aw;
                        if (e.gscore <= 0 || e.gscore <= a.score - o.pen_clip5) { a.qb = s.qbeg - e.qle; a.rb = s.rbeg - e.tle; a.truesc = a.score; }   // local extension
                        else { a.qb = 0; a.rb = s.rbeg - e.gtle; a.truesc = e.gscore; }                                                                  // to-end extension
                    } else {
                        aw1 = aw;
                        if (e.gscore <= 0 || e.gscore <= a.score - o.pen_clip3) { a.qe = qe + e.qle; a.re = re + e.tle; a.truesc += a.score - sc0; }
                        else { a.qe = l_query; a.re = re + e.gtle; a.truesc += e.gscore - sc0; }
                    }
                    ++side;
                }
                if (out) break;
                int cov = 0;   // seedcov
                for (int i = 0; i < n; ++i) {
                    DSeed t = sd[i];
                    if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
                }
                a.seedcov = cov;
                a.w = aw0 > aw1 ? aw0 : aw1;
                a.seedlen0 = s.len;
                a.frac_rep = c.frac_rep;
                a.is_alt = c.is_alt;   // (mem_align1_core sets it from the contig after the dedup: the same flag)
                av[n_av++] = a;
                --k;
            }
            if (!out) ++ci;
        }
        // (UNIT, verdict 2 = the pending call's live interval outgrew the lane's window: the saved state still describes that call; the unit is listed for k_ext_wround)
        if (!out) { if (UNIT) A.nreg_u[id] = n_av; else n_regs[r] = n_av; }
        else if (out == 1) {
            ExtSt st;
            st.w0 = k | side << 24 | tri << 25 | (aw0 != o.w ? 1 : 0) << 26; st.narrow = narrow; st.sc0 = sc0; st.w1 = ci | n_av << 16;
            if (UNIT) A.est_u[id] = st; else est[r] = st;
            av[n_av] = a;
        } else cells = 0;   // the read is redone from scratch: its cells are counted there
    }
    *key_out = key;
    *cells_out += cells;
    return out;
}

// wave-wide: the lanes whose read queued a call (out == 1) / is left to the wave kernel (out == 2) append it to the lists
__device__ __forceinline__ void ext_append(int out, int r, int key, int lane, int32_t* next_count, int32_t* next_list, int32_t* next_key, int32_t* defer_count, int32_t* defer_list) {
    const u64 em = __ballot(out == 1);
    if (em) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(next_count, (int32_t)__popcll(em));
        basep = wave_readlane(basep, 0);
        if (out == 1) { const int p = basep + lanes_below(em, lane); next_list[p] = r; if (next_key) next_key[p] = key; }
    }
    const u64 dm = __ballot(out == 2);
    if (dm) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(defer_count, (int32_t)__popcll(dm));
        basep = wave_readlane(basep, 0);
        if (out == 2) defer_list[basep + lanes_below(dm, lane)] = r;
    }
}
// One round.  Lane g runs the queued ksw_extend2 call of read order[g], then the read's control flow up to its next call (ext_control).
// A read that queues a call appends itself to next_list (its bin in next_key); one that is left to the wave kernel, or queues a call
// in the last round (next_* = the deferred list), appends itself to the deferred list.  (Round 0 — every read up to its first call —
// runs at the end of k_chain_lane, k_chain.h.)
template <bool UNIT>
__global__ void __launch_bounds__(64) k_ext_round(DIndex ix, DOpts o, const int32_t* __restrict__ range, int32_t* __restrict__ slice_ctr, const int32_t* __restrict__ order, ExtArgs A,
                                                   int32_t* __restrict__ next_count, int32_t* __restrict__ next_list, int32_t* __restrict__ next_key,
                                                   int32_t* __restrict__ defer_count, int32_t* __restrict__ defer_list, DCounters* __restrict__ ctr) {
    __shared__ uint32_t ehl[64 * 64];
    const int lane = LANE();
    const int first = range[0], last = range[1];
    u64 cells = 0;
    for (;;) {   // persistent waves, slices handed out in order: the expensive bins start first, the cheap ones fill the tail
        int nb = 0;
        if (lane == 0) nb = atomicAdd(slice_ctr, 1);
        const int blk = wave_readlane(nb, 0);
        if (first + blk * 64 >= last) break;
        const int g = first + blk * 64 + lane;
        const int r = g < last ? order[g] : -1;
        int out = 0, key = 0;
        if (r >= 0) out = ext_control<true, UNIT>(ix, o, A, r, ehl, lane, &key, &cells);
        ext_append(out, r, key, lane, next_count, next_list, next_key, defer_count, defer_list);
    }
    if (ctr) {
        uint32_t lo = (uint32_t)cells;   // < 2^32 cells per lane
        u64 tot = (u64)(uint32_t)wave_sum_i32((int)(lo >> 16)) << 16;
        tot += (u64)(uint32_t)wave_sum_i32((int)(lo & 0xffff));
        if (lane == 0 && tot) atomicAdd(&LH_CTR(ctr)->ext_cells, tot);
    }
}


// ---- THE LONG QUEUE (new in r04): the reads the wave kernel chained — many seeds, many chains: reads on the copies of a repeat family, tens to
// hundreds of regions each — through rounds as well.  The wave-per-read kernel runs a read's ksw_extend2 calls one after the other with a
// wave-wide DP of ~100 lane-instructions per cell; in the rounds the calls of 64 different reads run side by side at ~25.  But
// mem_chain2aln is sequential per read — a seed is skipped when an earlier region of the read contains it — and a read with a hundred
// regions would need two hundred rounds (measured: 150 rounds of 60,000 jobs, a quarter of the chip, slower than the wave kernel).  That
// dependence is almost never there ACROSS chains: a region found at one copy of the repeat does not contain a seed at another copy.  So the
// unit of work is the CHAIN (ext_control<.., UNIT>): every chain of a read runs as if it were alone, its regions go to its own slots, and
// the test against earlier regions sees only the chain's own.  Afterwards k_ext_merge checks, per read, what the units assumed — no
// extended seed of a chain is contained in a region of an EARLIER chain (mem_chain2aln's own test, same arithmetic) — and moves the regions
// together in chain order: the result is then exactly what the sequential program produces (an extended seed's DP depends on the seed
// alone; a seed skipped inside its chain is skipped in the sequential program too, which sees those regions and more).  A read that fails
// the check, or holds a DP a lane cannot (ext_control's verdict 2), or is still queued when the rounds stop, is extended from scratch by
// k_extend, as are all listed reads when they are few (unique sequence: 0.2 % of the reads) and reads with a chain of more than
// LH_EXT_PREP_MAXN seeds (low-complexity sequence).  k_ext_prep does per chain what k_extend does before it extends (the reference
// window, the order of the seeds), one LANE per chain, and lists the units; k_ext_round0 runs every unit up to its first DP.
#ifndef LH_EXT_PREP_MAXN
#define LH_EXT_PREP_MAXN 64
#endif
__global__ void __launch_bounds__(64) k_ext_prep(DIndex ix, DOpts o, const int32_t* __restrict__ list, const int32_t* __restrict__ range, int min_long, ExtArgs A,
                                                  int32_t* __restrict__ srt_w, i64* __restrict__ rmax_w, int32_t* __restrict__ u_read_w, int32_t* __restrict__ long_list,
                                                  int32_t* __restrict__ long_count, int32_t* __restrict__ fb_list, int32_t* __restrict__ fb_count, int32_t* __restrict__ ulist,
                                                  int32_t* __restrict__ ucount, DCounters* __restrict__ ctr) {
    const int lane = LANE();
    const int first = range[0], last = range[1];
    if (last - first < min_long) {
        for (int i = first + (int)blockIdx.x * 64 + lane; i < last; i += (int)gridDim.x * 64) fb_list[i - first] = list[i];
        if (blockIdx.x == 0 && lane == 0) *fb_count = last - first;
        return;
    }
    const i64 l_pac = ix.l_pac;
    unsigned win = 0, nchs = 0;   // (< 2^32 window bases per lane)
    for (int it = first + (int)blockIdx.x; it < last; it += (int)gridDim.x) {
        const int r = list[it];
        const i64 base = A.seed_off[r];
        const int S = (int)(A.seed_off[r + 1] - base);
        const int nch = A.n_chains[r];
        int l_query = (int)(A.seq_off[r + 1] - A.seq_off[r]);
        if (l_query > LH_MAXLEN) l_query = 0;
        int big = 0;
        for (int ci = lane; ci < nch; ci += 64) big |= A.chains[base + ci].n > LH_EXT_PREP_MAXN;
        if (S >= 32768 || __any(big)) {
            if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = r;
            continue;
        }
        for (int cb = 0; cb < nch; cb += 64) {
            const int ci = cb + lane;
            int n = 0;
            if (ci < nch) {
                const DChain c = A.chains[base + ci];
                n = c.n;
                u_read_w[base + ci] = r;
                A.nreg_u[base + ci] = 0;
                if (n > 0) {
                    const DSeed* sd = A.cseeds + base + c.seed_start;
                    int32_t* srt = srt_w + base + c.seed_start;
                    int32_t* done = A.sdone + base + c.seed_start;
                    i64 r0 = l_pac << 1, r1 = 0;   // max possible span
                    for (int i = 0; i < n; ++i) {
                        const DSeed t = sd[i];
                        const i64 b = t.rbeg - (t.qbeg + dev_cal_max_gap(o, t.qbeg));
                        const i64 e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + dev_cal_max_gap(o, l_query - t.qbeg - t.len));
                        r0 = r0 < b ? r0 : b;
                        r1 = r1 > e ? r1 : e;
                    }
                    i64 rmax0 = r0 > 0 ? r0 : 0, rmax1 = r1 < l_pac << 1 ? r1 : l_pac << 1;
                    const DSeed s0 = sd[0];
                    if (rmax0 < l_pac && l_pac < rmax1) {   // crossing the forward-reverse boundary; then choose one side
                        if (s0.rbeg < l_pac) rmax1 = l_pac;
                        else rmax0 = l_pac;
                    }
                    dev_fetch_clamp(ix, &rmax0, s0.rbeg, &rmax1);
                    win += (unsigned)(rmax1 - rmax0);
                    rmax_w[2 * (base + ci)] = rmax0; rmax_w[2 * (base + ci) + 1] = rmax1;
                    for (int i = 0; i < n; ++i) {   // by seed score (= len) then index, ascending
                        const DSeed t = sd[i];
                        int rank = 0;
                        for (int u = 0; u < n; ++u) { const DSeed x = sd[u]; rank += (x.len < t.len) || (x.len == t.len && u < i); }
                        srt[rank] = i;
                        done[i] = 1;
                    }
                }
            }
            const u64 um = __ballot(n > 0);   // the chunk's units: one reservation per wave
            if (um) {
                int bp = 0;
                if (lane == 0) bp = atomicAdd(ucount, (int)__popcll(um));
                bp = wave_readlane(bp, 0);
                if (n > 0) ulist[bp + lanes_below(um, lane)] = (int)(base + ci);
            }
        }
        if (lane == 0) { nchs += (unsigned)nch; A.rflag[r] = 0; long_list[atomicAdd(long_count, 1)] = r; }
    }
    if (ctr) {
        u64 wtot = (u64)(uint32_t)wave_sum_i32((int)(win >> 16)) << 16;
        wtot += (u64)(uint32_t)wave_sum_i32((int)(win & 0xffff));
        const int ctot = wave_sum_i32((int)nchs);
        if (lane == 0 && (wtot || ctot)) { atomicAdd(&LH_CTR(ctr)->win_bases, wtot); atomicAdd(&LH_CTR(ctr)->n_chain_ext, (u64)ctot); }
    }
}
// every listed unit up to its first ksw_extend2 call that needs a DP (one lane per unit)
__global__ void __launch_bounds__(64) k_ext_round0(DIndex ix, DOpts o, const int32_t* __restrict__ list, const int32_t* __restrict__ count, ExtArgs A,
                                                    int32_t* __restrict__ next_count, int32_t* __restrict__ next_list, int32_t* __restrict__ next_key, DCounters* __restrict__ ctr) {
    const int lane = LANE();
    const int n = *count;
    u64 cells = 0;
    for (int g0 = (int)blockIdx.x * 64; g0 < n; g0 += (int)gridDim.x * 64) {
        const int g = g0 + lane;
        const int id = g < n ? list[g] : -1;
        int out = 0, key = 0;
        if (id >= 0) out = ext_control<false, true>(ix, o, A, id, nullptr, lane, &key, &cells);
        ext_append(out, id, key, lane, next_count, next_list, next_key, nullptr, nullptr);   // (a unit has no verdict 2: ext_control flags its read instead)
    }
    if (ctr) {
        uint32_t lo = (uint32_t)cells;
        u64 tot = (u64)(uint32_t)wave_sum_i32((int)(lo >> 16)) << 16;
        tot += (u64)(uint32_t)wave_sum_i32((int)(lo & 0xffff));
        if (lane == 0 && tot) atomicAdd(&LH_CTR(ctr)->ext_cells, tot);
    }
}
// the calls a lane could not hold, one WAVE per unit; the unit's next call goes back to the lanes' queue
__global__ void __launch_bounds__(64) k_ext_wround(DIndex ix, DOpts o, const int32_t* __restrict__ list, const int32_t* __restrict__ count, int32_t* __restrict__ wnext, ExtArgs A,
                                                    int32_t* __restrict__ next_count, int32_t* __restrict__ next_list, int32_t* __restrict__ next_key, DCounters* __restrict__ ctr) {
    __shared__ uint8_t q[LH_MAXLEN + 6];
    const int lane = LANE();
    const int n = *count;
    u64 cells = 0;
    for (;;) {   // (r05) calls handed out one at a time: a fixed share per wave left the kernel waiting for its unluckiest wave (a seventh of the resident waves busy on average)
        int it = 0;
        if (lane == 0) it = atomicAdd(wnext, 1);
        it = wave_readlane(it, 0);
        if (it >= n) break;
        const int id = list[it];
        const int r = A.u_read[id];
        const i64 off = A.seq_off[r];
        int l_query = (int)(A.seq_off[r + 1] - off);
        if (l_query > LH_MAXLEN) l_query = 0;
        WAVE_SYNC();
        for (int i = lane; i < l_query; i += 64) q[i] = A.seq[off + i];
        WAVE_SYNC();
        int key = 0;
        const int out = ext_control<true, true, true>(ix, o, A, id, (uint32_t*)q, lane, &key, &cells);
        ext_append(lane == 0 ? out : 0, id, key, lane, next_count, next_list, next_key, nullptr, nullptr);
    }
    if (lane == 0 && ctr && cells) atomicAdd(&LH_CTR(ctr)->ext_cells, cells);
}
// the units still queued when the rounds stop: their reads are extended from scratch
__global__ void __launch_bounds__(256) k_ext_flush(const int32_t* __restrict__ count, const int32_t* __restrict__ list, ExtArgs A) {
    const int n = *count;
    for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < n; g += (int)(gridDim.x * blockDim.x)) atomicOr(&A.rflag[A.u_read[list[g]]], 1);
}
// per read of the long list: the check of what its units assumed, then the regions moved together in chain order (or the read listed for k_extend)
#ifndef LH_EXT_MERGE_CAP
#define LH_EXT_MERGE_CAP 512   // regions of a read copied to LDS for the check (the rest is read from the scratch records; tests build with 6)
#endif
struct ExtMReg { i64 rb, re; int32_t qb, qe, seedlen0, w; };
__global__ void __launch_bounds__(64) k_ext_merge(DOpts o, const int32_t* __restrict__ long_list, const int32_t* __restrict__ long_count, ExtArgs A, DReg* __restrict__ tmp,
                                                   int32_t* __restrict__ defer_count, int32_t* __restrict__ defer_list, int32_t* __restrict__ why) {   // why[0..2] (diagnostics): reads handed on because a unit flagged them / too many regions / the check failed
    __shared__ ExtMReg P[LH_EXT_MERGE_CAP];
    const int lane = LANE();
    const int n_items = *long_count;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const int r = long_list[it];
        const i64 base = A.seed_off[r];
        const int nch = A.n_chains[r];
        int l_query = (int)(A.seq_off[r + 1] - A.seq_off[r]);
        if (l_query > LH_MAXLEN) l_query = 0;
        DReg* const av = A.regs + A.reg_off[r];
        DReg* const tv = tmp + A.reg_off[r];
        int bad = A.rflag[r] != 0, reason = 0;
        int total = 0;
        WAVE_SYNC();
        if (!bad) {
            for (int cb = 0; cb < nch; cb += 64) {   // the regions of every chain, in chain order, to LDS (what the test reads) and to the scratch (whole records)
                const int ci = cb + lane;
                int m = 0, ss = 0;
                if (ci < nch) { m = A.nreg_u[base + ci]; ss = A.chains[base + ci].seed_start; }
                const int incl = wave_scan_add_i32(m);
                const int off = total + incl - m;
                total += wave_readlane(incl, 63);
                // (r05) a read with more regions than the LDS copy holds — 1 or 2 reads of some 400 k-pair batches on repeat copies — used to be left to the wave-per-read
                // kernel, ONE wave extending its 600 chains one after the other: 70 ms at the end of K4 in the batches that have such a read.  The test below reads the
                // regions beyond the LDS copy from the scratch records instead
                for (int j = 0; j < m; ++j) {
                    const DReg p = av[ss + j];
                    tv[off + j] = p;
                    if (off + j < LH_EXT_MERGE_CAP) {
                        ExtMReg e;
                        e.rb = p.rb; e.re = p.re; e.qb = p.qb; e.qe = p.qe; e.seedlen0 = p.seedlen0; e.w = p.w;
                        P[off + j] = e;
                    }
                }
            }
        }
        WAVE_SYNC();
        if (!bad) {
            int hit = 0, before = 0;
            for (int cb = 0; cb < nch; cb += 64) {
                const int ci = cb + lane;
                int m = 0;
                if (ci < nch) m = A.nreg_u[base + ci];
                const int incl = wave_scan_add_i32(m);
                const int off = before + incl - m;   // the regions of the earlier chains: P[0, off)
                before += wave_readlane(incl, 63);
                if (ci < nch && m > 0 && off > 0) {
                    const DChain c = A.chains[base + ci];
                    const DSeed* sd = A.cseeds + base + c.seed_start;
                    const int32_t* done = A.sdone + base + c.seed_start;
                    for (int i = 0; i < c.n && !hit; ++i) {
                        if (!done[i]) continue;   // skipped inside its own chain: skipped by the sequential program as well
                        const DSeed s = sd[i];
                        for (int j = 0; j < off; ++j) {
                            ExtMReg p;
                            if (j < LH_EXT_MERGE_CAP) p = P[j];
                            else { const DReg& g = tv[j]; p.rb = g.rb; p.re = g.re; p.qb = g.qb; p.qe = g.qe; p.seedlen0 = g.seedlen0; p.w = g.w; }
                            if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;   // not fully contained
                            if (s.len - p.seedlen0 > .1 * l_query) continue;
                            int qd = s.qbeg - p.qb; i64 rd = s.rbeg - p.rb;
                            int max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                            int w = max_gap < p.w ? max_gap : p.w;
                            if (qd - rd < w && rd - qd < w) { hit = 1; break; }
                            qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                            max_gap = dev_cal_max_gap(o, qd < rd ? qd : (int)rd);
                            w = max_gap < p.w ? max_gap : p.w;
                            if (qd - rd < w && rd - qd < w) { hit = 1; break; }
                        }
                    }
                }
            }
            bad = __any(hit);   // (the sequential program would have looked at the chain's other seeds before deciding: the wave kernel does)
            if (bad) reason = 2;
        }
        if (bad) {
            if (lane == 0) { defer_list[atomicAdd(defer_count, 1)] = r; atomicAdd(&why[reason], 1); }
            continue;
        }
        WAVE_SYNC();
        for (int j = lane; j < total; j += 64) av[j] = tv[j];
        if (lane == 0) A.n_regs[r] = total;
    }
}

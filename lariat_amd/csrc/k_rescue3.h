// k_rescue3.h — (r06) K6: a window the size of its hit.  mem_matesw (gobwa.go:286-325 -> bwamem_pair.c) hands ksw_align2 a window of pes_high - pes_low + l_ms
// rows (lariat: ~830) for a mate of 150 bases; the hit, when there is one, is a diagonal stretch of ~150 rows of it.  k_resc_sw's forward pass ran every
// row: 4.2 M cells per pair on repeat families, at the packed-16 VALU issue rate (profiles/r06_valu_rate.log) — it could only get faster by running fewer.
// k_resc_cert decides, per job and BEFORE any DP, a row range [rlo, rlo + rn) of the window whose ksw_u8 result is provably the whole window's.
//
// The argument.  ksw_u8's H (the value the row maximum, te and qe are read from) is a maximum over alignment paths — a subset of the textbook paths (the lazy-F
// quirk drops some), each scored as the textbook scores it.  Running the same kernel on a sub-range of rows is the maximum over the admissible paths INSIDE
// that range.  So if every textbook path with a cell outside the range scores below a value S_in that some path inside reaches, every cell whose H is >= S_in
// has the same H in both runs and no cell outside reaches S_in: the first row of the maximum, its smallest column and the score are the same (the reverse
// pass — untouched — starts from them).  What is needed is an upper bound on "any path with a cell outside", from quantities cheaper than the DP:
//   * a path is diagonal pieces joined by gaps; a gap between pieces on diagonals d and d' costs at least 6 + |d - d'| (o >= 6, e >= 1), a piece on d at
//     most the best ungapped segment of d;
//   * a maximal run of l matches on a diagonal holds l - 4 exact 5-mer matches (none if l < 5).  With a = 1 and b >= 4 a segment over runs l_i .. l_j
//     scores at most sum(l_u) - 4 (j - i), so (segment - 6) <= sum over its runs with l_u >= 5 of (l_u - 4), minus 2: the pieces of a diagonal with h
//     5-mer hits are worth at most V(d) = max(0, h - 2) to a path that pays 6 of a gap for each of them.  The hit counts of ALL diagonals of a window
//     are one pass of its rows through a 1,024-entry table of the mate's 5-mers — no per-cell work;
//   * d0 = the diagonal with the most hits is scanned exactly: K0 = its best ungapped segment (a path inside: S_in >= K0) and X8 = the most that
//     disjoint segments of d0 are worth to a path that leaves d0 between them (each return costs two gaps, 12, and two diagonals of extension, 2:
//     8 of it charged here) — max over segment sets of sum - 8 (n - 1);
//   * a path with a cell more than w diagonals off d0 either never touches d0 — at most 6 + Vside, Vside = sum of V over d != d0 — or walks from d0
//     across w + 1 diagonals: at most X8 + Vside - (w + 1).
// With 6 + Vside < K0 and w = X8 - K0 + Vside both are below K0: rows [d0 - w, d0 + qlen + w) hold every cell within w diagonals of d0, and the job runs on
// them alone.  6 + Vside + V(d0) < min_seed_len: no path reaches min_seed_len, no region comes of the attempt, no DP at all.  Anything else — another strong
// diagonal, a low-complexity mate whose 5-mers hit everywhere, scoring outside the derivation's assumptions — runs the whole window as before.
// oracle/bwa_mem.cpp: rescue_probe_cert counts the same certificate (with exact per-diagonal values) on the oracle's job list and checks what it promises
// against the full DP's result: profiles/r06_rescue_probe_config4.log — 97.9 % of configs[4]'s attempts restricted, mean w 10, 25 % of the reference's cells.
// rescue_cells still counts the cells the REFERENCE evaluates (k_resc_apply: from the job's tlen); rescue_cells_exec counts what ran here.
#pragma once
#include "k_rescue2.h"

#define LH_RC_NDIAG (LH_RJ_TMAX + LH_MAXLEN + 8)
#define LH_RC_KEYS (LH_RJ_NB * 32)   // forward order: striping x row class (32 rows per class)
struct RescCertLds {
    uint32_t head[1024];                   // the mate's 5-mers: last column + 1 that starts one (0: none)
    uint32_t hist[LH_RC_NDIAG / 4 + 1];    // hits per diagonal, one byte each (a diagonal holds at most qlen - 4 < 256)
    uint8_t nxt[LH_MAXLEN + 6];            // column -> the previous column + 1 with the same 5-mer
    uint8_t q[LH_MAXLEN + 6];              // the query as aligned (reverse complement of the mate)
    uint8_t tgt[LH_RJ_TMAX + 16];
};
// whether the certificate's derivation covers the scoring (a = 1, mismatches cost at least 4, gaps at least 6 + length)
__device__ __forceinline__ int resc_cert_ok(const DOpts& o) {
    return o.a == 1 && o.b >= 4 && o.o_del >= 6 && o.o_ins >= 6 && o.e_del >= 1 && o.e_ins >= 1 && o.min_seed_len >= 8;
}

// one diagonal of the window exactly: cells (i0 + c, k0 + c), c < n, as match bits mw[] (bit c of word c / 64), matches +1, mismatches -b.  K: the best ungapped
// segment; e: the cell its FIRST maximum ends at; s: the start of the SHORTEST segment ending there with that sum (what ksw_align2's reverse pass reports: the
// cell after the last one at which the running sum was back at its floor); Y2: max over sets of TWO OR MORE disjoint segments of (sum - R (n - 1)), -2^20 if
// there is none.  Run by run: a run of r matches or z mismatches updates the recurrences in closed form (cell by cell they are
//   h = max(h + s, 0), best = max(best, h);   o2 = max(o2, max(best, v2) - R) + s [best, v2 as they were before the cell], v2 = max(v2, o2)).
__device__ __forceinline__ void resc_diag_scan(const u64* mw, int n, int b, int R, int* K_out, int* e_out, int* s_out, int* Y2_out) {
    int h = 0, best = 0, e = -1, zero_at = 0, s0 = 0, o2 = -(1 << 20), v2 = -(1 << 20);
    int pos = 0;
    while (pos < n) {
        const int left = n - pos, inw = 64 - (pos & 63);
        const int valid = left < inw ? left : inw;
        const u64 wv = mw[pos >> 6] >> (pos & 63);
        const int m12 = best > v2 ? best : v2;
        if (wv & 1) {
            int r = ~wv ? __ffsll((unsigned long long)~wv) - 1 : 64;
            r = r < valid ? r : valid;
            const int a2 = o2 > m12 - R ? o2 : m12 - R;
            o2 = a2 + r; v2 = v2 > o2 ? v2 : o2;
            h += r;
            pos += r;
            if (h > best) { best = h; e = pos - 1; s0 = zero_at; }
        } else {
            int z = wv ? __ffsll((unsigned long long)wv) - 1 : 64;
            z = z < valid ? z : valid;
            const int f = o2 - b * z, g = m12 - R - b;
            {   // (the first cell of the run may be the second segment of a set: the sets' best so far sees it)
                const int o1 = (o2 > m12 - R ? o2 : m12 - R) - b;
                v2 = v2 > o1 ? v2 : o1;
            }
            o2 = f > g ? f : g;
            h -= b * z;
            pos += z;
            if (h <= 0) { h = 0; zero_at = pos; }
        }
    }
    *K_out = best; *e_out = e; *s_out = s0; *Y2_out = v2;
}

// one wave per job: J.rlo, J.rn (rn = 0: no forward pass — rlo = -1: the result is written here, rlo = 0: there is none; rn = tlen: the whole window)
// weaken (lh_diag_rescue_sw only; 0 in the pipeline): each bit switches one term of the certificate off, so that the tests can show that the crafted case
// it guards against then comes out wrong — 1: the paths off d0 (6 + Vside < K0), 2: two or more pieces of d0 (Y2), 4: the distance condition, 8: the strip's width
__global__ void __launch_bounds__(64) k_resc_cert(DIndex ix, DOpts o, i64 n_jobs, RJob* __restrict__ jobs, const uint8_t* __restrict__ seq, int weaken) {
    __shared__ RescCertLds S;
    const int lane = LANE();
    const int ok = resc_cert_ok(o);
    for (int i = lane; i < 1024; i += 64) S.head[i] = 0;
    WAVE_SYNC();
    // a wave takes a contiguous share of the jobs: the jobs of a pair follow each other and share their mate, whose 5-mer table is then built once
    const i64 per = (n_jobs + gridDim.x - 1) / gridDim.x, j_lo = per * blockIdx.x, j_hi = j_lo + per < n_jobs ? j_lo + per : n_jobs;
    i64 tab_q0 = -1;
    int tab_qlen = 0;
    auto qcode = [&](int k) { return (uint32_t)S.q[k] | (uint32_t)S.q[k + 1] << 2 | (uint32_t)S.q[k + 2] << 4 | (uint32_t)S.q[k + 3] << 6 | (uint32_t)S.q[k + 4] << 8; };
    for (i64 j = j_lo; j < j_hi; ++j) {
        RJob& J = jobs[j];
        const int qlen = J.qlen, tlen = J.tlen;
        if (!ok || qlen < 5 || tlen < 5) { if (lane == 0) { J.rlo = 0; J.rn = (int16_t)tlen; } continue; }
        const i64 t0 = J.t0, q0 = J.q0;
        WAVE_SYNC();   // the previous job's arrays are no longer in use
        const int same_q = q0 == tab_q0 && qlen == tab_qlen;
        if (!same_q) {
            for (int k = lane; k + 5 <= tab_qlen; k += 64) S.head[qcode(k)] = 0;   // (the table back to empty: only the entries the previous mate set)
            WAVE_SYNC();
            for (int k = lane; k < qlen; k += 64) S.q[k] = (uint8_t)(3 - seq[q0 - k]);   // (jobs are only made for mates without an ambiguous base)
        }
        {   // the window's bases, four per byte of the 2-bit reference (on the reverse strand: the complement of the forward bases, read backwards)
            const int fwd = t0 < ix.l_pac;
            const i64 f0 = fwd ? t0 : (ix.l_pac << 1) - t0 - tlen;   // the forward positions [f0, f0 + tlen) hold the window
            const i64 b_lo = f0 >> 2, b_hi = (f0 + tlen - 1) >> 2;
            for (i64 b = b_lo + lane; b <= b_hi; b += 64) {
                const uint32_t byte = ix.pac[b];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const i64 fp = 4 * b + u;
                    const int base = (int)(byte >> ((3 - u) << 1)) & 3;
                    const i64 i = fwd ? fp - f0 : f0 + tlen - 1 - fp;
                    if (fp >= f0 && fp < f0 + tlen) S.tgt[i] = (uint8_t)(fwd ? base : 3 - base);
                }
            }
        }
        const int nd = tlen + qlen - 1, nw = (nd + 3) / 4;
        for (int w = lane; w < nw; w += 64) S.hist[w] = 0;
        WAVE_SYNC();
        if (!same_q) {   // the mate's 5-mers, chained per code
            for (int k = lane; k + 5 <= qlen; k += 64) S.nxt[k] = (uint8_t)atomicExch(&S.head[qcode(k)], (uint32_t)(k + 1));
            tab_q0 = q0; tab_qlen = qlen;
            WAVE_SYNC();
        }
        // every 5-mer of the window against them: one count per (row, column) pair of equal 5-mers, on the pair's diagonal.  A lane takes a run of consecutive rows
        // and rolls their code (one base read per row)
        int over = 0;
        {
            const int n5 = tlen - 4, per_lane = (n5 + 63) / 64;
            const int i_lo = lane * per_lane, i_hi = i_lo + per_lane < n5 ? i_lo + per_lane : n5;
            uint32_t code = 0;
            if (i_lo < i_hi) code = (uint32_t)S.tgt[i_lo] << 2 | (uint32_t)S.tgt[i_lo + 1] << 4 | (uint32_t)S.tgt[i_lo + 2] << 6 | (uint32_t)S.tgt[i_lo + 3] << 8;
            for (int i = i_lo; i < i_hi; ++i) {
                code = code >> 2 | (uint32_t)S.tgt[i + 4] << 8;
                int hd = (int)S.head[code], steps = 0;
                while (hd) {
                    const int k = hd - 1, di = i - k + (qlen - 1);
                    atomicAdd(&S.hist[di >> 2], 1u << (8 * (di & 3)));
                    hd = S.nxt[k];
                    if (++steps > 48) { over = 1; break; }   // a low-complexity mate: the whole window
                }
            }
        }
        over = __any(over);
        WAVE_SYNC();
        if (over) { if (lane == 0) { J.rlo = 0; J.rn = (int16_t)tlen; } continue; }
        // the diagonal with the most hits (the smallest such diagonal), and V summed over all diagonals
        int bestkey = 0, vsum = 0;
        for (int w = lane; w < nw; w += 64) {
            const uint32_t hw = S.hist[w];
            if (hw == 0) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int hcount = (int)((hw >> (8 * u)) & 0xffu);
                const int key = hcount << 12 | (4095 - (4 * w + u));
                bestkey = bestkey > key ? bestkey : key;
                vsum += hcount > 2 ? hcount - 2 : 0;
            }
        }
        bestkey = wave_max_i32(bestkey);
        const int h0 = bestkey >> 12, di0 = bestkey ? 4095 - (bestkey & 4095) : 0, d0 = di0 - (qlen - 1);
        const int vside = wave_readlane(wave_scan_add_i32(vsum), 63) - (h0 > 2 ? h0 - 2 : 0);
        // V(d) = (hits - 2)+ by distance from d0: whether the V within every distance D sums to less than D (a chain of pieces off d0 pays at least its farthest
        // piece's distance in gap extension).  The sum within D never exceeds Vside, so only distances up to Vside can fail; and from Vside >= 250 on no window
        // passes the first condition (6 + Vside < K0 <= 250), whatever this one says
        const uint8_t* hb = (const uint8_t*)S.hist;
        const int maxD = di0 > nd - 1 - di0 ? di0 : nd - 1 - di0;
        const int scanD = vside < maxD ? vside : maxD;
        int near_fail = 0;
        if (vside < 250) {
            int cum = 0;
            for (int base = 1; base <= scanD; base += 64) {
                const int D = base + lane;
                int v = 0;
                if (D <= scanD) {
                    if (di0 - D >= 0) { const int x = hb[di0 - D]; v += x > 2 ? x - 2 : 0; }
                    if (di0 + D < nd) { const int x = hb[di0 + D]; v += x > 2 ? x - 2 : 0; }
                }
                const int incl = wave_scan_add_i32(v) + cum;
                if (D <= scanD && incl >= D) near_fail = 1;
                cum = wave_readlane(incl, 63);
            }
            near_fail = __any(near_fail);
        }
        const int vall = vside + (h0 > 2 ? h0 - 2 : 0);
        const int minsc = o.min_seed_len * o.a;
        if (6 + vall < minsc) { if (lane == 0) { J.rlo = 0; J.rn = 0; } continue; }   // no path reaches min_seed_len
        // d0 exactly
        const int k0 = d0 < 0 ? -d0 : 0, i0 = d0 < 0 ? 0 : d0;
        int n = qlen - k0 < tlen - i0 ? qlen - k0 : tlen - i0;
        n = n < 0 ? 0 : n;
        u64 mw[(LH_MAXLEN + 63) / 64];
#pragma unroll
        for (int u = 0; u < (LH_MAXLEN + 63) / 64; ++u) {
            const int c = 64 * u + lane;
            mw[u] = __ballot(c < n && S.q[k0 + (c < n ? c : 0)] == S.tgt[i0 + (c < n ? c : 0)]);
        }
        int K0, e0, s0, Y2;
        resc_diag_scan(mw, n, o.b, 8, &K0, &e0, &s0, &Y2);
        if (K0 >= minsc && (6 + vside < K0 || (weaken & 1)) && (Y2 < K0 || (weaken & 2)) && (!near_fail || (weaken & 4))) {
            // class A: every path with a gap scores below K0 — the result is d0's best segment, no DP in either direction (rlo = -1: settled)
            if (lane == 0) {
                J.score = (int16_t)K0; J.te = (int16_t)(i0 + e0); J.qe = (int16_t)(k0 + e0); J.tb = (int16_t)(i0 + s0); J.qb = (int16_t)(k0 + s0);
                J.rows2 = (int16_t)(e0 - s0 + 1); J.rlo = -1; J.rn = 0;
            }
            continue;
        }
        const int X8 = K0 > Y2 ? K0 : Y2;
        int rlo = 0, rn = tlen;
        if (6 + vside < K0) {
            const int w = (weaken & 8) ? 0 : X8 - K0 + vside;
            int r0 = d0 - w, r1 = d0 + qlen + w;
            r0 = r0 < 0 ? 0 : r0;
            r1 = r1 > tlen ? tlen : r1;
            if (r1 > r0) { rlo = r0; rn = r1 - r0; }
        }
        if (lane == 0) { J.rlo = (int16_t)rlo; J.rn = (int16_t)rn; }
    }
}

// the forward passes in the order (striping, rows to run): a wave's eight jobs share their striping and last about equally long.  A block takes a contiguous
// share of the jobs; SCATTER = false: histogram (one atomic per block and key) and the cells that will run, SCATTER = true (after k_resc_offsets1): places.
__device__ __forceinline__ int resc_key1(const RJob& jb) { return ((jb.qlen + 15) / 16) * 32 + ((jb.rn + 31) >> 5 < 31 ? (jb.rn + 31) >> 5 : 31); }
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_resc_bucket1(i64 n_jobs, const RJob* __restrict__ jobs, int32_t* __restrict__ hist1, int32_t* __restrict__ bcur1, const int32_t* __restrict__ bstart1,
                                                       int32_t* __restrict__ order, DCounters* __restrict__ ctr) {
    __shared__ int32_t sh_n[LH_RC_KEYS], sh_base[LH_RC_KEYS];
    __shared__ unsigned long long sh_cells;
    for (int k = threadIdx.x; k < LH_RC_KEYS; k += blockDim.x) sh_n[k] = 0;
    if (threadIdx.x == 0) sh_cells = 0;
    __syncthreads();
    const i64 per = (n_jobs + gridDim.x - 1) / gridDim.x, j0 = per * blockIdx.x, j1 = j0 + per < n_jobs ? j0 + per : n_jobs;
    unsigned long long cells = 0;
    for (i64 j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const RJob& jb = jobs[j];
        if (jb.rn <= 0) continue;
        atomicAdd(&sh_n[resc_key1(jb)], 1);
        cells += (unsigned long long)(16 * ((jb.qlen + 15) / 16)) * (unsigned long long)jb.rn;
    }
    if (!SCATTER && cells) atomicAdd(&sh_cells, cells);
    __syncthreads();
    if (!SCATTER) {
        for (int k = threadIdx.x; k < LH_RC_KEYS; k += blockDim.x) if (sh_n[k]) atomicAdd(&hist1[k], sh_n[k]);
        if (threadIdx.x == 0 && ctr && sh_cells) atomicAdd(&LH_CTR(ctr)->rescue_cells_exec, (u64)sh_cells);
        return;
    }
    for (int k = threadIdx.x; k < LH_RC_KEYS; k += blockDim.x) { sh_base[k] = sh_n[k] ? bstart1[k] + atomicAdd(&bcur1[k], sh_n[k]) : 0; sh_n[k] = 0; }
    __syncthreads();
    for (i64 j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const RJob& jb = jobs[j];
        if (jb.rn <= 0) continue;
        const int key = resc_key1(jb);
        order[sh_base[key] + atomicAdd(&sh_n[key], 1)] = (int32_t)j;
    }
}
// starts of the keys in the order array: padded to a multiple of 8 where the striping changes (a wave's jobs share it), dense inside a striping, longest first;
// bstart[s] (what k_resc_sw reads) = the start of striping s
__global__ void k_resc_offsets1(RMeta* __restrict__ meta, const int32_t* __restrict__ hist1, int32_t* __restrict__ bstart1, int32_t* __restrict__ bcur1, i64* __restrict__ peek_host) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int s = 0; s < LH_RJ_NB; ++s) {
        meta->bstart[s] = acc;
        for (int c = 31; c >= 0; --c) { const int k = s * 32 + c; bstart1[k] = acc; bcur1[k] = 0; acc += hist1[k]; }
        acc = (acc + 7) & ~7;
    }
    meta->bstart[LH_RJ_NB] = acc;
    if (peek_host) peek_host[0] = acc;
}
// the reverse passes' share of rescue_cells_exec (their rows are known once they have run)
__global__ void __launch_bounds__(256) k_resc_cells2(i64 n_jobs, const RJob* __restrict__ jobs, int minsc, DCounters* __restrict__ ctr) {
    __shared__ unsigned long long sh_cells;
    if (threadIdx.x == 0) sh_cells = 0;
    __syncthreads();
    unsigned long long cells = 0;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n_jobs; j += (i64)gridDim.x * blockDim.x) {
        const RJob& jb = jobs[j];
        if (jb.score >= minsc && jb.te >= 0 && jb.qe >= 0 && jb.rlo >= 0) cells += (unsigned long long)(16 * ((jb.qe + 1 + 15) / 16)) * (unsigned long long)jb.rows2;   // (rlo < 0: settled without a pass)
    }
    if (cells) atomicAdd(&sh_cells, cells);
    __syncthreads();
    if (threadIdx.x == 0 && ctr && sh_cells) atomicAdd(&LH_CTR(ctr)->rescue_cells_exec, (u64)sh_cells);
}

// k_aln.h — K7: region -> alignment.  Per candidate: BWA mem_reg2aln (band inference, bwa_gen_cigar2 -> ksw_global2 with
// traceback, NM, clip ops; go/src/gobwa/gobwa.go:400-415,449-488), bns_fetch_seq (gobwa.go:50-80) and lariat's
// GetAlignments CIGAR walk (go/src/inference/lariat.go:1552-1704): matches / mismatches / indels / soft clips /
// mismatch loci / log_alignment_probability / the best-17 filter.
//
// Four kernels (r04).  k_aln_prep (a lane per read: placeholders, best scores, the read of every candidate slot) and k_aln_flat (one LANE PER
// CANDIDATE) settle every candidate that needs no DP — query and reference spans of equal length and an inferred band of 0, i.e.
// bwa_gen_cigar2's "no gap; no need to do DP" branch, or a diagonal that provably beats every gapped path: the great majority — and list the
// others; k_aln_grp runs FOUR of those per wave, the banded global alignment with traceback in a band of 7 that is proved sufficient, and
// hands what it cannot settle to k_aln (one WAVE per candidate, the band mem_reg2aln asks for).
// Measured: one wave per read for everything was a chain of ~10 dependent phases per candidate (11.7 ms per 2 M reads).
#pragma once
#include "k_rescue2.h"
#include "k_extend2.h"   // LaneTgt

#define LH_MAXT 704                       // reference bases staged per candidate (re - rb)
#define LH_ZSLAB (LH_MAXT * 256)          // direction bytes per resident wave
#define LH_ZLDS 8192                      // ... kept in LDS when the band is narrow enough (150 rows x 54 columns)

struct DCand {   // device-side result arrays (one entry per candidate unless noted)
    i64* cand_off;   // [n_reads+1] (input)
    int32_t* rid; i64* pos; i64* aend; i64* rb; i64* re; uint8_t* reversed; int32_t* score; int32_t* qb; int32_t* qe; int32_t* nm;
    int32_t* matches; int32_t* mismatches; int32_t* indels; int32_t* soft_clipped; int32_t* soft_clipped_length; uint8_t* in_filtered;
    int32_t* n_cigar; uint32_t* cigar;   // LH_MAX_CIGAR slots per candidate
    int32_t* n_mm; int32_t* mm_ref; int32_t* mm_read;   // LH_MAX_MM slots per candidate
    // a candidate with more mismatch loci than slots (a long noisy read: up to one per base) continues in a pool shared by the batch: locus
    // k >= LH_MAX_MM of candidate c is entry mm_xoff[c] + k - LH_MAX_MM of mm_xref / mm_xread (space for the rest of the read is taken
    // from *mm_xctr when slot LH_MAX_MM is first needed)
    int32_t* mm_xoff; int32_t* mm_xref; int32_t* mm_xread; int32_t* mm_xctr; int32_t mm_xcap;
    double* lap;
    int32_t* read_len;   // per candidate (psuedoCountAlignmentScore needs len(read_seq))
};

__device__ __forceinline__ int dev_infer_bw(int l1, int l2, int score, int a, int q, int r) {
    int w;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;   // to get equal alignment length, we need at least two gaps
    w = (int)(((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.));
    int d = l1 - l2; d = d < 0 ? -d : d;
    if (w < d) w = d;
    return w;
}

// lariat.go:599-624 scoreAlignment(aln, nil, 0) - improper  ==  log_alignment_probability (lariat.go:1691)
__device__ __forceinline__ double dev_single_score(int mismatches, int indels, int soft_clipped, int soft_clipped_length) {
    double score = 0.0;
    score += (double)(mismatches * -2 + indels * -3);
    if (soft_clipped > 0) {
        score -= 5.0 * (double)soft_clipped;
        score -= (double)soft_clipped_length * 0.5;
    }
    return score;
}

__global__ void __launch_bounds__(64) k_aln(DIndex ix, DOpts o, int n_reads, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                             const i64* __restrict__ reg_off, const DReg* __restrict__ regs, const int32_t* __restrict__ n_regs, DCand R, i64 cand_cap,
                                             uint8_t* __restrict__ zpool, int32_t* __restrict__ status, DCounters* __restrict__ ctr,
                                             const int32_t* __restrict__ slow_r, const int32_t* __restrict__ slow_ci, const int32_t* __restrict__ slow_count) {
    __shared__ uint8_t q[LH_MAXLEN + 6];
    __shared__ uint8_t tref[LH_MAXT];
    __shared__ uint32_t cg[LH_MAX_CIGAR + 4];
    __shared__ uint32_t cgo[LH_MAX_CIGAR + 4];
    __shared__ int32_t sh[8];
    __shared__ uint8_t zl[LH_ZLDS];   // the direction bytes of a narrow band: lane 0's traceback is a chain of dependent reads (from the HBM slab: ~1 us each)
    int lane = LANE();
    uint8_t* const zg = zpool + (size_t)blockIdx.x * LH_ZSLAB;
    u64 cells = 0;
    const int n_items = *slow_count;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int r = slow_r[item], ci = slow_ci[item];
        i64 off = seq_off[r];
        int l_query = (int)(seq_off[r + 1] - off);
        if (l_query > LH_MAXLEN) l_query = 0;
        WAVE_SYNC();
        for (int i = lane; i < l_query; i += 64) q[i] = seq[off + i];
        WAVE_SYNC();
        const DReg* av = regs + reg_off[r];
        int n = n_regs[r];
        i64 c0 = R.cand_off[r];
        int st = 0;
        int best = 0;   // (pool overflow and reads without regions were settled by k_aln_fast)
        for (int i = lane; i < n; i += 64) { int s = av[i].score; best = best > s ? best : s; }
        best = wave_max_i32(best);
        {
            DReg ar = av[ci];
            i64 c = c0 + ci;
            int qb = ar.qb, qe = ar.qe, lq = qe - qb;
            i64 rb = ar.rb, re = ar.re;
            int rlen = (int)(re - rb);
            int rev = rb >= ix.l_pac;
            // oriented views: reverse BOTH sequences on the reverse strand so that indels are left-aligned on the forward strand
            int qoff = rev ? qe - 1 : qb, qstep = rev ? -1 : 1;
            i64 t0 = rev ? re - 1 : rb;
            int tstep = rev ? -1 : 1;
            int valid = lq > 0 && rb < re && !(rb < ix.l_pac && re > ix.l_pac) && rlen <= LH_MAXT;
            WAVE_SYNC();
            if (valid)
                for (int i = lane; i < rlen; i += 64) tref[i] = (uint8_t)dev_ref_base(ix, t0 + (i64)tstep * i);
            WAVE_SYNC();
            // mem_reg2aln: band inference and up to 3 global alignments with doubling band
            int tmp = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_del, o.e_del);
            int w2 = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_ins, o.e_ins);
            w2 = w2 > tmp ? w2 : tmp;
            if (w2 > o.w) w2 = w2 < ar.w ? w2 : ar.w;
            int score = 0, last_sc = -(1 << 30), n_cigar = 0, NM = -1, it = 0;
            int overflow = 0;
            do {
                w2 = w2 < o.w << 2 ? w2 : o.w << 2;
                n_cigar = 0; NM = -1;
                if (valid) {
                    if (lq == rlen && w2 == 0) {   // no gap; no need to do DP
                        int sc = 0;
                        for (int i = lane; i < lq; i += 64) {
                            int tb = tref[i], qv = q[qoff + qstep * i];
                            sc += (tb > 3 || qv > 3) ? -1 : (tb == qv ? o.a : -o.b);
                        }
                        score = wave_sum_i32(sc);
                        if (lane == 0) cg[0] = (uint32_t)lq << 4 | 0;
                        n_cigar = 1;
                        WAVE_SYNC();
                    } else {
                        int max_ins = (int)((double)(((lq + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
                        int max_del = (int)((double)(((lq + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
                        int max_gap = max_ins > max_del ? max_ins : max_del;
                        max_gap = max_gap > 1 ? max_gap : 1;
                        int dl = rlen - lq; dl = dl < 0 ? -dl : dl;
                        int w = (max_gap + dl + 1) >> 1;
                        w = w < w2 ? w : w2;
                        int min_w = dl + 3;
                        w = w > min_w ? w : min_w;
                        const int n_col_ = lq < 2 * w + 1 ? lq : 2 * w + 1;
                        uint8_t* const z = rlen * n_col_ <= LH_ZLDS ? zl : zg;
                        score = 2 * w + 1 <= 64 ? wave_ksw_global2_band(ix, o, q, qoff, qstep, lq, t0, tstep, rlen, w, z, lane, &cells)
                                               : wave_ksw_global2(ix, o, q, qoff, qstep, lq, t0, tstep, rlen, w, z, lane, &cells);
                        WAVE_SYNC();
                        if (lane == 0) {   // backtrack
                            const int n_col = n_col_;
                            int which = 0, nc = 0, ovf = 0;
                            int i = rlen - 1, k = (i + w + 1 < lq ? i + w + 1 : lq) - 1;
                            // ops are produced last-to-first; push_cigar merges equal neighbours
                            while (i >= 0 && k >= 0) {
                                which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
                                int op, len = 1;
                                if (which == 0) { op = 0; --i; --k; }
                                else if (which == 1) { op = 2; --i; }
                                else { op = 1; --k; }
                                if (nc == 0 || op != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_MAX_CIGAR) cg[nc++] = (uint32_t)len << 4 | op; else ovf = 1; }
                                else cg[nc - 1] += (uint32_t)len << 4;
                            }
                            if (i >= 0) { if (nc == 0 || 2 != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_MAX_CIGAR) cg[nc++] = (uint32_t)(i + 1) << 4 | 2; else ovf = 1; } else cg[nc - 1] += (uint32_t)(i + 1) << 4; }
                            if (k >= 0) { if (nc == 0 || 1 != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_MAX_CIGAR) cg[nc++] = (uint32_t)(k + 1) << 4 | 1; else ovf = 1; } else cg[nc - 1] += (uint32_t)(k + 1) << 4; }
                            for (int u = 0; u < nc >> 1; ++u) { uint32_t t = cg[u]; cg[u] = cg[nc - 1 - u]; cg[nc - 1 - u] = t; }   // reverse CIGAR
                            sh[0] = nc; sh[1] = ovf;
                        }
                        WAVE_SYNC();
                        n_cigar = sh[0]; overflow |= sh[1];
                    }
                    // NM: mismatches inside M runs (lane-parallel compare + ballot) + inserted + deleted bases (terminal D excluded)
                    {
                        int x = 0, y = 0, n_mm = 0, n_gap = 0;
                        for (int k = 0; k < n_cigar; ++k) {
                            int op = cg[k] & 0xf, len = (int)(cg[k] >> 4);
                            if (op == 0) {
                                for (int t0 = 0; t0 < len; t0 += 64) {
                                    int t = t0 + lane;
                                    int mm = t < len && q[qoff + qstep * (x + t)] != tref[y + t];
                                    n_mm += __popcll(__ballot(mm));
                                }
                                x += len; y += len;
                            } else if (op == 2) {
                                if (k > 0 && k < n_cigar - 1) n_gap += len;
                                y += len;
                            } else if (op == 1) { x += len; n_gap += len; }
                        }
                        NM = n_mm + n_gap;
                    }
                }
                if (score == last_sc || w2 == o.w << 2) break;   // it is possible that global alignment and local alignment give different scores
                last_sc = score;
                w2 <<= 1;
            } while (++it < 3 && score < ar.truesc - o.a);
            // position and clipping (lane 0 builds the final CIGAR in LDS), then lariat's CIGAR walk with lanes over the bases of each M run
            int is_rev;
            i64 posf = dev_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
            WAVE_SYNC();
            if (lane == 0) {
                int nc = n_cigar, s0 = 0, no = 0, ovf = 0;
                if (nc > 0) {   // squeeze out leading or trailing deletions (pos is shifted upstream, but lariat ignores that pos)
                    if ((cg[0] & 0xf) == 2) { s0 = 1; nc--; }
                    else if ((cg[nc - 1] & 0xf) == 2) nc--;
                }
                int clip5 = 0, clip3 = 0;
                if (qb != 0 || qe != l_query) { clip5 = is_rev ? l_query - qe : qb; clip3 = is_rev ? qb : l_query - qe; }
                if (clip5) cgo[no++] = (uint32_t)clip5 << 4 | 3;
                for (int u = 0; u < nc; ++u) { if (no < LH_MAX_CIGAR) cgo[no++] = cg[s0 + u]; else ovf = 1; }
                if (clip3) { if (no < LH_MAX_CIGAR) cgo[no++] = (uint32_t)clip3 << 4 | 3; else ovf = 1; }
                sh[3] = no; sh[4] = ovf;
            }
            WAVE_SYNC();
            int no = sh[3];
            overflow |= sh[4];
            i64 coff = ix.contig_off[ar.rid];
            // InterpretAlign (gobwa.go:339-371)
            i64 Offset = rb < ix.l_pac ? rb - coff : ix.l_pac * 2 - 1 - rb - coff;
            i64 End = re < ix.l_pac ? re - coff : ix.l_pac * 2 - 1 - re - coff;
            i64 refStart = Offset, refEnd = End;
            if (is_rev) { refStart = End + 1; refEnd = Offset + 1; }
            // CIGAR walk in READ orientation (lariat.go:1591-1632).  refSeq[k] == base at fwd||rev coordinate rb + k.
            int matches = 0, indels = 0, indel_length = 0, soft_clipping = 0, soft_clipping_length = 0, refSeqOffset = 0, readOffset = 0, nmm = 0, mm_ovf = 0, xo = -2;
            int32_t* mref = R.mm_ref + (size_t)c * LH_MAX_MM;
            int32_t* mread = R.mm_read + (size_t)c * LH_MAX_MM;
            int refLen = (int)(refEnd - refStart);
            for (int u = 0; u < no; ++u) {
                uint32_t cv = cgo[is_rev ? no - 1 - u : u];
                int op = cv & 0xf, len = (int)(cv >> 4);
                if (op == 0) {
                    matches += len;
                    for (int t0 = 0; t0 < len; t0 += 64) {
                        int t = t0 + lane, k = refSeqOffset + t;
                        int mm = 0;
                        if (t < len && k < refLen && readOffset + t < l_query) {
                            int rbase = (valid && k < rlen) ? tref[rev ? rlen - 1 - k : k] : 255;
                            mm = rbase != q[readOffset + t];
                        }
                        u64 bm = __ballot(mm);
                        if (nmm + (int)__popcll(bm) > LH_MAX_MM && xo == -2) {   // the slots run out here: the rest of the loci go to the pool
                            int v = 0;
                            if (lane == 0) {
                                v = atomicAdd(R.mm_xctr, l_query - LH_MAX_MM);
                                if (v + l_query - LH_MAX_MM > R.mm_xcap) v = -1;
                                R.mm_xoff[c] = v;
                            }
                            xo = wave_readlane(v, 0);
                        }
                        if (mm) {
                            int slot = nmm + lanes_below(bm, lane);
                            if (slot < LH_MAX_MM) { mref[slot] = is_rev ? (int)refEnd - k : k + (int)refStart; mread[slot] = readOffset + t; }
                            else if (xo >= 0) { R.mm_xref[xo + slot - LH_MAX_MM] = is_rev ? (int)refEnd - k : k + (int)refStart; R.mm_xread[xo + slot - LH_MAX_MM] = readOffset + t; }
                        }
                        nmm += __popcll(bm);
                    }
                    refSeqOffset += len; readOffset += len;
                } else if (op == 1) { indels += 1; indel_length += len; readOffset += len; }
                else if (op == 2) { indels += 1; indel_length += len; refSeqOffset += len; }
                else if (op == 3) { soft_clipping += 1; soft_clipping_length += len; readOffset += len; }
            }
            if (nmm > LH_MAX_MM && xo < 0) { mm_ovf = 1; nmm = LH_MAX_MM; }   // the pool is full as well
            if (lane < no) R.cigar[(size_t)c * LH_MAX_CIGAR + lane] = cgo[lane];
            if (lane == 0) {
                int rid = dev_pos2rid(ix, posf);
                int mismatches = NM - indel_length;
                matches -= mismatches;
                if (mismatches < 0) mismatches = 0;
                i64 pos = Offset, aend = End;
                if (pos != -1 && is_rev) { pos = End + 1; aend = Offset + 1; }
                R.rid[c] = rid; R.pos[c] = pos; R.aend[c] = aend; R.rb[c] = rb; R.re[c] = re; R.reversed[c] = (uint8_t)is_rev; R.score[c] = ar.score;
                R.qb[c] = qb; R.qe[c] = qe; R.nm[c] = NM; R.matches[c] = matches; R.mismatches[c] = mismatches; R.indels[c] = indels;
                R.soft_clipped[c] = soft_clipping; R.soft_clipped_length[c] = soft_clipping_length;
                R.in_filtered[c] = ar.score >= best - o.aln_score_delta;
                R.n_cigar[c] = no; R.n_mm[c] = nmm; R.read_len[c] = l_query;
                R.lap[c] = (dev_single_score(mismatches, indels, soft_clipping, soft_clipping_length) + o.improper_pair_penalty) - o.improper_pair_penalty;
                if (overflow) st |= LH_ST_CIGAR_OVERFLOW;
                if (mm_ovf) st |= LH_ST_MM_OVERFLOW;
                if (!valid) st |= LH_ST_TOO_LONG;
            }
        }
        if (lane == 0 && st) atomicOr(&status[r], st);   // other waves may hold other candidates of the same read
    }
    if (lane == 0 && ctr && cells) atomicAdd(&LH_CTR(ctr)->glob_cells, cells);
}

// ---- (r06) the second look at a candidate with equal spans and FIVE OR SIX mismatches (a third of what k_aln_flat lists on repeat families).  As in aln_fast_cand the diagonal wins when no path with gaps reaches it at any diagonal
// cell (ties go to the diagonal).  Such a path is the diagonal with EXCURSIONS, each leaving the diagonal and coming back to it; the excursions lie in disjoint
// column ranges, so their gains add, and it is enough that no single excursion gains.  An excursion over the columns [x, y) with R gap runs, I inserted and I
// deleted bases has  gain = (a+b) (M - own) - I (a + e_ins + e_del) - opens(R),  M = the diagonal's mismatches in [x, y), own = its own mismatches.
//   R = 2, g >= g0:   gain <= loss - g a - c(g) <= 0 by the choice of g0 (as before; g0 is 5 / 6 for 5 / 6 mismatches with the usual penalties).
//   R = 2, g <  g0:   the exact running maximum over both orders of the two gaps (C1; shifts 1 .. g0-1 instead of 1 .. 2).
//   R >= 3:           split the columns [x, y) by their TEXT base: deleted (I columns, H <= min(I, m) of them mismatches of the diagonal) or aligned in one of the
//                     R - 1 pieces between two runs.  A piece at shift s, |s| < g0, saves (mismatches of the diagonal over its text columns) - (its own) <= B,
//                     the largest such saving of any stretch of any of those shifted diagonals — a second running maximum in the same sweep.  So
//                     M - own <= min(m, (R-1) B + H), I >= max(H, ceil(R/2), 2), opens(R) >= o_ins + o_del + (R-2) min(o_ins, o_del): a few dozen (R, H) pairs
//                     to look at (C2).  A piece at a shift of g0 or more needs I >= g0: gain <= (a+b) m - g0 (a + e_ins + e_del) - opens(3), checked as well.
// With the usual penalties C2 holds for m = 5, 6 when B <= 1 (m = 7 would need B = 0: not looked at); low-complexity sequence, whose shifted diagonals match, fails C1 or C2
// and takes the DP as before.  LH_K7_WEAK (test builds only, tests/hipemu `weak1` / `weak2`) leaves out one of the two checks: the crafted cases of
// tests/test_emu_front.py must then come out wrong.
#ifndef LH_K7_WEAK
#define LH_K7_WEAK 0
#endif
__device__ __forceinline__ int aln_deep_check(const DIndex& ix, const DOpts& o, const uint32_t* q4, i64 qp0, i64 rb, int lq, int loss, int m, int t_first, int t_last, int eligible) {
    if (!eligible || m > 6 || o.o_ins < 1 || o.o_del < 1) return 0;   // (seven would need B = 0 with the usual penalties: some shifted diagonal always matches somewhere)
    int g0 = 1;
    while (o.o_ins + o.o_del + g0 * (o.e_ins + o.e_del + o.a) < loss) { if (++g0 > 8) return 0; }
    const int ab = o.a + o.b, per = o.a + o.e_ins + o.e_del, omin = o.o_ins < o.o_del ? o.o_ins : o.o_del;
    if (ab * m - g0 * per - (o.o_ins + o.o_del + omin) > 0) return 0;   // (an excursion of three or more runs with a piece g0 or more columns out)
    int K[14], KT[7], B = 0;   // K[2 (g-1) + v]: running maxima of C1 (v = 0: insertion first, 1: deletion first); KT[g-1]: the text-indexed one of the deletion-first diagonal
#pragma unroll
    for (int i = 0; i < 14; ++i) K[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) KT[i] = 0;
    const int e_beg = (t_first - g0 > 0 ? t_first - g0 : 0) & ~7;
    for (int e8 = e_beg; e8 <= t_last + 1; e8 += 8) {
        const u64 qq = (u64)dev_nib8(q4, qp0 + e8) | (u64)dev_nib8(q4, qp0 + e8 + 8) << 32;
        const u64 tt = (u64)dev_nib8(ix.tn, rb + e8) | (u64)dev_nib8(ix.tn, rb + e8 + 8) << 32;
        const u64 xm = qq ^ tt;
        uint32_t mmask = 0;   // bit j: the diagonal's pair e8 + j is a mismatch (pairs beyond the last one are not)
#pragma unroll
        for (int j = 0; j < 16; ++j) mmask |= (uint32_t)(((xm >> (4 * j)) & 0xf) != 0 && e8 + j <= t_last) << j;
#pragma unroll
        for (int g = 1; g <= 7; ++g) {
            if (g >= g0) continue;
            if (LH_K7_WEAK == 1 && g > 2) continue;
            const int lim = g * o.a + o.o_ins + o.o_del + g * (o.e_ins + o.e_del);   // g a + c(g)
            const u64 xi = (qq >> (4 * g)) ^ tt, xd = qq ^ (tt >> (4 * g));       // q[k + g] against t[k]; q[k] against t[k + g]
            int ki = K[2 * g - 2], kd = K[2 * g - 1], kt = KT[g - 1];
            for (int u = 0; u < 8; ++u) {
                const int e = e8 + u;
                if (e > t_last + 1 || e + g > lq) break;
                const int tail = __popc(mmask >> u & ((1u << g) - 1u));
                if (ab * ((ki > kd ? ki : kd) + tail) > lim) return 0;
                if (e + g < lq) {
                    const int m0 = mmask >> u & 1, mg = mmask >> (u + g) & 1;
                    const int bi = ((xi >> (4 * u)) & 0xf) != 0, bd = ((xd >> (4 * u)) & 0xf) != 0;
                    ki += m0 - bi; ki = ki > 0 ? ki : 0;
                    kd += m0 - bd; kd = kd > 0 ? kd : 0;
                    kt += mg - bd; kt = kt > 0 ? kt : 0;
                    B = B > ki ? B : ki; B = B > kt ? B : kt;
                }
            }
            K[2 * g - 2] = ki; K[2 * g - 1] = kd; KT[g - 1] = kt;
        }
    }
    if (LH_K7_WEAK == 2) B = 0;
    for (int R = 3; o.o_ins + o.o_del + (R - 2) * omin < ab * m; ++R)
        for (int h = 0; h <= m; ++h) {
            int I = (R + 1) >> 1;
            I = I > h ? I : h; I = I > 2 ? I : 2;
            int av = (R - 1) * B + h;
            av = av < m ? av : m;
            if (ab * av - I * per - (o.o_ins + o.o_del + (R - 2) * omin) > 0) return 0;
        }
    return 1;
}

// one candidate of a read without DP, by ONE lane: everything k_aln would write for it, or 1 = it needs the DP (nothing final written).
// q = the read's bytes, off = its offset in the batch's 4-bit stream q4, best = the read's best region score
template <int DEEP>
__device__ __forceinline__ int aln_fast_cand(const DIndex& ix, const DOpts& o, const DCand& R, const uint8_t* q, const uint32_t* q4, i64 off, int l_query, const DReg& ar, i64 c,
                                             int best, int32_t* status_r, unsigned* proven_cells) {
        const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
        const i64 rb = ar.rb, re = ar.re;
        const int rlen = (int)(re - rb);
        const int valid = lq > 0 && rb < re && !(rb < ix.l_pac && re > ix.l_pac) && rlen <= LH_MAXT;
        int tmp = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_del, o.e_del);
        int w2 = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_ins, o.e_ins);
        w2 = w2 > tmp ? w2 : tmp;
        if (w2 > o.w) w2 = w2 < ar.w ? w2 : ar.w;
        w2 = w2 < o.w << 2 ? w2 : o.w << 2;
        if (!(valid && lq == rlen)) return 1;
        if (w2 != 0) {
            // Equal spans but an inferred band > 0 (BWA's test is only "fewer than two gaps' worth of penalty lost").  The
            // banded global alignment is still the plain diagonal whenever no path with gaps can beat it: such a path has
            // g >= 1 inserted and g deleted bases, so it scores at most (L-g)*a - (o_ins + g*e_ins) - (o_del + g*e_del),
            // largest at g = 1.  If the diagonal's own score S0 reaches that bound the DP ends on the diagonal (ties are
            // resolved towards the diagonal in ksw_global2: d = M >= E ? 0 : 1, then h >= F), every retry of mem_reg2aln
            // returns the same score, and the CIGAR is lq M.  With the default scoring this settles 3 mismatches
            // (bound: mm*(a+b) <= a + oe_ins + oe_del).  S0 needs the bases, so it is checked after the compare loop.
        }
        // no gap: CIGAR = [clip] lq M [clip]; one pass over the bases gives NM and lariat's mismatch loci.
        // The aligned pairs are q[qb + t] vs the base at fwd||rev coordinate rb + t (the wave kernel's oriented views
        // pair the same bases in the opposite order on the reverse strand).
        int is_rev;
        const i64 posf = dev_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
        const i64 coff = ix.contig_off[ar.rid];
        const i64 Offset = rb < ix.l_pac ? rb - coff : ix.l_pac * 2 - 1 - rb - coff;   // InterpretAlign (gobwa.go:339-371)
        const i64 End = re < ix.l_pac ? re - coff : ix.l_pac * 2 - 1 - re - coff;
        i64 refStart = Offset, refEnd = End;
        if (is_rev) { refStart = End + 1; refEnd = Offset + 1; }
        int32_t* mref = R.mm_ref + (size_t)c * LH_MAX_MM;
        int32_t* mread = R.mm_read + (size_t)c * LH_MAX_MM;
        int nmm = 0, n_amb = 0, xo = -2;   // n_amb: mismatching pairs with an ambiguous base (scored -1, not -b)
        int t_first = lq, t_last = -1;     // the first and the last mismatching pair
        // 32 pairs per round trip, eight per XOR: the batch's 4-bit reads against the 4-bit text (both strands: rb + t is a text position);
        // without those tables the same words are put together from the bytes and the 2-bit reference
        LaneTgt tg;
        const bool packed = ix.tn && q4;
        tg.init(ix, rb, 1);
        for (int t0 = 0; t0 < lq; t0 += 32) {
            uint32_t qx[4], tx[4];
            if (packed) {
                const i64 qp = off + qb + t0, tp = rb + t0;
                const uint32_t* qa = q4 + (qp >> 3);
                const uint32_t* ta = ix.tn + (tp >> 3);
                uint32_t qr[5], tr[5];
                for (int k = 0; k < 5; ++k) { qr[k] = qa[k]; tr[k] = ta[k]; }   // (both streams are padded past their ends)
                const int qs = (int)(qp & 7) * 4, ts = (int)(tp & 7) * 4;
                for (int k = 0; k < 4; ++k) {
                    qx[k] = qs ? (qr[k] >> qs) | (qr[k + 1] << (32 - qs)) : qr[k];
                    tx[k] = ts ? (tr[k] >> ts) | (tr[k + 1] << (32 - ts)) : tr[k];
                }
            } else {
                for (int k = 0; k < 4; ++k) {
                    qx[k] = 0; tx[k] = 0;
                    for (int u = 0; u < 8 && t0 + 8 * k + u < lq; ++u) { qx[k] |= (uint32_t)(q[qb + t0 + 8 * k + u] & 0xf) << (4 * u); tx[k] |= (uint32_t)tg.base(t0 + 8 * k + u) << (4 * u); }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tb = t0 + 8 * k;
                if (tb >= lq) break;
                const uint32_t qw = qx[k];
                uint32_t xw = (qw ^ tx[k]) & (lq - tb < 8 ? (1u << (4 * (lq - tb))) - 1u : 0xffffffffu);
                while (xw) {
                    const int u = (__ffs((int)xw) - 1) >> 2, t = tb + u;
                    xw &= ~(0xfu << (4 * u));
                    if (nmm < LH_MAX_MM) { mref[nmm] = is_rev ? (int)refEnd - t : t + (int)refStart; mread[nmm] = qb + t; }
                    else {   // (rare: the pool, see DCand)
                        if (DEEP) return 1;   // (far beyond what the second look can settle: leave the pool to the DP kernels)
                        if (xo == -2) {
                            xo = atomicAdd(R.mm_xctr, l_query - LH_MAX_MM);
                            if (xo + l_query - LH_MAX_MM > R.mm_xcap) xo = -1;
                            R.mm_xoff[c] = xo;
                        }
                        if (xo >= 0) { R.mm_xref[xo + nmm - LH_MAX_MM] = is_rev ? (int)refEnd - t : t + (int)refStart; R.mm_xread[xo + nmm - LH_MAX_MM] = qb + t; }
                    }
                    nmm++;
                    n_amb += ((qw >> (4 * u)) & 0xf) > 3;
                    t_first = t_first < t ? t_first : t; t_last = t;
                }
            }
        }
        const int NM = nmm;
        if (w2 != 0) {
            const int S0 = lq * o.a - (nmm - n_amb) * (o.a + o.b) - n_amb * (o.a + 1);
            const int bound = (lq - 1) * o.a - (o.o_ins + o.e_ins) - (o.o_del + o.e_del);
            if (S0 < bound) {
                // A FEW MORE MISMATCHES (new in r03: 37 % of the candidates that used to need the wave DP).  The diagonal still wins when no
                // path with gaps reaches it at ANY diagonal cell: let D(t) be the diagonal's score up to pair t and G(t) the best score of a
                // path to (t, t) with gaps; ksw_global2 picks M at (t, t) iff H(t-1,t-1) + s >= E, F (ties towards the diagonal), and
                // E(t,t), F(t,t) <= G(t), so G(t) <= D(t) for all t leaves score S0 and the CIGAR lq M, in every band and every retry.
                // A path back on the diagonal has as many inserted as deleted bases.  (1) Three or more gap runs: at least two runs on one
                // side, so >= 2 bases each way, cost >= C3 and at most t + 1 - 2 aligned pairs: below D(t) whenever the diagonal's total
                // loss is <= 2a + C3.  (2) One run each way of g bases, cost c(g) = o_ins + o_del + g (e_ins + e_del): the detour replaces
                // the pairs [x, e + g) by the pairs of the diagonal shifted by g over [x, e); its gain is (a + b) x (mismatches of the main
                // diagonal in [x, e + g) - mismatches of the shifted one in [x, e)) - g a - c(g) <= loss - g a - c(g): never positive
                // for g >= g0.  (3) For g < g0, both shifts: the largest gain over all x <= e is a running maximum (Kadane) over the two
                // mismatch patterns, compared eight bases per word against the 4-bit text — this is where low-complexity sequence, whose
                // shifted diagonals do match, is told apart.  Reads with an ambiguous base in the span take the DP.
                const int loss = lq * o.a - S0;
                const int c3a = 2 * o.o_ins + 2 * o.e_ins + o.o_del + 2 * o.e_del, c3b = 2 * o.o_del + 2 * o.e_del + o.o_ins + 2 * o.e_ins;
                int okd = packed && n_amb == 0 && loss <= 2 * o.a + (c3a < c3b ? c3a : c3b);
                if (!DEEP) {
                    int g0 = 1;
                    while (okd && o.o_ins + o.o_del + g0 * (o.e_ins + o.e_del + o.a) < loss) { if (++g0 > 5) okd = 0; }
                    if (okd && g0 > 3) okd = 0;   // (shifts of one and two bases are checked below: enough for every loss the bound of (1) admits with BWA's usual penalties)
                    if (okd) {
                        // shifts 1 and 2, both directions, in one sweep from the first to the last mismatch of the main diagonal (outside, every gain is
                        // zero or falling): sixteen bases of read and text per step in two 64-bit words, the shifted diagonals are shifts of those
                        const i64 qp0 = off + qb;
                        int K[4] = {0, 0, 0, 0};   // running maxima: (g = 1, insertion first), (1, deletion first), (2, ins), (2, del)
                        const int e_beg = (t_first - 1 > 0 ? t_first - 1 : 0) & ~7;
                        for (int e8 = e_beg; e8 <= t_last + 1 && okd; e8 += 8) {
                            const u64 qq = (u64)dev_nib8(q4, qp0 + e8) | (u64)dev_nib8(q4, qp0 + e8 + 8) << 32;
                            const u64 tt = (u64)dev_nib8(ix.tn, rb + e8) | (u64)dev_nib8(ix.tn, rb + e8 + 8) << 32;
                            const u64 xm = qq ^ tt;   // (nibbles past the spans are never looked at: see the limits below)
                            const u64 xs[4] = {(qq >> 4) ^ tt, qq ^ (tt >> 4), (qq >> 8) ^ tt, qq ^ (tt >> 8)};
                            for (int u = 0; u < 8; ++u) {
                                const int e = e8 + u;
                                if (e > t_last + 1) break;
                                const int m0 = ((xm >> (4 * u)) & 0xf) != 0, m1 = ((xm >> (4 * u + 4)) & 0xf) != 0;
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const int g = 1 + (v >> 1);
                                    if (g >= g0 || e + g > lq) continue;
                                    const int lim = g * o.a + o.o_ins + o.o_del + g * (o.e_ins + o.e_del);   // g a + c(g)
                                    const int tail = m0 + (g == 2 ? m1 : 0);
                                    if ((o.a + o.b) * (K[v] + tail) > lim) okd = 0;
                                    const int mb = ((xs[v] >> (4 * u)) & 0xf) != 0;
                                    K[v] += m0 - mb;
                                    K[v] = K[v] > 0 ? K[v] : 0;
                                }
                            }
                        }
                    }
                } else okd = aln_deep_check(ix, o, q4, off + qb, rb, lq, loss, nmm, t_first, t_last, packed && n_amb == 0);
                if (!okd) return 1;   // a gapped path could win: run the DP (k_aln)
            }
            // the DP cells mem_reg2aln would have evaluated (telemetry stays comparable with the reference's work)
            int wq = w2, it = 0, last_sc = -(1 << 30);
            do {
                wq = wq < o.w << 2 ? wq : o.w << 2;
                int max_ins = (int)((double)(((lq + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
                int max_del = (int)((double)(((lq + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
                int max_gap = max_ins > max_del ? max_ins : max_del;
                max_gap = max_gap > 1 ? max_gap : 1;
                int w = (max_gap + 1) >> 1;
                w = w < wq ? w : wq;
                w = w > 3 ? w : 3;
                {   // sum over rows i of min(i + w + 1, lq) - max(i - w, 0)   (rlen == lq here)
                    const int n1 = lq - w > 0 ? lq - w : 0, m = lq - 1 - w > 0 ? lq - 1 - w : 0;
                    *proven_cells += (unsigned)(n1 * (w + 1) + n1 * (n1 - 1) / 2 + (lq - n1) * lq - m * (m + 1) / 2);
                }
                if (S0 == last_sc || wq == o.w << 2) break;
                last_sc = S0;
                wq <<= 1;
            } while (++it < 3 && S0 < ar.truesc - o.a);
        }
        int mm_ovf = 0;
        if (nmm > LH_MAX_MM && xo < 0) { mm_ovf = 1; nmm = LH_MAX_MM; }   // (NM keeps the true count)
        int clip5 = 0, clip3 = 0, no = 0;
        if (qb != 0 || qe != l_query) { clip5 = is_rev ? l_query - qe : qb; clip3 = is_rev ? qb : l_query - qe; }
        uint32_t* cgo = R.cigar + (size_t)c * LH_MAX_CIGAR;
        if (clip5) cgo[no++] = (uint32_t)clip5 << 4 | 3;
        cgo[no++] = (uint32_t)lq << 4 | 0;
        if (clip3) cgo[no++] = (uint32_t)clip3 << 4 | 3;
        const int soft_clipping = (clip5 > 0) + (clip3 > 0), soft_clipping_length = clip5 + clip3;
        int mismatches = NM, matches = lq - NM;
        const int rid = dev_pos2rid(ix, posf);
        i64 pos = Offset, aend = End;
        if (pos != -1 && is_rev) { pos = End + 1; aend = Offset + 1; }
        R.rid[c] = rid; R.pos[c] = pos; R.aend[c] = aend; R.rb[c] = rb; R.re[c] = re; R.reversed[c] = (uint8_t)is_rev; R.score[c] = ar.score;
        R.qb[c] = qb; R.qe[c] = qe; R.nm[c] = NM; R.matches[c] = matches; R.mismatches[c] = mismatches; R.indels[c] = 0;
        R.soft_clipped[c] = soft_clipping; R.soft_clipped_length[c] = soft_clipping_length;
        R.in_filtered[c] = ar.score >= best - o.aln_score_delta;
        R.n_cigar[c] = no; R.n_mm[c] = nmm; R.read_len[c] = l_query;
        R.lap[c] = (dev_single_score(mismatches, 0, soft_clipping, soft_clipping_length) + o.improper_pair_penalty) - o.improper_pair_penalty;
        if (mm_ovf) atomicOr(status_r, LH_ST_MM_OVERFLOW);
    return 0;
}

// ---- the candidates that need no DP, one LANE PER CANDIDATE (r04, late; until then a lane per read — 13 of 64 lanes active on unique sequence, where a read has
// one to five candidates, and a wave per read beside it for reads with dozens).  k_aln_prep, a lane per read: the read's placeholder when it has no
// region (lariat.go:1737-1750,1773-1785), its best region score, and the read of every candidate slot (-1: the candidate pool cannot hold the read).
// k_aln_flat: aln_fast_cand for candidate c; the ones that need the DP are listed for k_aln_grp. ----
__global__ void __launch_bounds__(256) k_aln_prep(DOpts o, int n_reads, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off, const DReg* __restrict__ regs,
                                                   const int32_t* __restrict__ n_regs, DCand R, i64 cand_cap, int32_t* __restrict__ status, int32_t* __restrict__ best_r,
                                                   int32_t* __restrict__ cand_rd) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int l_query = (int)(seq_off[r + 1] - seq_off[r]);
    if (l_query > LH_MAXLEN) l_query = 0;
    const DReg* av = regs + reg_off[r];
    const int n = n_regs[r];
    const i64 c0 = R.cand_off[r];
    if (c0 + (n > 0 ? n : 1) > cand_cap) {   // candidate pool exhausted: flag and skip (host returns LH_E_CAPACITY)
        atomicOr(&status[r], LH_ST_POOL_OVERFLOW);
        for (i64 c = c0; c < c0 + n && c < cand_cap; ++c) cand_rd[c] = -1;
    } else if (n == 0) {   // placeholder: contig "", pos -1, aend 0, score 0
        const i64 c = c0;
        R.rid[c] = -1; R.pos[c] = -1; R.aend[c] = 0; R.rb[c] = -1; R.re[c] = -1; R.reversed[c] = 0; R.score[c] = 0; R.qb[c] = 0; R.qe[c] = 0;
        R.nm[c] = 0; R.matches[c] = 0; R.mismatches[c] = 0; R.indels[c] = 0; R.soft_clipped[c] = 0; R.soft_clipped_length[c] = 0;
        R.in_filtered[c] = 1; R.n_cigar[c] = 0; R.n_mm[c] = 0; R.read_len[c] = l_query;
        R.lap[c] = (dev_single_score(0, 0, 0, 0) + o.improper_pair_penalty) - o.improper_pair_penalty;
        cand_rd[c] = -1;
    } else {
        int best = 0;
        for (int i = 0; i < n; ++i) { const int sc = av[i].score; best = best > sc ? best : sc; cand_rd[c0 + i] = r; }
        best_r[r] = best;
    }
}
__global__ void __launch_bounds__(256) k_aln_flat(DIndex ix, DOpts o, i64 n_cand, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                   const DReg* __restrict__ regs, DCand R, int32_t* __restrict__ status, int32_t* __restrict__ slow_r, int32_t* __restrict__ slow_ci,
                                                   int32_t* __restrict__ slow_count, DCounters* __restrict__ ctr, const uint32_t* __restrict__ q4, const int32_t* __restrict__ best_r,
                                                   const int32_t* __restrict__ cand_rd) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = LANE();
    unsigned proven_cells = 0;   // cells of global DPs whose outcome is known without running them (aln_fast_cand)
    int slow = 0, r = -1, ci = 0;
    if (c < n_cand) r = cand_rd[c];
    if (r >= 0) {
        const i64 off = seq_off[r];
        int l_query = (int)(seq_off[r + 1] - off);
        if (l_query > LH_MAXLEN) l_query = 0;
        ci = (int)(c - R.cand_off[r]);
        const DReg ar = regs[reg_off[r] + ci];
        slow = aln_fast_cand<0>(ix, o, R, seq + off, q4, off, l_query, ar, c, best_r[r], &status[r], &proven_cells);
    }
    if (ctr) {
        u64 tot = (u64)(uint32_t)wave_sum_i32((int)(proven_cells >> 16)) << 16;
        tot += (u64)(uint32_t)wave_sum_i32((int)(proven_cells & 0xffff));
        if (lane == 0 && tot) atomicAdd(&LH_CTR(ctr)->glob_cells, tot);
    }
    const u64 sb = __ballot(slow);   // the wave reserves list space once (same-address atomics are slow)
    if (sb) {
        int basep = 0;
        if (lane == 0) { basep = atomicAdd(slow_count, (int)__popcll(sb)); if (ctr) atomicAdd(&LH_CTR(ctr)->n_glob_listed, (u64)__popcll(sb)); }
        basep = wave_readlane(basep, 0) + lanes_below(sb, lane);
        if (slow) { slow_r[basep] = r; slow_ci[basep] = ci; }
    }
}

// (r06) the listed candidates once more, a lane each, with aln_deep_check: what it settles is written as k_aln_flat would have, the others are listed again
__global__ void __launch_bounds__(256) k_aln_flat2(DIndex ix, DOpts o, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                    const DReg* __restrict__ regs, DCand R, int32_t* __restrict__ status, const int32_t* __restrict__ in_r, const int32_t* __restrict__ in_ci,
                                                    const int32_t* __restrict__ in_count, int32_t* __restrict__ slow_r, int32_t* __restrict__ slow_ci, int32_t* __restrict__ slow_count,
                                                    DCounters* __restrict__ ctr, const uint32_t* __restrict__ q4, const int32_t* __restrict__ best_r) {
    const int lane = LANE();
    const int n_items = *in_count;
    for (int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {
        const int item = base + (int)threadIdx.x;
        unsigned proven_cells = 0;
        int slow = 0, r = -1, ci = 0;
        if (item < n_items) {
            r = in_r[item]; ci = in_ci[item];
            const i64 off = seq_off[r];
            int l_query = (int)(seq_off[r + 1] - off);
            if (l_query > LH_MAXLEN) l_query = 0;
            const DReg ar = regs[reg_off[r] + ci];
            slow = aln_fast_cand<1>(ix, o, R, seq + off, q4, off, l_query, ar, R.cand_off[r] + ci, best_r[r], &status[r], &proven_cells);
        }
        if (ctr) {
            u64 tot = (u64)(uint32_t)wave_sum_i32((int)(proven_cells >> 16)) << 16;
            tot += (u64)(uint32_t)wave_sum_i32((int)(proven_cells & 0xffff));
            if (lane == 0 && tot) atomicAdd(&LH_CTR(ctr)->glob_cells, tot);
        }
        const u64 sb = __ballot(slow);
        if (sb) {
            int basep = 0;
            if (lane == 0) { basep = atomicAdd(slow_count, (int)__popcll(sb)); if (ctr) atomicAdd(&LH_CTR(ctr)->n_glob_exec, (u64)__popcll(sb)); }
            basep = wave_readlane(basep, 0) + lanes_below(sb, lane);
            if (slow) { slow_r[basep] = r; slow_ci[basep] = ci; }
        }
    }
}

// ---- FOUR listed candidates per wave (new in r04): the banded global alignment in a group of 16 lanes ----
// The band mem_reg2aln asks for is wide (a read with five mismatches: w = 21, 43 columns; measured on the repeat-family input: 15 % of the
// listed candidates have at most 16 columns, 55 % between 33 and 64), the alignment it finds is not: almost always it stays within a few
// columns of the main diagonal.  A NARROW band of wn = 7 (15 columns) gives the same score AND the same traceback as any wider one when
// every path that leaves the narrow band scores below the narrow band's optimum Sn.  Such a path holds at least wn + 1 inserted bases at
// some point (or wn + 1 deleted ones) and must come back to the end cell, dq = lq - rlen columns off the main diagonal: at most
//   U = max( a (lq - (wn+1)) - (o_ins + e_ins (wn+1)) - [o_del + e_del (wn+1-dq)]+ ,  a (rlen - (wn+1)) - (o_del + e_del (wn+1)) - [o_ins + e_ins (wn+1+dq)]+ )
// even if every aligned pair matched.  Sn > U: the wide DP's optimum is Sn as well (the narrow band's cells are a subset), its traceback is
// an optimal path, hence inside the narrow band; along that path every value the traceback's choices look at is the same in both DPs on the
// chosen side and no larger in the narrow one on the other side (a cell value of the narrow band never exceeds the wide band's), and
// ksw_global2 resolves ties by fixed preference (diagonal, then E, then F), so the same choice is made at every cell: the same CIGAR.  The
// same holds for each wider band of mem_reg2aln's retries, which therefore return Sn again and stop.  Measured: 83 % of the listed
// candidates of the repeat-family input and 99.9 % of the headline's are settled here; the others (Sn <= U, a reference span beyond
// LH_GRP_T, more than LH_GRP_CIG operations) are listed for k_aln, which starts them afresh.
// (r05) GL = 16: four candidates per wave, band up to 7 (as r04).  GL = 32: two per wave, band up to 15 — for what the first kernel hands on: with five or more mismatches
// Sn <= U(7) whatever the candidate looks like (U(wn) = lq - 2 (wn + 1) - 6 with the default scoring: 22 below the perfect score at wn = 7, 38 at wn = 15), a sixth of the
// candidates on repeat families; they ran one per wave in k_aln, 43 columns of 64 lanes, as long as the other five sixths here.
#define LH_GRP_T 176
#define LH_GRP_CIG 32
template <int GL>
__global__ void __launch_bounds__(64) k_aln_grp(DIndex ix, DOpts o, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off, const i64* __restrict__ reg_off,
                                                 const DReg* __restrict__ regs, const int32_t* __restrict__ n_regs, DCand R, int32_t* __restrict__ status, DCounters* __restrict__ ctr,
                                                 const int32_t* __restrict__ slow_r, const int32_t* __restrict__ slow_ci, const int32_t* __restrict__ slow_count,
                                                 int32_t* __restrict__ wide_r, int32_t* __restrict__ wide_ci, int32_t* __restrict__ wide_count) {
    constexpr int NG = 64 / GL, GRP_W = GL / 2 - 1, GRP_Z = LH_GRP_T * (2 * GRP_W + 1);
    __shared__ uint8_t q_[NG][LH_MAXLEN + 6];
    __shared__ uint8_t tref_[NG][LH_GRP_T];
    __shared__ uint32_t cg_[NG][LH_GRP_CIG + 4];
    __shared__ uint32_t cgo_[NG][LH_GRP_CIG + 4];
    __shared__ int32_t sh_[NG][8];
    __shared__ uint8_t z_[NG][GRP_Z];
    const int lane = LANE(), g = lane / GL, d = lane & (GL - 1);
    uint8_t* const q = q_[g];
    uint8_t* const tref = tref_[g];
    uint32_t* const cg = cg_[g];
    uint32_t* const cgo = cgo_[g];
    int32_t* const sh = sh_[g];
    uint8_t* const z = z_[g];
    u64 cells = 0;   // (the group's first lane: the cells of the DPs mem_reg2aln asks for, whatever band was run)
    const int n_items = *slow_count;
    for (int base = blockIdx.x * NG; base < n_items; base += gridDim.x * NG) {
        const int item = base + g;
        const int has = item < n_items;
        int r = 0, ci = 0, l_query = 0, n = 0;
        i64 off = 0, c0 = 0;
        const DReg* av = regs;
        if (has) {
            r = slow_r[item]; ci = slow_ci[item];
            off = seq_off[r];
            l_query = (int)(seq_off[r + 1] - off);
            if (l_query > LH_MAXLEN) l_query = 0;
            av = regs + reg_off[r];
            n = n_regs[r];
            c0 = R.cand_off[r];
        }
        WAVE_SYNC();
        for (int i = d; i < l_query; i += GL) q[i] = seq[off + i];
        int best = 0;
        for (int i = d; i < n; i += GL) { int s = av[i].score; best = best > s ? best : s; }
        best = grpN_max_i32<GL>(best);
        DReg ar;
        ar.rb = ar.re = 0; ar.qb = ar.qe = ar.rid = ar.score = ar.truesc = ar.w = 0;
        if (has) ar = av[ci];
        const i64 c = c0 + ci;
        const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
        const i64 rb = ar.rb, re = ar.re;
        const int rlen = (int)(re - rb);
        const int rev = rb >= ix.l_pac;
        const int qoff = rev ? qe - 1 : qb, qstep = rev ? -1 : 1;
        const i64 t0 = rev ? re - 1 : rb;
        const int tstep = rev ? -1 : 1;
        const int dq = lq - rlen, adq = dq < 0 ? -dq : dq;
        const int valid = has && lq > 0 && rb < re && !(rb < ix.l_pac && re > ix.l_pac) && re - rb <= LH_GRP_T;
        int towide = has && !valid;
        if (valid)
            for (int i = d; i < rlen; i += GL) tref[i] = (uint8_t)dev_ref_base(ix, t0 + (i64)tstep * i);
        WAVE_SYNC();
        int tmp = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_del, o.e_del);
        int w2 = dev_infer_bw(lq, rlen, ar.truesc, o.a, o.o_ins, o.e_ins);
        w2 = w2 > tmp ? w2 : tmp;
        if (w2 > o.w) w2 = w2 < ar.w ? w2 : ar.w;
        int score = 0, last_sc = -(1 << 30), n_cigar = 0, proved = 0, wn = 0;
        u64 ccells = 0;   // this candidate's DP cells (counted when it is settled here: k_aln counts its own)
        int act = valid;   // mem_reg2aln's loop is still running for this candidate
        for (int it = 0; it < 3; ++it) {
            if (!__any(act)) break;
            int w = 0, run = 0, ccw = -1;
            if (act) {
                w2 = w2 < o.w << 2 ? w2 : o.w << 2;
                if (lq == rlen && w2 == 0) { towide = 1; act = 0; }   // ("no gap; no need to do DP": never listed by the kernels above; k_aln has the branch)
                else {
                    int max_ins = (int)((double)(((lq + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
                    int max_del = (int)((double)(((lq + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
                    int max_gap = max_ins > max_del ? max_ins : max_del;
                    max_gap = max_gap > 1 ? max_gap : 1;
                    w = (max_gap + adq + 1) >> 1;
                    w = w < w2 ? w : w2;
                    const int min_w = adq + 3;
                    w = w > min_w ? w : min_w;
                    ccw = w;   // (the cells ksw_global2 evaluates with the band it was asked for: counted below, by the group)
                    if (!proved) {
                        wn = w < GRP_W ? w : GRP_W;
                        if (wn < adq + 3) { towide = 1; act = 0; }   // (the narrow band would not reach the end cell with BWA's own margin)
                        else run = 1;
                    }
                }
            }
            {   // (r05: the group's lanes share the rows of the count; it was the group's first lane walking all of them, a tenth of the kernel)
                int part = 0;
                if (ccw >= 0)
                    for (int i = d; i < rlen; i += GL) {
                        const int beg = i > ccw ? i - ccw : 0, end = i + ccw + 1 < lq ? i + ccw + 1 : lq;
                        if (end > beg) part += end - beg;
                    }
                ccells += (u64)grpN_sum_i32<GL>(part);
            }
            const int tl_max = wave_max_i32(run ? rlen : 0);
            if (tl_max > 0) {
                const int sc = grp_ksw_global2_band<GL>(o, q, qoff, qstep, lq, tref, rlen, wn, z, lane, run, tl_max);
                if (run) score = sc;
            }
            WAVE_SYNC();
            if (run && wn < w) {
                const int ui = o.a * (lq - (wn + 1)) - (o.o_ins + o.e_ins * (wn + 1)) - (wn + 1 - dq > 0 ? o.o_del + o.e_del * (wn + 1 - dq) : 0);
                const int ud = o.a * (rlen - (wn + 1)) - (o.o_del + o.e_del * (wn + 1)) - (wn + 1 + dq > 0 ? o.o_ins + o.e_ins * (wn + 1 + dq) : 0);
                if (score > (ui > ud ? ui : ud)) proved = 1;
                else { towide = 1; act = 0; run = 0; }
            }
            if (run && d == 0) {   // backtrack, by the group's first lane (ops are produced last-to-first; equal neighbours merge)
                const int n_col = lq < 2 * wn + 1 ? lq : 2 * wn + 1;
                int which = 0, nc = 0, ovf = 0;
                int i = rlen - 1, k = (i + wn + 1 < lq ? i + wn + 1 : lq) - 1;
                while (i >= 0 && k >= 0) {
                    which = z[i * n_col + (k - (i > wn ? i - wn : 0))] >> (which << 1) & 3;
                    int op;
                    if (which == 0) { op = 0; --i; --k; }
                    else if (which == 1) { op = 2; --i; }
                    else { op = 1; --k; }
                    if (nc == 0 || op != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_GRP_CIG) cg[nc++] = 1u << 4 | op; else ovf = 1; }
                    else cg[nc - 1] += 1u << 4;
                }
                if (i >= 0) { if (nc == 0 || 2 != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_GRP_CIG) cg[nc++] = (uint32_t)(i + 1) << 4 | 2; else ovf = 1; } else cg[nc - 1] += (uint32_t)(i + 1) << 4; }
                if (k >= 0) { if (nc == 0 || 1 != (int)(cg[nc - 1] & 0xf)) { if (nc < LH_GRP_CIG) cg[nc++] = (uint32_t)(k + 1) << 4 | 1; else ovf = 1; } else cg[nc - 1] += (uint32_t)(k + 1) << 4; }
                for (int u = 0; u < nc >> 1; ++u) { uint32_t t = cg[u]; cg[u] = cg[nc - 1 - u]; cg[nc - 1 - u] = t; }
                sh[0] = nc; sh[1] = ovf;
            }
            WAVE_SYNC();
            if (run) {
                n_cigar = sh[0];
                if (sh[1]) { towide = 1; act = 0; }
            }
            if (act) {   // (a proved candidate's wider retry returns the same score: this is where its loop ends)
                if (score == last_sc || w2 == o.w << 2) act = 0;
                else {
                    last_sc = score;
                    w2 <<= 1;
                    if (!(it + 1 < 3 && score < ar.truesc - o.a)) act = 0;
                }
            }
        }
        const int fin = valid && !towide;
        // NM: mismatches inside M runs + inserted + deleted bases (terminal D excluded)
        int NM = 0;
        {
            int x = 0, y = 0, n_mm = 0, n_gap = 0;
            if (fin)
                for (int k = 0; k < n_cigar; ++k) {
                    const int op = cg[k] & 0xf, len = (int)(cg[k] >> 4);
                    if (op == 0) {
                        for (int t = d; t < len; t += GL) n_mm += q[qoff + qstep * (x + t)] != tref[y + t];
                        x += len; y += len;
                    } else if (op == 2) {
                        if (k > 0 && k < n_cigar - 1) n_gap += len;
                        y += len;
                    } else if (op == 1) { x += len; n_gap += len; }
                }
            NM = grpN_sum_i32<GL>(n_mm) + n_gap;
        }
        int is_rev = 0;
        i64 posf = 0;
        if (fin) posf = dev_depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
        WAVE_SYNC();
        if (fin && d == 0) {
            int nc = n_cigar, s0 = 0, no = 0;
            if (nc > 0) {   // squeeze out leading or trailing deletions
                if ((cg[0] & 0xf) == 2) { s0 = 1; nc--; }
                else if ((cg[nc - 1] & 0xf) == 2) nc--;
            }
            int clip5 = 0, clip3 = 0;
            if (qb != 0 || qe != l_query) { clip5 = is_rev ? l_query - qe : qb; clip3 = is_rev ? qb : l_query - qe; }
            if (clip5) cgo[no++] = (uint32_t)clip5 << 4 | 3;
            for (int u = 0; u < nc; ++u) cgo[no++] = cg[s0 + u];
            if (clip3) cgo[no++] = (uint32_t)clip3 << 4 | 3;
            sh[3] = no;
        }
        WAVE_SYNC();
        const int no = fin ? sh[3] : 0;
        static_assert(LH_GRP_CIG + 2 <= LH_MAX_CIGAR, "a group's CIGAR with both clips fits the candidate's slots");
        const i64 coff = fin ? ix.contig_off[ar.rid] : 0;
        const i64 Offset = rb < ix.l_pac ? rb - coff : ix.l_pac * 2 - 1 - rb - coff;   // InterpretAlign (gobwa.go:339-371)
        const i64 End = re < ix.l_pac ? re - coff : ix.l_pac * 2 - 1 - re - coff;
        i64 refStart = Offset, refEnd = End;
        if (is_rev) { refStart = End + 1; refEnd = Offset + 1; }
        // CIGAR walk in READ orientation (lariat.go:1591-1632): the group's lanes over the bases of each M run
        int matches = 0, indels = 0, indel_length = 0, soft_clipping = 0, soft_clipping_length = 0, refSeqOffset = 0, readOffset = 0, nmm = 0;
        int32_t* mref = R.mm_ref + (size_t)c * LH_MAX_MM;
        int32_t* mread = R.mm_read + (size_t)c * LH_MAX_MM;
        const int refLen = (int)(refEnd - refStart);
        const int no_max = wave_max_i32(no);
        for (int u = 0; u < no_max; ++u) {
            const uint32_t cv = u < no ? cgo[is_rev ? no - 1 - u : u] : 15u;
            const int op = cv & 0xf, len = (int)(cv >> 4);
            const int mlen = op == 0 ? len : 0;
            const int lmax = wave_max_i32(mlen);
            for (int t0 = 0; t0 < lmax; t0 += GL) {
                const int t = t0 + d, k = refSeqOffset + t;
                int mm = 0;
                if (t < mlen && k < refLen && readOffset + t < l_query) {
                    const int rbase = k < rlen ? tref[rev ? rlen - 1 - k : k] : 255;
                    mm = rbase != q[readOffset + t];
                }
                const uint32_t bm = (uint32_t)(__ballot(mm) >> (g * GL)) & (GL == 32 ? 0xffffffffu : 0xffffu);
                if (nmm + (int)__popc(bm) > LH_MAX_MM) towide = 1;   // (more loci than slots: k_aln knows the pool)
                else if (mm) {
                    const int slot = nmm + (int)__popc(bm & ((1u << d) - 1u));
                    mref[slot] = is_rev ? (int)refEnd - k : k + (int)refStart; mread[slot] = readOffset + t;
                }
                nmm += (int)__popc(bm);
            }
            if (op == 0) { matches += len; refSeqOffset += len; readOffset += len; }
            else if (op == 1) { indels += 1; indel_length += len; readOffset += len; }
            else if (op == 2) { indels += 1; indel_length += len; refSeqOffset += len; }
            else if (op == 3) { soft_clipping += 1; soft_clipping_length += len; readOffset += len; }
        }
        if (fin && !towide) {
            cells += ccells;
            for (int k = d; k < no; k += GL) R.cigar[(size_t)c * LH_MAX_CIGAR + k] = cgo[k];
            if (d == 0) {
                const int rid = dev_pos2rid(ix, posf);
                int mismatches = NM - indel_length;
                matches -= mismatches;
                if (mismatches < 0) mismatches = 0;
                i64 pos = Offset, aend = End;
                if (pos != -1 && is_rev) { pos = End + 1; aend = Offset + 1; }
                R.rid[c] = rid; R.pos[c] = pos; R.aend[c] = aend; R.rb[c] = rb; R.re[c] = re; R.reversed[c] = (uint8_t)is_rev; R.score[c] = ar.score;
                R.qb[c] = qb; R.qe[c] = qe; R.nm[c] = NM; R.matches[c] = matches; R.mismatches[c] = mismatches; R.indels[c] = indels;
                R.soft_clipped[c] = soft_clipping; R.soft_clipped_length[c] = soft_clipping_length;
                R.in_filtered[c] = ar.score >= best - o.aln_score_delta;
                R.n_cigar[c] = no; R.n_mm[c] = nmm; R.read_len[c] = l_query;
                R.lap[c] = (dev_single_score(mismatches, indels, soft_clipping, soft_clipping_length) + o.improper_pair_penalty) - o.improper_pair_penalty;
            }
        }
        // the candidates this kernel does not settle: listed for k_aln (the wave reserves the space once)
        const u64 tb = __ballot(towide && d == 0);
        if (tb) {
            int bp = 0;
            if (lane == 0) bp = atomicAdd(wide_count, (int)__popcll(tb));
            bp = wave_readlane(bp, 0) + (int)__popcll(tb & ((1ull << lane) - 1ull));
            if (towide && d == 0) { wide_r[bp] = r; wide_ci[bp] = ci; }
        }
    }
    if (d == 0 && ctr && cells) atomicAdd(&LH_CTR(ctr)->glob_cells, cells);
}

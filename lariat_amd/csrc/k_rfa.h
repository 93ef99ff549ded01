// k_rfa.h — K8: lariat's per-barcode inference on device, one wavefront per barcode.
// Follows go/src/inference/lariat.go: tagBestAlignments (:1466-1549), inferMolecules (:1370-1408),
// markBestAlignmentForReadInMolecule (:1410-1463), scrapMolecules (:1061-1086), the RFA optimizer
// (optimizer/optimizer.go:15-27 -> GenerateMove :1135-1167, fastScore :1179-1307, acceptMove :1331-1368),
// estimateMapQualities (:867-992 incl. moleculeMapqProbabilitySums :767-790, updateAlignmentsMoleculeStatus :687-719,
// calculateLogMoleculePenalty :792-825), markDuplicates (:655-685) and split.go:29-158 (CheckSplitReads).
//
// Device layout: the pointer-heavy Go structures become dense per-barcode tables in an HBM slab owned by the wave:
//   plist[]            filtered candidates grouped by contig (first-seen order) and Go-sorted by position
//   molecule m         = a contiguous run of plist (gap > 50 kb starts a new one); after scrapMolecules renumbered 0..M-1
//   bestT[m*R + r]     best_alignment_for_read of molecule m for local read r (-1: nil)            (OrderedAlignmentMap.Get)
//   act_store[aoff[m]..+alen[m]]  active_alignments of m in OrderedAlignmentMap order (swap-delete / append)
//   act_cand[r], act_slot[r]      the read's single active candidate and its slot in its molecule's list
// The molecule-vs-molecule move scoring (fastScore) is evaluated one SINK per lane, so each lane keeps the reference's
// summation order; the winner is a lexicographic wave reduction (score, sink size, first index) = GenerateMove's fold.
#pragma once
#include "k_aln.h"

struct DInf {   // per-candidate / per-read inference outputs (device)
    uint8_t *active, *is_proper, *bwa_pick, *active_molecule, *duplicate;
    int32_t *molecule_id, *mapq;
    double *mol_diff, *mol_conf, *sum_move;
    i64* mate;          // global candidate index or -1
    int32_t* cand_read; // global read index of the candidate (filled by k_rfa's init)
    // per read
    i64 *active_idx, *second_best_idx, *split_idx;
    double *second_best_score, *as_score, *split_second_best, *split_score;
    int32_t* split_mapq;
};

struct DTieRng {   // xoshiro256** seeded by splitmix64 (see oracle/lariat_oracle.h: Go's math/rand stream is not reproducible offline)
    u64 s0, s1, s2, s3;
};
__device__ __forceinline__ u64 dev_splitmix(u64& x) {
    u64 z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void dev_rng_seed(DTieRng& g, u64 seed) { g.s0 = dev_splitmix(seed); g.s1 = dev_splitmix(seed); g.s2 = dev_splitmix(seed); g.s3 = dev_splitmix(seed); }
__device__ __forceinline__ u64 dev_rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ double dev_rng_f64(DTieRng& g) {
    u64 r = dev_rotl(g.s1 * 5, 7) * 9, t = g.s1 << 17;
    g.s2 ^= g.s0; g.s3 ^= g.s1; g.s1 ^= g.s2; g.s0 ^= g.s3; g.s2 ^= t; g.s3 = dev_rotl(g.s3, 45);
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

// lariat.go:1102-1133
__device__ __forceinline__ int dev_is_pair(const DCand& R, i64 a, i64 b) {
    if (R.reversed[a] == R.reversed[b] || R.rid[a] != R.rid[b]) return 0;
    i64 fwd = R.reversed[a] ? b : a, rev = R.reversed[a] ? a : b;
    i64 dist = R.pos[rev] - R.pos[fwd];
    return dist >= -35 && dist < 750;
}

// lariat.go:599-624; a or m may be -1 (nil)
__device__ __forceinline__ double dev_score_aln(const DCand& R, const DInf& S, double improper, i64 a, i64 m, double lmp) {
    double score = 0.0;
    if (a >= 0) {
        score += (double)(R.mismatches[a] * -2 + R.indels[a] * -3);
        if (R.soft_clipped[a] > 0) { score -= 5.0 * (double)R.soft_clipped[a]; score -= (double)R.soft_clipped_length[a] * 0.5; }
    }
    if (m >= 0) {
        score += (double)(R.mismatches[m] * -2 + R.indels[m] * -3);
        if (R.soft_clipped[m] > 0) { score -= 5.0 * (double)R.soft_clipped[m]; score -= (double)R.soft_clipped_length[m] * 0.5; }
    }
    if (m < 0 || a < 0 || !dev_is_pair(R, a, m)) score += improper;
    if (a >= 0 && !S.active_molecule[a]) score += lmp;
    return score;
}
// lariat.go:590-597
__device__ __forceinline__ double dev_pseudo_score(const DCand& R, i64 a, double lmp) {
    double score = 0.0;
    score -= 10.0;
    score -= ((double)R.read_len[a] - 25.0) * 0.5;
    score += lmp;
    return score;
}

struct RfaTab {   // carved from the wave's slab
    int32_t* plist;      // [NCf] local candidate ids
    int32_t* molraw;     // [NCf] raw molecule id of plist entry
    int32_t* mstart;     // [NCf+1] raw molecule -> first plist index
    int32_t* newid;      // [NCf] raw molecule -> id after scrap or -1
    int32_t* nreads;     // [NCf] raw: number of distinct reads (best_alignment_for_read.Len())
    double* sval;        // [NCf]
    int32_t* seen_rid;   // [ncont+2]
    int32_t* ccnt;       // [ncont+2]
    int32_t* coff;       // [ncont+2]
    // after scrap (M molecules)
    int32_t* seg0;       // [M] plist range of molecule
    int32_t* seg1;
    int32_t* nbest;      // [M]
    int32_t* aoff;       // [M]
    int32_t* alen;       // [M]
    int32_t* act_store;  // [NCf]
    int32_t* act_cand;   // [R]
    int32_t* act_slot;   // [R]
    int32_t* tdel;       // [R]
    int32_t* tset;       // [R]
    int32_t* mflag;      // [M] active_molecule
    double* P;           // [M]
    int32_t* bestT;      // [M*R]
};

// isActiveMolecule, lariat.go:1309-1319
__device__ __forceinline__ int dev_mol_active(int alen, int nbest, int change) {
    double active = (double)(alen + change), potential = (double)nbest;
    if (active <= 4) return 0;
    if (active / potential < 0.1) return 0;
    return 1;
}

// fastScore(source, sink) (lariat.go:1179-1307); returns the score change, *num = reads with an alternative in the sink.
// If tdel != NULL also records the reads that would move (toDelete/toSet).
__device__ __forceinline__ double dev_fast_score(const DCand& R, const DInf& S, const RfaTab& T, i64 c_lo, int r0, int nR, int src, int snk, double lup,
                                                 int* num_out, int32_t* tdel, int32_t* tset, int* nmove) {
    double change = 0, alignment_change = 0;
    int num = 0, nm = 0;
    int n = T.alen[src], ao = T.aoff[src];
    for (int s = 0; s < n; ++s) {
        int a = T.act_store[ao + s];
        int lr = S.cand_read[c_lo + a] - r0;   // local read id
        int t = T.bestT[(size_t)snk * nR + lr];
        if (t >= 0) {
            int ml = lr ^ 1;
            int sm = T.act_cand[ml];
            int source_has_mate = sm >= 0 && S.molecule_id[c_lo + sm] == src;
            int source_has_mate_pair = source_has_mate && dev_is_pair(R, c_lo + a, c_lo + sm);
            int tm = T.bestT[(size_t)snk * nR + ml];
            int sink_has_mate_pair = tm >= 0 && dev_is_pair(R, c_lo + t, c_lo + tm) && source_has_mate;
            if (!source_has_mate_pair || (source_has_mate && sink_has_mate_pair)) {
                if (tdel) { tdel[nm] = lr; tset[nm] = t; }
                nm++;
            }
            alignment_change += R.lap[c_lo + t] - R.lap[c_lo + a];
            if (source_has_mate_pair && !sink_has_mate_pair) alignment_change += lup / 2.0;
            else if (!source_has_mate_pair && sink_has_mate_pair) alignment_change -= lup / 2.0;
            num++;
        }
    }
    int sb = dev_mol_active(T.alen[src], T.nbest[src], 0), sa = dev_mol_active(T.alen[src], T.nbest[src], -num);
    if (!sa && sb) change -= (double)T.nbest[src] * -0.5;
    int kb = dev_mol_active(T.alen[snk], T.nbest[snk], 0), ka = dev_mol_active(T.alen[snk], T.nbest[snk], num);
    if (ka && !kb) change += (double)T.nbest[snk] * -0.5;
    if (T.alen[src] - num == 0 && num > 0) change -= -3.0;
    if (T.alen[snk] == 0 && num > 0) change += -3.0;
    change += alignment_change;
    *num_out = num;
    if (nmove) *nmove = nm;
    return change;
}

#define LH_SPLIT_MAX 64
#define LH_RFA_SORT_LDS 1536   // filtered candidates of a barcode whose position sort is staged in LDS (18 KB)

__global__ void __launch_bounds__(64) k_rfa(DIndex ix, DOpts o, int n_bc, const int32_t* __restrict__ bc_pair_off, const uint8_t* __restrict__ bc_do_rfa,
                                             const u64* __restrict__ name_seed, const i64* __restrict__ cen_start, const i64* __restrict__ cen_end, DCand R, DInf S, i64 cand_cap,
                                             uint8_t* __restrict__ slab_pool, i64 slab_bytes, int32_t* __restrict__ status) {
    __shared__ int32_t shi[8];
    __shared__ double shd[4];
    __shared__ i64 spos[LH_RFA_SORT_LDS];
    __shared__ int32_t sidx[LH_RFA_SORT_LDS];
    int lane = LANE();
    uint8_t* slab = slab_pool + (size_t)blockIdx.x * (size_t)slab_bytes;
    const double improper = o.improper_pair_penalty;
    for (int bc = blockIdx.x; bc < n_bc; bc += gridDim.x) {
        int p0 = bc_pair_off[bc], p1 = bc_pair_off[bc + 1];
        int nR = 2 * (p1 - p0), r0 = 2 * p0;
        i64 c_lo = R.cand_off[r0], c_hi = R.cand_off[r0 + nR];
        int NC = (int)(c_hi - c_lo);
        WAVE_SYNC();
        if (c_hi > cand_cap) continue;   // flagged by k_aln
        // ---- init per-candidate and per-read state (Alignment defaults, lariat.go:1655-1689) ----
        for (int r = lane; r < nR; r += 64) {
            for (i64 g = R.cand_off[r0 + r]; g < R.cand_off[r0 + r + 1]; ++g) {
                S.active[g] = 0; S.is_proper[g] = 0; S.bwa_pick[g] = 0; S.active_molecule[g] = 0; S.duplicate[g] = 0; S.molecule_id[g] = -1; S.mapq[g] = 0;
                S.mol_diff[g] = 0; S.mol_conf[g] = 0.00075 * 0.025; S.sum_move[g] = 1.0; S.mate[g] = -1; S.cand_read[g] = r0 + r;
            }
            S.active_idx[r0 + r] = -1; S.second_best_idx[r0 + r] = -1; S.split_idx[r0 + r] = -1; S.second_best_score[r0 + r] = 0; S.as_score[r0 + r] = 0;
            S.split_second_best[r0 + r] = 0; S.split_score[r0 + r] = 0; S.split_mapq[r0 + r] = 0;
        }
        WAVE_SYNC();
        // ---- tagBestAlignments: one lane per pair.  Read 2 of a pair is always "touched" by read 1 (every read has >= 1
        // filtered candidate), so only read 1's scan decides; its RNG stream is seeded from the read name. ----
        for (int p = p0 + lane; p < p1; p += 64) {
            int ra = 2 * p, rb = 2 * p + 1;
            DTieRng rng;
            dev_rng_seed(rng, name_seed[p]);
            double best = -1.7976931348623157e308;
            i64 ba = -1, bm = -1;
            for (i64 a = R.cand_off[ra]; a < R.cand_off[ra + 1]; ++a) {
                if (!R.in_filtered[a]) continue;
                for (i64 m = R.cand_off[rb]; m < R.cand_off[rb + 1]; ++m) {
                    if (!R.in_filtered[m]) continue;
                    double total = dev_score_aln(R, S, improper, a, m, 0.0) + (dev_rng_f64(rng) / 2.0);
                    if (total > best) { best = total; ba = a; bm = m; }
                }
            }
            S.active[ba] = 1; S.bwa_pick[ba] = 1;
            if (dev_is_pair(R, ba, bm)) { S.is_proper[ba] = 1; S.is_proper[bm] = 1; }
            S.active[bm] = 1; S.bwa_pick[bm] = 1;
        }
        WAVE_SYNC();
        // ---- positions: filtered candidates grouped by contig in first-seen order (lane 0), then Go-sorted by pos ----
        // slab carve (sizes depend on NC, nR)
        size_t so = 0;
        RfaTab T;
#define CARVE(ptr, type, count) { so = (so + 7) & ~(size_t)7; T.ptr = (type*)(slab + so); so += sizeof(type) * (size_t)(count); }
        int ncmax = ix.n_contigs + 2;
        CARVE(plist, int32_t, NC) CARVE(molraw, int32_t, NC) CARVE(mstart, int32_t, NC + 1) CARVE(newid, int32_t, NC) CARVE(nreads, int32_t, NC)
        CARVE(sval, double, NC) CARVE(seen_rid, int32_t, ncmax) CARVE(ccnt, int32_t, ncmax) CARVE(coff, int32_t, ncmax + 1)
        CARVE(seg0, int32_t, NC) CARVE(seg1, int32_t, NC) CARVE(nbest, int32_t, NC) CARVE(aoff, int32_t, NC) CARVE(alen, int32_t, NC)
        CARVE(act_store, int32_t, NC) CARVE(act_cand, int32_t, nR) CARVE(act_slot, int32_t, nR) CARVE(tdel, int32_t, nR) CARVE(tset, int32_t, nR)
        CARVE(mflag, int32_t, NC) CARVE(P, double, NC)
        so = (so + 7) & ~(size_t)7;
        T.bestT = (int32_t*)(slab + so);
        size_t best_cap = ((size_t)slab_bytes > so) ? ((size_t)slab_bytes - so) / 4 : 0;
#undef CARVE
        if (so > (size_t)slab_bytes) {   // barcode too large for the slab
            if (lane == 0) status[r0] |= LH_ST_POOL_OVERFLOW;
            continue;
        }
        if (lane == 0) {
            int ncont = 0, nf = 0;
            for (int a = 0; a < NC; ++a) {
                if (!R.in_filtered[c_lo + a]) continue;
                int rid = R.rid[c_lo + a], k;
                for (k = 0; k < ncont; ++k) if (T.seen_rid[k] == rid) break;
                if (k == ncont) { T.seen_rid[ncont] = rid; T.ccnt[ncont] = 0; ncont++; }
                T.ccnt[k]++; nf++;
            }
            int acc = 0;
            for (int k = 0; k < ncont; ++k) { T.coff[k] = acc; acc += T.ccnt[k]; T.ccnt[k] = 0; }
            T.coff[ncont] = acc;
            for (int a = 0; a < NC; ++a) {
                if (!R.in_filtered[c_lo + a]) continue;
                int rid = R.rid[c_lo + a], k;
                for (k = 0; k < ncont; ++k) if (T.seen_rid[k] == rid) break;
                T.plist[T.coff[k] + T.ccnt[k]++] = a;
            }
            shi[0] = ncont; shi[1] = nf;
        }
        WAVE_SYNC();
        int ncont = shi[0], NCf = shi[1];
        // sort.Sort(ByPosition) per contig (lariat.go:1545-1547), one lane per contig; keys staged in LDS when they fit
        if (NCf <= LH_RFA_SORT_LDS) {
            for (int i = lane; i < NCf; i += 64) { sidx[i] = T.plist[i]; spos[i] = R.pos[c_lo + T.plist[i]]; }
            WAVE_SYNC();
            for (int k = lane; k < ncont; k += 64) {
                int b0 = T.coff[k], n = T.coff[k + 1] - b0;
                i64* kp = spos + b0;
                int32_t* ip = sidx + b0;
                dev_gosort(n, [&](int i, int j) { return kp[i] < kp[j]; },
                           [&](int i, int j) { i64 t = kp[i]; kp[i] = kp[j]; kp[j] = t; int u = ip[i]; ip[i] = ip[j]; ip[j] = u; });
            }
            WAVE_SYNC();
            for (int i = lane; i < NCf; i += 64) T.plist[i] = sidx[i];
        } else {
            for (int k = lane; k < ncont; k += 64) {
                int32_t* pl = T.plist + T.coff[k];
                int n = T.coff[k + 1] - T.coff[k];
                dev_gosort(n, [&](int i, int j) { return R.pos[c_lo + pl[i]] < R.pos[c_lo + pl[j]]; }, [&](int i, int j) { int t = pl[i]; pl[i] = pl[j]; pl[j] = t; });
            }
        }
        WAVE_SYNC();
        int do_rfa = bc_do_rfa[bc] != 0;
        int M = 0;
        if (do_rfa) {
            // ---- inferMolecules: a gap > 50 kb (or a new contig list) starts a molecule ----
            if (lane == 0) {
                int m = -1;
                for (int k = 0; k < ncont; ++k)
                    for (int i = T.coff[k]; i < T.coff[k + 1]; ++i) {
                        if (i == T.coff[k] || R.pos[c_lo + T.plist[i]] - R.pos[c_lo + T.plist[i - 1]] > 50000) { ++m; T.mstart[m] = i; }
                        T.molraw[i] = m;
                    }
                T.mstart[m + 1] = NCf;
                shi[2] = m + 1;
            }
            WAVE_SYNC();
            int Mraw = shi[2];
            // ---- markBestAlignmentForReadInMolecule, step 1: best pair score of every entry inside its molecule ----
            for (int i = lane; i < NCf; i += 64) {
                int a = T.plist[i], m = T.molraw[i];
                int mate_read = S.cand_read[c_lo + a] ^ 1;
                double best = -1.7976931348623157e308;
                int found = 0;
                for (int j = T.mstart[m]; j < T.mstart[m + 1]; ++j) {
                    int b = T.plist[j];
                    if (S.cand_read[c_lo + b] != mate_read) continue;
                    found = 1;
                    double s = dev_score_aln(R, S, improper, c_lo + a, c_lo + b, 0.0);
                    if (s > best) best = s;
                }
                T.sval[i] = found ? best : R.lap[c_lo + a];
            }
            WAVE_SYNC();
            // distinct reads and "has an active alignment" per raw molecule (lane per molecule)
            for (int m = lane; m < Mraw; m += 64) {
                int nr = 0, has = 0;
                for (int i = T.mstart[m]; i < T.mstart[m + 1]; ++i) {
                    int a = T.plist[i], rd = S.cand_read[c_lo + a], first = 1;
                    for (int j = T.mstart[m]; j < i; ++j) if (S.cand_read[c_lo + T.plist[j]] == rd) { first = 0; break; }
                    nr += first;
                    has |= S.active[c_lo + a];
                }
                T.nreads[m] = nr; T.newid[m] = has;
            }
            WAVE_SYNC();
            // ---- scrapMolecules: keep molecules with an active alignment, renumber ----
            if (lane == 0) {
                int cnt = 0, ao = 0;
                for (int m = 0; m < Mraw; ++m) {
                    if (T.newid[m]) {
                        T.newid[m] = cnt; T.seg0[cnt] = T.mstart[m]; T.seg1[cnt] = T.mstart[m + 1]; T.nbest[cnt] = T.nreads[m]; T.aoff[cnt] = ao; T.alen[cnt] = 0;
                        ao += T.nreads[m]; cnt++;
                    } else T.newid[m] = -1;
                }
                shi[3] = cnt;
                shi[4] = ((size_t)cnt * (size_t)nR > best_cap) ? 1 : 0;
            }
            WAVE_SYNC();
            M = shi[3];
            if (shi[4]) {   // molecule table does not fit the slab
                if (lane == 0) status[r0] |= LH_ST_POOL_OVERFLOW;
                continue;
            }
            for (size_t x = lane; x < (size_t)M * nR; x += 64) T.bestT[x] = -1;
            for (int r = lane; r < nR; r += 64) { T.act_cand[r] = -1; T.act_slot[r] = -1; }
            for (int i = lane; i < NCf; i += 64) {
                int nm = T.newid[T.molraw[i]];
                S.molecule_id[c_lo + T.plist[i]] = nm;
            }
            WAVE_SYNC();
            // step 2: per molecule, reads in first-occurrence order: best alignment (earliest maximum) and the active list
            for (int m = lane; m < M; m += 64) {
                int b0 = T.seg0[m], b1 = T.seg1[m], na = 0;
                for (int i = b0; i < b1; ++i) {
                    int a = T.plist[i], rd = S.cand_read[c_lo + a], first = 1;
                    for (int j = b0; j < i; ++j) if (S.cand_read[c_lo + T.plist[j]] == rd) { first = 0; break; }
                    if (!first) continue;
                    double best = -1.7976931348623157e308;
                    int bi = -1, act = -1;
                    for (int j = i; j < b1; ++j) {
                        int b = T.plist[j];
                        if (S.cand_read[c_lo + b] != rd) continue;
                        if (T.sval[j] > best) { best = T.sval[j]; bi = b; }
                        if (S.active[c_lo + b]) act = b;
                    }
                    int lr = rd - r0;
                    T.bestT[(size_t)m * nR + lr] = bi;
                    if (act >= 0) { T.act_store[T.aoff[m] + na] = act; T.act_cand[lr] = act; T.act_slot[lr] = na; na++; }
                }
                T.alen[m] = na;
            }
            WAVE_SYNC();
            // setMoleculeDifferences(candidate_molecules, false) before the optimizer (lariat.go:503): alignments that are
            // active NOW keep this value even if a later move deactivates them
            for (int m = lane; m < M; m += 64) {
                int diffs = 0;
                for (int k = 0; k < T.alen[m]; ++k) diffs += R.mismatches[c_lo + T.act_store[T.aoff[m] + k]];
                double diff = (double)diffs / (double)T.alen[m];
                for (int k = 0; k < T.alen[m]; ++k) S.mol_diff[c_lo + T.act_store[T.aoff[m] + k]] = diff;
            }
            WAVE_SYNC();
            // ---- optimizer.Optimize(opt, 1, 2, 4*M): 8*M greedy molecule moves ----
            int source = 0;
            for (int it = 0; it < 8 * M; ++it) {
                if (T.alen[source] == 0) { source = (source + 1) % M; continue; }
                double bs = -1.7976931348623157e308;
                int bl = -1, bi = 0x7fffffff;
                for (int i = lane; i < M; i += 64) {
                    if (i == source) continue;
                    int num;
                    double sc = dev_fast_score(R, S, T, c_lo, r0, nR, source, i, improper, &num, (int32_t*)0, (int32_t*)0, (int*)0);
                    if (num > 0 && (sc > bs || (sc == bs && T.alen[i] > bl))) { bs = sc; bl = T.alen[i]; bi = i; }
                }
                for (int msk = 32; msk >= 1; msk >>= 1) {   // lexicographic max of (score, sink size), first index on full ties
                    double os = __shfl_xor(bs, msk);
                    int ol = __shfl_xor(bl, msk), oi = __shfl_xor(bi, msk);
                    int take = oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && (ol > bl || (ol == bl && oi < bi))));
                    if (take) { bs = os; bl = ol; bi = oi; }
                }
                if (bi != 0x7fffffff && (bs > 0 || (bs == 0 && bl > T.alen[source]))) {
                    if (lane == 0) {   // acceptMove: recompute the move list, then apply it in order
                        int num, nmv;
                        dev_fast_score(R, S, T, c_lo, r0, nR, source, bi, improper, &num, T.tdel, T.tset, &nmv);
                        for (int k = 0; k < nmv; ++k) {
                            int lr = T.tdel[k], t = T.tset[k];
                            int a = T.act_cand[lr], slot = T.act_slot[lr];
                            int last = T.alen[source] - 1;
                            if (T.alen[source] > 1) {
                                int mv = T.act_store[T.aoff[source] + last];
                                T.act_store[T.aoff[source] + slot] = mv;
                                T.act_slot[S.cand_read[c_lo + mv] - r0] = slot;
                            }
                            T.alen[source] = last;
                            T.act_store[T.aoff[bi] + T.alen[bi]] = t;
                            T.act_slot[lr] = T.alen[bi]; T.act_cand[lr] = t;
                            T.alen[bi]++;
                            S.active[c_lo + a] = 0; S.active[c_lo + t] = 1;
                        }
                    }
                }
                WAVE_SYNC();
                source = (source + 1) % M;
            }
            // ---- moleculeMapqProbabilitySums ----
            for (int s = 0; s < M; ++s) {
                for (int t = lane; t < M; t += 64) {
                    int num;
                    T.P[t] = t == s ? 0.0 : pow(10.0, dev_fast_score(R, S, T, c_lo, r0, nR, s, t, improper, &num, (int32_t*)0, (int32_t*)0, (int*)0));
                }
                WAVE_SYNC();
                for (int k = lane; k < T.alen[s]; k += 64) {
                    int a = T.act_store[T.aoff[s] + k], lr = S.cand_read[c_lo + a] - r0;
                    double sum = S.sum_move[c_lo + a];
                    for (int t = 0; t < M; ++t)
                        if (t != s && T.bestT[(size_t)t * nR + lr] >= 0) sum += T.P[t];
                    S.sum_move[c_lo + a] = sum;
                }
                WAVE_SYNC();
            }
            // ---- updateAlignmentsMoleculeStatus: confidences, differences, active molecules ----
            for (int m = lane; m < M; m += 64) {
                double conf = (double)T.alen[m] / (double)T.nbest[m];
                int soft = 0, diffs = 0;
                for (int k = 0; k < T.alen[m]; ++k) {
                    int a = T.act_store[T.aoff[m] + k];
                    if (R.soft_clipped[c_lo + a] > 0) soft++;
                    diffs += R.mismatches[c_lo + a];
                }
                double diff = (double)diffs / (double)T.alen[m];
                for (int k = 0; k < T.alen[m]; ++k) {
                    int a = T.act_store[T.aoff[m] + k];
                    S.mol_conf[c_lo + a] = conf; S.mol_diff[c_lo + a] = diff;
                }
                T.mflag[m] = (T.alen[m] - soft > 4 && conf > 0.1) ? 1 : 0;
            }
            WAVE_SYNC();
            for (int i = lane; i < NCf; i += 64) {
                int a = T.plist[i], m = S.molecule_id[c_lo + a];
                if (m != -1) S.active_molecule[c_lo + a] = (uint8_t)T.mflag[m];
            }
            WAVE_SYNC();
        }
        // ---- calculateLogMoleculePenalty ----
        if (lane == 0) {
            double lmp = 0.0;
            if (do_rfa && M > 0) {
                double dnaLength = 1000.0;
                for (int m = 0; m < M; ++m) {
                    if (T.mflag[m]) {
                        i64 smallest = 0x7fffffffffffffffll, biggest = -1;
                        for (int k = 0; k < T.alen[m]; ++k) {
                            i64 p = R.pos[c_lo + T.act_store[T.aoff[m] + k]];
                            if (p > biggest) biggest = p;
                            if (p < smallest) smallest = p;
                        }
                        if (biggest >= smallest) dnaLength += (double)(biggest - smallest) + 1000.0;
                    } else {
                        for (int k = 0; k < T.alen[m]; ++k) {
                            i64 g = c_lo + T.act_store[T.aoff[m] + k];
                            dnaLength += (double)(R.aend[g] - R.pos[g]) * 2.0;
                        }
                    }
                }
                lmp = log10(dnaLength / o.genome_length * 0.05);
            }
            shd[0] = lmp;
        }
        WAVE_SYNC();
        double lmp = shd[0];
        // ---- per read: link active mates (lariat.go:892-900).  Done for all reads before any scoring that reads mate links. ----
        for (int r = lane; r < nR; r += 64) {
            i64 act = -1;
            for (i64 a = R.cand_off[r0 + r]; a < R.cand_off[r0 + r + 1]; ++a) if (R.in_filtered[a] && S.active[a]) act = a;
            S.active_idx[r0 + r] = act;
        }
        WAVE_SYNC();
        for (int r = lane; r < nR; r += 64) {
            i64 a = S.active_idx[r0 + r], m = S.active_idx[r0 + (r ^ 1)];
            S.mate[a] = m;
        }
        WAVE_SYNC();
        // ---- estimateMapQualities per read (lariat.go:887-990), one lane per read ----
        for (int r = lane; r < nR; r += 64) {
            int gr = r0 + r, gm = r0 + (r ^ 1);
            i64 a0 = R.cand_off[gr], a1 = R.cand_off[gr + 1], m0 = R.cand_off[gm], m1 = R.cand_off[gm + 1];
            double top[15];
            int ntop = 0;
#define TOP_PUSH(v_)                                                                   \
    {                                                                                  \
        double v = (v_);                                                               \
        int k_ = ntop < 15 ? ntop : 15;                                                \
        if (ntop < 15 || v > top[14]) {                                                \
            if (ntop < 15) ntop++; else k_ = 14;                                       \
            while (k_ > 0 && top[k_ - 1] < v) { top[k_] = top[k_ - 1]; k_--; }         \
            top[k_] = v;                                                               \
        }                                                                              \
    }
            // pseudo-count entry (appendPsuedocountAlignmentScore): first filtered alignment of the read + best single mate
            i64 first = -1;
            for (i64 a = a0; a < a1; ++a) if (R.in_filtered[a]) { first = a; break; }
            double bestSingle = -1.7976931348623157e308;
            for (i64 m = m0; m < m1; ++m) {
                if (!R.in_filtered[m]) continue;
                double s = dev_score_aln(R, S, improper, -1, m, lmp);
                if (s > bestSingle) bestSingle = s;
            }
            double pseudo = bestSingle + dev_pseudo_score(R, first, lmp);
            TOP_PUSH(pseudo)
            for (i64 a = a0; a < a1; ++a) {   // best pair score of every alignment
                if (!R.in_filtered[a]) continue;
                double best = -1.7976931348623157e308;
                for (i64 m = m0; m < m1; ++m) {
                    if (!R.in_filtered[m]) continue;
                    double s = dev_score_aln(R, S, improper, a, m, lmp);
                    if (s > best) best = s;
                }
                TOP_PUSH(best)
            }
            // second best (lariat.go:917-943); sets mate_alignment of the inactive alignment it selects
            double sb_raw = pseudo, sb_lp = -1000.0;
            i64 sb_aln = -1;
            for (i64 a = a0; a < a1; ++a) {
                if (!R.in_filtered[a] || S.active[a]) continue;
                for (i64 m = m0; m < m1; ++m) {
                    if (!R.in_filtered[m]) continue;
                    double s = dev_score_aln(R, S, improper, a, m, lmp);
                    if (s > sb_lp) { sb_lp = s; sb_raw = dev_score_aln(R, S, improper, a, m, 0.0); sb_aln = a; S.mate[a] = m; }
                }
            }
            i64 act = S.active_idx[gr];
            S.second_best_idx[gr] = sb_aln; S.second_best_score[gr] = sb_raw;
            S.as_score[gr] = dev_score_aln(R, S, improper, act, S.mate[act], 0.0);
            double total = 0;
            for (int k = 0; k < ntop; ++k) total += pow(10.0, top[k]);
            for (i64 a = a0; a < a1; ++a) {
                if (!R.in_filtered[a]) continue;
                double score = dev_score_aln(R, S, improper, a, S.mate[a], lmp);
                double mapq = -10.0 * log10(1.0 - pow(10.0, score) / total);
                double mmq = -10.0 * log10(1.0 - (1.0 / S.sum_move[a]));
                mapq = (mapq != mapq || mmq != mmq) ? mapq + mmq : (mapq < mmq ? mapq : mmq);   // math.Min propagates NaN
                mapq = (mapq != mapq) ? mapq : (60.0 < mapq ? 60.0 : mapq);
                i64 cs = -1, ce = -1;
                if (R.rid[a] >= 0 && cen_start[R.rid[a]] >= 0) { cs = cen_start[R.rid[a]]; ce = cen_end[R.rid[a]]; }
                if (R.pos[a] > cs && R.pos[a] <= ce) mapq = 0.0;
                S.mapq[a] = (mapq != mapq) ? (int)0x80000000 : (int)mapq;
            }
#undef TOP_PUSH
        }
        WAVE_SYNC();
        // ---- markDuplicates: first-seen wins on (read1?, reversed, contig, pos, mate contig, mate pos) in read order ----
        for (int r = lane; r < nR; r += 64) {
            i64 a = S.active_idx[r0 + r], m = S.mate[a];
            int dup = 0;
            for (int q = 0; q < r && !dup; ++q) {
                if ((q & 1) != (r & 1)) continue;
                i64 b = S.active_idx[r0 + q], bm = S.mate[b];
                if (R.reversed[a] == R.reversed[b] && R.rid[a] == R.rid[b] && R.pos[a] == R.pos[b] && R.rid[m] == R.rid[bm] && R.pos[m] == R.pos[bm]) dup = 1;
            }
            S.duplicate[a] = (uint8_t)dup;
        }
        WAVE_SYNC();
        // ---- CheckSplitReads / GetSplitAlignment over the unfiltered candidates (split.go) ----
        for (int r = lane; r < nR; r += 64) {
            int gr = r0 + r;
            i64 P = S.active_idx[gr];
            if (R.pos[P] == -1) continue;
            int Ps = R.qb[P], Pe = R.qe[P];
            if (Ps > Pe) { int t = Ps; Ps = Pe; Pe = t; }
            if ((Pe - Ps) > R.read_len[P] - 15) continue;
            i64 cidx[LH_SPLIT_MAX];
            int ncand = 0, ovf = 0;
            for (i64 sc = R.cand_off[gr]; sc < R.cand_off[gr + 1]; ++sc) {
                if (S.active[sc] || R.pos[sc] == -1) continue;
                int Ss = R.qb[sc], Se = R.qe[sc], overlap;
                if (Ss > Se) { int t = Ss; Ss = Se; Se = t; }
                if ((Ps < Ss && Pe > Se) || (Ss < Ps && Se > Pe)) continue;
                else if (Ps < Ss) overlap = Pe - Ss;
                else overlap = Se - Ps;
                if (overlap < (Se - Ss) / 2) {
                    int prop = dev_is_pair(R, sc, S.mate[P]);
                    S.is_proper[sc] = (uint8_t)prop;
                    if (R.score[sc] >= 36 || prop) { if (ncand < LH_SPLIT_MAX) cidx[ncand++] = sc; else ovf = 1; }
                }
            }
            if (ovf) status[gr] |= LH_ST_POOL_OVERFLOW;
            if (ncand == 0) continue;
            dev_gosort(ncand, [&](int i, int j) { return R.score[cidx[i]] > R.score[cidx[j]]; }, [&](int i, int j) { i64 t = cidx[i]; cidx[i] = cidx[j]; cidx[j] = t; });
            i64 c = cidx[0];
            double mapq;
            double second_best = dev_score_aln(R, S, improper, P, -1, 0.0) + dev_pseudo_score(R, c, 0.0);
            if (ncand > 1) { mapq = (double)(R.score[cidx[0]] - R.score[cidx[1]]); second_best = dev_score_aln(R, S, improper, P, cidx[1], 0.0); }
            else mapq = (double)R.score[cidx[0]];
            i64 cs = -1, ce = -1;
            if (R.rid[c] >= 0 && cen_start[R.rid[c]] >= 0) { cs = cen_start[R.rid[c]]; ce = cen_end[R.rid[c]]; }
            if (R.pos[c] > cs && R.pos[c] <= ce) mapq = 0.0;
            if (mapq > 60) mapq = 60;
            S.mapq[c] = (int)mapq;
            S.split_idx[gr] = c; S.split_mapq[gr] = (int)mapq; S.split_second_best[gr] = second_best;
            S.split_score[gr] = dev_score_aln(R, S, improper, c, S.mate[P], 0.0);
        }
        WAVE_SYNC();
    }
}

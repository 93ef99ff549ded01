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

// ---- Go's math/rand source (rng.go), the jitter stream of tagBestAlignments (lariat.go:1486 rand.New(rand.NewSource(seed)),
// :1499,:1510 random.Float64()/2.0).  Additive lagged Fibonacci x[n] = x[n-607] + x[n-273] mod 2^64; Seed fills the 607-entry
// state from a Lehmer stream (x' = 48271 x mod 2^31-1, three values per entry at shifts 40/20/0) XORed with the table rngCooked
// (go_rng_cooked.inc: derived from its definition by tools/gen_go_rng_cooked.py, pinned by Go's Seed(1) value stream).
// A read draws once per (alignment, mate alignment) combination — a handful — so the device never builds the state for
// them: draw k <= 273 is vec0[334-k] + vec0[607-k] of the SEEDED state (neither slot has been overwritten yet), and a seeded
// entry is three jumps of the Lehmer stream (powers of the multiplier: go_rng_lehmer_pow.inc).  Reads with more draws
// materialise the state in the wave's slab (one u64 [607][64 lanes] ring) and step it as rng.go does.
#ifdef LH_EMU
static const u64 lh_go_cooked[607] = {
#include "go_rng_cooked.inc"
};
static const uint32_t lh_go_lpow[1848] = {
#include "go_rng_lehmer_pow.inc"
};
#else
__device__ const u64 lh_go_cooked[607] = {
#include "go_rng_cooked.inc"
};
__device__ const uint32_t lh_go_lpow[1848] = {
#include "go_rng_lehmer_pow.inc"
};
#endif
#define LH_GO_LEN 607
#define LH_GO_TAP 273
#define LH_GO_FAST_DRAWS 256   // draws a read may make on the state-free path (<= 273; the slack covers Float64's redraws)
#define LH_GO_RING_BYTES ((size_t)LH_GO_LEN * 64 * 8)
struct DGoRng {
    uint32_t x0;   // normalised seed: the Lehmer stream's x_0
    int k;         // draws made
    u64* ring;     // null: state-free path; else this lane's state, entry i at ring[i * 64]
    int tap, feed;
};
__device__ __forceinline__ uint32_t dev_go_mulmod31(uint32_t a, uint32_t b) {   // a * b mod (2^31 - 1)
    u64 p = (u64)a * (u64)b;
    u64 r = (p & 0x7fffffffull) + (p >> 31);
    r = (r & 0x7fffffffull) + (r >> 31);
    if (r >= 0x7fffffffull) r -= 0x7fffffffull;
    return (uint32_t)r;
}
__device__ __forceinline__ u64 dev_go_vec0(uint32_t x0, int i) {   // rngSource.Seed's vec[i]
    u64 a = dev_go_mulmod31(x0, lh_go_lpow[21 + 3 * i]), b = dev_go_mulmod31(x0, lh_go_lpow[22 + 3 * i]), c = dev_go_mulmod31(x0, lh_go_lpow[23 + 3 * i]);
    return ((a << 40) ^ (b << 20) ^ c) ^ lh_go_cooked[i];
}
__device__ __forceinline__ void dev_go_seed(DGoRng& g, u64 seed_bits, u64* ring) {
    i64 seed = (i64)seed_bits % 2147483647ll;   // rng.go Seed: int64 remainder (sign of the dividend), then made positive
    if (seed < 0) seed += 2147483647ll;
    if (seed == 0) seed = 89482311;
    g.x0 = (uint32_t)seed; g.k = 0; g.ring = ring; g.tap = 0; g.feed = LH_GO_LEN - LH_GO_TAP;
    if (ring) for (int i = 0; i < LH_GO_LEN; ++i) ring[(size_t)i * 64] = dev_go_vec0(g.x0, i);
}
__device__ __forceinline__ u64 dev_go_u64(DGoRng& g) {   // rngSource.Uint64
    ++g.k;
    if (!g.ring) return dev_go_vec0(g.x0, LH_GO_LEN - LH_GO_TAP - g.k) + dev_go_vec0(g.x0, LH_GO_LEN - g.k);   // k <= 273 (caller's contract)
    if (--g.tap < 0) g.tap += LH_GO_LEN;
    if (--g.feed < 0) g.feed += LH_GO_LEN;
    u64 x = g.ring[(size_t)g.feed * 64] + g.ring[(size_t)g.tap * 64];
    g.ring[(size_t)g.feed * 64] = x;
    return x;
}
__device__ __forceinline__ double dev_go_f64(DGoRng& g) {   // Rand.Float64: float64(Int63()) / (1<<63), redrawn if it rounds to 1
    for (;;) {
        double f = (double)(i64)(dev_go_u64(g) & 0x7fffffffffffffffull) * (1.0 / 9223372036854775808.0);
        if (f != 1.0) return f;
    }
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
    int32_t* seen_rid;   // [ncont+2]   (contig tables live here when the index has more contigs than the LDS copy holds)
    int32_t* ccnt;       // [ncont+2]
    int32_t* coff;       // [ncont+2]
    int32_t* kidx;       // [NC] local candidate -> contig slot (first-seen order) or -1 (not filtered)
    int32_t* molc;       // [NC] local candidate -> raw molecule or -1
    int32_t* ppos;       // [NC] local candidate -> plist index
    int32_t* rdl;        // [NCf] local read of plist entry
    int32_t* firstf;     // [NCf] entry is its read's first occurrence inside its molecule
    int32_t* actc;       // [NCf+1] step 1: exclusive count of active entries; step 2: first entries' active candidate or -1
    int32_t* psum;       // [NCf+1] step 1: exclusive count of first entries; step 2: of first entries with an active candidate
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
    u64 *dk0, *dk1, *dk2, *dk3;   // [R] markDuplicates keys
    int32_t* htab;       // [<= 4R] open-addressing table over the keys
    int32_t* gstk;       // [3 * LH_GOSORT_STK * 64] the lanes' stacks of dev_gosort
    i64* spl;            // (k_rfa_post) [LH_SPLIT_MAX * 64] a read's split candidates, entry i of lane l at spl[i * 64 + l] (a private array of 64 was 512 B of scratch per lane)
    int32_t* bestT;      // [R*M] best_alignment_for_read of molecule m for local read r at [r*M + m]; bit 30: it pairs with
                         // the molecule's best alignment of the mate read (static after markBest); -1: nil
};
#define RFA_T_MASK 0x3fffffff
#define RFA_T_PAIR 0x40000000

// isActiveMolecule, lariat.go:1309-1319
__device__ __forceinline__ int dev_mol_active(int alen, int nbest, int change) {
    double active = (double)(alen + change), potential = (double)nbest;
    if (active <= 4) return 0;
    if (active / potential < 0.1) return 0;
    return 1;
}

#define LH_SPLIT_MAX 64
#ifndef LH_RFA_SORT_LDS
#define LH_RFA_SORT_LDS 768   // filtered candidates of a barcode whose position sort is staged in LDS (9 KB: 16 single-wave blocks per CU)
#endif
#define LH_RFA_LDS_BYTES (LH_RFA_SORT_LDS * 12)

static_assert(LH_RFA_LDS_BYTES >= 15 * 64 * (int)sizeof(double), "estimateMapQualities keeps a read's top-15 scores per lane in lds_raw (top[k * 64 + lane])");
#define LH_RFA_NCONT_LDS 1024  // contig slots of a barcode kept in LDS while grouping (index with more contigs: slab copy)
#define LH_RFA_SRC_CHUNK 256   // source-molecule alignments staged per pass of fastScore
#ifndef LH_RFA_MOL_LDS_MIN
#ifndef LH_RFA_TIE_LDS
#define LH_RFA_TIE_LDS (LH_RFA_LDS_BYTES / 4 < 4096 ? LH_RFA_LDS_BYTES / 4 : 4096)   // entries of a contig list with two equal positions that Go's algorithm sorts as 32-bit words in LDS (test builds: 100)
#endif
#ifndef LH_RFA_NET_BLOCK
#define LH_RFA_NET_BLOCK 1024   // places of the position sort's network that run in LDS at a time (a power of two; test builds: 64)
#endif
static_assert(LH_RFA_NET_BLOCK >= 64 && (LH_RFA_NET_BLOCK & (LH_RFA_NET_BLOCK - 1)) == 0 && LH_RFA_NET_BLOCK * 8 <= LH_RFA_LDS_BYTES, "the network's block lives in lds_raw");
#define LH_RFA_MOL_LDS_MIN 24   // entries of a raw molecule from which on markBest's first step scans the molecule from LDS (up to LH_RFA_LDS_BYTES / 24 entries)
#endif

// fastScore(source, sink) (lariat.go:1179-1307), evaluated by the whole wave: lane L scores sink `snk` (< 0: idle lane).
// Everything that depends on the source alone (the read, whether its mate is active in the source and pairs with it,
// its log alignment probability) is computed once by the lanes in parallel and staged in LDS; every lane then walks the
// source's active alignments in OrderedAlignmentMap order, so each sink's sums keep the reference's order of additions.
// Returns the score change, *num_out = reads with an alternative in the sink; with `record` the lane also writes the
// reads that would move (toDelete / toSet) to T.tdel / T.tset and their number to *nmove.
// (r06, late) `staged`: the source's alignments are in LDS already — the caller has scored another 64 sinks of the SAME source since nothing changed (a source of up to
// LH_RFA_SRC_CHUNK alignments: longer ones are staged chunk by chunk on every call).  Staging was repeated per 64 sinks: half of a call on a barcode of hundreds of molecules.
__device__ __forceinline__ double dev_fast_score_w(const DCand& R, const DInf& S, const RfaTab& T, i64 c_lo, int r0, int M, int src, int snk, double lup,
                                                   int32_t* sLr, int32_t* sFl, double* sLap, int* num_out, int record, int* nmove, int staged = 0) {
    const int lane = LANE();
    double change = 0, alignment_change = 0;
    int num = 0, nm = 0;
    const int n = T.alen[src], ao = T.aoff[src];
    const int reuse = staged && n <= LH_RFA_SRC_CHUNK;
    for (int cb = 0; cb < n; cb += LH_RFA_SRC_CHUNK) {
        int cn = n - cb < LH_RFA_SRC_CHUNK ? n - cb : LH_RFA_SRC_CHUNK;
        if (!reuse) {
            EMU_SYNC();   // the previous chunk has been consumed
            for (int s = lane; s < cn; s += 64) {
                int a = T.act_store[ao + cb + s];
                int lr = S.cand_read[c_lo + a] - r0;
                int sm = T.act_cand[lr ^ 1];
                int hm = sm >= 0 && S.molecule_id[c_lo + sm] == src;          // source_has_mate
                int hp = hm && dev_is_pair(R, c_lo + a, c_lo + sm);           // source_has_mate_pair
                sLr[s] = lr; sFl[s] = hm | hp << 1; sLap[s] = R.lap[c_lo + a];
            }
            WAVE_SYNC();
        }
        if (snk >= 0) {
            // (r06, late) four of the source's reads at a time: their table entries, then the log probabilities those name, are read before the first sum is touched — the
            // additions stay in the map's order; a table read and a dependent gather per read, one after the other, were the step of this loop on large barcodes
            for (int s0 = 0; s0 < cn; s0 += 4) {
                int tq4[4];
                double lap4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) tq4[u] = s0 + u < cn ? T.bestT[(size_t)sLr[s0 + u] * M + snk] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) lap4[u] = tq4[u] >= 0 ? R.lap[c_lo + (tq4[u] & RFA_T_MASK)] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tq = tq4[u], s = s0 + u;
                    if (tq >= 0) {
                        int t = tq & RFA_T_MASK, fl = sFl[s];
                        int hm = fl & 1, hp = fl >> 1;
                        int skp = (tq & RFA_T_PAIR) && hm;                    // sink_has_mate_pair
                        if (!hp || (hm && skp)) {
                            if (record) { T.tdel[nm] = sLr[s]; T.tset[nm] = t; }
                            nm++;
                        }
                        alignment_change += lap4[u] - sLap[s];
                        if (hp && !skp) alignment_change += lup / 2.0;
                        else if (!hp && skp) alignment_change -= lup / 2.0;
                        num++;
                    }
                }
            }
        }
    }
    if (snk >= 0) {
        int sb = dev_mol_active(T.alen[src], T.nbest[src], 0), sa = dev_mol_active(T.alen[src], T.nbest[src], -num);
        if (!sa && sb) change -= (double)T.nbest[src] * -0.5;
        int kb = dev_mol_active(T.alen[snk], T.nbest[snk], 0), ka = dev_mol_active(T.alen[snk], T.nbest[snk], num);
        if (ka && !kb) change += (double)T.nbest[snk] * -0.5;
        if (T.alen[src] - num == 0 && num > 0) change -= -3.0;
        if (T.alen[snk] == 0 && num > 0) change += -3.0;
        change += alignment_change;
    }
    *num_out = num;
    if (nmove) *nmove = nm;
    return change;
}

__device__ __forceinline__ u64 dev_mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// tagBestAlignments for one pair, as written (lariat.go:1474-1543): the combinations one after the other, one draw each
__device__ __forceinline__ void dev_tag_pair_serial(const DCand& R, const DInf& S, double improper, u64 seed, int p, int nA, int nM, u64* ring) {
    const int ra = 2 * p, rb = 2 * p + 1;
    DGoRng rng;
    dev_go_seed(rng, seed, ring);
    double best = -1.7976931348623157e308;
    i64 ba = -1, bm = -1;
    for (i64 a = R.cand_off[ra]; a < R.cand_off[ra + 1]; ++a) {
        if (!R.in_filtered[a]) continue;
        for (i64 m = R.cand_off[rb]; m < R.cand_off[rb + 1]; ++m) {
            if (!R.in_filtered[m]) continue;
            double total = dev_score_aln(R, S, improper, a, m, 0.0) + (dev_go_f64(rng) / 2.0);
            if (total > best) { best = total; ba = a; bm = m; }
        }
    }
    S.active[ba] = 1; S.bwa_pick[ba] = 1;
    if (dev_is_pair(R, ba, bm)) { S.is_proper[ba] = 1; S.is_proper[bm] = 1; }
    S.active[bm] = 1; S.bwa_pick[bm] = 1;
    (void)nA; (void)nM;
}
#ifndef LH_RFA_TAG_WAVE
#define LH_RFA_TAG_WAVE 128   // combinations of a pair from which on the wave scores it together
#define LH_RFA_MAPQ_WAVE 128  // ... and from which on the wave estimates the read's map qualities together (candidate counts, filtered or not)
#define LH_RFA_MQ_CHUNK 384   // mate alignments staged per turn (24 B each in lds_raw)
#endif
static_assert(24 * LH_RFA_MQ_CHUNK <= LH_RFA_LDS_BYTES, "mate staging of estimateMapQualities");


// ---- K8 in pieces (r05).  What a barcode's wave did for its heavy pairs and reads one after the other — tagBestAlignments for a pair with thousands of
// (alignment, mate alignment) combinations, estimateMapQualities for a read with a hundred alignments — needs nothing of the barcode's tables: a pair's
// candidates and its name's seed; a read's candidates, the active links and the barcode's molecule penalty.  Those run as kernels of their own, a wave
// per listed PAIR / READ over the whole batch (k_rfa_tag_w, k_rfa_mq_w), and with them the defaults (k_rfa_init: a thread per candidate) and the light pairs
// (k_rfa_tag: a thread per pair).  The barcode program (k_rfa) starts at the positions and ends with the light reads' map qualities; markDuplicates and the
// split reads, which read what estimateMapQualities leaves for EVERY read of the barcode, follow in k_rfa_post.  Scratch of the wave kernels: the slabs
// of the barcode program, free while they run.
__global__ void __launch_bounds__(256) k_rfa_init(int n_reads, DCand R, DInf S, i64 cand_cap) {
    // Alignment defaults, lariat.go:1655-1689 (S.cand_read was written by k_aln_prep)
    const i64 n_cand = R.cand_off[n_reads] < cand_cap ? R.cand_off[n_reads] : cand_cap;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n_cand; g += (i64)gridDim.x * blockDim.x) {
        S.active[g] = 0; S.is_proper[g] = 0; S.bwa_pick[g] = 0; S.active_molecule[g] = 0; S.duplicate[g] = 0; S.molecule_id[g] = -1; S.mapq[g] = 0;
        S.mol_diff[g] = 0; S.mol_conf[g] = 0.00075 * 0.025; S.sum_move[g] = 1.0; S.mate[g] = -1;
    }
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (i64)gridDim.x * blockDim.x) {
        S.active_idx[r] = -1; S.second_best_idx[r] = -1; S.split_idx[r] = -1; S.second_best_score[r] = 0; S.as_score[r] = 0;
        S.split_second_best[r] = 0; S.split_score[r] = 0; S.split_mapq[r] = 0;
        for (i64 g = R.cand_off[r]; g < R.cand_off[r + 1] && g < cand_cap; ++g) S.cand_read[g] = (int32_t)r;
    }
}
// tagBestAlignments, one thread per pair (lariat.go:1474-1543).  Read 2 of a pair is always "touched" by read 1 (every read has >= 1 filtered candidate), so
// only read 1's scan decides; its jitter stream is Go's, seeded from the read name: a pair with up to LH_RFA_TAG_WAVE combinations draws that often, within
// what the state-free generator gives.  The others (a read on a repeat family has tens of candidates, its mate as many: thousands of combinations, each with
// its own draw) are listed for k_rfa_tag_w.
__global__ void __launch_bounds__(256) k_rfa_tag(DOpts o, int n_pairs, const u64* __restrict__ name_seed, DCand R, DInf S, i64 cand_cap, int32_t* __restrict__ hp_list,
                                                  int32_t* __restrict__ hp_count) {
    static_assert(LH_RFA_TAG_WAVE <= LH_GO_FAST_DRAWS, "a light pair's draws come from the state-free generator");
    const int p = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int hv = 0;
    if (p < n_pairs && R.cand_off[2 * p + 2] <= cand_cap) {
        int nA = 0, nM = 0;
        for (i64 a = R.cand_off[2 * p]; a < R.cand_off[2 * p + 1]; ++a) nA += R.in_filtered[a] != 0;
        for (i64 m = R.cand_off[2 * p + 1]; m < R.cand_off[2 * p + 2]; ++m) nM += R.in_filtered[m] != 0;
        hv = (i64)nA * nM > LH_RFA_TAG_WAVE;
        if (!hv) dev_tag_pair_serial(R, S, o.improper_pair_penalty, name_seed[p], p, nA, nM, (u64*)nullptr);
    }
    const u64 mh = __ballot(hv);
    if (mh) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(hp_count, (int32_t)__popcll(mh));
        basep = wave_readlane(basep, 0);
        if (hv) hp_list[basep + lanes_below(mh, lane)] = p;
    }
}
// ... a pair with many combinations, by a whole wave: 64 draws of Go's generator and 64 combinations per turn
__global__ void __launch_bounds__(64) k_rfa_tag_w(DOpts o, const u64* __restrict__ name_seed, DCand R, DInf S, uint8_t* __restrict__ slab_pool, i64 slab_bytes,
                                                   const int32_t* __restrict__ hp_list, const int32_t* __restrict__ hp_count, int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[8 * 1024];
    const int lane = LANE();
    uint8_t* slab = slab_pool + (size_t)blockIdx.x * (size_t)slab_bytes;
    const double improper = o.improper_pair_penalty;
    const int n_items = *hp_count;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const int p = hp_list[item];
            WAVE_SYNC();   // the previous pair's ring and lists have been read
            if ((size_t)(R.cand_off[2 * p + 2] - R.cand_off[2 * p]) * 12 + 16 > (size_t)slab_bytes || LH_GO_RING_BYTES > (size_t)slab_bytes) {   // (thousands of candidates in a slab of a few KB: refused)
                if (lane == 0) status[2 * p] |= LH_ST_POOL_OVERFLOW;
                continue;
            }
            const i64 c_lo = R.cand_off[2 * p];
            // the pair's filtered candidates and their single-read scores (scoreAlignment's two sums: exact multiples of 0.5), compacted in the slab
            const i64 a0 = R.cand_off[2 * p], a1 = R.cand_off[2 * p + 1], m1 = R.cand_off[2 * p + 2];
            int32_t* const fa = (int32_t*)slab;                                   // [nA | nM] candidate ids
            double* const fs = (double*)(slab + (((size_t)(m1 - a0) * 4 + 7) & ~(size_t)7));   // [nA | nM] scores
            int nA = 0, nM = 0;
            for (int side = 0; side < 2; ++side) {   // read 1's candidates, then read 2's behind them
                const i64 lo = side ? a1 : a0, hi = side ? m1 : a1;
                int cnt = 0;
                for (i64 cb = lo; cb < hi; cb += 64) {
                    const i64 c = cb + lane;
                    const int ok = c < hi && R.in_filtered[c];
                    const u64 mk = __ballot(ok);
                    if (ok) {
                        double sc = (double)(R.mismatches[c] * -2 + R.indels[c] * -3);
                        if (R.soft_clipped[c] > 0) { sc -= 5.0 * (double)R.soft_clipped[c]; sc -= (double)R.soft_clipped_length[c] * 0.5; }
                        const int at = nA + cnt + lanes_below(mk, lane);
                        fa[at] = (int32_t)(c - c_lo); fs[at] = sc;
                    }
                    cnt += __popcll(mk);
                }
                if (side) nM = cnt; else nA = cnt;
            }
            WAVE_SYNC();
            // Go's generator, 64 draws per turn: x[n] = x[n - 607] + x[n - 273]; the last 1024 values in LDS, the seeded state from its definition
            u64* const xb = (u64*)lds_raw;
            i64 sd = (i64)name_seed[p] % 2147483647ll;
            if (sd < 0) sd += 2147483647ll;
            if (sd == 0) sd = 89482311;
            const uint32_t x0 = (uint32_t)sd;
            const i64 K = (i64)nA * nM;
            double best = -1.7976931348623157e308;
            i64 bk = 0x7fffffffffffffffll;
            int redraw = 0;
            for (i64 k0 = 0; k0 < K; k0 += 64) {
                const i64 k = k0 + lane, n = k + 1;
                const u64 va = n > LH_GO_LEN ? xb[(n - LH_GO_LEN) & 1023] : dev_go_vec0(x0, (int)(n <= 334 ? 334 - n : 941 - n));
                const u64 vb = n > LH_GO_TAP ? xb[(n - LH_GO_TAP) & 1023] : dev_go_vec0(x0, (int)(LH_GO_LEN - n));
                const u64 x = va + vb;
                xb[n & 1023] = x;
                if (k < K) {
                    const double f = (double)(i64)(x & 0x7fffffffffffffffull) * (1.0 / 9223372036854775808.0);
                    redraw |= f == 1.0;   // Float64 draws again then (once in 2^53 draws): every later draw moves: the serial form
                    const int ia = (int)(k / nM), im = (int)(k - (i64)ia * nM);
                    const i64 a = c_lo + fa[ia], m = c_lo + fa[nA + im];
                    double t = fs[ia] + fs[nA + im];
                    if (!dev_is_pair(R, a, m)) t += improper;
                    t += 0.0;   // (scoreAlignment adds log_molecule_penalty = 0.0 for an alignment outside an active molecule)
                    const double total = t + f / 2.0;
                    if (total > best) { best = total; bk = k; }
                }
                WAVE_SYNC();
            }
            if (__any(redraw)) {
                if (lane == 0) dev_tag_pair_serial(R, S, improper, name_seed[p], p, nA, nM, (u64*)slab + lane);
                WAVE_SYNC();
                continue;
            }
            for (int msk = 32; msk >= 1; msk >>= 1) {   // the first maximum in draw order
                const double ob = __shfl_xor(best, msk);
                const i64 ok = shfl_xor_i64(bk, msk);
                if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
            }
            if (lane == 0) {
                const i64 ba = c_lo + fa[(int)(bk / nM)], bm = c_lo + fa[nA + (int)(bk % nM)];
                S.active[ba] = 1; S.bwa_pick[ba] = 1;
                if (dev_is_pair(R, ba, bm)) { S.is_proper[ba] = 1; S.is_proper[bm] = 1; }
                S.active[bm] = 1; S.bwa_pick[bm] = 1;
            }
            WAVE_SYNC();
    }
}

// One wavefront per barcode; barcodes are handed out through the device counter *bc_next (their costs differ widely).
// A barcode whose tables do not fit the wave's slab (the reader caps a work unit at 30,000 pairs, fastqreader/reader.go:205,
// far above the common few hundred) is appended to ovf_list and processed by a second launch whose few waves own much
// larger slabs (work_list = that list); only there an overflow is final (LH_ST_POOL_OVERFLOW).
// the barcode program's tables, carved from the wave's slab in this order (sizes follow the barcode's candidates NC, reads nR, the index's contigs and the hash bits): one list
// for the carve itself (k_rfa) and for its size (k_rfa_order routes a barcode that cannot fit the regular slab to a larger one BEFORE any wave has looked at it)
#define LH_RFA_CARVE_LIST(X)                                                                                                                                             \
        X(plist, int32_t, NC) X(molraw, int32_t, NC) X(mstart, int32_t, NC + 1) X(newid, int32_t, NC) X(nreads, int32_t, NC)                                              \
        X(sval, double, NC) X(seen_rid, int32_t, ncmax) X(ccnt, int32_t, ncmax) X(coff, int32_t, ncmax + 1)                                                               \
        X(kidx, int32_t, NC) X(molc, int32_t, NC) X(ppos, int32_t, NC) X(rdl, int32_t, NC) X(firstf, int32_t, NC)                                                         \
        X(actc, int32_t, NC + 1) X(psum, int32_t, NC + 1)                                                                                                                 \
        X(seg0, int32_t, NC) X(seg1, int32_t, NC) X(nbest, int32_t, NC) X(aoff, int32_t, NC) X(alen, int32_t, NC)                                                         \
        X(act_store, int32_t, NC) X(act_cand, int32_t, nR) X(act_slot, int32_t, nR) X(tdel, int32_t, nR) X(tset, int32_t, nR)                                             \
        X(mflag, int32_t, NC) X(P, double, NC)                                                                                                                            \
        X(dk0, u64, nR) X(dk1, u64, nR) X(dk2, u64, nR) X(dk3, u64, nR) X(htab, int32_t, (size_t)1 << hbits)                                                              \
        X(gstk, int32_t, 3 * LH_GOSORT_STK * 64)
__device__ __forceinline__ size_t rfa_carve_bytes(int NC, int nR, int ncmax) {
    int hbits = 6;
    while ((1 << hbits) < 2 * nR) ++hbits;
    size_t so = 0;
#define LH_RFA_SZ(ptr, type, count) { so = (so + 7) & ~(size_t)7; so += sizeof(type) * (size_t)(count); }
    LH_RFA_CARVE_LIST(LH_RFA_SZ)
#undef LH_RFA_SZ
    return (so + 7) & ~(size_t)7;
}

// (r06) the order in which the barcode programs take their barcodes: most candidates first.  A barcode's time grows faster than its candidate count (the position sorts,
// the molecule x read tables), a wave takes one barcode at a time, and the launch lasts until its last barcode is done: in index order the largest barcode of a batch
// could start last (kernel_ms_by_step of the repeats leg: 119 / 93 / 90 ms on three read sets).  A counting sort by size class (a quarter of an octave of the
// candidate count), one block; order[n_bc] = n_bc, the list's length as k_rfa's work_count wants it.
// all[0 .. n_bc): every barcode, largest first (k_rfa_post's list); big[]: those whose tables cannot fit a regular slab (slab_bytes), largest first — they go straight to the
// first tier's slabs, beside the others, instead of being turned away by the first launch and started when it has ended (a batch with log-normal barcode sizes: 61 ms, then 77 ms
// for the turned-away ones) —; rest[]: the others.  counts = {n_bc, n_big, n_bc - n_big}.
__global__ void __launch_bounds__(256) k_rfa_order(int n_bc, const int32_t* __restrict__ bc_pair_off, DCand R, int32_t* __restrict__ all, int32_t* __restrict__ big, int32_t* __restrict__ rest,
                                                    int32_t* __restrict__ counts, i64 slab_bytes, int ncmax) {
    __shared__ int32_t cnt[256], start[256];
    const int t = threadIdx.x;
    cnt[t] = 0;
    __syncthreads();
    auto cls_of = [&](int bc) {
        const int p0 = bc_pair_off[bc], p1 = bc_pair_off[bc + 1];
        const i64 nc = R.cand_off[2 * (i64)p1] - R.cand_off[2 * (i64)p0];
        const u64 x = (u64)(nc > 0 ? nc : 0) + 1;
        const int lg = 63 - __clzll((long long)x);                       // floor(log2(x)), x >= 1
        const int frac = lg >= 2 ? (int)((x >> (lg - 2)) & 3) : 0;
        const int c = 4 * lg + frac;
        // "cannot fit": the fixed tables, plus the molecule x read table at one molecule per four reads (M x nR words, M unknown until the molecules are made: nR^2 bytes
        // is M = nR / 4; a library's barcodes hold a molecule per three to ten read pairs) — a barcode routed without need only takes one of the larger slabs, one turned
        // away late starts again when the first launch has ended
        const size_t nr_ = (size_t)(2 * (p1 - p0));
        const int is_big = nc < 0x7fffffff && rfa_carve_bytes((int)nc, 2 * (p1 - p0), ncmax) + nr_ * nr_ > (size_t)slab_bytes;
        return (is_big ? 0 : 128) + 127 - (c < 127 ? c : 127);           // the big ones first, large classes first
    };
    for (int bc = t; bc < n_bc; bc += 256) atomicAdd(&cnt[cls_of(bc)], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0, nb = 0;
        for (int c = 0; c < 256; ++c) { start[c] = acc; acc += cnt[c]; if (c == 127) nb = acc; }
        counts[0] = n_bc; counts[1] = nb; counts[2] = n_bc - nb;
    }
    __syncthreads();
    const int n_big = start[128];
    for (int bc = t; bc < n_bc; bc += 256) {
        const int c = cls_of(bc), at = atomicAdd(&start[c], 1);
        all[at] = bc;
        if (c < 128) big[at] = bc; else rest[at - n_big] = bc;
    }
}

// development aid (tools/prof_rfa.sh builds a library with -DLH_RFA_PROF): shader-clock time per phase of the barcode program, summed over the waves
#ifdef LH_RFA_PROF
__device__ unsigned long long lh_rfa_prof[24];
#define RFA_PROF(k_) { const unsigned long long now_ = (unsigned long long)clock64(); if (lane == 0) atomicAdd(&lh_rfa_prof[k_], now_ - prof_t_); prof_t_ = (unsigned long long)clock64(); }
#else
#define RFA_PROF(k_)
#endif
#ifndef LH_RFA_WAVES
#ifdef LH_RA_HIST
__device__ unsigned long long lh_rfa_hist[40];
__device__ unsigned long long lh_rfa_hist2[8];
__device__ int lh_rfa_bcstat[4096][8];   // per barcode (index mod 4096): 10 ns ticks, candidates, filtered, raw molecules, largest raw molecule, contigs, molecules, largest contig list   // position sort: contig lists, the longest, those beyond the LDS buffer's usual / 32-bit-key length, sum of squares, lists with a tie
#endif
#define LH_RFA_WAVES 4   // waves per SIMD the register budget is sized for (128 VGPRs + 64 spilled: the kernel waits on memory, 4 waves hide more of it than 2 waves of 190 registers)
#endif
__global__ void __launch_bounds__(64, LH_RFA_WAVES) k_rfa(DIndex ix, DOpts o, int n_bc, const int32_t* __restrict__ bc_pair_off, const uint8_t* __restrict__ bc_do_rfa,
                                             const u64* __restrict__ name_seed, const i64* __restrict__ cen_start, const i64* __restrict__ cen_end, DCand R, DInf S, i64 cand_cap,
                                             uint8_t* __restrict__ slab_pool, i64 slab_bytes, int32_t* __restrict__ status, int32_t* __restrict__ bc_next,
                                             const int32_t* __restrict__ work_list, const int32_t* __restrict__ work_count, int32_t* __restrict__ ovf_list,
                                             int32_t* __restrict__ ovf_count, int32_t* __restrict__ hr_list, int32_t* __restrict__ hr_count, double* __restrict__ bc_lmp) {
    __shared__ int32_t shi[8];
    __shared__ double shd[4];
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LH_RFA_LDS_BYTES];   // one buffer, re-used phase by phase
    i64* const spos = (i64*)lds_raw;                                               // position sort
    int32_t* const sidx = (int32_t*)(lds_raw + 8 * LH_RFA_SORT_LDS);
    int32_t* const sLr = (int32_t*)lds_raw;                                        // fastScore source staging
    int32_t* const sFl = sLr + LH_RFA_SRC_CHUNK;
    double* const sLap = (double*)(sFl + LH_RFA_SRC_CHUNK);
    const int lane = LANE();
    uint8_t* slab = slab_pool + (size_t)blockIdx.x * (size_t)slab_bytes;
    const double improper = o.improper_pair_penalty;
    const int n_work = work_list ? *work_count : n_bc;
#define RFA_OVERFLOW()                                                                                 \
    {                                                                                                  \
        if (lane == 0) {                                                                               \
            if (ovf_list) ovf_list[atomicAdd(ovf_count, 1)] = bc;                                      \
            else status[r0] |= LH_ST_POOL_OVERFLOW;                                                    \
        }                                                                                              \
        continue;                                                                                      \
    }
    int wd_main = 1 << 24;
    for (;;) {
        LH_WATCH(o.wd, wd_main, 12, break)
        if (lane == 0) shi[5] = atomicAdd(bc_next, 1);
        WAVE_SYNC();
        const int widx = shi[5];
        WAVE_SYNC();
        if (widx >= n_work) break;
        const int bc = work_list ? work_list[widx] : widx;
        int p0 = bc_pair_off[bc], p1 = bc_pair_off[bc + 1];
        int nR = 2 * (p1 - p0), r0 = 2 * p0;
        i64 c_lo = R.cand_off[r0], c_hi = R.cand_off[r0 + nR];
        int NC = (int)(c_hi - c_lo);
        if (c_hi > cand_cap) continue;   // flagged by k_aln
#ifdef LH_RFA_PROF
        unsigned long long prof_t_ = (unsigned long long)clock64();
#endif
        RFA_PROF(2)
#ifdef LH_RA_HIST
        const unsigned long long bc_t0_ = (unsigned long long)wall_clock64();
#endif
        // ---- slab carve (sizes depend on NC, nR) ----
        size_t so = 0;
        RfaTab T;
#define CARVE(ptr, type, count) { so = (so + 7) & ~(size_t)7; T.ptr = (type*)(slab + so); so += sizeof(type) * (size_t)(count); }
        int ncmax = ix.n_contigs + 2;
        int hbits = 6;
        while ((1 << hbits) < 2 * nR) ++hbits;
        LH_RFA_CARVE_LIST(CARVE)
        so = (so + 7) & ~(size_t)7;
        T.bestT = (int32_t*)(slab + so);
        size_t best_cap = ((size_t)slab_bytes > so) ? ((size_t)slab_bytes - so) / 4 : 0;
#undef CARVE
        if (so > (size_t)slab_bytes) RFA_OVERFLOW()   // barcode too large for the slab
        // (measured, r05: guessing from NC and nR that the molecule x read table will not fit — to spare a barcode three phases in a tier it outgrows — sends on too many that do fit:
        // one candidate in four to sixteen starts a molecule, and a barcode sent on waits for one of fewer waves; 4,000 x 100 pairs 108 -> 155 ms, 1,000 x 400 pairs 0.44 -> 1.1 s)
        // ---- positions: filtered candidates grouped by contig in first-seen order, candidate order inside a contig ----
        int ncont = 0, NCf = 0;
        if (ncmax <= LH_RFA_NCONT_LDS && NC > 256) {
            // (r06) a large barcode, a contig table that fits LDS: the contigs' first-seen order from one atomic minimum per candidate (the smallest candidate index
            // that names the contig), slot sizes from one atomic add, and the stable placement from the lanes that share a slot — found with one ballot per bit of
            // the slot number, not one loop turn per slot present in the chunk (a read on repeat families names thirty contigs: the three passes were a sixth of the
            // kernel for a barcode of 10,000 candidates, most of it loop turns and a 36-deep search per candidate).
            int32_t* const fs = (int32_t*)lds_raw;                          // [ncmax] by contig + 1: its first candidate, then its slot
            int32_t* const scnt2 = (int32_t*)lds_raw + LH_RFA_NCONT_LDS;    // [ncont] slot sizes, then running write positions
            for (int x = lane; x < ncmax; x += 64) fs[x] = 0x7fffffff;
            WAVE_SYNC();
            for (int a = lane; a < NC; a += 64) if (R.in_filtered[c_lo + a]) atomicMin(&fs[R.rid[c_lo + a] + 1], a);   // (+ 1: a read without a region has a placeholder on contig -1, a group of its own)
            WAVE_SYNC();
            // the contigs present, in the order of their first candidates: rank by counting (a few dozen of them)
            for (int base = 0; base < ncmax; base += 64) {
                const int x = base + lane;
                const int f = x < ncmax ? fs[x] : 0x7fffffff;
                int rank = -1;
                if (f != 0x7fffffff) { rank = 0; for (int y = 0; y < ncmax; ++y) rank += fs[y] < f; }
                const u64 pm = __ballot(rank >= 0);
                if (rank >= 0) { T.seen_rid[rank] = x; T.ccnt[x] = rank; }
                ncont += __popcll(pm);
            }
            WAVE_SYNC();
            for (int x = lane; x < ncmax; x += 64) fs[x] = fs[x] != 0x7fffffff ? T.ccnt[x] : -1;   // contig -> slot
            for (int k = lane; k < ncont; k += 64) scnt2[k] = 0;
            WAVE_SYNC();
            for (int a = lane; a < NC; a += 64) {
                const int k = R.in_filtered[c_lo + a] ? fs[R.rid[c_lo + a] + 1] : -1;
                T.kidx[a] = k;
                if (k >= 0) atomicAdd(&scnt2[k], 1);
            }
            WAVE_SYNC();
            for (int base = 0; base < ncont; base += 64) {   // slot offsets; scnt2 becomes the running write position
                const int k = base + lane;
                const int c = k < ncont ? scnt2[k] : 0;
                const int inc = wave_scan_add_i32(c);
                EMU_SYNC();
                if (k < ncont) { T.coff[k] = NCf + inc - c; scnt2[k] = NCf + inc - c; }
                NCf += wave_readlane(inc, 63);
            }
            if (lane == 0) T.coff[ncont] = NCf;
            WAVE_SYNC();
            int kbits = 1;
            while ((1 << kbits) < ncont) ++kbits;
            for (int base = 0; base < NC; base += 64) {   // stable placement: candidate order inside a slot
                const int a = base + lane;
                const int k = a < NC ? T.kidx[a] : -1;
                const int valid = k >= 0;
                u64 peers = __ballot(valid);
                for (int b = 0; b < kbits; ++b) { const u64 bb = __ballot(valid && ((k >> b) & 1)); peers &= ((k >> b) & 1) ? bb : ~bb; }
                const int at = valid ? scnt2[k] : 0;
                EMU_SYNC();
                if (valid) {
                    T.plist[at + lanes_below(peers, lane)] = a;
                    if (lanes_below(peers, lane) == 0) scnt2[k] = at + __popcll(peers);
                }
                WAVE_SYNC();
            }
        } else {
        // pass A: contig slot of every candidate (first-seen numbering) and the slot sizes
        int32_t* const seen = ncmax <= LH_RFA_NCONT_LDS ? (int32_t*)lds_raw : T.seen_rid;
        int32_t* const scnt = ncmax <= LH_RFA_NCONT_LDS ? (int32_t*)lds_raw + LH_RFA_NCONT_LDS : T.ccnt;
        for (int base = 0; base < NC; base += 64) {
            int a = base + lane;
            int valid = a < NC && R.in_filtered[c_lo + a];
            int rid = valid ? R.rid[c_lo + a] : 0;
            int k = -1;
            if (valid) for (int j = ncont - 1; j >= 0; --j) if (seen[j] == rid) { k = j; break; }
            u64 un = __ballot(valid && k < 0);
            while (un) {   // contigs met for the first time, in candidate order
                int leader = __ffsll((unsigned long long)un) - 1;
                int lrid = wave_readlane(rid, leader);
                if (lane == 0) { seen[ncont] = lrid; scnt[ncont] = 0; }
                if (valid && k < 0 && rid == lrid) k = ncont;
                ncont++;
                un = __ballot(valid && k < 0);
            }
            WAVE_SYNC();
            if (a < NC) T.kidx[a] = valid ? k : -1;
            u64 rem = __ballot(valid);
            while (rem) {
                int kk = wave_readlane(k, __ffsll((unsigned long long)rem) - 1);
                u64 m = __ballot(valid && k == kk);
                if (lane == 0) scnt[kk] += __popcll(m);
                rem &= ~m;
            }
            WAVE_SYNC();
        }
        for (int base = 0; base < ncont; base += 64) {   // slot offsets; scnt becomes the running write position
            int k = base + lane;
            int c = k < ncont ? scnt[k] : 0;
            int inc = wave_scan_add_i32(c);
            EMU_SYNC();
            if (k < ncont) { T.coff[k] = NCf + inc - c; scnt[k] = NCf + inc - c; }
            NCf += wave_readlane(inc, 63);
        }
        if (lane == 0) T.coff[ncont] = NCf;
        WAVE_SYNC();
        // pass B: stable placement
        for (int base = 0; base < NC; base += 64) {
            int a = base + lane;
            int k = a < NC ? T.kidx[a] : -1;
            int valid = k >= 0;
            u64 rem = __ballot(valid);
            while (rem) {
                int kk = wave_readlane(k, __ffsll((unsigned long long)rem) - 1);
                u64 m = __ballot(valid && k == kk);
                int at = scnt[kk];
                EMU_SYNC();
                if (valid && k == kk) T.plist[at + lanes_below(m, lane)] = a;
                if (lane == 0) scnt[kk] = at + __popcll(m);
                EMU_SYNC();
                rem &= ~m;
            }
        }
        WAVE_SYNC();
        }
        RFA_PROF(3)
        // sort.Sort(ByPosition) per contig (lariat.go:1545-1547), one lane per contig; keys staged in LDS when they fit
        if (NCf <= LH_RFA_SORT_LDS && ncont <= 8) {
            // Few contigs: the wave sorts one contig at a time by ranking (every lane counts the smaller keys of its elements).
            // Go's sort is unstable, but with all keys different there is only one sorted order; a contig with two equal
            // positions is sorted again by the serial restatement of Go's algorithm.
            for (int i = lane; i < NCf; i += 64) { sidx[i] = T.plist[i]; spos[i] = R.pos[c_lo + T.plist[i]]; }
            WAVE_SYNC();
            for (int k = 0; k < ncont; ++k) {
                int b0 = T.coff[k], n = T.coff[k + 1] - b0;
                i64* kp = spos + b0;
                int32_t* ip = sidx + b0;
                int tie = 0;
                for (int e = lane; e < n; e += 64) {
                    i64 key = kp[e];
                    int rank = 0;
                    for (int j = 0; j < n; ++j) { i64 kj = kp[j]; rank += kj < key; tie |= (kj == key) & (j != e); }
                    T.plist[b0 + rank] = ip[e];
                }
                if (__any(tie)) {
                    WAVE_SYNC();
                    if (lane == 0)
                        dev_gosort(n, [&](int i, int j) { return kp[i] < kp[j]; },
                                   [&](int i, int j) { i64 t = kp[i]; kp[i] = kp[j]; kp[j] = t; int u = ip[i]; ip[i] = ip[j]; ip[j] = u; }, T.gstk + lane, 64);
                    WAVE_SYNC();
                    for (int e = lane; e < n; e += 64) T.plist[b0 + e] = ip[e];
                }
            }
        } else if (NCf <= LH_RFA_SORT_LDS) {
            for (int i = lane; i < NCf; i += 64) { sidx[i] = T.plist[i]; spos[i] = R.pos[c_lo + T.plist[i]]; }
            WAVE_SYNC();
            for (int k = lane; k < ncont; k += 64) {
                int b0 = T.coff[k], n = T.coff[k + 1] - b0;
                i64* kp = spos + b0;
                int32_t* ip = sidx + b0;
                dev_gosort(n, [&](int i, int j) { return kp[i] < kp[j]; },
                           [&](int i, int j) { i64 t = kp[i]; kp[i] = kp[j]; kp[j] = t; int u = ip[i]; ip[i] = ip[j]; ip[j] = u; }, T.gstk + lane, 64);
            }
            WAVE_SYNC();
            for (int i = lane; i < NCf; i += 64) T.plist[i] = sidx[i];
        } else {
            // More candidates than the LDS buffer holds (a barcode on repeat families has thousands; the reader's cap is 30,000 pairs).  Contig by contig (r05):
            // a contig's list that fits the buffer is staged there and sorted by ranking, as above — with all keys different there is only one sorted order; two
            // equal positions: Go's algorithm, serially, in LDS —; a longer one is sorted in memory with the ranges of Go's quickSort spread over the lanes
            // (lh_sort.h: wave_gosort; keys next to the list in sval, free until the molecules are scored; molraw / rdl / firstf hold the queue of ranges).
            // (All of them at once in memory, the r04 form, was 10 % of the kernel on the repeat input: the first levels of every quickSort are one lane's.)
            i64* const kpg = (i64*)T.sval;
            int32_t* const pl = T.plist;
            for (int k = 0; k < ncont; ++k) {
                const int b0 = T.coff[k], n = T.coff[k + 1] - b0;
#ifdef LH_RA_HIST
                if (lane == 0) { atomicAdd(&lh_rfa_hist2[0], 1ull); atomicMax(&lh_rfa_hist2[1], (unsigned long long)n); if (n > LH_RFA_SORT_LDS) atomicAdd(&lh_rfa_hist2[2], 1ull); if (n > LH_RFA_LDS_BYTES / 4) atomicAdd(&lh_rfa_hist2[3], 1ull); atomicAdd(&lh_rfa_hist2[4], (unsigned long long)n * n); }
#endif
                if (n < 2) continue;
                if (n > 64 && n < (1 << 20)) {
                    // (r06) a list of more than a wave's worth is put in order by a sorting NETWORK on packed words ((position - smallest) << 20 | place in the list), in the
                    // slab (sval, free until the molecules are scored) — instead of by ranking (every lane counting the smaller keys
                    // of its elements: n^2 / 64 steps; 2,000 entries of one contig, common for a 1,000-pair barcode with a twentieth of its pairs on repeat families, were
                    // 62,000 steps a list and the position sorts a third of the kernel).  With all keys different there is only one sorted order, whatever the algorithm;
                    // two equal positions (adjacent after the sort): Go's algorithm on the list as it was, below.
                    u64* const bk = (u64*)kpg + b0;   // (always in the slab: a pointer that is LDS for short lists and memory for long ones is a flat one — that form faulted on the device, and this one measured faster)
                    WAVE_SYNC();   // the previous contig's keys have been read
                    RFA_PROF(1)
                    // the positions, gathered ONCE (eight independent reads a lane in flight) into the slab next to the list, and the smallest of them
                    i64 mn = 0x7fffffffffffffffll;
                    for (int i0 = lane; i0 < n; i0 += 64 * 8) {
                        int ca[8];
                        i64 x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) ca[u] = i0 + 64 * u < n ? pl[b0 + i0 + 64 * u] : -1;
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = ca[u] >= 0 ? R.pos[c_lo + ca[u]] : 0x7fffffffffffffffll;
#pragma unroll
                        for (int u = 0; u < 8; ++u) { if (ca[u] >= 0) bk[i0 + 64 * u] = (u64)x[u]; mn = mn < x[u] ? mn : x[u]; }
                    }
                    mn = wave_min_i64(mn);
                    RFA_PROF(22)
                    if (n <= LH_RFA_NET_BLOCK) {
                        // up to 1,024 keys: the network runs in LDS (a pass is a round trip to LDS, not to the slab: thirty lists a barcode, 36 - 55 passes each) and its result is
                        // copied to the slab, where everything that follows reads it — through its own pointer, never one that is LDS for some lists and memory for others
                        u64* const lk = (u64*)lds_raw;
                        for (int i = lane; i < n; i += 64) lk[i] = (u64)((i64)bk[i] - mn) << 20 | (u64)i;
                        WAVE_SYNC();
                        RFA_PROF(23)
                        wave_bitonic_u64(lk, n, lane);
                        for (int i = lane; i < n; i += 64) bk[i] = lk[i];
                        WAVE_SYNC();
                    } else {
                        // longer: the network block by block through LDS, a few passes over the list in the slab between them (lh_sort.h)
                        for (int i = lane; i < n; i += 64) bk[i] = (u64)((i64)bk[i] - mn) << 20 | (u64)i;
                        WAVE_SYNC();
                        RFA_PROF(23)
                        wave_bitonic_u64_blocks<LH_RFA_NET_BLOCK>(bk, n, (u64*)lds_raw, lane);
                    }
                    RFA_PROF(19)
                    int tie = 0;
                    for (int e = lane; e < n; e += 64) {
                        const u64 w = bk[e];
                        if (e + 1 < n) tie |= (bk[e + 1] >> 20) == (w >> 20);
                        T.molraw[b0 + e] = pl[b0 + (int)(w & 0xfffffu)];
                    }
                    const int tied = __any(tie);
#ifdef LH_RA_HIST
                    if (tied && lane == 0) atomicAdd(&lh_rfa_hist2[5], 1ull);
#endif
                    WAVE_SYNC();
                    if (!tied) {
                        for (int e = lane; e < n; e += 64) pl[b0 + e] = T.molraw[b0 + e];
                        WAVE_SYNC();
                        RFA_PROF(20)
                        continue;
                    }
                    RFA_PROF(20)
                    if (n <= LH_RFA_TIE_LDS) {
                        // Two equal positions: Go's algorithm (its ranges spread over the lanes) on the list AS IT WAS — but on one word per entry in LDS: Less only asks which of two
                        // positions is smaller, so a position's dense RANK (from the network's order: equal positions share one) answers for it, and (rank << 12 | place) is a
                        // 32-bit word that a Swap moves whole.  Up to 2,304 entries instead of 768 (12 bytes each: position and place), and lists of 800 - 2,300 entries — two reads
                        // at the same offset of a 300-bp family tie on EVERY copy — were sorted in memory, two dependent reads per comparison.
                        int carry = 0;
                        for (int e0 = 0; e0 < n; e0 += 64) {
                            const int e = e0 + lane;
                            int fl = 0;
                            u64 w = 0;
                            if (e < n) { w = bk[e]; fl = e > 0 && (bk[e - 1] >> 20) != (w >> 20); }
                            const int rk = carry + wave_scan_add_i32(fl);
                            if (e < n) T.rdl[b0 + (int)(w & 0xfffffu)] = rk;   // (rdl: free until inferMolecules)
                            carry = wave_readlane(rk, 63);
                        }
                        WAVE_SYNC();
                        uint32_t* const g32 = (uint32_t*)lds_raw;
                        for (int i = lane; i < n; i += 64) g32[i] = (uint32_t)T.rdl[b0 + i] << 12 | (uint32_t)i;
                        if (lane == 0) { shi[6] = 0; shi[7] = n; }
                        WAVE_SYNC();
                        wave_gosort(1, shi + 6, [&](int i, int j) { return (g32[i] >> 12) < (g32[j] >> 12); },
                                    [&](int i, int j) { const uint32_t t = g32[i]; g32[i] = g32[j]; g32[j] = t; }, T.molraw, T.rdl, T.firstf, T.molc, T.ppos);   // (molc, ppos: set after the sorts)
                        WAVE_SYNC();
                        for (int e = lane; e < n; e += 64) T.molc[b0 + e] = pl[b0 + (int)(g32[e] & 0xfffu)];   // (molc: set after the sorts)
                        WAVE_SYNC();
                        for (int e = lane; e < n; e += 64) pl[b0 + e] = T.molc[b0 + e];
                        WAVE_SYNC();
                        RFA_PROF(21)
                        continue;
                    }
                    {
                        // (r06, late) longer still — several thousand entries on one contig, a 400-pair barcode on repeat families — and two equal positions: Go's algorithm
                        // on (rank << 20 | place) words in the slab, where the network's keys were: its long ranges are partitioned by the whole wave there (coalesced passes),
                        // and every range of up to a block's length is sorted in LDS with the depth the long sort has left it.  (It was Go's algorithm on positions and
                        // places in memory, a lane a range: a sixth of the kernel on such barcodes.)
                        int carry = 0;
                        for (int e0 = 0; e0 < n; e0 += 64) {
                            const int e = e0 + lane;
                            int fl = 0;
                            u64 w = 0;
                            if (e < n) { w = bk[e]; fl = e > 0 && (bk[e - 1] >> 20) != (w >> 20); }
                            const int rk = carry + wave_scan_add_i32(fl);
                            if (e < n) T.rdl[b0 + (int)(w & 0xfffffu)] = rk;
                            carry = wave_readlane(rk, 63);
                        }
                        WAVE_SYNC();
                        for (int i = lane; i < n; i += 64) bk[i] = (u64)(uint32_t)T.rdl[b0 + i] << 20 | (u64)i;
                        WAVE_SYNC();
                        auto less_g = [&](int i, int j) { return (bk[i] >> 20) < (bk[j] >> 20); };
                        auto swp_g = [&](int i, int j) { const u64 t = bk[i]; bk[i] = bk[j]; bk[j] = t; };
                        const int nq = wave_gosort_split(0, n, LH_RFA_NET_BLOCK, less_g, swp_g, T.molraw, T.rdl, T.firstf, T.molc, T.ppos);
                        const int qo = n / 16 + 64;   // (the ranges of the sorts in LDS wait behind those of the long one: fewer than n / 25 of the latter, at most a block's length of the former)
                        u64* const lk = (u64*)lds_raw;
                        for (int q = 0; q < nq; ++q) {
                            const int a = T.molraw[q], len = T.rdl[q] - a, d = T.firstf[q];
                            if (len < 2) continue;
                            if (len > LH_RFA_NET_BLOCK) {   // (its depth used up before it was short: Go's heap sort, where it lies)
                                WAVE_SYNC();
                                if (lane == 0) { shi[6] = a; shi[7] = a + len; }
                                WAVE_SYNC();
                                wave_gosort(1, shi + 6, less_g, swp_g, T.molraw + qo, T.rdl + qo, T.firstf + qo, T.molc, T.ppos, d);
                                WAVE_SYNC();
                                continue;
                            }
                            WAVE_SYNC();
                            for (int i = lane; i < len; i += 64) lk[i] = bk[a + i];
                            if (lane == 0) { shi[6] = 0; shi[7] = len; }
                            WAVE_SYNC();
                            wave_gosort(1, shi + 6, [&](int i, int j) { return (lk[i] >> 20) < (lk[j] >> 20); }, [&](int i, int j) { const u64 t = lk[i]; lk[i] = lk[j]; lk[j] = t; },
                                        T.molraw + qo, T.rdl + qo, T.firstf + qo, T.molc, T.ppos, d);
                            WAVE_SYNC();
                            for (int i = lane; i < len; i += 64) bk[a + i] = lk[i];
                        }
                        WAVE_SYNC();
                        for (int e = lane; e < n; e += 64) T.molc[b0 + e] = pl[b0 + (int)(bk[e] & 0xfffffu)];
                        WAVE_SYNC();
                        for (int e = lane; e < n; e += 64) pl[b0 + e] = T.molc[b0 + e];
                        WAVE_SYNC();
                        RFA_PROF(21)
                        continue;
                    }
                }
                if (n > LH_RFA_SORT_LDS) {
                    for (int i = lane; i < n; i += 64) kpg[b0 + i] = R.pos[c_lo + pl[b0 + i]];
                    WAVE_SYNC();
                    wave_gosort(1, T.coff + k, [&](int i, int j) { return kpg[i] < kpg[j]; },
                                [&](int i, int j) { i64 t = kpg[i]; kpg[i] = kpg[j]; kpg[j] = t; int u = pl[i]; pl[i] = pl[j]; pl[j] = u; }, T.molraw, T.rdl, T.firstf, T.molc, T.ppos);
                    WAVE_SYNC();
                    continue;
                }
                WAVE_SYNC();   // the previous contig's keys have been read
                for (int i = lane; i < n; i += 64) { const int a = pl[b0 + i]; sidx[i] = a; spos[i] = R.pos[c_lo + a]; }
                WAVE_SYNC();
                int tie = 0;
                for (int e = lane; e < n; e += 64) {
                    const i64 key = spos[e];
                    int rank = 0;
                    for (int j = 0; j < n; ++j) { const i64 kj = spos[j]; rank += kj < key; tie |= (kj == key) & (j != e); }
                    pl[b0 + rank] = sidx[e];
                }
                if (__any(tie)) {
                    // (r05: two reads drawn on one 300-bp family at the same offset have equal positions on EVERY copy — most of a repeat barcode's larger contigs have a tie.
                    // Go's algorithm then, with the ranges of its quickSort spread over the lanes, on the keys in LDS; it was one lane's)
#ifdef LH_RA_HIST
                    if (lane == 0) atomicAdd(&lh_rfa_hist2[5], 1ull);
#endif
                    WAVE_SYNC();
                    if (lane == 0) { shi[6] = 0; shi[7] = n; }
                    WAVE_SYNC();
                    wave_gosort(1, shi + 6, [&](int i, int j) { return spos[i] < spos[j]; },
                                [&](int i, int j) { i64 t = spos[i]; spos[i] = spos[j]; spos[j] = t; int u = sidx[i]; sidx[i] = sidx[j]; sidx[j] = u; }, T.molraw, T.rdl, T.firstf, T.molc, T.ppos);
                    WAVE_SYNC();
                    for (int e = lane; e < n; e += 64) pl[b0 + e] = sidx[e];
                }
            }
        }
        WAVE_SYNC();
        RFA_PROF(4)
        int do_rfa = bc_do_rfa[bc] != 0;
        int M = 0;
        if (do_rfa) {
            for (int a = lane; a < NC; a += 64) { T.molc[a] = -1; T.ppos[a] = -1; }
            WAVE_SYNC();
            // ---- inferMolecules: a gap > 50 kb (or a new contig list) starts a molecule ----
            int Mraw = 0;
            for (int base = 0; base < NCf; base += 64) {
                int i = base + lane, st = 0, a = -1;
                if (i < NCf) {
                    a = T.plist[i];
                    st = i == T.coff[T.kidx[a]];
                    if (!st) st = R.pos[c_lo + a] - R.pos[c_lo + T.plist[i - 1]] > 50000;
                }
                u64 mk = __ballot(st);
                int mi = Mraw + __popcll(mk & ((2ull << lane) - 1)) - 1;
                if (i < NCf) {
                    T.molraw[i] = mi; T.molc[a] = mi; T.ppos[a] = i; T.rdl[i] = S.cand_read[c_lo + a] - r0;
                    T.firstf[i] = -1;   // (not looked at yet: the large molecules' entries are, below, before the others)
                    if (st) T.mstart[mi] = i;
                }
                Mraw += __popcll(mk);
            }
            if (lane == 0) T.mstart[Mraw] = NCf;
            WAVE_SYNC();
            RFA_PROF(16)
#ifdef LH_RA_HIST
            {
                int mxs = 0;
                for (int q = lane; q < Mraw; q += 64) { const int z = T.mstart[q + 1] - T.mstart[q]; mxs = mxs > z ? mxs : z; }
                mxs = wave_max_i32(mxs);
                int mxc = 0;
                for (int q = lane; q < ncont; q += 64) { const int z = T.coff[q + 1] - T.coff[q]; mxc = mxc > z ? mxc : z; }
                mxc = wave_max_i32(mxc);
                if (lane == 0) { lh_rfa_bcstat[bc & 4095][3] = Mraw; lh_rfa_bcstat[bc & 4095][4] = mxs; lh_rfa_bcstat[bc & 4095][7] = mxc; }
            }
#endif
            // ---- markBestAlignmentForReadInMolecule, step 1: best pair score of every entry inside its molecule; the
            // entry's read is counted once per molecule (its first occurrence); molecules with an active alignment ----
            // (r05) LARGE raw molecules first, from LDS.  On repeat families a read pair has a hundred alignments each within one 50-kb neighbourhood: one raw molecule of
            // several hundred entries, in which every entry of the read is scored against every entry of its mate — n_a x n_m evaluations of ten scattered fields each,
            // half of this kernel's time.  The molecule's entries (position, contig, strand, read, the four counts the score is made of: 24 B) are staged once and every
            // lane scans them for ITS entry: the same set of mate entries, the same expression evaluated in the same order per pair (lariat.go:599-624 with lmp = 0: adding
            // 0.0 to a sum that cannot be -0.0 is left out), the maximum of the same values; "first" = no earlier entry of the read in the molecule, as below.
            {
                struct MEnt { i64 pos; int32_t rid; uint32_t rd; int32_t s2, spare; };   // s2: twice the entry's own part of a pair's score (below)
                static_assert(sizeof(MEnt) == 24, "24 bytes per staged entry");
                MEnt* const me = (MEnt*)lds_raw;
                // (r06, late) next to the records, the entries' read numbers alone, sixteen bits each: a lane looking for its mate's entries (and for earlier ones of its own
                // read) reads EIGHT of them per round trip to LDS and fetches a 24-byte record only where the low sixteen bits match — one entry in ten on repeat families, where
                // the scan of whole records, a dependent LDS read per entry, was two fifths of the kernel on large barcodes
                constexpr int ME_FIT = (LH_RFA_LDS_BYTES / 26) & ~7;
#ifdef LH_RFA_ME_CAP   // (test builds: tiles short enough for the suite's molecules to need several)
                constexpr int ME_CAP = (LH_RFA_ME_CAP < ME_FIT ? LH_RFA_ME_CAP : ME_FIT) & ~7;
#else
                constexpr int ME_CAP = ME_FIT;
#endif
                static_assert(ME_CAP >= 8 && ME_CAP % 8 == 0 && ME_CAP * 26 <= LH_RFA_LDS_BYTES, "records and read numbers of a tile share lds_raw");
                uint16_t* const mrd = (uint16_t*)(lds_raw + ME_CAP * 24);   // (16-byte aligned: ME_CAP is a multiple of 8)
                auto load_ent = [&](int i) {
                    const i64 ca = c_lo + T.plist[i];
                    MEnt e;
                    e.pos = R.pos[ca]; e.rid = R.rid[ca]; e.rd = (uint32_t)T.rdl[i] | (uint32_t)(R.reversed[ca] != 0) << 31;
                    const int sc = R.soft_clipped[ca];
                    e.s2 = R.mismatches[ca] * -4 + R.indels[ca] * -6 - (sc > 0 ? 10 * sc + R.soft_clipped_length[ca] : 0);   // twice (-2 mm - 3 id - 5 sc - 0.5 scl)
                    e.spare = 0;
                    return e;
                };
                // (r06) EVERY molecule's entries go through LDS, not only the large ones: the entries of a barcode are cut into tiles of whole molecules (they are contiguous
                // in position order), a tile is staged once and every lane scans ITS entry's molecule inside it — a handful to a few dozen LDS records where it was as many
                // dependent reads of ten scattered fields from memory per entry (46 % of the kernel on repeat families, 37 % on 1,000-pair barcodes with a twentieth of their
                // pairs on them).  A molecule longer than a tile goes through tiles of its own, as in r05.
                // (r06, late) a pair's score (lariat.go:599-624 with lmp = 0) is the sum of the two entries' own parts — whole and half numbers (-2 per mismatch, -3 per indel,
                // -5 per clipped end, -0.5 per clipped base), every partial sum exact in a double whatever the order — plus the improper-pair penalty as the LAST addition
                // when the pair is not proper.  So the largest score over a set of mates is max(half the largest integer sum over the proper ones, half the largest over the
                // improper ones + penalty) — the same doubles the expression as written yields (x -> x / 2 + penalty does not decrease in x) — and the loop over the mates is
                // integer work: it was thirty double-precision operations a pair, most of this step on repeat families.
                auto pair_proper = [&](const MEnt& A, int arev, const MEnt& B) {
                    int pr = 0;
                    if (arev != (int)(B.rd >> 31) && A.rid == B.rid) { const i64 dist = arev ? A.pos - B.pos : B.pos - A.pos; pr = dist >= -35 && dist < 750; }
                    return pr;
                };
                constexpr int S2_NONE = -(1 << 30);
                auto best_of = [&](int bp, int bi) {
                    double best = -1.7976931348623157e308;
                    if (bp != S2_NONE) best = (double)bp * 0.5;
                    if (bi != S2_NONE) { const double x = (double)bi * 0.5 + improper; if (x > best) best = x; }
                    return best;
                };
                // entries [j0, j1) of the staged tile against entry A (read lr, mate lrm); the tile's entry j is the molecule's entry jb + j, A is its entry e
                auto scan_tile = [&](const MEnt& A, int lr, int lrm, int arev, int j0, int j1, int jb, int e, int& bp, int& bi, int& first) {
                    const uint32_t km = (uint32_t)lrm & 0xffffu, ko = (uint32_t)lr & 0xffffu;
                    for (int j8 = j0 & ~7; j8 < j1; j8 += 8) {
                        const u64* const wp = (const u64*)(mrd + j8);
                        const u64 ww[2] = {wp[0], wp[1]};
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const uint32_t id = (uint32_t)(ww[u >> 2] >> ((u & 3) * 16)) & 0xffffu;
                            const int j = j8 + u;
                            if ((id == km || id == ko) && j >= j0 && j < j1) {
                                const MEnt B = me[j];
                                const int lj = (int)(B.rd & 0x7fffffffu);
                                if (lj == lrm) {
                                    const int s2 = A.s2 + B.s2;
                                    if (pair_proper(A, arev, B)) bp = bp > s2 ? bp : s2;
                                    else bi = bi > s2 ? bi : s2;
                                } else if (lj == lr && jb + j < e) first = 0;
                            }
                        }
                    }
                };
                for (int i0 = 0; i0 < NCf;) {
                    const int m0 = T.molraw[i0];
                    const int msz = T.mstart[m0 + 1] - i0;   // (i0 is the first entry of molecule m0)
                    if (msz > ME_CAP) {
                        // a molecule longer than the buffer goes through it in tiles: every entry keeps its maximum so far in sval, "a mate's entry seen" in psum (free until
                        // the counting below) and "first" in firstf
                        const int ms0 = i0;
                        for (int t0 = 0; t0 < msz; t0 += ME_CAP) {
                            const int tn = msz - t0 < ME_CAP ? msz - t0 : ME_CAP;
                            WAVE_SYNC();   // the previous tile's entries have been read
                            for (int j = lane; j < tn; j += 64) { const MEnt E = load_ent(ms0 + t0 + j); me[j] = E; mrd[j] = (uint16_t)(E.rd & 0xffffu); }
                            WAVE_SYNC();
                            for (int e0 = 0; e0 < msz; e0 += 64) {
                                const int e = e0 + lane;
                                if (e < msz) {
                                    const int i = ms0 + e;
                                    const MEnt A = (e >= t0 && e < t0 + tn) ? me[e - t0] : load_ent(i);
                                    const int lr = (int)(A.rd & 0x7fffffffu), lrm = lr ^ 1, arev = (int)(A.rd >> 31);
                                    double best = -1.7976931348623157e308;
                                    int found = 0, first = 1, bp = S2_NONE, bi = S2_NONE;
                                    if (t0 > 0) { best = T.sval[i]; found = T.psum[i]; first = T.firstf[i]; }
                                    scan_tile(A, lr, lrm, arev, 0, tn, t0, e, bp, bi, first);
                                    if (bp != S2_NONE || bi != S2_NONE) { found = 1; const double x = best_of(bp, bi); if (x > best) best = x; }
                                    if (t0 + tn >= msz) { T.sval[i] = found ? best : R.lap[c_lo + T.plist[i]]; T.firstf[i] = first; }
                                    else { T.sval[i] = best; T.psum[i] = found; T.firstf[i] = first; }
                                }
                            }
                        }
                        i0 += msz;
                        continue;
                    }
                    // whole molecules from i0 on: up to the start of the molecule that holds entry i0 + ME_CAP
                    int i1 = NCf;
                    if (i0 + ME_CAP < NCf) i1 = T.mstart[T.molraw[i0 + ME_CAP]];
                    const int tn = i1 - i0;
                    WAVE_SYNC();   // the previous tile's entries have been read
                    for (int j = lane; j < tn; j += 64) { const MEnt E = load_ent(i0 + j); me[j] = E; mrd[j] = (uint16_t)(E.rd & 0xffffu); }
                    WAVE_SYNC();
                    RFA_PROF(17)
                    for (int e0 = 0; e0 < tn; e0 += 64) {
                        const int e = e0 + lane;
                        if (e < tn) {
                            const int i = i0 + e, m = T.molraw[i];
                            const int j0 = T.mstart[m] - i0, j1 = T.mstart[m + 1] - i0;
                            const MEnt A = me[e];
                            const int lr = (int)(A.rd & 0x7fffffffu), lrm = lr ^ 1, arev = (int)(A.rd >> 31);
                            int first = 1, bp = S2_NONE, bi = S2_NONE;
                            scan_tile(A, lr, lrm, arev, j0, j1, 0, e, bp, bi, first);
                            T.sval[i] = bp != S2_NONE || bi != S2_NONE ? best_of(bp, bi) : R.lap[c_lo + T.plist[i]];
                            T.firstf[i] = first;
                        }
                    }
                    RFA_PROF(18)
                    i0 = i1;
                }
                WAVE_SYNC();
            }
            int nfirst = 0, nactive = 0;
            for (int base = 0; base < NCf; base += 64) {
                int i = base + lane, first = 0, isact = 0;
                const int pre = i < NCf ? T.firstf[i] : -1;
                if (i < NCf && pre >= 0) { first = pre; isact = S.active[c_lo + T.plist[i]] != 0; }
                u64 mf = __ballot(first), ma = __ballot(isact);
                if (i < NCf) { T.psum[i] = nfirst + lanes_below(mf, lane); T.actc[i] = nactive + lanes_below(ma, lane); }
                nfirst += __popcll(mf); nactive += __popcll(ma);
            }
            if (lane == 0) { T.psum[NCf] = nfirst; T.actc[NCf] = nactive; }
            WAVE_SYNC();
            RFA_PROF(5)
            // ---- scrapMolecules: keep molecules with an active alignment, renumber ----
            int ao = 0;
            for (int base = 0; base < Mraw; base += 64) {
                int m = base + lane;
                int keep = m < Mraw && T.actc[T.mstart[m + 1]] - T.actc[T.mstart[m]] > 0;
                int nr = keep ? T.psum[T.mstart[m + 1]] - T.psum[T.mstart[m]] : 0;
                u64 mk = __ballot(keep);
                int id = M + lanes_below(mk, lane);
                int inc = wave_scan_add_i32(nr);
                if (keep) { T.newid[m] = id; T.seg0[id] = T.mstart[m]; T.seg1[id] = T.mstart[m + 1]; T.nbest[id] = nr; T.aoff[id] = ao + inc - nr; T.alen[id] = 0; }
                else if (m < Mraw) T.newid[m] = -1;
                M += __popcll(mk);
                ao += wave_readlane(inc, 63);
            }
            WAVE_SYNC();
            if ((size_t)M * (size_t)nR > best_cap) RFA_OVERFLOW()   // molecule table does not fit the slab
            for (size_t x = lane; x < (size_t)M * nR; x += 64) T.bestT[x] = -1;
            for (int r = lane; r < nR; r += 64) { T.act_cand[r] = -1; T.act_slot[r] = -1; T.dk0[r] = 0; }   // (dk0: k_rfa_post's; here the read's sink groups, below)
            for (int i = lane; i < NCf; i += 64) S.molecule_id[c_lo + T.plist[i]] = T.newid[T.molraw[i]];
            WAVE_SYNC();
            // step 2: per molecule, reads in first-occurrence order: best alignment (earliest maximum) and the active list
            int nact = 0;
            for (int base = 0; base < NCf; base += 64) {
                int i = base + lane;
                int m = -1, isf = 0, act = -1;
                if (i < NCf) { m = T.newid[T.molraw[i]]; isf = T.firstf[i] && m >= 0; }
                if (isf) {
                    int mr = T.molraw[i], g = r0 + T.rdl[i];
                    double best = -1.7976931348623157e308;
                    int bi = -1, bpos = 0x7fffffff, apos = -1;
                    // (four candidates' molecule numbers, places and flags are fetched before any is looked at, then the values of those that are in the molecule: the loop
                    // was a chain of dependent reads per candidate; the candidates are still taken in order)
                    const i64 cb0 = R.cand_off[g], cb1 = R.cand_off[g + 1];
                    for (i64 b4 = cb0; b4 < cb1; b4 += 4) {
                        int mm[4], pj[4], ac[4];
                        double sv4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int ok = b4 + u < cb1;
                            const int lb = (int)(b4 + u - c_lo);
                            mm[u] = ok ? T.molc[lb] : -2; pj[u] = ok ? T.ppos[lb] : 0; ac[u] = ok ? (int)S.active[b4 + u] : 0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) sv4[u] = mm[u] == mr ? T.sval[pj[u]] : 0.0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (mm[u] != mr) continue;
                            const int lb = (int)(b4 + u - c_lo), j = pj[u];
                            const double sv = sv4[u];
                            if (sv > best || (sv == best && j < bpos)) { best = sv; bi = lb; bpos = j; }
                            if (ac[u] && j > apos) { apos = j; act = lb; }
                        }
                    }
                    T.bestT[(size_t)T.rdl[i] * M + m] = bi;
                    // (r06, late) which groups of 64 molecules hold an alignment of the read at all (bit (m / 64) mod 64): fastScore is asked about 64 sinks at a time, and a
                    // group in which none of the source's reads has an alignment cannot hold a sink with a read to move
                    atomicOr((unsigned long long*)&T.dk0[T.rdl[i]], 1ull << ((m >> 6) & 63));
                }
                int hasact = isf && act >= 0;
                u64 mk = __ballot(hasact);
                if (i < NCf) { T.actc[i] = hasact ? act : -1; T.psum[i] = nact + lanes_below(mk, lane); }
                nact += __popcll(mk);
            }
            if (lane == 0) T.psum[NCf] = nact;
            WAVE_SYNC();
            for (int i = lane; i < NCf; i += 64) {
                int m = T.newid[T.molraw[i]];
                if (m < 0 || !T.firstf[i]) continue;
                int lr = T.rdl[i];
                int t = T.bestT[(size_t)lr * M + m] & RFA_T_MASK, tm = T.bestT[(size_t)(lr ^ 1) * M + m];
                if (tm >= 0 && dev_is_pair(R, c_lo + t, c_lo + (tm & RFA_T_MASK))) T.bestT[(size_t)lr * M + m] = t | RFA_T_PAIR;
                int act = T.actc[i];
                if (act >= 0) {
                    int slot = T.psum[i] - T.psum[T.seg0[m]];
                    T.act_store[T.aoff[m] + slot] = act; T.act_cand[lr] = act; T.act_slot[lr] = slot;
                }
            }
            for (int m = lane; m < M; m += 64) T.alen[m] = T.psum[T.seg1[m]] - T.psum[T.seg0[m]];
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
            RFA_PROF(6)
            // ---- optimizer.Optimize(opt, 1, 2, 4*M): 8*M greedy molecule moves ----
            // The optimizer is deterministic: once M consecutive turns (every molecule tried once as the source) accept no
            // move, the state can no longer change and the remaining turns are no-ops, so they are not executed.
            // the source staged once (a call with idle lanes), and the groups of 64 sinks in which one of its reads has an alignment: the union of its reads' group bits
            auto stage_source = [&](int src) -> u64 {
                if (M <= 64 || T.alen[src] > LH_RFA_SRC_CHUNK) return ~0ull;
                int num;
                dev_fast_score_w(R, S, T, c_lo, r0, M, src, -1, improper, sLr, sFl, sLap, &num, 0, (int*)0);
                u64 mine = 0;
                for (int k = lane; k < T.alen[src]; k += 64) mine |= T.dk0[sLr[k]];
                u64 all = 0;
                for (int b = 0; b < 64 && b * 64 < M; ++b)
                    if (__ballot((int)(mine >> b & 1))) all |= 1ull << b;
                if (M > 64 * 64) all = ~0ull;   // (the bits wrap around: every group may be meant)
                return all;
            };
            int source = 0, idle = 0;
            for (int it = 0; it < 8 * M && idle < M; ++it) {
                ++idle;
                if (T.alen[source] == 0) { source = (source + 1) % M; continue; }
                double bs = -1.7976931348623157e308;
                int bl = -1, bi = 0x7fffffff;
                const u64 groups = stage_source(source);
                const int staged = groups != ~0ull;
                for (int sb = 0; sb < M; sb += 64) {
                    if (!(groups >> ((sb >> 6) & 63) & 1)) continue;   // (no sink here has a read of the source to move: num = 0 for all of them)
                    int i = sb + lane, num;
                    int snk = (i < M && i != source) ? i : -1;
                    double sc = dev_fast_score_w(R, S, T, c_lo, r0, M, source, snk, improper, sLr, sFl, sLap, &num, 0, (int*)0, staged);
                    if (snk >= 0 && num > 0 && (sc > bs || (sc == bs && T.alen[i] > bl))) { bs = sc; bl = T.alen[i]; bi = i; }
                }
                for (int msk = 32; msk >= 1; msk >>= 1) {   // lexicographic max of (score, sink size), first index on full ties
                    double os = __shfl_xor(bs, msk);
                    int ol = __shfl_xor(bl, msk), oi = __shfl_xor(bi, msk);
                    int take = oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && (ol > bl || (ol == bl && oi < bi))));
                    if (take) { bs = os; bl = ol; bi = oi; }
                }
                if (bi != 0x7fffffff && (bs > 0 || (bs == 0 && bl > T.alen[source]))) {
                    int num, nmv;   // acceptMove: recompute the move list (every lane evaluates the winner, lane 0 records), apply in order
                    dev_fast_score_w(R, S, T, c_lo, r0, M, source, bi, improper, sLr, sFl, sLap, &num, lane == 0, &nmv, staged);
                    WAVE_SYNC();
                    idle = 0;
                    if (lane == 0) {
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
            RFA_PROF(7)
            // ---- moleculeMapqProbabilitySums ----
            for (int s = 0; s < M; ++s) {
                const u64 groups = stage_source(s);
                const int staged = groups != ~0ull;
                for (int sb = 0; sb < M; sb += 64) {
                    if (!(groups >> ((sb >> 6) & 63) & 1)) continue;   // (P[t] is read below only where a read of s has an alignment in t: none in this group)
                    int t = sb + lane, num;
                    double sc = dev_fast_score_w(R, S, T, c_lo, r0, M, s, (t < M && t != s) ? t : -1, improper, sLr, sFl, sLap, &num, 0, (int*)0, staged);
                    if (t < M) T.P[t] = t == s ? 0.0 : pow(10.0, sc);
                }
                WAVE_SYNC();
                RFA_PROF(8)
                for (int k = lane; k < T.alen[s]; k += 64) {
                    int a = T.act_store[T.aoff[s] + k], lr = S.cand_read[c_lo + a] - r0;
                    double sum = S.sum_move[c_lo + a];
                    const u64 mine = M > 64 && M <= 64 * 64 ? T.dk0[lr] : ~0ull;   // (the read's own groups: elsewhere its row of the table is empty)
                    for (int t0 = 0; t0 < M; t0 += 8) {   // (eight table entries read at a time; the sums in the molecules' order, as written)
                        if (!(mine >> ((t0 >> 6) & 63) & 1)) { t0 = (t0 | 63) - 7; continue; }
                        int q8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) q8[u] = t0 + u < M ? T.bestT[(size_t)lr * M + t0 + u] : -1;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (t0 + u != s && q8[u] >= 0) sum += T.P[t0 + u];
                    }
                    S.sum_move[c_lo + a] = sum;
                }
                WAVE_SYNC();
                RFA_PROF(2)
            }
            RFA_PROF(8)
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
        // ---- calculateLogMoleculePenalty (lariat.go:1061-1086) ----
        // Every term of dnaLength is an integer-valued double far below 2^53, so the sum is exact in any order: the lanes share a
        // molecule's alignments (it was one lane walking all of them, 16 % of the kernel's time), the wave adds up.
        {
            double part = 0.0;
            if (do_rfa && M > 0) {
                for (int m = 0; m < M; ++m) {
                    const int al = T.alen[m], ao = T.aoff[m];
                    if (T.mflag[m]) {
                        i64 smallest = 0x7fffffffffffffffll, biggest = -1;
                        for (int k = lane; k < al; k += 64) {
                            i64 p = R.pos[c_lo + T.act_store[ao + k]];
                            if (p > biggest) biggest = p;
                            if (p < smallest) smallest = p;
                        }
                        for (int msk = 32; msk >= 1; msk >>= 1) {
                            const i64 ob = shfl_xor_i64(biggest, msk), os = shfl_xor_i64(smallest, msk);
                            if (ob > biggest) biggest = ob;
                            if (os < smallest) smallest = os;
                        }
                        if (lane == 0 && biggest >= smallest) part += (double)(biggest - smallest) + 1000.0;
                    } else {
                        for (int k = lane; k < al; k += 64) {
                            i64 g = c_lo + T.act_store[ao + k];
                            part += (double)(R.aend[g] - R.pos[g]) * 2.0;
                        }
                    }
                }
            }
            for (int msk = 32; msk >= 1; msk >>= 1) part += __shfl_xor(part, msk);
            if (lane == 0) shd[0] = (do_rfa && M > 0) ? log10((1000.0 + part) / o.genome_length * 0.05) : 0.0;
        }
        WAVE_SYNC();
        double lmp = shd[0];
        RFA_PROF(9)
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
        // ---- estimateMapQualities per read (lariat.go:887-990).  A read with few (alignment, mate alignment) combinations: one lane; the others
        // (a read on a repeat family and its mate: thousands) are listed and done by the whole wave below, a lane per alignment ----
        // the reads with many combinations: listed for k_rfa_mq_w (with the barcode, whose molecule penalty they need)
        if (lane == 0) bc_lmp[bc] = lmp;
        for (int rb_ = 0; rb_ < nR; rb_ += 64) {
            const int r = rb_ + lane;
            int hv = 0;
            if (r < nR) hv = (R.cand_off[r0 + r + 1] - R.cand_off[r0 + r]) * (R.cand_off[r0 + (r ^ 1) + 1] - R.cand_off[r0 + (r ^ 1)]) > LH_RFA_MAPQ_WAVE;
            const u64 mh = __ballot(hv);
            if (mh) {
                int basep = 0;
                if (lane == 0) basep = atomicAdd(hr_count, (int32_t)__popcll(mh));
                basep = wave_readlane(basep, 0);
                if (hv) { const int at = basep + lanes_below(mh, lane); hr_list[2 * at] = r0 + r; hr_list[2 * at + 1] = bc; }
            }
        }
        WAVE_SYNC();
        RFA_PROF(10)
        for (int r = lane; r < nR; r += 64) {
            int gr = r0 + r, gm = r0 + (r ^ 1);
            i64 a0 = R.cand_off[gr], a1 = R.cand_off[gr + 1], m0 = R.cand_off[gm], m1 = R.cand_off[gm + 1];
            if ((a1 - a0) * (m1 - m0) > LH_RFA_MAPQ_WAVE) continue;
            double* const top = (double*)lds_raw;   // the read's 15 best scores, lane-strided in LDS (the phases that staged things there are over): no scratch
#define TOPV(k_) top[(k_) * 64 + lane]
            int ntop = 0;
#define TOP_PUSH(v_)                                                                   \
    {                                                                                  \
        double v = (v_);                                                               \
        int k_ = ntop < 15 ? ntop : 15;                                                \
        if (ntop < 15 || v > TOPV(14)) {                                               \
            if (ntop < 15) ntop++; else k_ = 14;                                       \
            while (k_ > 0 && TOPV(k_ - 1) < v) { TOPV(k_) = TOPV(k_ - 1); k_--; }      \
            TOPV(k_) = v;                                                              \
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
            for (int k = 0; k < ntop; ++k) total += pow(10.0, TOPV(k));
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
#undef TOPV
        }
        WAVE_SYNC();
        RFA_PROF(11)
#ifdef LH_RA_HIST
        if (lane == 0) {   // barcodes by log2 of the time k_rfa spent on them (10 ns ticks), and by candidates
            const unsigned long long dt_ = (unsigned long long)wall_clock64() - bc_t0_;
            int b_ = 0;
            while (b_ < 31 && (dt_ >> (b_ + 1))) ++b_;
            atomicAdd(&lh_rfa_hist[b_], 1ull);
            atomicMax(&lh_rfa_hist[32], dt_);
            atomicAdd(&lh_rfa_hist[33], dt_);
            atomicAdd(&lh_rfa_hist[34], (unsigned long long)NC); atomicAdd(&lh_rfa_hist[35], (unsigned long long)NCf); atomicAdd(&lh_rfa_hist[36], (unsigned long long)M);
            atomicAdd(&lh_rfa_hist[37], (unsigned long long)so); atomicAdd(&lh_rfa_hist[38], (unsigned long long)M * (unsigned long long)nR);
            int* st_ = lh_rfa_bcstat[bc & 4095];
            st_[0] = (int)dt_; st_[1] = NC; st_[2] = NCf; st_[5] = ncont; st_[6] = M;
        }
#endif
    }
}

// estimateMapQualities for a read with many (alignment, mate alignment) combinations (lariat.go:887-990), by a whole wave: a lane per alignment, the mate's
// alignments staged in LDS; hr_list holds (read, barcode) pairs, bc_lmp the barcode's log molecule penalty (calculateLogMoleculePenalty, left by k_rfa)
__global__ void __launch_bounds__(64) k_rfa_mq_w(DOpts o, const i64* __restrict__ cen_start, const i64* __restrict__ cen_end, DCand R, DInf S, uint8_t* __restrict__ slab_pool,
                                                  i64 slab_bytes, const int32_t* __restrict__ hr_list, const int32_t* __restrict__ hr_count, const double* __restrict__ bc_lmp,
                                                  int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LH_RFA_LDS_BYTES];
    const int lane = LANE();
    uint8_t* slab = slab_pool + (size_t)blockIdx.x * (size_t)slab_bytes;
    const double improper = o.improper_pair_penalty;
    const int n_items = *hr_count;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const int gr = hr_list[2 * item], gm = gr ^ 1;
            const double lmp = bc_lmp[hr_list[2 * item + 1]];
            const i64 a0 = R.cand_off[gr], a1 = R.cand_off[gr + 1], m0 = R.cand_off[gm], m1 = R.cand_off[gm + 1];
            WAVE_SYNC();   // the previous read's staging has been read
            if ((size_t)(a1 - a0 + 2) * 8 > (size_t)slab_bytes) { if (lane == 0) status[gr] |= LH_ST_POOL_OVERFLOW; continue; }
            const i64 c_lo = m0;
            double* const sc_all = (double*)slab;   // the read's scores: the pseudo-count entry, then one per alignment
            // the mate's filtered alignments staged in LDS, LH_RFA_MQ_CHUNK at a time: single-read score, strand, contig, position
            double* const ms = (double*)lds_raw;                                    // [CH]
            i64* const mp = (i64*)(lds_raw + 8 * LH_RFA_MQ_CHUNK);                  // [CH]
            int32_t* const mi = (int32_t*)(lds_raw + 16 * LH_RFA_MQ_CHUNK);         // [CH] candidate (global - c_lo)
            int32_t* const mr = (int32_t*)(lds_raw + 20 * LH_RFA_MQ_CHUNK);         // [CH] rid << 1 | reversed
            // pass over this read's alignments, 64 at a time: best pair score of each (and where), with the mate chunks as the inner loop
            double pseudo = 0.0, g_sb = -1000.0;   // second best: the running maximum over the inactive alignments before this lane's
            i64 sb_aln = -1;
            double sb_raw = 0.0;
            int n_sc = 0;
            i64 first = -1;
            for (i64 ab = a0; ab < a1; ab += 64) {
                const i64 a = ab + lane;
                const int ok = a < a1 && R.in_filtered[a];
                double sa = 0.0, best = -1.7976931348623157e308, best0 = 0.0;
                i64 bm = -1;
                int a_rev = 0, a_rid = 0, a_am = 0;
                i64 a_pos = 0;
                if (ok) {
                    sa = (double)(R.mismatches[a] * -2 + R.indels[a] * -3);
                    if (R.soft_clipped[a] > 0) { sa -= 5.0 * (double)R.soft_clipped[a]; sa -= (double)R.soft_clipped_length[a] * 0.5; }
                    a_rev = R.reversed[a]; a_rid = R.rid[a]; a_pos = R.pos[a]; a_am = S.active_molecule[a];
                }
                double bsingle = -1.7976931348623157e308;
                for (i64 mb = m0; mb < m1; mb += LH_RFA_MQ_CHUNK) {
                    WAVE_SYNC();   // the previous chunk has been read
                    int nch = 0;
                    for (i64 cb = mb; cb < m1 && cb < mb + LH_RFA_MQ_CHUNK; cb += 64) {   // stage (filtered ones only, compacted, in order)
                        const i64 c = cb + lane;
                        const int okm = c < m1 && c < mb + LH_RFA_MQ_CHUNK && R.in_filtered[c];
                        const u64 mk = __ballot(okm);
                        if (okm) {
                            double sc = (double)(R.mismatches[c] * -2 + R.indels[c] * -3);
                            if (R.soft_clipped[c] > 0) { sc -= 5.0 * (double)R.soft_clipped[c]; sc -= (double)R.soft_clipped_length[c] * 0.5; }
                            const int at = nch + lanes_below(mk, lane);
                            ms[at] = sc; mp[at] = R.pos[c]; mi[at] = (int32_t)(c - c_lo); mr[at] = R.rid[c] << 1 | (R.reversed[c] ? 1 : 0);
                        }
                        nch += __popcll(mk);
                    }
                    WAVE_SYNC();
                    for (int j = 0; j < nch; ++j) {
                        const double sm = ms[j];
                        const int rr = mr[j];
                        const i64 pm = mp[j];
                        if (ab == a0) { const double s1 = (0.0 + sm) + improper; if (s1 > bsingle) bsingle = s1; }   // scoreAlignment(nil, mate, lmp)
                        if (ok) {
                            int pair = 0;
                            if ((rr & 1) != a_rev && (rr >> 1) == a_rid) { const i64 dist = a_rev ? a_pos - pm : pm - a_pos; pair = dist >= -35 && dist < 750; }
                            double t = sa + sm;
                            if (!pair) t += improper;
                            const double t0 = t + 0.0;
                            if (!a_am) t += lmp;
                            if (t > best) { best = t; best0 = t0; bm = c_lo + mi[j]; }
                        }
                    }
                }
                // (the read's first filtered alignment, for the pseudo-count entry: in the first 64-chunk that has one)
                { const u64 fk = __ballot(ok); if (first < 0 && fk) first = ab + (__ffsll((unsigned long long)fk) - 1); }
                if (ab == a0) pseudo = bsingle;
                // this chunk's scores into the list
                { const u64 mk = __ballot(ok); if (ok) sc_all[1 + n_sc + lanes_below(mk, lane)] = best; n_sc += __popcll(mk); }
                // second best (lariat.go:917-943): inactive alignments in order, a strict running maximum; every alignment that raises it gets its mate link
                {
                    const int cand = ok && !S.active[a];
                    double v = cand ? best : -1.7976931348623157e308;
                    double pm_ = v;   // inclusive prefix maximum over the lanes
                    for (int d = 1; d < 64; d <<= 1) { const double o_ = __shfl_up(pm_, d); if (lane >= d && o_ > pm_) pm_ = o_; }
                    double ex = __shfl_up(pm_, 1);
                    if (lane == 0) ex = -1.7976931348623157e308;
                    const double before = ex > g_sb ? ex : g_sb;
                    const int raises = cand && v > before;
                    if (raises) S.mate[a] = bm;
                    const u64 rk = __ballot(raises);
                    if (rk) {
                        const int last = 63 - __clzll((unsigned long long)rk);
                        sb_aln = shfl_i64(a, last);
                        sb_raw = __shfl(best0, last);
                        g_sb = __shfl(v, last);
                    }
                }
            }
            WAVE_SYNC();
            pseudo = __shfl(pseudo, 0) + dev_pseudo_score(R, first, lmp);
            if (lane == 0) sc_all[0] = pseudo;
            n_sc += 1;
            WAVE_SYNC();
            if (sb_aln < 0) sb_raw = pseudo;
            // the 15 best scores, largest first (sort.Float64s, then the last 15 from the top)
            double total = 0.0;
            for (int k = 0; k < 15 && k < n_sc; ++k) {
                double v = -1.7976931348623157e308;
                int at = -1;
                for (int i = lane; i < n_sc; i += 64) { const double x = sc_all[i]; if (x > v || (x == v && at < 0)) { v = x; at = i; } }
                for (int msk = 32; msk >= 1; msk >>= 1) {
                    const double ov = __shfl_xor(v, msk);
                    const int oa = __shfl_xor(at, msk);
                    if (oa >= 0 && (at < 0 || ov > v || (ov == v && oa < at))) { v = ov; at = oa; }
                }
                total += pow(10.0, v);
                WAVE_SYNC();
                if (lane == 0) sc_all[at] = -1.7976931348623157e308;   // taken
                WAVE_SYNC();
            }
            const i64 act = S.active_idx[gr];
            if (lane == 0) {
                S.second_best_idx[gr] = sb_aln; S.second_best_score[gr] = sb_raw;
                S.as_score[gr] = dev_score_aln(R, S, improper, act, S.mate[act], 0.0);
            }
            for (i64 a = a0 + lane; a < a1; a += 64) {
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
            WAVE_SYNC();
    }
}


// markDuplicates (lariat.go:655-685) and CheckSplitReads (split.go:29-158), a wave per barcode, after every read's map qualities are in
__global__ void __launch_bounds__(64, LH_RFA_WAVES) k_rfa_post(DIndex ix, DOpts o, int n_bc, const int32_t* __restrict__ bc_pair_off, const i64* __restrict__ cen_start,
                                                  const i64* __restrict__ cen_end, DCand R, DInf S, i64 cand_cap, uint8_t* __restrict__ slab_pool, i64 slab_bytes,
                                                  int32_t* __restrict__ status, int32_t* __restrict__ bc_next, const int32_t* __restrict__ work_list,
                                                  const int32_t* __restrict__ work_count, int32_t* __restrict__ ovf_list, int32_t* __restrict__ ovf_count) {
    __shared__ int32_t shi[8];
    const int lane = LANE();
    uint8_t* slab = slab_pool + (size_t)blockIdx.x * (size_t)slab_bytes;
    const double improper = o.improper_pair_penalty;
    const int n_work = work_list ? *work_count : n_bc;
    int wd_main = 1 << 24;
    for (;;) {
        LH_WATCH(o.wd, wd_main, 12, break)
        if (lane == 0) shi[5] = atomicAdd(bc_next, 1);
        WAVE_SYNC();
        const int widx = shi[5];
        WAVE_SYNC();
        if (widx >= n_work) break;
        const int bc = work_list ? work_list[widx] : widx;
        const int p0 = bc_pair_off[bc], p1 = bc_pair_off[bc + 1];
        const int nR = 2 * (p1 - p0), r0 = 2 * p0;
        if (R.cand_off[r0 + nR] > cand_cap) continue;   // flagged by k_aln
        int hbits = 6;
        while ((1 << hbits) < 2 * nR) ++hbits;
        size_t so = 0;
        RfaTab T;
#define CARVE(ptr, type, count) { so = (so + 7) & ~(size_t)7; T.ptr = (type*)(slab + so); so += sizeof(type) * (size_t)(count); }
        CARVE(dk0, u64, nR) CARVE(dk1, u64, nR) CARVE(dk2, u64, nR) CARVE(dk3, u64, nR) CARVE(htab, int32_t, (size_t)1 << hbits)
        CARVE(gstk, int32_t, 3 * LH_GOSORT_STK * 64) CARVE(spl, i64, LH_SPLIT_MAX * 64)
#undef CARVE
        if (so > (size_t)slab_bytes) RFA_OVERFLOW()   // barcode too large for the slab
        // ---- markDuplicates: first-seen wins on (read1?, reversed, contig, pos, mate contig, mate pos) in read order ----
        {   // open-addressing table over the keys; a slot ends up holding the smallest read index of its key
            const int hmask = (1 << hbits) - 1;
            for (int x = lane; x <= hmask; x += 64) T.htab[x] = -1;
            for (int r = lane; r < nR; r += 64) {
                i64 a = S.active_idx[r0 + r], m = S.mate[a];
                T.dk0[r] = (u64)R.pos[a]; T.dk1[r] = (u64)R.pos[m];
                T.dk2[r] = (u64)(uint32_t)R.rid[a] << 32 | (u64)(R.reversed[a] != 0) << 1 | (u64)(r & 1);
                T.dk3[r] = (u64)(uint32_t)R.rid[m];
            }
            WAVE_SYNC();
            for (int r = lane; r < nR; r += 64) {
                u64 k0 = T.dk0[r], k1 = T.dk1[r], k2 = T.dk2[r], k3 = T.dk3[r];
                int slot = (int)(dev_mix64(k0 ^ dev_mix64(k1 ^ dev_mix64(k2 ^ dev_mix64(k3)))) & (u64)hmask);
                int wd_ins = hmask + 2;
                for (;;) {
                    LH_WATCH(o.wd, wd_ins, 13, break)
                    int cur = atomicCAS(&T.htab[slot], -1, r);
                    if (cur == -1) break;
                    if (T.dk0[cur] == k0 && T.dk1[cur] == k1 && T.dk2[cur] == k2 && T.dk3[cur] == k3) { atomicMin(&T.htab[slot], r); break; }
                    slot = (slot + 1) & hmask;
                }
            }
            WAVE_SYNC();
            for (int r = lane; r < nR; r += 64) {
                u64 k0 = T.dk0[r], k1 = T.dk1[r], k2 = T.dk2[r], k3 = T.dk3[r];
                int slot = (int)(dev_mix64(k0 ^ dev_mix64(k1 ^ dev_mix64(k2 ^ dev_mix64(k3)))) & (u64)hmask);
                int dup = 0;
                int wd_look = hmask + 2;
                for (;;) {
                    LH_WATCH(o.wd, wd_look, 14, break)
                    int cur = atomicAdd(&T.htab[slot], 0);   // read at L2, where the atomics above landed
                    if (T.dk0[cur] == k0 && T.dk1[cur] == k1 && T.dk2[cur] == k2 && T.dk3[cur] == k3) { dup = cur != r; break; }
                    slot = (slot + 1) & hmask;
                }
                S.duplicate[S.active_idx[r0 + r]] = (uint8_t)dup;
            }
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
            i64* const cidx = T.spl + lane;   // entry i at cidx[i * 64]
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
                    if (R.score[sc] >= 36 || prop) { if (ncand < LH_SPLIT_MAX) cidx[64 * ncand++] = sc; else ovf = 1; }
                }
            }
            if (ovf) status[gr] |= LH_ST_POOL_OVERFLOW;
            if (ncand == 0) continue;
            dev_gosort(ncand, [&](int i, int j) { return R.score[cidx[64 * i]] > R.score[cidx[64 * j]]; }, [&](int i, int j) { i64 t = cidx[64 * i]; cidx[64 * i] = cidx[64 * j]; cidx[64 * j] = t; }, T.gstk + lane, 64);
            i64 c = cidx[0];
            double mapq;
            double second_best = dev_score_aln(R, S, improper, P, -1, 0.0) + dev_pseudo_score(R, c, 0.0);
            if (ncand > 1) { mapq = (double)(R.score[cidx[0]] - R.score[cidx[64]]); second_best = dev_score_aln(R, S, improper, P, cidx[64], 0.0); }
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
#undef RFA_OVERFLOW


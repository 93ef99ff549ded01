// lh_dev.h — device-side common definitions for the gfx950 kernels.
//
// Compiles under hipcc (--offload-arch=gfx950; the product) and, with -DLH_EMU, under g++ against
// tests/hipemu/hip_emu.h (a TEST-ONLY SPMD emulator used by the CPU test-suite; never shipped).
//
// Execution model used by every per-read / per-pair / per-barcode kernel here: ONE 64-lane wavefront per
// work item, workgroup = 1 wave (blockDim.x == 64).  Control flow is wave-uniform; lanes either cooperate
// (occurrence-block loads, DP rows, scans, ballots) or redundantly evaluate the same scalar state, and
// only lane 0 stores scalar results.  WAVE_SYNC() orders lane-0 stores before the wave's later loads.
#pragma once
#include <stdint.h>

#ifdef LH_EMU
#include "hip_emu.h"
#define LH_LAUNCH(kernel, grid, block, stream, ...)                                          \
    do {                                                                                     \
        if (getenv("LH_EMU_TRACE")) fprintf(stderr, "emu launch %s grid=%d\n", #kernel, (int)(grid)); \
        emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); });                \
    } while (0)
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
#else
#include <hip/hip_runtime.h>
#define LH_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

#define LH_WAVE 64
#define LH_MAXLEN 250            // LH_MAX_READ_LEN
#ifndef LH_MAX_INTV
#define LH_MAX_INTV 64           // SMEM intervals kept per read in its regular slots; a read with more is seeded again into a big slab (k_smem4.h, BIG) — tests build with 4
#endif
#define LH_BIG_INTV 1024         // ... which holds this many per read (mem_collect_intv yields at most ~250 for 250 bases outside pathological re-seeding); beyond: LH_ST_INTV_OVERFLOW
#define LH_MAX_CIGAR 64          // cigar ops per candidate
#define LH_MAX_MM 64             // mismatch loci per candidate
#define LH_RESCUE_SLOTS 50       // opt->max_matesw / gobwa.go:287

#ifdef LH_EMU
#define WAVE_SYNC() (emu::note(__FILE__, __LINE__), __syncthreads())
#else
#define WAVE_SYNC() __syncthreads()
#endif
#define LANE() ((int)(threadIdx.x & 63))
// EMU_SYNC(): orders one lane's LDS store after the other lanes' earlier LDS loads.  A hardware wave issues its LDS
// operations in program order for all lanes at once, so nothing is needed there; the emulator's fibers need a rendezvous.
#ifdef LH_EMU
#define EMU_SYNC() WAVE_SYNC()
#else
#define EMU_SYNC() __builtin_amdgcn_wave_barrier()
#endif

typedef uint64_t u64;
typedef int64_t i64;

// LH_UNI(cond): the branch condition of single-lane (lane-0) sequential code, made provably wave-uniform so that hipcc
// emits scalar branches instead of exec-mask loops (deeply nested divergent loops proved fragile on gfx950/ROCm 7.2).
#ifdef LH_EMU
#define LH_UNI(c) (c)
#else
#define LH_UNI(c) (__builtin_amdgcn_readfirstlane((int)(c)) != 0)
#endif

// watchdog: loops that should be short bump a budget; on exhaustion they record a site code in the PIPELINE's slots (DOpts::wd: 32 words owned
// by the lh_context whose kernels these are, read back and cleared with that context's result) and bail out
#define LH_WD_SLOTS 32
#define LH_WATCH(wdp, budget, code, action) if (--(budget) < 0) { (wdp)[code] = 1; action; }
#ifdef LH_NOWATCH_SORT
#define LH_WATCH_S(wdp, budget, code, action)
#else
#define LH_WATCH_S LH_WATCH
#endif
#ifdef LH_NOWATCH_DEDUP
#define LH_WATCH_D(wdp, budget, code, action)
#else
#define LH_WATCH_D LH_WATCH
#endif

// status bits per read
#define LH_ST_INTV_OVERFLOW 1
#define LH_ST_TOO_LONG 2
#define LH_ST_CIGAR_OVERFLOW 4
#define LH_ST_MM_OVERFLOW 8
#define LH_ST_POOL_OVERFLOW 16

struct DIndex {
    // FM-index occurrence table, re-laid out on load from <prefix>.bwt (which interleaves 4 x u64 counts with 128 2-bit
    // symbols per 64 B): per 64 symbols one 32-B record [4 x u32 counts before the block | u64 high bit-plane | u64 low
    // bit-plane] (symbol s of the block = bit s of the planes).  One 32-B read and three 64-bit popcounts per bwt_occ4.
    const uint4* occ;
    const u64* sb;           // absolute counts at every 2^sb_shift-th symbol (the u32 counts are relative to them); null if there is one super-block
    int32_t sb_shift;        // 31 (LH_SB_SHIFT overrides it for tests)
    // only with a fully resident suffix array (sa_intv == 1), else null: the inverse suffix array and the text fwd||rev with 4 bits
    // per base (8 per word, base p at bits 4*(p&7) of word p>>3; padded with 0xF, two words before and three after).  K1 uses them
    // to follow a UNIQUE match through the text instead of through the FM-index (k_smem4.h).
    const u64* isa;
    const uint32_t* tn;
    // lcp[r] = bases shared by the suffixes of rows r - 1 and r, capped at 255 (rows 0, 1 and n + 1: 0); next to isa / tn, else null.
    // K1 (pass 1) reads from it whether every entry of a forward list is unique at the point where its longest one ends (k_smem4.h).
    const uint8_t* lcp;
    // plcp[p] = the most bases the suffix of the text (fwd||rev) at position p shares with any OTHER suffix, capped at 255: max(lcp[r], lcp[r + 1]) at
    // r = isa[p].  Next to isa / tn / lcp, else null.  K1 decides from it — by text position, without a row — whether a match is unique (k_smem4.h).
    const uint8_t* plcp;
    // rep_t: one bit per position of the text fwd||rev, next to lcp, else null: the LH_BLOOM_K-mer that starts there occurs again elsewhere
    // (its row shares LH_BLOOM_K bases with a neighbouring row).  K1's pass 2 reads from it that a re-seeding inside a unique SMEM
    // cannot yield a seed (k_smem4.h, S4_P2_PROBE).
    const u64* rep_t;
    // the bi-interval of every 12-mer (packed like K1's list entries, 16 B each): bwt_seed_strategy1's walks start there
    const void* kmer12;
    // the bi-interval of EVERY string of 1 .. ktree_levels bases (only next to isa / tn; packed like kmer12's entries): level L
    // (strings of L bases) starts at entry (4^L - 4) / 3, a string's index inside its level is its code with base t at bits
    // 2t..2t+1.  K1 reads the result of a bwt_extend that yields a short match here — one 16-B read, cacheable for the
    // shortest — instead of computing it from two occurrence records that lie far apart (k_smem4.h).  Null: not built.
    const void* ktree;
    int32_t ktree_levels;
    // sweep filters (k_smem4.h), only next to isa / tn: blocked Bloom filters (one 64-bit word, 4 bits per key) over the
    // LH_BLOOM_K-mers of the text fwd||rev: bloom1 = every k-mer that occurs, bloom2 = every k-mer that occurs at least twice
    const u64* bloom1;
    const u64* bloom2;
    uint32_t bloom1_words, bloom2_words;
    const u64* sa;           // sampled SA, sa[0] = -1
    const uint8_t* pac;      // 2-bit forward reference, MSB first
    const i64* contig_off;   // [n_contigs]
    const int32_t* contig_len;
    const uint8_t* contig_alt;   // [n_contigs] bntann1_t.is_alt (the .alt file, bwa_idx_load BWA_IDX_ALL: gobwa.go:130); null: no ALT contigs
    const int32_t* rid_bins; // contig holding forward position (b << rid_bin_shift), b = 0 .. l_pac >> shift: bns_pos2rid in one read + a short walk
    int32_t rid_bin_shift;
    u64 primary, L2[5], seq_len;
    i64 l_pac;
    int32_t sa_intv, n_contigs;
};

struct DOpts {   // mem_opt_t + lariat knobs, POD
    int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop, T;
    int32_t min_seed_len, min_chain_weight, max_chain_extend, split_width, max_occ, max_chain_gap, max_ins, max_mem_intv, max_matesw;
    float split_factor, mask_level, drop_ratio, XA_drop_ratio, mask_level_redun, mapQ_coef_len;
    int32_t pes_low, pes_high, rescue_score_delta, rescue_max_hits, aln_score_delta, run_inference;
    double improper_pair_penalty, genome_length;
    int8_t mat[25];
    int32_t* wd;   // the pipeline's watchdog slots (LH_WATCH)
};

struct DIntv { u64 x0, x1, x2, info; };   // bwtintv_t

struct DReg {   // mem_alnreg_t
    i64 rb, re;
    int32_t qb, qe, rid, score, truesc, sub, csub, w, seedcov, secondary, seedlen0, n_comp, is_alt;
    float frac_rep;
};

// telemetry counters: LH_CTR_SLOTS copies on separate 128-B lines, indexed by blockIdx, summed by the host
// (same-address atomics serialize at ~12 ns each: one shared copy cost k_extend ~70 ms per 2M waves)
#define LH_CTR_SLOTS 64
struct DCounters { u64 n_ext, n_lf, n_sa, win_bases, n_chain_ext, ext_cells, glob_cells, n_rescue, rescue_cells, n_ext_exec[3], n_ktree[3], n_bt, rescue_cells_exec, n_glob_listed, n_glob_exec; };   // n_ext_exec: bwt_extend calls K1 really executed on the occurrence table, per pass; n_ktree: those it read from the k-mer tree table
#define LH_CTR(ctr) ((ctr) + (blockIdx.x & (LH_CTR_SLOTS - 1)))

// ------------------------------------------------------------------ lane helpers
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src); hi = __shfl(hi, src);
    return (u64)hi << 32 | lo;
}
__device__ __forceinline__ i64 shfl_i64(i64 v, int src) { return (i64)shfl_u64((u64)v, src); }
__device__ __forceinline__ i64 shfl_xor_i64(i64 v, int m) {
    uint32_t lo = (uint32_t)(u64)v, hi = (uint32_t)((u64)v >> 32);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
    return (i64)((u64)hi << 32 | lo);
}
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int d) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, d); hi = __shfl_up(hi, d);
    return (u64)hi << 32 | lo;
}
#ifdef LH_EMU
__device__ __forceinline__ int wave_max_i32(int v) {
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v > o ? v : o; }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = v < o ? v : o; }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// inclusive prefix max over lanes
__device__ __forceinline__ int wave_scan_max_i32(int v, int lane) {
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v = v > o ? v : o; }
    return v;
}
// inclusive prefix sum over lanes
__device__ __forceinline__ int wave_scan_add_i32(int v) {
    int lane = LANE();
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}
__device__ __forceinline__ int wave_shr1_i32(int v, int fill) { int o = __shfl_up(v, 1); return LANE() == 0 ? fill : o; }   // lane i <- lane i-1
__device__ __forceinline__ int wave_readlane(int v, int l) { return __shfl(v, l); }                                          // l must be wave-uniform
#else
// gfx9 DPP: row_shr:n = 0x110+n, row_bcast:15 = 0x142 (row_mask 0xA), row_bcast:31 = 0x143 (row_mask 0xC), wave_shr:1 = 0x138.
// Disabled / out-of-range lanes keep `old`, which carries the identity.  One VALU op per step instead of a ds_bpermute round trip.
#define LH_DPP(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xF, false)
__device__ __forceinline__ int wave_scan_max_i32(int v, int) {
    const int ID = (int)0x80000000;
    int t;
    t = LH_DPP(ID, v, 0x111, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x112, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x114, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x118, 0xF); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x142, 0xA); v = v > t ? v : t;
    t = LH_DPP(ID, v, 0x143, 0xC); v = v > t ? v : t;
    return v;
}
__device__ __forceinline__ int wave_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ int wave_shr1_i32(int v, int fill) { return LH_DPP(fill, v, 0x138, 0xF); }
__device__ __forceinline__ int wave_max_i32(int v) { return wave_readlane(wave_scan_max_i32(v, 0), 63); }
__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }
__device__ __forceinline__ int wave_scan_add_i32(int v) {   // inclusive prefix sum over lanes
    v += LH_DPP(0, v, 0x111, 0xF);
    v += LH_DPP(0, v, 0x112, 0xF);
    v += LH_DPP(0, v, 0x114, 0xF);
    v += LH_DPP(0, v, 0x118, 0xF);
    v += LH_DPP(0, v, 0x142, 0xA);
    v += LH_DPP(0, v, 0x143, 0xC);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += LH_DPP(0, v, 0x111, 0xF);
    v += LH_DPP(0, v, 0x112, 0xF);
    v += LH_DPP(0, v, 0x114, 0xF);
    v += LH_DPP(0, v, 0x118, 0xF);
    v += LH_DPP(0, v, 0x142, 0xA);
    v += LH_DPP(0, v, 0x143, 0xC);
    return wave_readlane(v, 63);
}
#endif
// 64-bit maximum / minimum over the wave
#ifdef LH_EMU
__device__ __forceinline__ i64 wave_max_i64(i64 v) { for (int m = 32; m >= 1; m >>= 1) { i64 o = shfl_xor_i64(v, m); v = v > o ? v : o; } return v; }
__device__ __forceinline__ i64 wave_min_i64(i64 v) { for (int m = 32; m >= 1; m >>= 1) { i64 o = shfl_xor_i64(v, m); v = v < o ? v : o; } return v; }
#else
// (DPP on both halves: six steps of two moves, a compare and a select instead of twelve ds_bpermute round trips)
#define LH_DPP64_STEP(ctrl, rmask, id_hi, id_lo, cmp)                                                       \
    {                                                                                                      \
        const int lo_ = LH_DPP((int)(id_lo), (int)(uint32_t)(u64)v, ctrl, rmask);                          \
        const int hi_ = LH_DPP((int)(id_hi), (int)((u64)v >> 32), ctrl, rmask);                            \
        const i64 t_ = (i64)((u64)(uint32_t)hi_ << 32 | (u64)(uint32_t)lo_);                               \
        v = v cmp t_ ? v : t_;                                                                             \
    }
__device__ __forceinline__ i64 wave_max_i64(i64 v) {
    LH_DPP64_STEP(0x111, 0xF, 0x80000000u, 0u, >) LH_DPP64_STEP(0x112, 0xF, 0x80000000u, 0u, >) LH_DPP64_STEP(0x114, 0xF, 0x80000000u, 0u, >)
    LH_DPP64_STEP(0x118, 0xF, 0x80000000u, 0u, >) LH_DPP64_STEP(0x142, 0xA, 0x80000000u, 0u, >) LH_DPP64_STEP(0x143, 0xC, 0x80000000u, 0u, >)
    return (i64)((u64)(uint32_t)wave_readlane((int)((u64)v >> 32), 63) << 32 | (u64)(uint32_t)wave_readlane((int)(uint32_t)(u64)v, 63));
}
__device__ __forceinline__ i64 wave_min_i64(i64 v) {
    LH_DPP64_STEP(0x111, 0xF, 0x7fffffffu, 0xffffffffu, <) LH_DPP64_STEP(0x112, 0xF, 0x7fffffffu, 0xffffffffu, <) LH_DPP64_STEP(0x114, 0xF, 0x7fffffffu, 0xffffffffu, <)
    LH_DPP64_STEP(0x118, 0xF, 0x7fffffffu, 0xffffffffu, <) LH_DPP64_STEP(0x142, 0xA, 0x7fffffffu, 0xffffffffu, <) LH_DPP64_STEP(0x143, 0xC, 0x7fffffffu, 0xffffffffu, <)
    return (i64)((u64)(uint32_t)wave_readlane((int)((u64)v >> 32), 63) << 32 | (u64)(uint32_t)wave_readlane((int)(uint32_t)(u64)v, 63));
}
#endif
__device__ __forceinline__ int lanes_below(u64 mask, int lane) { return __popcll(mask & ((1ull << lane) - 1)); }

// DPP lane exchanges inside a 16-lane row (1 VALU op instead of an LDS-crossbar ds_bpermute round trip)
#ifdef LH_EMU
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return __shfl_xor(v, 1); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return __shfl_xor(v, 2); }
__device__ __forceinline__ uint32_t dpp_half_mirror(uint32_t v) { return __shfl(v, (LANE() & ~7) | (7 - (LANE() & 7))); }
__device__ __forceinline__ uint32_t dpp_ror8(uint32_t v) { return __shfl_xor(v, 8); }
#else
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ uint32_t dpp_half_mirror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); }   // row_half_mirror
__device__ __forceinline__ uint32_t dpp_ror8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); }   // row_ror:8
#endif
#ifdef LH_EMU
template <int K> __device__ __forceinline__ uint32_t dpp_quad_bcast(uint32_t v) { return __shfl(v, (LANE() & ~3) | K); }
#else
template <int K> __device__ __forceinline__ uint32_t dpp_quad_bcast(uint32_t v) {   // quad_perm [K,K,K,K]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K | K << 2 | K << 4 | K << 6, 0xF, 0xF, true);
}
#endif
template <int K> __device__ __forceinline__ u64 dpp_quad_bcast_u64(u64 v) { return (u64)dpp_quad_bcast<K>((uint32_t)(v >> 32)) << 32 | dpp_quad_bcast<K>((uint32_t)v); }
__device__ __forceinline__ u64 dpp_ror8_u64(u64 v) { return (u64)dpp_ror8((uint32_t)(v >> 32)) << 32 | dpp_ror8((uint32_t)v); }

// ------------------------------------------------------------------ k-mer membership filters
#define LH_BLOOM_K 19   // = bwa's default min_seed_len: the filters are only consulted when opts.min_seed_len >= LH_BLOOM_K
// key: base j of the k-mer (j = 0 the first one) at bits 2j..2j+1
__device__ __forceinline__ void dev_bloom_slot(u64 key, uint32_t n_words, uint32_t* word, u64* mask) {
    u64 h = key * 0x9E3779B97F4A7C15ull;
    *word = __umulhi((uint32_t)(h >> 32), n_words);
    uint32_t g = (uint32_t)(h >> 16) * 0x85EBCA6Bu;
    g ^= g >> 15;
    *mask = 1ull << (g & 63) | 1ull << ((g >> 6) & 63) | 1ull << ((g >> 12) & 63) | 1ull << ((g >> 18) & 63);
}

// ------------------------------------------------------------------ FM-index primitives (restated from BWA bwt.c)
// packed per-base counts of the 16 symbols of w under the 2-bit-position mask `valid` (0x55555555 = all 16)
// bwt_occ4: occurrences of each base in BWT[0..k]
__device__ __forceinline__ void dev_occ4(const DIndex& ix, u64 k, u64 cnt[4]) {
    if (k == (u64)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= ix.primary);
    const uint4* p = ix.occ + ((k >> 6) << 1);
    uint4 h = p[0], d = p[1];
    int m = (int)(k & 63) + 1;                     // symbols of the block that count
    u64 mask = ~0ull >> (64 - m);
    u64 hi = ((u64)d.y << 32 | d.x) & mask, lo = ((u64)d.w << 32 | d.z) & mask;
    u64 i3 = hi & lo, i2 = hi ^ i3, i1 = lo ^ i3;
    int n3 = __popcll(i3), n2 = __popcll(i2), n1 = __popcll(i1);
    cnt[0] = (u64)h.x + (u64)(m - n1 - n2 - n3); cnt[1] = (u64)h.y + (u64)n1; cnt[2] = (u64)h.z + (u64)n2; cnt[3] = (u64)h.w + (u64)n3;
    if (ix.sb) { const u64* sb = ix.sb + ((k >> ix.sb_shift) << 2); cnt[0] += sb[0]; cnt[1] += sb[1]; cnt[2] += sb[2]; cnt[3] += sb[3]; }
}

// counts of one record up to and including symbol (k & 63): the popcount half of bwt_occ4
__device__ __forceinline__ void dev_occ4_rec(const DIndex& ix, u64 k, uint4 h, uint4 d, u64 cnt[4]) {
    int m = (int)(k & 63) + 1;
    u64 mask = ~0ull >> (64 - m);
    u64 hi = ((u64)d.y << 32 | d.x) & mask, lo = ((u64)d.w << 32 | d.z) & mask;
    u64 i3 = hi & lo, i2 = hi ^ i3, i1 = lo ^ i3;
    int n3 = __popcll(i3), n2 = __popcll(i2), n1 = __popcll(i1);
    cnt[0] = (u64)h.x + (u64)(m - n1 - n2 - n3); cnt[1] = (u64)h.y + (u64)n1; cnt[2] = (u64)h.z + (u64)n2; cnt[3] = (u64)h.w + (u64)n3;
    if (ix.sb) { const u64* sb = ix.sb + ((k >> ix.sb_shift) << 2); cnt[0] += sb[0]; cnt[1] += sb[1]; cnt[2] += sb[2]; cnt[3] += sb[3]; }
}

// bwt_2occ4(k, l): the two positions of an interval usually fall into the same 64-symbol record once the interval is
// small; the record is then read once (L1 does not merge the two in-flight misses, and K1 runs at the request-rate limit
// of the memory system, so the second request is not free).
__device__ __forceinline__ void dev_2occ4(const DIndex& ix, u64 k, u64 l, u64 tk[4], u64 tl[4]) {
    u64 l2 = l - (l >= ix.primary);
    const uint4* pl = ix.occ + ((l2 >> 6) << 1);
    uint4 hl = pl[0], dl = pl[1];
    if (k == (u64)-1) { tk[0] = tk[1] = tk[2] = tk[3] = 0; }
    else {
        u64 k2 = k - (k >= ix.primary);
        uint4 hk = hl, dk = dl;
        if ((k2 >> 6) != (l2 >> 6)) { const uint4* pk = ix.occ + ((k2 >> 6) << 1); hk = pk[0]; dk = pk[1]; }
        dev_occ4_rec(ix, k2, hk, dk, tk);
    }
    dev_occ4_rec(ix, l2, hl, dl, tl);
}

// bwt_extend restricted to the one base `c` the caller follows: returns ok[c]
__device__ __forceinline__ DIntv dev_extend_c(const DIndex& ix, const DIntv& ik, int c, int is_back) {
    u64 tk[4], tl[4];
    u64 xa = is_back ? ik.x0 : ik.x1;   // x[!is_back]
    u64 xb = is_back ? ik.x1 : ik.x0;   // x[is_back]
    dev_2occ4(ix, xa - 1, xa - 1 + ik.x2, tk, tl);
    u64 s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
    u64 acc = xb + ((xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) ? 1 : 0);   // ok[3].x[is_back]
    u64 o3 = acc, o2 = o3 + s3, o1 = o2 + s2, o0 = o1 + s1;
    u64 na = c == 0 ? ix.L2[0] + 1 + tk[0] : c == 1 ? ix.L2[1] + 1 + tk[1] : c == 2 ? ix.L2[2] + 1 + tk[2] : ix.L2[3] + 1 + tk[3];
    u64 nb = c == 0 ? o0 : c == 1 ? o1 : c == 2 ? o2 : o3;
    DIntv o;
    o.x2 = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
    if (is_back) { o.x0 = na; o.x1 = nb; } else { o.x1 = na; o.x0 = nb; }
    o.info = 0;
    return o;
}

__device__ __forceinline__ DIntv dev_set_intv(const DIndex& ix, int c) {
    DIntv ik;
    ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c + 1] - ix.L2[c]; ik.x1 = ix.L2[3 - c] + 1; ik.info = 0;
    return ik;
}

// bwt_sa: walk inverse-Psi until a sampled row; *n_lf counts LF steps
__device__ __forceinline__ u64 dev_sa(const DIndex& ix, u64 k, int* n_lf) {
    u64 sa = 0, mask = (u64)ix.sa_intv - 1;
    int steps = 0;
    while (k & mask) {
        ++sa; ++steps;
        if (k == ix.primary) { k = 0; continue; }
        u64 x = k - (k > ix.primary);
        uint4 pl = ix.occ[((x >> 6) << 1) + 1];
        int sh = (int)(x & 63);
        uint32_t c = (uint32_t)((((u64)pl.y << 32 | pl.x) >> sh & 1) << 1 | (((u64)pl.w << 32 | pl.z) >> sh & 1));   // bwt_B0
        u64 cnt[4];
        dev_occ4(ix, k, cnt);
        k = ix.L2[c] + (c == 0 ? cnt[0] : c == 1 ? cnt[1] : c == 2 ? cnt[2] : cnt[3]);
    }
    *n_lf = steps;
    return sa + ix.sa[k / ix.sa_intv];
}

// ------------------------------------------------------------------ bntseq.c
__device__ __forceinline__ int dev_pac(const uint8_t* pac, i64 l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }
// base at coordinate p of the fwd||rev reference
__device__ __forceinline__ int dev_ref_base(const DIndex& ix, i64 p) {
    return p < ix.l_pac ? dev_pac(ix.pac, p) : 3 - dev_pac(ix.pac, (ix.l_pac << 1) - 1 - p);
}
// the reference bases of one lane's extension / alignment, read 16 at a time from the 2-bit forward array
struct LaneTgt {
    const uint32_t* pac32;
    i64 idx0, widx;
    int dir, comp;
    uint32_t w;
    __device__ __forceinline__ void init(const DIndex& ix, i64 p0, int tstep) {
        pac32 = (const uint32_t*)ix.pac;
        int fwd = p0 < ix.l_pac;
        idx0 = fwd ? p0 : (ix.l_pac << 1) - 1 - p0;
        dir = fwd ? tstep : -tstep;
        comp = fwd ? 0 : 3;
        widx = -1; w = 0;
    }
    __device__ __forceinline__ int base(int i) {
        i64 ii = idx0 + (i64)dir * i;
        i64 wi = ii >> 4;
        if (wi != widx) { w = pac32[wi]; widx = wi; }
        return (int)((w >> (8 * (int)((ii >> 2) & 3) + (int)((~ii & 3) << 1))) & 3) ^ comp;
    }
};

// ------------------------------------------------------------------ ungapped extension profile (K3's routing, K4's shortcuts)
// eight 4-bit symbols from position p of a packed stream (symbol i at bits 4*(i&7) of word i>>3; p >= -16: the streams are padded)
__device__ __forceinline__ uint32_t dev_nib8(const uint32_t* __restrict__ a, i64 p) {
    const i64 w = p >> 3;
    const int sh = (int)(p & 7) * 4;
    const uint32_t lo = a[w], hi = a[w + 1];
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}
// The diagonal of a ksw_extend2 call: query base k = q[qoff + qstep * k] against the text at tc0 + tstep * k (fwd||rev coordinates),
// k = 0 .. qlen - 1, scored like the DP scores a diagonal step.  P = score lost against all-matches, sc_run = h0 + the diagonal's
// score, mx / mxk = the first maximum of the running score over k (mx = h0, mxk = -1: never above h0).  The walk stops when P reaches
// p_cap or the running score drops to zero (done = 0).  With the 4-bit text (DIndex::tn) and the batch's 4-bit reads (q4; q4pos = the
// stream position of q[0]) eight bases are one XOR; otherwise base by base from the 2-bit reference.
struct DiagScan { int done, P, sc_run, mx, mxk; };
__device__ __forceinline__ DiagScan dev_diag_scan(const DIndex& ix, const DOpts& o, const uint8_t* __restrict__ q, const uint32_t* __restrict__ q4, i64 q4pos, int qoff, int qstep,
                                                  int qlen, i64 tc0, int tstep, int h0, int p_cap) {
    DiagScan d;
    d.P = 0; d.sc_run = h0; d.mx = h0; d.mxk = -1; d.done = 0;
    int k = 0;
    if (ix.tn && q4) {
        const i64 qp = q4pos + qoff;
        for (; k < qlen; k += 8) {
            const int nb = qlen - k < 8 ? qlen - k : 8;
            uint32_t qw, tw, x;
            if (qstep > 0) {   // symbol j of the words = step k + j
                qw = dev_nib8(q4, qp + k); tw = dev_nib8(ix.tn, tc0 + k);
                x = (qw ^ tw) & (nb < 8 ? (1u << (4 * nb)) - 1u : 0xffffffffu);
            } else {           // symbol j = step k + 7 - j
                qw = dev_nib8(q4, qp - k - 7); tw = dev_nib8(ix.tn, tc0 - k - 7);
                x = (qw ^ tw) & (nb < 8 ? ~((1u << (4 * (8 - nb))) - 1u) : 0xffffffffu);
            }
            if (x == 0) {   // nb matches: the running score rises with every step
                d.sc_run += nb * o.a;
                if (d.sc_run > d.mx) { d.mx = d.sc_run; d.mxk = k + nb - 1; }
                continue;
            }
            int stop = 0;
            for (int u = 0; u < nb; ++u) {
                const int j = qstep > 0 ? u : 7 - u;
                const int qv = (int)((qw >> (4 * j)) & 0xf), ne = (int)((x >> (4 * j)) & 0xf);
                const int sc = qv > 3 ? -1 : (ne ? -o.b : o.a);
                d.P += o.a - sc;
                d.sc_run += sc;
                if (d.P >= p_cap || d.sc_run <= 0) { stop = 1; break; }
                if (d.sc_run > d.mx) { d.mx = d.sc_run; d.mxk = k + u; }
            }
            if (stop) return d;
        }
        d.done = 1;
        return d;
    }
    LaneTgt tg;
    tg.init(ix, tc0, tstep);
    for (; k < qlen; ++k) {
        int qv = q[qoff + qstep * k], tb = tg.base(k);
        int sc = qv > 3 ? -1 : (tb == qv ? o.a : -o.b);
        d.P += o.a - sc;
        d.sc_run += sc;
        if (d.P >= p_cap || d.sc_run <= 0) return d;
        if (d.sc_run > d.mx) { d.mx = d.sc_run; d.mxk = k; }
    }
    d.done = 1;
    return d;
}
// one 64-bit value into page-locked host memory the device can write: a read-back that does not queue behind the bulk transfers of the copy engines
__global__ void k_peek_i64(const i64* __restrict__ src, i64* __restrict__ dst_host) { *dst_host = *src; }
__global__ void k_peek_i32(const int32_t* __restrict__ src, i64* __restrict__ dst_host) { dst_host[0] = *src; }
__global__ void k_peek_i64_i32(const i64* __restrict__ src, const int32_t* __restrict__ src2, i64* __restrict__ dst_host) { dst_host[0] = *src; dst_host[1] = *src2; }
// the batch's reads as a 4-bit stream (base i of the batch buffer at symbol i; non-bases = 4), two words of padding in front
__global__ void __launch_bounds__(256) k_pack_reads(const uint8_t* __restrict__ seq, i64 n_bases, uint32_t* __restrict__ q4) {
    const i64 nw = (n_bases + 7) / 8;   // (the batch buffer is 8-aligned and padded past n_bases)
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < nw + 2; w += (i64)gridDim.x * blockDim.x) {
        u64 v8 = 0x0404040404040404ull;
        if (w < nw) __builtin_memcpy(&v8, __builtin_assume_aligned(seq + w * 8, 8), 8);
        uint32_t v = 0;
        for (int b = 0; b < 8; ++b) {
            const i64 i = w * 8 + b;
            uint32_t c = i < n_bases ? (uint32_t)((v8 >> (8 * b)) & 0xff) : 4;
            v |= (c > 4 ? 4u : c) << (4 * b);
        }
        q4[w] = v;
    }
}

__device__ __forceinline__ int dev_pos2rid(const DIndex& ix, i64 pos_f) {
    int left, mid, right;
    if (pos_f >= ix.l_pac) return -1;
    if (ix.rid_bins) {   // the contig of the bin's first base, then forward: the same contig bns_pos2rid's binary search finds
        int rid = ix.rid_bins[pos_f >> ix.rid_bin_shift];
        while (rid + 1 < ix.n_contigs && pos_f >= ix.contig_off[rid + 1]) ++rid;
        return rid;
    }
    left = 0; mid = 0; right = ix.n_contigs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix.contig_off[mid]) {
            if (mid == ix.n_contigs - 1) break;
            if (pos_f < ix.contig_off[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
__device__ __forceinline__ i64 dev_depos(const DIndex& ix, i64 pos, int* is_rev) {
    return (*is_rev = (pos >= ix.l_pac)) ? (ix.l_pac << 1) - 1 - pos : pos;
}
__device__ __forceinline__ int dev_intv2rid(const DIndex& ix, i64 rb, i64 re) {
    int is_rev, rid_b, rid_e;
    if (rb < ix.l_pac && re > ix.l_pac) return -2;
    rid_b = dev_pos2rid(ix, dev_depos(ix, rb, &is_rev));
    rid_e = rb < re ? dev_pos2rid(ix, dev_depos(ix, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}
// bns_fetch_seq's clamping: [*beg,*end) restricted to the contig (on the strand of mid) that contains mid
__device__ __forceinline__ int dev_fetch_clamp(const DIndex& ix, i64* beg, i64 mid, i64* end) {
    int is_rev;
    if (*end < *beg) { i64 t = *beg; *beg = *end; *end = t; }
    int rid = dev_pos2rid(ix, dev_depos(ix, mid, &is_rev));
    i64 far_beg = ix.contig_off[rid], far_end = far_beg + ix.contig_len[rid];
    if (is_rev) { i64 t = far_beg; far_beg = (ix.l_pac << 1) - far_end; far_end = (ix.l_pac << 1) - t; }
    *beg = *beg > far_beg ? *beg : far_beg;
    *end = *end < far_end ? *end : far_end;
    return rid;
}

__device__ __forceinline__ int dev_cal_max_gap(const DOpts& o, int qlen) {
    int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}

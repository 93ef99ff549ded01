// k_diag.h — diagnostic microbenchmark: rate of independent random reads of `granule`-byte blocks (64 B = one BWA
// occurrence block) from a table in HBM.  SURVEY.md §8d: the 64-B-granule random rate is the practical ceiling of the
// FM-index walks (K1/K2) and must be measured, not assumed from the streaming figure.
#pragma once
#include "lh_dev.h"

__global__ void __launch_bounds__(256) k_diag_random_read(const uint4* __restrict__ table, u64 n_blocks, int vec_per_block, int per_thread, u64 seed,
                                                          uint32_t* __restrict__ sink) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x = seed + t * 0x9e3779b97f4a7c15ull;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;   // xorshift64*
        u64 b = (x * 0x2545F4914F6CDD1Dull) % n_blocks;
        const uint4* p = table + b * (u64)vec_per_block;
        for (int v = 0; v < vec_per_block; ++v) { uint4 d = p[v]; acc ^= d.x ^ d.y ^ d.z ^ d.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;   // keeps the loads alive
}

// Go's math/rand stream as K8 produces it (k_rfa.h): lane 0 draws on the state-free path (first min(n, 273) values), lane 1
// with the state ring; parity tests compare both with the oracle's restatement of rng.go and with Go's known Seed(1) values.
__global__ void __launch_bounds__(64) k_diag_go_rand(u64 seed, int n, u64* __restrict__ ring, u64* __restrict__ out_fast, u64* __restrict__ out_ring,
                                                     double* __restrict__ f_ring) {
    const int lane = LANE();
    if (lane == 0) {
        DGoRng g; dev_go_seed(g, seed, nullptr);
        for (int i = 0; i < n && i < LH_GO_TAP; ++i) out_fast[i] = dev_go_u64(g);
    } else if (lane == 1) {
        DGoRng g; dev_go_seed(g, seed, ring + lane);
        for (int i = 0; i < n; ++i) out_ring[i] = dev_go_u64(g);
        DGoRng h; dev_go_seed(h, seed, ring + lane);
        for (int i = 0; i < n; ++i) f_ring[i] = dev_go_f64(h);
    }
}

// Self-check of a resident index with a dense suffix array (sampled rows r = k * stride): (1) suffix sa[r] < suffix sa[r+1]
// by direct comparison of the text; (2) the BWT symbol stored for row r+1 is the text base before its suffix, and the
// LF-mapping through the occurrence table lands on the row of that longer suffix (isa).  out: [0] rows checked,
// [1] order violations, [2] BWT / LF violations.  Size-independent property test for indexes too large for the oracle.
__global__ void __launch_bounds__(256) k_diag_index_check(DIndex ix, u64 stride, unsigned long long* __restrict__ out) {
    const u64 n = ix.seq_len;
    unsigned long long checked = 0, bad_order = 0, bad_lf = 0;
    for (u64 r = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * stride; r < n; r += (u64)gridDim.x * blockDim.x * stride) {
        u64 a = r == 0 ? n : ix.sa[r], b = ix.sa[r + 1];
        ++checked;
        if (a != n) {
            int verdict = 0;
            for (u64 i = 0; i < (1u << 20) && !verdict; ++i) {
                if (a + i >= n) verdict = 1;            // a is a proper prefix of b: smaller
                else if (b + i >= n) verdict = -1;
                else {
                    int ca = dev_ref_base(ix, (i64)(a + i)), cb = dev_ref_base(ix, (i64)(b + i));
                    if (ca != cb) verdict = ca < cb ? 1 : -1;
                }
            }
            if (verdict <= 0) ++bad_order;
        }
        u64 k = r + 1;
        if (k != ix.primary) {
            u64 x = k - (k > ix.primary);
            uint4 pl = ix.occ[((x >> 6) << 1) + 1];
            int sh = (int)(x & 63);
            int c = (int)((((u64)pl.y << 32 | pl.x) >> sh & 1) << 1 | (((u64)pl.w << 32 | pl.z) >> sh & 1));
            u64 cnt[4];
            dev_occ4(ix, k, cnt);
            u64 lf = ix.L2[c] + cnt[c];
            if (b == 0 || c != dev_ref_base(ix, (i64)(b - 1)) || (ix.isa && ix.isa[b - 1] != lf) || ix.sa[lf] != b - 1) ++bad_lf;
        } else if (b != 0) ++bad_lf;
    }
    if (checked) atomicAdd(&out[0], checked);
    if (bad_order) atomicAdd(&out[1], bad_order);
    if (bad_lf) atomicAdd(&out[2], bad_lf);
}

// order-sensitive checksums of the LCP array and of the k-mer tree table (tests: the builder's key-derived tables against the
// generic text-derived ones)
__global__ void __launch_bounds__(256) k_diag_digest(const uint8_t* __restrict__ lcp, u64 n_lcp, const u64* __restrict__ tree, u64 n_tree_words, unsigned long long* __restrict__ out) {
    unsigned long long a = 0, b = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_lcp; i += (u64)gridDim.x * blockDim.x) a += (i + 1) * (u64)lcp[i];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_tree_words; i += (u64)gridDim.x * blockDim.x) b += (i * 0x9E3779B97F4A7C15ull + 1) * tree[i];
    if (a) atomicAdd(&out[0], a);
    if (b) atomicAdd(&out[1], b);
}

// Go's sort.Sort as K8 runs it (lh_sort.h): `n_sorts` index spaces of keys[] sorted (a) one sort per lane by the serial restatement,
// into perm_serial, (b) all at once by the wave-wide one, into perm_wave.  Both must issue Go's Less / Swap sequence per range: equal
// keys end up where Go leaves them.  q*: the wave-wide sort's queue of ranges (n ints each); sa / sb: the places a whole-wave doPivot swaps (n / 2 each).
__global__ void __launch_bounds__(64) k_diag_gosort(int n_sorts, const int32_t* __restrict__ first, i64* __restrict__ keys_a, int32_t* __restrict__ perm_serial,
                                                    i64* __restrict__ keys_b, int32_t* __restrict__ perm_wave, int32_t* __restrict__ qa, int32_t* __restrict__ qb,
                                                    int32_t* __restrict__ qd, int32_t* __restrict__ sa, int32_t* __restrict__ sb) {
    const int lane = LANE();
    for (int k = lane; k < n_sorts; k += 64) {
        i64* kp = keys_a + first[k];
        int32_t* ip = perm_serial + first[k];
        dev_gosort(first[k + 1] - first[k], [&](int i, int j) { return kp[i] < kp[j]; },
                   [&](int i, int j) { i64 t = kp[i]; kp[i] = kp[j]; kp[j] = t; int u = ip[i]; ip[i] = ip[j]; ip[j] = u; }, qa + lane, 64);
    }
    WAVE_SYNC();
    wave_gosort(n_sorts, first, [&](int i, int j) { return keys_b[i] < keys_b[j]; },
                [&](int i, int j) { i64 t = keys_b[i]; keys_b[i] = keys_b[j]; keys_b[j] = t; int u = perm_wave[i]; perm_wave[i] = perm_wave[j]; perm_wave[j] = u; }, qa, qb, qd, sa, sb);
}

// Go's sort.Sort of ONE index space the way K8 sorts a contig list too long for LDS: the ranges longer than `limit` partitioned by the whole wave (wave_gosort_split),
// every range left sorted on its own with the depth the long sort has left it (K8 does that from LDS; here where it lies)
__global__ void __launch_bounds__(64) k_diag_gosort_split(int n, i64* __restrict__ keys, int32_t* __restrict__ perm, int limit, int32_t* __restrict__ qa, int32_t* __restrict__ qb,
                                                          int32_t* __restrict__ qd, int32_t* __restrict__ sa, int32_t* __restrict__ sb) {
    __shared__ int32_t fr[2];
    const int lane = LANE();
    auto less = [&](int i, int j) { return keys[i] < keys[j]; };
    auto swp = [&](int i, int j) { i64 t = keys[i]; keys[i] = keys[j]; keys[j] = t; int u = perm[i]; perm[i] = perm[j]; perm[j] = u; };
    const int nq = wave_gosort_split(0, n, limit, less, swp, qa, qb, qd, sa, sb);
    const int qo = n / 16 + 64;
    for (int q = 0; q < nq; ++q) {
        const int a = qa[q], b = qb[q], d = qd[q];
        if (b - a < 2) continue;
        WAVE_SYNC();
        if (lane == 0) { fr[0] = a; fr[1] = b; }
        WAVE_SYNC();
        wave_gosort(1, fr, less, swp, qa + qo, qb + qo, qd + qo, sa, sb, d);
        WAVE_SYNC();
    }
}

// K8's sorting network for distinct keys (lh_sort.h): n words sorted in LDS when they fit a block, otherwise block by block with passes in memory between them
// (block: 64 or 1,024 places; block 0: the whole list in memory, one pass per step — the form the blocks replace)
__global__ void __launch_bounds__(64) k_diag_bitonic(int n, u64* __restrict__ a, int block) {
    __shared__ u64 lk[1024];
    const int lane = LANE();
    if (block == 0) wave_bitonic_u64(a, n, lane);
    else if (n <= block) {
        for (int i = lane; i < n; i += 64) lk[i] = a[i];
        WAVE_SYNC();
        wave_bitonic_u64(lk, n, lane);
        for (int i = lane; i < n; i += 64) a[i] = lk[i];
    } else if (block == 64) wave_bitonic_u64_blocks<64>(a, n, lk, lane);
    else wave_bitonic_u64_blocks<1024>(a, n, lk, lane);
}

// klib's introsort as the region sorts run it (lh_sort.h): every index space of keys[] as packed words (key << 11 | index, compared above the index) sorted (a) by one lane
// (dev_introsort), (b) by the wave (wave_introsort_i64); both must leave equal keys where ks_introsort does.  At most LH_DIAG_ISORT_MAX elements per sort.
#define LH_DIAG_ISORT_MAX 1024
__global__ void __launch_bounds__(64) k_diag_introsort(int n_sorts, const int32_t* __restrict__ first, const i64* __restrict__ keys, int32_t* __restrict__ perm_serial,
                                                       int32_t* __restrict__ perm_wave, int32_t* __restrict__ wd) {
    __shared__ i64 lk[LH_DIAG_ISORT_MAX];
    __shared__ uint16_t pa[LH_DIAG_ISORT_MAX], pb[LH_DIAG_ISORT_MAX];
    const int lane = LANE();
    for (int c = blockIdx.x; c < n_sorts; c += gridDim.x) {
        const int f0 = first[c], n = first[c + 1] - f0;
        if (n > LH_DIAG_ISORT_MAX) continue;
        WAVE_SYNC();
        for (int k = lane; k < n; k += 64) lk[k] = keys[f0 + k] << 11 | (i64)k;
        WAVE_SYNC();
        if (lane == 0) dev_introsort(n, lk, [&](i64 x, i64 y) { return (x >> 11) < (y >> 11); }, wd);
        WAVE_SYNC();
        for (int k = lane; k < n; k += 64) perm_serial[f0 + k] = (int)(lk[k] & 2047);
        WAVE_SYNC();
        for (int k = lane; k < n; k += 64) lk[k] = keys[f0 + k] << 11 | (i64)k;
        WAVE_SYNC();
        wave_introsort_i64<LH_DIAG_ISORT_MAX / 64>(n, lk, 11, lane, pa, pb, wd);
        WAVE_SYNC();
        for (int k = lane; k < n; k += 64) perm_wave[f0 + k] = (int)(lk[k] & 2047);
    }
}

// K6's exact shortcut (k_rescue2.h: resc_dedup_incremental) against what it replaces, on arbitrary region lists: per case a list of regions
// (6 values each: rb, re, qb, qe, score, rid) and one more region b.  The wave makes the list clean as the pipeline does (mem_sort_dedup_patch
// once), then runs (a) the call as written on list + b in memory, (b) the incremental form on the list in LDS.  verdict[c]: 0 = equal, 1 = the
// incremental form declined (equal keys), 2 = THEY DIFFER, 3 = the cleaned list has equal re or contigs that interleave (the pipeline would not use the incremental form), 4 = THE CALL AS WRITTEN ON THE
// LIST IN LDS (resc_dedup_lds; run for every case) DIFFERS from the one in memory — in a field, or in which of two identical hits it kept.
__global__ void __launch_bounds__(64) k_diag_resc_dedup(DIndex ix, DOpts o, int n_cases, const int32_t* __restrict__ first, const i64* __restrict__ vals, const i64* __restrict__ bvals,
                                                        DReg* __restrict__ ra, DReg* __restrict__ rb_, DReg* __restrict__ tmp, int32_t* __restrict__ ia, int32_t* __restrict__ verdict,
                                                        int32_t* __restrict__ n_out) {
    __shared__ RescList W;
    __shared__ RescScratchT<LH_RA_CAP> S;
    const int lane = LANE();
    for (int c = blockIdx.x; c < n_cases; c += gridDim.x) {
        WAVE_SYNC();
        const int f0 = first[c], n0 = first[c + 1] - f0, stride = f0 + c;   // (room for one more entry per case)
        DReg *A = ra + stride, *B = rb_ + stride, *T = tmp + stride;
        int32_t* I = ia + stride + c;
        auto mk = [&](const i64* v) { DReg g; g.rb = v[0]; g.re = v[1]; g.qb = (int)v[2]; g.qe = (int)v[3]; g.score = (int)v[4]; g.rid = (int)v[5]; g.truesc = 0; g.sub = 0; g.csub = 0; g.w = 0;
                                      g.seedcov = 0; g.secondary = -1; g.seedlen0 = 0; g.n_comp = 0; g.is_alt = 0; g.frac_rep = 0; return g; };
        for (int k = lane; k < n0; k += 64) A[k] = mk(vals + (size_t)(f0 + k) * 6);
        WAVE_SYNC();
        u64 cells = 0;
        int n = wave_sort_dedup_patch(ix, o, nullptr, A, n0, I, T, 0, lane, &cells);   // the clean list: a list a further call leaves alone (with real coordinates one call does
        for (int it = 0; it < 16; ++it) {                                               // that; contigs that interleave can take more, and the pipeline runs such a list's calls as written)
            const int n2 = wave_sort_dedup_patch(ix, o, nullptr, A, n, I, T, 0, lane, &cells);
            if (n2 == n) break;
            n = n2;
        }
        DReg b = mk(bvals + (size_t)c * 6);
        b.seedcov = -1;
        if (n + 1 > LH_RA_CAP) { if (lane == 0) { verdict[c] = 3; n_out[c] = n; } continue; }
        // (seedcov = the entry's place in the clean list, -1 for b: WHICH of two identical hits a call keeps is part of its result)
        auto load_w = [&]() {
            WAVE_SYNC();
            for (int k = lane; k < n; k += 64) { const DReg g = A[k]; W.rb[k] = g.rb; W.re[k] = g.re; W.qb[k] = g.qb; W.qe[k] = g.qe; W.score[k] = g.score; W.rid[k] = g.rid; W.src[k] = k; }
            WAVE_SYNC();
        };
        for (int k = lane; k < n; k += 64) { A[k].seedcov = k; B[k] = A[k]; }
        WAVE_SYNC();
        // (a) as written: b goes in before the first entry with a smaller score (mem_matesw), then the call
        int pos = n;
        for (int k = 0; k < n; ++k) if (B[k].score < b.score) { pos = k; break; }
        WAVE_SYNC();
        if (lane == 0) { for (int k = n; k > pos; --k) B[k] = B[k - 1]; B[pos] = b; }
        WAVE_SYNC();
        const int n_full = wave_sort_dedup_patch(ix, o, nullptr, B, n + 1, I, T, 0, lane, &cells);
        // (c) as written, on the list in LDS (resc_dedup_lds): every case, equal keys or not
        int bad_lds = 0;
        {
            load_w();
            int smax = 0;
            if (resc_lds_keys_ok(W, n, &b, lane, &smax)) {
                const int n1 = resc_list_insert(W, n, b, lane);
                int harmful_lds = 0;
                const int n_lds = resc_dedup_lds(o, W, S, n1, smax, lane, c & 1, &harmful_lds);
                if (n_lds >= 0) {   // its marks and verdict against resc_list_ties' on the same list (a mark too many is allowed, see resc_dedup_lds)
                    int t0[(LH_RA_CAP + 63) / 64];
#pragma unroll
                    for (int u = 0; u < (LH_RA_CAP + 63) / 64; ++u) { const int k = u * 64 + lane; t0[u] = k < n_lds ? W.tied[k] : 0; }
                    WAVE_SYNC();
                    const int h2 = resc_list_ties(o, W, n_lds, lane);
#pragma unroll
                    for (int u = 0; u < (LH_RA_CAP + 63) / 64; ++u) { const int k = u * 64 + lane; if (k < n_lds && W.tied[k] && !t0[u]) bad_lds = 1; }
                    if (h2 && !harmful_lds) bad_lds = 1;
                }
                bad_lds |= n_lds != n_full;
                for (int k = lane; k < n_full && k < n_lds; k += 64)
                    bad_lds |= B[k].rb != W.rb[k] || B[k].re != W.re[k] || B[k].qb != W.qb[k] || B[k].qe != W.qe[k] || B[k].score != W.score[k] || B[k].rid != W.rid[k] || B[k].seedcov != W.src[k];
                bad_lds = __any(bad_lds);
            }
        }
        if (bad_lds) { if (lane == 0) { verdict[c] = 4; n_out[c] = n_full | n << 16; } continue; }
        load_w();
        // equal end positions in the clean list: the replay takes the incremental form only when every such tie is harmless (resc_list_ties marks the entries)
        if (resc_list_ties(o, W, n, lane) | resc_list_interleaved(o, W, n, lane)) { if (lane == 0) { verdict[c] = 3; n_out[c] = n; } continue; }
        WAVE_SYNC();
        // (b) incremental (an entry that IS b counts as rescued earlier in the same replay: the form's shortcut for a twin)
        for (int k = lane; k < n; k += 64)
            if (W.re[k] == b.re && W.rb[k] == b.rb && W.qb[k] == b.qb && W.qe[k] == b.qe && W.score[k] == b.score && W.rid[k] == b.rid) W.src[k] = -1;
        WAVE_SYNC();
        int app = 0;
        const int n_inc = resc_dedup_incremental(o, W, n, b, lane, &app);
        if (n_inc >= 0 && app) resc_list_sort(W, n_inc, lane);
        WAVE_SYNC();
        int bad = 0;
        if (n_inc >= 0) {
            bad = n_inc != n_full;
            for (int k = lane; k < n_full && k < n_inc; k += 64)
                bad |= B[k].rb != W.rb[k] || B[k].re != W.re[k] || B[k].qb != W.qb[k] || B[k].qe != W.qe[k] || B[k].score != W.score[k] || B[k].rid != W.rid[k];
        }
        bad = __any(bad);
        if (lane == 0) { verdict[c] = n_inc < 0 ? 1 : (bad ? 2 : 0); n_out[c] = n_full | n << 16; }
    }
}

// synth.cpp — lh_synth_genome / lh_synth_reads: the synthetic workloads of bench.py and the tests (no
// genome but PhiX exists offline, so every benchmark genome and read set is generated).  Host-only, threaded, and
// reproducible from the seed whatever the thread count (every genome block / barcode owns its generator).
// The read model is lariat_amd/synth.py's (linked reads: a few long molecules per barcode, FR pairs, an error ramp along
// the read, rare indels) at the speed a 50 M-pair run needs; names are not materialised, name_seed is a hash of the pair index.
#include <math.h>
#include <stdio.h>
#include <zlib.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/lariat_hip.h"

extern "C" int lh_set_error_(int code, const char* msg);

namespace {
typedef uint64_t u64;
typedef int64_t i64;

struct Rng {   // xoshiro256**
    u64 s[4];
    static u64 splitmix(u64& x) {
        u64 z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    Rng(u64 seed, u64 stream) {
        u64 x = seed ^ (stream * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
        for (int i = 0; i < 4; ++i) s[i] = splitmix(x);
    }
    static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    u64 next() {
        u64 r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    i64 below(i64 n) { return n <= 1 ? 0 : (i64)(uni() * (double)n); }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
};

template <class F> void parallel_for(i64 n, int threads, F f) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    if ((i64)threads > n) threads = (int)(n > 0 ? n : 1);
    std::atomic<i64> next(0);
    auto work = [&]() { for (;;) { i64 i = next.fetch_add(1); if (i >= n) break; f(i); } };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

inline int pac_get(const uint8_t* pac, i64 l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }
}  // namespace

extern "C" int lh_synth_genome(uint64_t seed, double gc, int64_t l_pac, uint8_t* pac, int32_t threads) {
    if (!pac || l_pac <= 0 || gc <= 0 || gc >= 1) return lh_set_error_(LH_E_ARG, "lh_synth_genome: bad argument");
    memset(pac, 0, (size_t)(l_pac / 4 + 1));
    const i64 BLK = 1 << 20;   // bases per block, a multiple of 4: blocks own whole bytes
    const uint32_t tA = (uint32_t)((1 - gc) / 2 * 65536.0), tC = tA + (uint32_t)(gc / 2 * 65536.0), tG = tC + (uint32_t)(gc / 2 * 65536.0);
    parallel_for((l_pac + BLK - 1) / BLK, threads, [&](i64 blk) {
        Rng g(seed, (u64)blk);
        i64 b0 = blk * BLK, b1 = b0 + BLK < l_pac ? b0 + BLK : l_pac;
        for (i64 p = b0; p < b1; p += 4) {
            u64 r = g.next();
            uint8_t byte = 0;
            for (int k = 0; k < 4 && p + k < b1; ++k) {
                uint32_t v = (uint32_t)(r >> (16 * k)) & 0xffff;
                uint8_t base = v < tA ? 0 : v < tC ? 1 : v < tG ? 2 : 3;
                byte |= base << ((3 - k) << 1);
            }
            pac[p >> 2] = byte;
        }
    });
    return LH_OK;
}

extern "C" int lh_synth_reads(const uint8_t* pac, int64_t l_pac, int32_t n_contigs, const int64_t* contig_off, const int32_t* contig_len, uint64_t seed,
                              int32_t n_barcodes, int32_t pairs_per_barcode, int32_t len1, int32_t len2, double sub_lo, double sub_hi, double indel_rate,
                              double junk_frac, int32_t mol_min, int32_t mol_max, int32_t threads, uint8_t* seq, int64_t* seq_off, int32_t* bc_pair_off,
                              uint64_t* name_seed, int32_t* truth_rid, int64_t* truth_pos1, int64_t* truth_pos2) {
    if (!pac || !contig_off || !contig_len || !seq || !seq_off || !bc_pair_off || n_contigs <= 0 || n_barcodes <= 0 || pairs_per_barcode <= 0 || len1 < 30 ||
        len2 < 30 || len1 > LH_MAX_READ_LEN - 3 || len2 > LH_MAX_READ_LEN - 3 || mol_min < 1 || mol_max < mol_min)
        return lh_set_error_(LH_E_ARG, "lh_synth_reads: bad argument");
    for (int c = 0; c < n_contigs; ++c)
        if (contig_len[c] < 2000) return lh_set_error_(LH_E_ARG, "lh_synth_reads: contigs must hold at least 2000 bases");
    const i64 n_pairs = (i64)n_barcodes * pairs_per_barcode;
    const int maxlen = (len1 > len2 ? len1 : len2) + 3;
    std::vector<int32_t> lens((size_t)(2 * n_pairs));
    // reads are produced at their worst-case slots of `seq` first, then compacted in order
    const i64 slot = maxlen;
    std::vector<double> ccum(n_contigs);
    { double a = 0; for (int c = 0; c < n_contigs; ++c) { a += contig_len[c]; ccum[c] = a; } }
    parallel_for(n_barcodes, threads, [&](i64 b) {
        Rng g(seed, (u64)b);
        int K = mol_min + (int)g.below(mol_max - mol_min + 1);
        std::vector<int> mc(K);
        std::vector<i64> ms(K), ml(K);
        std::vector<double> mcum(K);
        double acc = 0;
        for (int k = 0; k < K; ++k) {
            double L = exp(log(50000.0) + 0.6 * g.normal());
            L = L < 10000 ? 10000 : L > 200000 ? 200000 : L;
            double u = g.uni() * ccum[n_contigs - 1];
            int c = (int)(std::lower_bound(ccum.begin(), ccum.end(), u) - ccum.begin());
            if (c >= n_contigs) c = n_contigs - 1;
            i64 cl = contig_len[c];
            i64 len = (i64)L < cl - 1000 ? (i64)L : cl - 1000;
            mc[k] = c; ml[k] = len; ms[k] = g.below(cl - len);
            acc += (double)len; mcum[k] = acc;
        }
        uint8_t tmp[2][LH_MAX_READ_LEN + 8];
        for (int q = 0; q < pairs_per_barcode; ++q) {
            i64 p = b * pairs_per_barcode + q;
            int k = (int)(std::lower_bound(mcum.begin(), mcum.end(), g.uni() * acc) - mcum.begin());
            if (k >= K) k = K - 1;
            int lo_ins = 200 > len1 ? 200 : len1; lo_ins = lo_ins > len2 ? lo_ins : len2;
            double insd = 350.0 + 50.0 * g.normal();
            i64 ins = (i64)(insd < lo_ins ? lo_ins : insd > 700 ? 700 : insd);
            i64 span = ml[k] - ins; span = span > 1 ? span : 1;
            i64 frag = ms[k] + g.below(span);
            if (frag + ins > contig_len[mc[k]]) frag = contig_len[mc[k]] - ins;
            bool flip = g.uni() < 0.5;
            int fwd_len = flip ? len2 : len1, rev_len = flip ? len1 : len2;
            i64 g0 = contig_off[mc[k]];
            i64 pos_f = frag, pos_r = frag + ins - rev_len;
            uint8_t* fw = tmp[flip ? 1 : 0];   // the mate on the forward strand: read 1 unless flipped
            uint8_t* rv = tmp[flip ? 0 : 1];
            for (int i = 0; i < fwd_len; ++i) fw[i] = (uint8_t)pac_get(pac, g0 + pos_f + i);
            for (int i = 0; i < rev_len; ++i) rv[i] = (uint8_t)(3 - pac_get(pac, g0 + pos_r + rev_len - 1 - i));
            if (truth_rid) truth_rid[p] = mc[k];
            if (truth_pos1) truth_pos1[p] = flip ? pos_r : pos_f;
            if (truth_pos2) truth_pos2[p] = flip ? pos_f : pos_r;
            for (int m = 0; m < 2; ++m) {
                uint8_t* s = tmp[m];
                int L = m == 0 ? len1 : len2;
                for (int i = 0; i < L; ++i) {   // substitutions: the error rate ramps along the read
                    double pe = sub_lo + (sub_hi - sub_lo) * (double)i / (double)(L - 1);
                    if (g.uni() < pe) s[i] = (uint8_t)((s[i] + 1 + g.below(3)) & 3);
                }
                if (indel_rate > 0 && g.uni() < indel_rate * 150.0 && L > 50) {
                    int at = 20 + (int)g.below(L - 40), ln = 1 + (int)g.below(3);
                    if (g.uni() < 0.5) { memmove(s + at, s + at + ln, (size_t)(L - at - ln)); L -= ln; }
                    else { memmove(s + at + ln, s + at, (size_t)(L - at)); for (int i = 0; i < ln; ++i) s[at + i] = (uint8_t)g.below(4); L += ln; }
                }
                if (junk_frac > 0 && g.uni() < junk_frac) for (int i = 0; i < L; ++i) s[i] = (uint8_t)g.below(4);
                lens[(size_t)(2 * p + m)] = L;
                memcpy(seq + (size_t)(2 * p + m) * slot, s, (size_t)L);
            }
            if (name_seed) { u64 x = seed ^ ((u64)p * 0x9E3779B97F4A7C15ull); name_seed[p] = Rng::splitmix(x); }
        }
    });
    seq_off[0] = 0;
    for (i64 r = 0; r < 2 * n_pairs; ++r) seq_off[r + 1] = seq_off[r] + lens[(size_t)r];
    for (i64 r = 0; r < 2 * n_pairs; ++r)   // in-order compaction: destinations never overtake sources
        if (seq_off[r] != r * slot) memmove(seq + seq_off[r], seq + (size_t)r * slot, (size_t)lens[(size_t)r]);
    for (int b = 0; b <= n_barcodes; ++b) bc_pair_off[b] = b * pairs_per_barcode;
    (void)l_pac;
    return LH_OK;
}

// The reads of lh_synth_reads as the reference's input: 9-line barcode-sorted FASTQ records (README.md:34-48 of the reference), gzip'ed when
// gz_level > 0.  Read 1 gets `trim` random bases in front (the run trims them again: -trim_length); qualities are 'I'; barcode b of the batch is
// the 16-mer of first_barcode + b in base 4 (sorted as the reference needs them) + "-1"; names are mol:<barcode>:<pair> (seven colon-separated
// fields as -simulated wants: mol:bc:chrom:ms:me:pos1:pos2, with the truth positions when given).
extern "C" int lh_synth_write_fastq9(const char* path, const uint8_t* seq, const int64_t* seq_off, const int32_t* bc_pair_off, int32_t n_barcodes, int64_t first_barcode,
                                     int32_t trim, int32_t gz_level, uint64_t seed, const int32_t* truth_rid, const int64_t* truth_pos1, const int64_t* truth_pos2) {
    if (!path || !seq || !seq_off || !bc_pair_off || n_barcodes <= 0 || trim < 0 || trim > 64) return lh_set_error_(LH_E_ARG, "lh_synth_write_fastq9: bad argument");
    gzFile gz = nullptr;
    FILE* fp = nullptr;
    if (gz_level > 0) {
        char mode[8];
        snprintf(mode, sizeof mode, "wb%d", gz_level > 9 ? 9 : gz_level);
        gz = gzopen(path, mode);
        if (gz) gzbuffer(gz, 1 << 20);
    } else fp = fopen(path, "wb");
    if (!gz && !fp) return lh_set_error_(LH_E_IO, "lh_synth_write_fastq9: cannot open the output file");
    std::vector<char> buf;
    buf.reserve(1 << 22);
    static const char B[5] = {'A', 'C', 'G', 'T', 'N'};
    Rng g(seed, 0x9f);
    bool ok = true;
    auto flush = [&]() {
        if (buf.empty()) return;
        if (gz) ok = ok && gzwrite(gz, buf.data(), (unsigned)buf.size()) == (int)buf.size();
        else ok = ok && fwrite(buf.data(), 1, buf.size(), fp) == buf.size();
        buf.clear();
    };
    char tmp[256];
    for (int32_t b = 0; b < n_barcodes && ok; ++b) {
        char bc[20];
        i64 v = first_barcode + b;
        for (int k = 15; k >= 0; --k) { bc[k] = B[v & 3]; v >>= 2; }
        bc[16] = '-'; bc[17] = '1'; bc[18] = 0;
        for (int32_t p = bc_pair_off[b]; p < bc_pair_off[b + 1]; ++p) {
            const i64 o1 = seq_off[2 * p], o2 = seq_off[2 * p + 1], o3 = seq_off[2 * p + 2];
            int n = snprintf(tmp, sizeof tmp, "@mol:%s:%d:0:0:%lld:%lld\n", bc, truth_rid ? truth_rid[p] : 0, (long long)(truth_pos1 ? truth_pos1[p] : p), (long long)(truth_pos2 ? truth_pos2[p] : p));
            buf.insert(buf.end(), tmp, tmp + n);
            for (int k = 0; k < trim; ++k) buf.push_back(B[g.next() & 3]);
            for (i64 i = o1; i < o2; ++i) buf.push_back(B[seq[i] > 4 ? 4 : seq[i]]);
            buf.push_back('\n');
            buf.insert(buf.end(), (size_t)(trim + (o2 - o1)), 'I');
            buf.push_back('\n');
            for (i64 i = o2; i < o3; ++i) buf.push_back(B[seq[i] > 4 ? 4 : seq[i]]);
            buf.push_back('\n');
            buf.insert(buf.end(), (size_t)(o3 - o2), 'I');
            buf.push_back('\n');
            buf.insert(buf.end(), bc, bc + 18);
            static const char tail[] = "\nIIIIIIIIIIIIIIII\nACGTACGT\nIIIIIIII\n";
            buf.insert(buf.end(), tail, tail + sizeof tail - 1);
            if (buf.size() > (3u << 20)) flush();
        }
    }
    flush();
    if (gz) ok = (gzclose(gz) == Z_OK) && ok;
    if (fp) ok = (fclose(fp) == 0) && ok;
    return ok ? LH_OK : lh_set_error_(LH_E_IO, "lh_synth_write_fastq9: write failed");
}

// index_build.cpp — lh_index_build: FM-index construction producing BWA-byte-compatible <prefix>.bwt/.sa/.pac/.ann/.amb
// (SURVEY.md §8f N3).  The formats are the ones the reference's fixture go/src/test/inputs/phix/PhiX.fa.* uses and
// bwa_idx_load (go/src/gobwa/gobwa.go:130) consumes: text = forward || reverse complement, sentinel smallest,
// BWT with '$' removed at `primary`, occurrence counts interleaved every 128 symbols, SA sampled every 32 rows.
//
// Host-side, multi-threaded: suffixes are bucketed by their first 12 symbols (counting sort) and each bucket is
// finished by a comparison sort over the 2-bit packed text, 32 symbols per compare word.  Ambiguous bases are replaced and
// recorded as bns_fasta2bntseq does (lh_reference_pack below).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lariat_hip.h"

namespace {
typedef uint64_t u64;
typedef int64_t i64;

struct Packed {   // 2-bit text, 32 symbols per u64, MSB first, padded with zeros
    std::vector<u64> w;
    i64 n;
    inline u64 word(i64 pos) const {   // 32 symbols starting at pos
        i64 wi = pos >> 5;
        int sh = (int)(pos & 31) << 1;
        u64 a = w[wi];
        if (!sh) return a;
        return (a << sh) | (w[wi + 1] >> (64 - sh));
    }
};

struct SufLess {
    const Packed* P;
    i64 n;
    bool operator()(i64 a, i64 b) const {
        if (a == b) return false;
        i64 la = n - a, lb = n - b, l = la < lb ? la : lb;
        i64 off = 0;
        while (off < l) {
            u64 wa = P->word(a + off), wb = P->word(b + off);
            if (wa != wb) {
                int lead = __builtin_clzll(wa ^ wb) >> 1;   // first differing symbol
                if (off + lead >= l) break;                  // difference lies in the padding
                return wa < wb;
            }
            off += 32;
        }
        return la < lb;   // one is a prefix of the other: the shorter (sentinel first) is smaller
    }
};

bool write_file(const std::string& p, const void* d, size_t n) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) return false;
    size_t w = n ? fwrite(d, 1, n, f) : 0;
    fclose(f);
    return w == n;
}
}  // namespace

// bntseq.c: nst_nt4_table (A C G T in either case -> 0..3, everything else 4)
static inline int nt4_of(uint8_t v) {
    if (v <= 4) return v;
    switch (v) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// bns_fasta2bntseq / add1 (bntseq.c) restated: the 2-bit forward reference and the table of "holes".  An ambiguous base is
// replaced by lrand48() & 3 — the C library's 48-bit linear congruential generator, seeded with srand48(11) once per index
// (the 11 is the seed field of the .ann header) and stepped once per ambiguous base in file order — and a run of the SAME
// ambiguity letter is one hole (offset, length, letter) of the .amb file; n_ambs counts a contig's holes for the .ann file.
// Bases come as nt4 codes (0..3, 4 = 'N') or as raw FASTA letters (any value > 4).
extern "C" int lh_reference_pack(int32_t n_contigs, const uint8_t* const* seqs, const int64_t* lens, uint8_t* pac, int32_t* n_ambs, int32_t max_holes,
                                 int64_t* hole_off, int32_t* hole_len, char* hole_char, int32_t* n_holes) {
    if (n_contigs <= 0 || !seqs || !lens || !pac || !n_holes) return LH_E_ARG;
    i64 l_pac = 0;
    for (int i = 0; i < n_contigs; ++i) l_pac += lens[i];
    memset(pac, 0, (size_t)(l_pac / 4 + 1));
    u64 x48 = ((u64)11 << 16) | 0x330E;   // srand48(11)
    i64 p = 0;
    int nh = 0;
    for (int c = 0; c < n_contigs; ++c) {
        int lasts = 0;   // add1(): `lasts` restarts at 0 for every sequence
        if (n_ambs) n_ambs[c] = 0;
        for (i64 i = 0; i < lens[c]; ++i, ++p) {
            const uint8_t raw = seqs[c][i];
            const int letter = raw <= 4 ? "ACGTN"[raw] : raw;
            int b = nt4_of(raw);
            if (b >= 4) {
                if (lasts == letter && nh > 0) { if (hole_len && nh <= max_holes) ++hole_len[nh - 1]; }
                else {
                    if (nh < max_holes) { if (hole_off) hole_off[nh] = p; if (hole_len) hole_len[nh] = 1; if (hole_char) hole_char[nh] = (char)letter; }
                    ++nh;
                    if (n_ambs) ++n_ambs[c];
                }
                x48 = (x48 * 0x5DEECE66Dull + 0xB) & 0xFFFFFFFFFFFFull;   // lrand48(): the high 31 bits of the new state
                b = (int)((x48 >> 17) & 3);
            }
            lasts = letter;
            pac[p >> 2] |= (uint8_t)(b << ((~p & 3) << 1));
        }
    }
    *n_holes = nh;
    return nh > max_holes && (hole_off || hole_len || hole_char) ? LH_E_CAPACITY : LH_OK;
}

extern "C" int lh_index_build(const char* prefix, int32_t n_contigs, const char* const* names, const uint8_t* const* nt4, const int64_t* lens, int32_t threads) {
    if (!prefix || n_contigs <= 0 || !names || !nt4 || !lens) return LH_E_ARG;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    i64 l_pac = 0;
    for (int i = 0; i < n_contigs; ++i) l_pac += lens[i];
    i64 n = 2 * l_pac;
    // text
    std::vector<uint8_t> T((size_t)n);
    std::vector<int32_t> n_ambs((size_t)n_contigs, 0), hole_len;
    std::vector<i64> hole_off;
    std::vector<char> hole_char;
    {
        std::vector<uint8_t> pac0((size_t)(l_pac / 4 + 1), 0);
        int32_t nh = 0;
        lh_reference_pack(n_contigs, nt4, lens, pac0.data(), n_ambs.data(), 0, nullptr, nullptr, nullptr, &nh);   // count the holes
        hole_off.resize((size_t)nh + 1); hole_len.resize((size_t)nh + 1); hole_char.resize((size_t)nh + 1);
        int rc = lh_reference_pack(n_contigs, nt4, lens, pac0.data(), n_ambs.data(), nh, hole_off.data(), hole_len.data(), hole_char.data(), &nh);
        if (rc) return rc;
        hole_off.resize((size_t)nh); hole_len.resize((size_t)nh); hole_char.resize((size_t)nh);
        for (i64 i = 0; i < l_pac; ++i) T[i] = pac0[i >> 2] >> ((~i & 3) << 1) & 3;
        for (i64 i = 0; i < l_pac; ++i) T[l_pac + i] = 3 - T[l_pac - 1 - i];
    }
    Packed P;
    P.n = n;
    P.w.assign((size_t)(n >> 5) + 3, 0);
    for (i64 i = 0; i < n; ++i) P.w[i >> 5] |= (u64)T[i] << ((31 - (i & 31)) << 1);
    // bucket by the first K symbols (zero padded)
    const int K = n > (1 << 24) ? 12 : 8;
    const size_t NB = (size_t)1 << (2 * K);
    std::vector<i64> bstart(NB + 1, 0);
    auto key_at = [&](i64 i) -> size_t { return (size_t)(P.word(i) >> (64 - 2 * K)); };
    for (i64 i = 0; i < n; ++i) bstart[key_at(i) + 1]++;
    for (size_t b = 0; b < NB; ++b) bstart[b + 1] += bstart[b];
    std::vector<i64> SA((size_t)n);
    {
        std::vector<i64> fill(bstart.begin(), bstart.end() - 1);
        for (i64 i = 0; i < n; ++i) SA[fill[key_at(i)]++] = i;
    }
    // finish every bucket
    {
        std::atomic<size_t> next(0);
        SufLess less{&P, n};
        auto work = [&]() {
            const size_t CH = 4096;
            for (;;) {
                size_t b0 = next.fetch_add(CH);
                if (b0 >= NB) break;
                size_t b1 = b0 + CH < NB ? b0 + CH : NB;
                for (size_t b = b0; b < b1; ++b)
                    if (bstart[b + 1] - bstart[b] > 1) std::sort(SA.begin() + bstart[b], SA.begin() + bstart[b + 1], less);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    // BWT ('$' row removed), primary, L2.  Full order = [sentinel suffix n] ++ SA.
    u64 primary = 0, L2[5] = {0, 0, 0, 0, 0};
    {
        u64 cnt[4] = {0, 0, 0, 0};
        for (i64 i = 0; i < n; ++i) ++cnt[T[i]];
        for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + cnt[c];
    }
    u64 n_occ = (u64)(n + 127) / 128 + 1;
    u64 words = (u64)((n + 15) >> 4) + n_occ * 8;
    std::vector<uint32_t> bwt((size_t)words, 0);
    {
        u64 c4[4] = {0, 0, 0, 0};
        size_t w = 0;
        uint32_t word = 0;
        i64 k = 0;   // index into the '$'-less BWT
        auto put = [&](uint8_t sym) {
            if (k % 128 == 0) { memcpy(&bwt[w], c4, 32); w += 8; }
            word |= (uint32_t)sym << ((15 - (k & 15)) << 1);
            if ((k & 15) == 15 || k == n - 1) { bwt[w++] = word; word = 0; }
            ++c4[sym];
            ++k;
        };
        put(T[n - 1]);   // row 0: the sentinel suffix, preceded by the last symbol
        for (i64 i = 0; i < n; ++i) {
            i64 s = SA[i];
            if (s == 0) { primary = (u64)i + 1; continue; }
            put(T[s - 1]);
        }
        memcpy(&bwt[w], c4, 32);
        w += 8;
        words = w;
    }
    // sampled SA
    const u64 intv = 32;
    u64 n_sa = ((u64)n + intv) / intv;
    std::vector<u64> sa((size_t)n_sa);
    sa[0] = (u64)-1;
    for (u64 j = 1; j < n_sa; ++j) sa[j] = (u64)SA[j * intv - 1];   // row j*32 of the full order = SA[j*32-1]
    // ---- files ----
    std::string p(prefix);
    {
        std::vector<uint8_t> o(40 + (size_t)words * 4);
        memcpy(o.data(), &primary, 8);
        memcpy(o.data() + 8, &L2[1], 32);
        memcpy(o.data() + 40, bwt.data(), (size_t)words * 4);
        if (!write_file(p + ".bwt", o.data(), o.size())) return LH_E_IO;
    }
    {
        std::vector<uint8_t> o(56 + (size_t)(n_sa - 1) * 8);
        memcpy(o.data(), &primary, 8);
        memcpy(o.data() + 8, &L2[1], 32);
        memcpy(o.data() + 40, &intv, 8);
        u64 sl = (u64)n;
        memcpy(o.data() + 48, &sl, 8);
        memcpy(o.data() + 56, sa.data() + 1, (size_t)(n_sa - 1) * 8);
        if (!write_file(p + ".sa", o.data(), o.size())) return LH_E_IO;
    }
    {
        std::vector<uint8_t> pac((size_t)(l_pac / 4 + 1), 0);
        for (i64 i = 0; i < l_pac; ++i) pac[i >> 2] |= T[i] << ((~i & 3) << 1);
        size_t nb = (size_t)(l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1);
        std::vector<uint8_t> o(pac.begin(), pac.begin() + nb);
        if (l_pac % 4 == 0) o.push_back(0);
        o.push_back((uint8_t)(l_pac % 4));
        if (!write_file(p + ".pac", o.data(), o.size())) return LH_E_IO;
    }
    {
        std::string s;
        char buf[4200];
        snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)l_pac, n_contigs, 11u);
        s += buf;
        i64 off = 0;
        for (int i = 0; i < n_contigs; ++i) {
            snprintf(buf, sizeof buf, "0 %s (null)\n%lld %d %d\n", names[i], (long long)off, (int)lens[i], (int)n_ambs[(size_t)i]);
            s += buf;
            off += lens[i];
        }
        if (!write_file(p + ".ann", s.data(), s.size())) return LH_E_IO;
        snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)l_pac, n_contigs, (unsigned)hole_off.size());
        std::string amb = buf;
        for (size_t h = 0; h < hole_off.size(); ++h) {   // bns_dump: "%lld %d %c"
            snprintf(buf, sizeof buf, "%lld %d %c\n", (long long)hole_off[h], (int)hole_len[h], hole_char[h]);
            amb += buf;
        }
        if (!write_file(p + ".amb", amb.data(), amb.size())) return LH_E_IO;
    }
    return LH_OK;
}

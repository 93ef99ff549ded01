// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// BWA index file formats (restated from lh3/bwa 0.7.17 bwt.c:bwt_restore_bwt/bwt_restore_sa,
// bntseq.c:bns_restore_core, bwtindex.c:bwt_bwtupdate_core/bwt_cal_sa; byte-checked against the
// reference fixture go/src/test/inputs/phix/PhiX.fa.{bwt,sa,pac,ann,amb} — SURVEY.md §4).
#include "bwa_oracle.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace orc {

void Counters::add(const Counters& o) {
    n_ext += o.n_ext; n_lf += o.n_lf; n_sa += o.n_sa; win_bases += o.win_bases; n_chain_ext += o.n_chain_ext;
    ext_cells += o.ext_cells; glob_cells += o.glob_cells; n_rescue += o.n_rescue; rescue_cells += o.rescue_cells; n_glob += o.n_glob;
    n_reads += o.n_reads; read_bases += o.read_bases; n_cand += o.n_cand;
}

static bool read_file(const std::string& p, std::vector<uint8_t>& out) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n);
    size_t r = n ? fread(out.data(), 1, n, f) : 0;
    fclose(f);
    return (long)r == n;
}

bool index_load(const std::string& prefix, Index& idx, std::string* err) {
    std::vector<uint8_t> d;
    // .bwt : primary, L2[1..4], then the occ-interleaved u32 array
    if (!read_file(prefix + ".bwt", d) || d.size() < 40) { if (err) *err = "cannot read " + prefix + ".bwt"; return false; }
    memcpy(&idx.primary, d.data(), 8);
    idx.L2[0] = 0;
    memcpy(&idx.L2[1], d.data() + 8, 32);
    idx.seq_len = idx.L2[4];
    idx.bwt_size = (d.size() - 40) >> 2;
    idx.bwt.resize(idx.bwt_size);
    memcpy(idx.bwt.data(), d.data() + 40, idx.bwt_size * 4);
    // .sa : primary, 4 skipped u64, sa_intv, seq_len, sa[1..]
    if (!read_file(prefix + ".sa", d) || d.size() < 56) { if (err) *err = "cannot read " + prefix + ".sa"; return false; }
    bwtint_t primary, sl, intv;
    memcpy(&primary, d.data(), 8);
    memcpy(&intv, d.data() + 40, 8);
    memcpy(&sl, d.data() + 48, 8);
    if (primary != idx.primary || sl != idx.seq_len) { if (err) *err = "SA-BWT inconsistency"; return false; }
    idx.sa_intv = (int)intv;
    idx.n_sa = (idx.seq_len + intv) / intv;
    if (d.size() != 56 + (idx.n_sa - 1) * 8) { if (err) *err = "bad .sa size"; return false; }
    idx.sa.resize(idx.n_sa);
    idx.sa[0] = (bwtint_t)-1;
    memcpy(idx.sa.data() + 1, d.data() + 56, (idx.n_sa - 1) * 8);
    // .ann
    {
        std::ifstream f(prefix + ".ann");
        if (!f) { if (err) *err = "cannot read " + prefix + ".ann"; return false; }
        std::string line;
        std::getline(f, line);
        long long lp; int ns; unsigned seed;
        if (sscanf(line.c_str(), "%lld %d %u", &lp, &ns, &seed) != 3) { if (err) *err = "bad .ann header"; return false; }
        idx.l_pac = lp; idx.seed = seed;
        idx.contigs.resize(ns);
        for (int i = 0; i < ns; ++i) {
            Contig& c = idx.contigs[i];
            std::getline(f, line);
            std::istringstream ss(line);
            ss >> c.gi >> c.name;
            std::string rest;
            std::getline(ss, rest);
            if (!rest.empty() && rest[0] == ' ') rest = rest.substr(1);
            c.anno = rest == "(null)" ? "" : rest;
            std::getline(f, line);
            long long off; int len, na;
            if (sscanf(line.c_str(), "%lld %d %d", &off, &len, &na) != 3) { if (err) *err = "bad .ann record"; return false; }
            c.offset = off; c.len = len; c.n_ambs = na; c.is_alt = 0;
        }
    }
    {   // .amb (bns_restore_core): "l_pac n_seqs n_holes" then "offset len letter" per hole
        std::ifstream f(prefix + ".amb");
        std::string line;
        if (f && std::getline(f, line))
            while (std::getline(f, line)) {
                long long o; int l; char ch;
                if (sscanf(line.c_str(), "%lld %d %c", &o, &l, &ch) == 3) idx.holes.push_back(Index::Hole{o, l, ch});
            }
    }
    {   // .alt (bntseq.c bns_restore, reached through bwa_idx_load(path, BWA_IDX_ALL): go/src/gobwa/gobwa.go:130): per line the first token up
        // to a tab / newline / CR is a contig name ('@' lines skipped); a name found in .ann marks that contig (the last one of that name)
        std::vector<uint8_t> a;
        if (read_file(prefix + ".alt", a)) {
            std::string tok;
            size_t i = 0;
            while (i < a.size()) {
                char ch = (char)a[i++];
                if (ch == '\t' || ch == '\n' || ch == '\r') {
                    if (tok.empty() || tok[0] != '@') {
                        int hit = -1;
                        for (size_t k = 0; k < idx.contigs.size(); ++k) if (idx.contigs[k].name == tok) hit = (int)k;
                        if (hit >= 0) idx.contigs[hit].is_alt = 1;
                    }
                    char c = ch;
                    while (c != '\n' && i < a.size()) c = (char)a[i++];
                    tok.clear();
                } else tok.push_back(ch);
            }
        }
    }
    // .pac (forward strand only)
    if (!read_file(prefix + ".pac", d)) { if (err) *err = "cannot read " + prefix + ".pac"; return false; }
    idx.pac.assign(d.begin(), d.begin() + (idx.l_pac / 4 + 1));
    if ((bwtint_t)idx.l_pac * 2 != idx.seq_len) { if (err) *err = "l_pac*2 != seq_len"; return false; }
    return true;
}

void index_build_naive(const std::vector<std::string>& names, const std::vector<std::vector<uint8_t>>& seqs, Index& idx) {
    idx = Index();
    int64_t off = 0;
    std::vector<uint8_t> T;
    // bntseq.c bns_fasta2bntseq / add1: an ambiguous base becomes lrand48() & 3 (srand48(bns->seed), seed 11, one draw per
    // ambiguous base in file order); a run of the same ambiguity letter is one hole.  Bases: nt4 codes (4 = 'N') or raw letters.
    uint64_t x48 = ((uint64_t)idx.seed << 16) | 0x330E;
    auto lrand48_ = [&]() { x48 = (x48 * 0x5DEECE66Dull + 0xBull) & ((1ull << 48) - 1); return (long)(x48 >> 17); };
    for (size_t i = 0; i < seqs.size(); ++i) {
        Contig c;
        c.offset = off; c.len = (int32_t)seqs[i].size(); c.n_ambs = 0; c.gi = 0; c.is_alt = 0; c.name = names[i];
        int lasts = 0;
        for (size_t j = 0; j < seqs[i].size(); ++j) {
            uint8_t raw = seqs[i][j];
            int letter = raw <= 4 ? "ACGTN"[raw] : raw;
            int b = raw <= 4 ? raw : (letter == 'A' || letter == 'a') ? 0 : (letter == 'C' || letter == 'c') ? 1 : (letter == 'G' || letter == 'g') ? 2
                                   : (letter == 'T' || letter == 't') ? 3 : 4;
            if (b >= 4) {
                if (lasts == letter) ++idx.holes.back().len;
                else { idx.holes.push_back(Index::Hole{c.offset + (int64_t)j, 1, (char)letter}); ++c.n_ambs; }
                b = (int)(lrand48_() & 3);
            }
            lasts = letter;
            T.push_back((uint8_t)b);
        }
        idx.contigs.push_back(c);
        off += c.len;
    }
    idx.l_pac = off;
    idx.pac.assign(idx.l_pac / 4 + 1, 0);
    for (int64_t i = 0; i < idx.l_pac; ++i) idx.pac[i >> 2] |= (T[i] & 3) << ((~i & 3) << 1);
    // forward || reverse complement
    int64_t n = idx.l_pac * 2;
    T.resize(n);
    for (int64_t i = 0; i < idx.l_pac; ++i) T[idx.l_pac + i] = 3 - T[idx.l_pac - 1 - i];
    std::vector<int64_t> SA(n + 1);
    for (int64_t i = 0; i <= n; ++i) SA[i] = i;
    const uint8_t* t = T.data();
    std::sort(SA.begin(), SA.end(), [&](int64_t a, int64_t b) {
        if (a == b) return false;
        int64_t la = n - a, lb = n - b, l = la < lb ? la : lb;
        int c = memcmp(t + a, t + b, l);
        if (c) return c < 0;
        return la < lb;   // the shorter suffix (sentinel reached first) sorts first
    });
    idx.seq_len = n;
    std::vector<uint8_t> B(n);   // BWT with the '$' row removed
    int64_t k = 0;
    bwtint_t cnt[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i <= n; ++i) {
        if (SA[i] == 0) { idx.primary = i; continue; }
        B[k++] = t[SA[i] - 1];
    }
    for (int64_t i = 0; i < n; ++i) ++cnt[t[i]];
    idx.L2[0] = 0;
    for (int c = 0; c < 4; ++c) idx.L2[c + 1] = idx.L2[c] + cnt[c];
    // occ-interleaved layout: every 128 symbols [4 x u64 running counts | 8 x u32 of 16 symbols], trailing counts block
    bwtint_t n_occ = (n + 127) / 128 + 1;
    idx.bwt_size = ((n + 15) >> 4) + n_occ * 8;
    idx.bwt.assign(idx.bwt_size, 0);
    bwtint_t c4[4] = {0, 0, 0, 0};
    size_t w = 0;
    uint32_t word = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (i % 128 == 0) { memcpy(&idx.bwt[w], c4, 32); w += 8; }
        word |= (uint32_t)B[i] << ((15 - (i & 15)) << 1);
        if ((i & 15) == 15 || i == n - 1) { idx.bwt[w++] = word; word = 0; }
        ++c4[B[i]];
    }
    memcpy(&idx.bwt[w], c4, 32);
    w += 8;
    idx.bwt_size = w;
    idx.bwt.resize(w);
    idx.sa_intv = 32;
    idx.n_sa = (n + 32) / 32;
    idx.sa.resize(idx.n_sa);
    for (bwtint_t j = 0; j < idx.n_sa; ++j) idx.sa[j] = (bwtint_t)SA[j * 32];
    idx.sa[0] = (bwtint_t)-1;
}

std::vector<uint8_t> image_bwt(const Index& idx) {
    std::vector<uint8_t> o(40 + idx.bwt_size * 4);
    memcpy(o.data(), &idx.primary, 8);
    memcpy(o.data() + 8, &idx.L2[1], 32);
    memcpy(o.data() + 40, idx.bwt.data(), idx.bwt_size * 4);
    return o;
}
std::vector<uint8_t> image_sa(const Index& idx) {
    std::vector<uint8_t> o(56 + (idx.n_sa - 1) * 8);
    memcpy(o.data(), &idx.primary, 8);
    memcpy(o.data() + 8, &idx.L2[1], 32);
    bwtint_t v = idx.sa_intv;
    memcpy(o.data() + 40, &v, 8);
    memcpy(o.data() + 48, &idx.seq_len, 8);
    memcpy(o.data() + 56, idx.sa.data() + 1, (idx.n_sa - 1) * 8);
    return o;
}
std::vector<uint8_t> image_pac(const Index& idx) {
    size_t nb = (idx.l_pac >> 2) + ((idx.l_pac & 3) == 0 ? 0 : 1);
    std::vector<uint8_t> o(idx.pac.begin(), idx.pac.begin() + nb);
    if (idx.l_pac % 4 == 0) o.push_back(0);
    o.push_back((uint8_t)(idx.l_pac % 4));
    return o;
}
std::string image_ann(const Index& idx) {
    char buf[1024];
    std::string s;
    snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)idx.l_pac, (int)idx.contigs.size(), idx.seed);
    s += buf;
    for (const Contig& c : idx.contigs) {
        snprintf(buf, sizeof buf, "%d %s", (int)c.gi, c.name.c_str());
        s += buf;
        s += c.anno.empty() ? " (null)\n" : " " + c.anno + "\n";
        snprintf(buf, sizeof buf, "%lld %d %d\n", (long long)c.offset, c.len, c.n_ambs);
        s += buf;
    }
    return s;
}
std::string image_amb(const Index& idx) {
    char buf[256];
    snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)idx.l_pac, (int)idx.contigs.size(), (unsigned)idx.holes.size());
    std::string s = buf;
    for (const Index::Hole& h : idx.holes) { snprintf(buf, sizeof buf, "%lld %d %c\n", (long long)h.offset, h.len, h.amb); s += buf; }
    return s;
}

// ---- bntseq.c ------------------------------------------------------------------
int64_t bns_depos(const Index& b, int64_t pos, int* is_rev) {
    return (*is_rev = (pos >= b.l_pac)) ? (b.l_pac << 1) - 1 - pos : pos;
}

int bns_pos2rid(const Index& b, int64_t pos_f) {
    int left, mid, right;
    if (pos_f >= b.l_pac) return -1;
    left = 0; mid = 0; right = (int)b.contigs.size();
    while (left < right) {   // binary search
        mid = (left + right) >> 1;
        if (pos_f >= b.contigs[mid].offset) {
            if (mid == (int)b.contigs.size() - 1) break;
            if (pos_f < b.contigs[mid + 1].offset) break;   // bracketed
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

int bns_intv2rid(const Index& b, int64_t rb, int64_t re) {
    int is_rev, rid_b, rid_e;
    if (rb < b.l_pac && re > b.l_pac) return -2;
    rid_b = bns_pos2rid(b, bns_depos(b, rb, &is_rev));
    rid_e = rb < re ? bns_pos2rid(b, bns_depos(b, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

static inline int get_pac(const uint8_t* pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

std::vector<uint8_t> bns_get_seq(const Index& b, int64_t beg, int64_t end) {
    std::vector<uint8_t> seq;
    int64_t l_pac = b.l_pac;
    if (end < beg) std::swap(beg, end);
    if (end > l_pac << 1) end = l_pac << 1;
    if (beg < 0) beg = 0;
    if (beg >= l_pac || end <= l_pac) {
        seq.resize(end - beg);
        int64_t l = 0;
        if (beg >= l_pac) {   // reverse strand
            int64_t beg_f = (l_pac << 1) - 1 - end;
            int64_t end_f = (l_pac << 1) - 1 - beg;
            for (int64_t k = end_f; k > beg_f; --k) seq[l++] = 3 - get_pac(b.pac.data(), k);
        } else {
            for (int64_t k = beg; k < end; ++k) seq[l++] = get_pac(b.pac.data(), k);
        }
    }   // else: bridging the forward-reverse boundary -> nothing
    return seq;
}

std::vector<uint8_t> bns_fetch_seq(const Index& b, int64_t* beg, int64_t mid, int64_t* end, int* rid) {
    int64_t far_beg, far_end;
    int is_rev;
    if (*end < *beg) std::swap(*beg, *end);
    *rid = bns_pos2rid(b, bns_depos(b, mid, &is_rev));
    far_beg = b.contigs[*rid].offset;
    far_end = far_beg + b.contigs[*rid].len;
    if (is_rev) {   // flip to the reverse strand
        int64_t tmp = far_beg;
        far_beg = (b.l_pac << 1) - far_end;
        far_end = (b.l_pac << 1) - tmp;
    }
    *beg = *beg > far_beg ? *beg : far_beg;
    *end = *end < far_end ? *end : far_end;
    return bns_get_seq(b, *beg, *end);
}

}  // namespace orc

// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, link or call anything in this directory.
//
// CPU restatement of the candidate-generation arithmetic that 10XGenomics/lariat
// reaches through cgo (go/src/gobwa/gobwa.go:59,130,151,164,181,244,253,291,315,404;
// prototypes go/src/gobwa/bwa_bridge.h:35-39).  The arithmetic itself lives in the
// un-vendored submodule go/src/gobwa/bwa -> github.com/lh3/bwa ("Apache2 branch",
// README.md:51; commit unpinned, API surface consistent with 0.7.17), which is
// ABSENT from /root/reference.  This file therefore restates BWA-MEM's published
// algorithm (bwt.c / bwamem.c / bwa.c / bntseq.c / ksw.c / ksort.h of 0.7.17).
//
// PARITY STATUS: pinned only by the reference's own fixtures:
//   * PhiX index bytes go/src/test/inputs/phix/PhiX.fa.{bwt,sa,pac,ann,amb}
//   * go/src/test/gobwa_test.go:18-25 (offset 210 / contig PhiX / exactly one hit)
//   * go/src/test/lariat_test.go:12-24 (empty read must not crash)
// Everything else (CIGARs, scores, MAPQ, RFA picks) is "parity unpinned": no
// reference test asserts it and the reference cannot be built here (no Go, no BWA).
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

namespace orc {

typedef uint64_t bwtint_t;

// ---- instrumentation: algorithmic-byte accounting ----------
extern std::atomic<uint64_t> g_rescue_probe[64];   // tools/rescue_probe.py (bwa_mem.cpp: rescue_probe)
extern int g_rescue_probe_on;
struct Counters {
    uint64_t n_ext = 0;       // bwt_extend calls (each = one bwt_2occ4 = 2 occ-block reads of 64 B)
    uint64_t n_lf = 0;        // LF-mapping steps inside bwt_sa
    uint64_t n_sa = 0;        // SA samples read
    uint64_t win_bases = 0;   // reference bases fetched for chain extension windows
    uint64_t n_chain_ext = 0; // chains extended
    uint64_t ext_cells = 0;   // ksw_extend2 DP cells
    uint64_t glob_cells = 0;  // ksw_global2 DP cells
    uint64_t n_rescue = 0;    // mem_matesw SW attempts
    uint64_t rescue_cells = 0;
    uint64_t n_glob = 0;      // candidates mem_reg2aln ran ksw_global2 for (every one but bwa_gen_cigar2's "no gap; no need to do DP")
    uint64_t n_reads = 0;
    uint64_t read_bases = 0;
    uint64_t n_cand = 0;
    void add(const Counters& o);
};

// ---- index ------------------------------------------------------------------
struct Contig { int64_t offset; int32_t len; int32_t n_ambs; uint32_t gi; int32_t is_alt; std::string name, anno; };

struct Index {
    // bwt_t
    bwtint_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0, bwt_size = 0;
    std::vector<uint32_t> bwt;   // occ-interleaved layout (bwt_bwtupdate_core), as stored in .bwt
    int sa_intv = 32;
    bwtint_t n_sa = 0;
    std::vector<bwtint_t> sa;    // sa[0] = -1
    // bntseq_t
    int64_t l_pac = 0;
    uint32_t seed = 11;
    std::vector<Contig> contigs;
    std::vector<uint8_t> pac;    // 2-bit, MSB first
    struct Hole { int64_t offset; int32_t len; char amb; };
    std::vector<Hole> holes;     // bntamb1_t: runs of one ambiguity letter (.amb)
};

// read <prefix>.bwt/.sa/.pac/.ann/.amb  (bwa_idx_load, gobwa.go:130)
bool index_load(const std::string& prefix, Index& idx, std::string* err);
// build from nt4 contig sequences by naive suffix sorting (small genomes only: KAT for the fixture bytes)
void index_build_naive(const std::vector<std::string>& names, const std::vector<std::vector<uint8_t>>& seqs, Index& idx);
// serialise to the byte-exact BWA file images
std::vector<uint8_t> image_bwt(const Index& idx);
std::vector<uint8_t> image_sa(const Index& idx);
std::vector<uint8_t> image_pac(const Index& idx);
std::string image_ann(const Index& idx);
std::string image_amb(const Index& idx);

// ---- options (mem_opt_init; gobwa.go:149-153 never modifies them) -----------
struct MemOpt {
    int a = 1, b = 4, o_del = 6, e_del = 1, o_ins = 6, e_ins = 1;
    int pen_unpaired = 17, pen_clip5 = 5, pen_clip3 = 5;
    int w = 100, zdrop = 100, T = 30;
    int min_seed_len = 19, min_chain_weight = 0, max_chain_extend = 1 << 30;
    float split_factor = 1.5f;
    int split_width = 10, max_occ = 500, max_chain_gap = 10000, max_ins = 10000;
    float mask_level = 0.50f, drop_ratio = 0.50f, XA_drop_ratio = 0.80f, mask_level_redun = 0.95f;
    float mapQ_coef_len = 50;
    int max_mem_intv = 20, max_matesw = 50;
    int8_t mat[25];
    MemOpt();
};

struct Intv { bwtint_t x[3]; uint64_t info; };   // bwtintv_t
struct Seed { int64_t rbeg; int32_t qbeg, len, score; };   // mem_seed_t
struct Chain {   // mem_chain_t
    int rid = 0; uint32_t w = 0; int kept = 0, first = -1, is_alt = 0; float frac_rep = 0; int64_t pos = 0;
    std::vector<Seed> seeds;
};
struct AlnReg {   // mem_alnreg_t
    int64_t rb = 0, re = 0; int qb = 0, qe = 0; int rid = 0; int score = 0, truesc = 0, sub = 0, alt_sc = 0, csub = 0, sub_n = 0;
    int w = 0, seedcov = 0, secondary = 0, secondary_all = 0, seedlen0 = 0; int n_comp = 0, is_alt = 0; float frac_rep = 0; uint64_t hash = 0;
};
struct Aln {   // mem_aln_t
    int64_t pos = -1; int rid = -1; int flag = 0; int is_rev = 0, is_alt = 0, mapq = 0, NM = 0;
    std::vector<uint32_t> cigar;   // len<<4|op, MIDSH=01234
    int score = 0, sub = 0, alt_sc = 0;
};
struct PeStat { int low = 0, high = 0, failed = 1; double avg = 0, std = 0; };   // mem_pestat_t

extern const uint8_t nst_nt4_table[256];

// bwt.c
void bwt_occ4(const Index& b, bwtint_t k, bwtint_t cnt[4]);
void bwt_2occ4(const Index& b, bwtint_t k, bwtint_t l, bwtint_t cntk[4], bwtint_t cntl[4]);
void bwt_extend(const Index& b, const Intv& ik, Intv ok[4], int is_back, Counters* c);
bwtint_t bwt_sa(const Index& b, bwtint_t k, Counters* c);
int bwt_smem1(const Index& b, int len, const uint8_t* q, int x, int min_intv, std::vector<Intv>& mem, Counters* c);
int bwt_seed_strategy1(const Index& b, int len, const uint8_t* q, int x, int min_len, int max_intv, Intv* mem, Counters* c);
// bntseq.c
int64_t bns_depos(const Index& b, int64_t pos, int* is_rev);
int bns_pos2rid(const Index& b, int64_t pos_f);
int bns_intv2rid(const Index& b, int64_t rb, int64_t re);
std::vector<uint8_t> bns_get_seq(const Index& b, int64_t beg, int64_t end);
std::vector<uint8_t> bns_fetch_seq(const Index& b, int64_t* beg, int64_t mid, int64_t* end, int* rid);
// bwamem.c
void mem_collect_intv(const MemOpt& o, const Index& b, int len, const uint8_t* seq, std::vector<Intv>& mem, Counters* c);
std::vector<Chain> mem_chain(const MemOpt& o, const Index& b, int len, const uint8_t* seq, Counters* c, std::vector<Intv>* intv_out = nullptr, std::vector<Seed>* seeds_out = nullptr);
int mem_chain_flt(const MemOpt& o, std::vector<Chain>& a);
void mem_chain2aln(const MemOpt& o, const Index& b, int l_query, const uint8_t* query, const Chain& c, std::vector<AlnReg>& av, Counters* cn);
int mem_sort_dedup_patch(const MemOpt& o, const Index* b, const uint8_t* query, std::vector<AlnReg>& a, Counters* cn);
std::vector<AlnReg> mem_align1_core(const MemOpt& o, const Index& b, int l_seq, const uint8_t* seq_nt4, Counters* c);
int mem_matesw(const MemOpt& o, const Index& b, const PeStat pes[4], const AlnReg& a, int l_ms, const uint8_t* ms, std::vector<AlnReg>& ma, Counters* c);
Aln mem_reg2aln(const MemOpt& o, const Index& b, int l_query, const uint8_t* query_nt4, const AlnReg& ar, Counters* c);
// ksw.c
int ksw_extend2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins,
                int w, int end_bonus, int zdrop, int h0, int* qle, int* tle, int* gtle, int* gscore, int* max_off, Counters* c);
int ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins,
                int w, int* n_cigar, std::vector<uint32_t>* cigar, Counters* c);
struct Kswr { int score = 0, te = -1, qe = -1, score2 = -1, te2 = -1, tb = -1, qb = -1; };
Kswr ksw_align2(int qlen, uint8_t* query, int tlen, uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, Counters* c);
// bwa.c
bool bwa_gen_cigar2(const MemOpt& o, int w_, const Index& b, int l_query, uint8_t* query, int64_t rb, int64_t re,
                    int* score, std::vector<uint32_t>* cigar, int* NM, Counters* c);

// klib ksort.h introsort (unstable; tie order reproduced by running the same algorithm)
template <class T, class Lt> void ks_introsort(size_t n, T* a, Lt lt);

}  // namespace orc

#include "ksort_impl.h"

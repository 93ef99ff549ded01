// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// CPU restatement of lariat's per-barcode align loop (go/src/inference/lariat.go:461-547 and callees,
// go/src/inference/split.go, ordered_map.go, ordered_alignment_map.go, go/src/optimizer/optimizer.go,
// go/src/gobwa/gobwa.go:226-371,400-488).
#pragma once
#include <cstdint>
#include <vector>

#include "bwa_oracle.h"

namespace orc {

struct LariatOpts {
    MemOpt mem;
    int pes_low = -35, pes_high = 500;     // gobwa.go:231-232
    int rescue_score_delta = 25;           // lariat.go:475
    int rescue_max_hits = 50;              // gobwa.go:287,311
    int aln_score_delta = 17;              // lariat.go:476
    double improper_pair_penalty = -4.0;   // main.go:10
    double genome_length = 3200000000.0;   // lariat.go:885
    bool run_inference = true;
};

// Alignment (lariat.go:68-117), one per candidate in `full` order
struct Cand {
    int id = 0;            // hit_id
    int read_id = 0, mate_id = 0;
    bool read1 = false;
    int rid = -1;          // contig ("" == -1)
    int64_t pos = -1, aend = 0;
    int64_t rb = -1, re = -1;
    bool reversed = false;
    int score = 0;
    int readmap_s = 0, readmap_e = 0;
    int nm = 0, matches = 0, mismatches = 0, indels = 0, soft_clipped = 0, soft_clipped_length = 0;
    std::vector<uint32_t> cigar;           // BAM-encoded
    std::vector<int> mismatchLocs, mismatchReadLocs;
    int read_len = 0;
    bool in_filtered = false;
    double log_alignment_probability = 0;
    // inference state
    bool active = false, is_proper = false, bwa_pick = false, active_molecule = false, duplicate = false;
    int molecule_id = -1, mapq = 0;
    double molecule_difference = 0, molecule_confidence = 0.00075 * 0.025, sum_move_probability_change = 1.0;
    int mate_alignment = -1;               // candidate index (barcode-local)
    int secondary = -1, primary = -1;
    // mapq_data
    int second_best = -1;
    double second_best_score = 0, md_score = 0;
    // the rest of MapQData (lariat.go:150-163), read by -debugBamTags only
    int md_copies = 0, md_copies_in_active = 0, md_unique_active = 0, md_copies_outside = 0, md_reads_in_molecule = 0, md_sb_molecule_reads = 0;
    bool md_sb_proper = false;
    double md_sb_molecule_confidence = 0;
    bool has_split_md = false;
    double split_second_best = 0, split_score = 0;
};

struct BarcodeResult {
    std::vector<Cand> cands;                      // `full` flattened, read_id-major
    std::vector<int> cand_off;                    // [n_reads+1]
    std::vector<std::vector<int>> alignments;     // filtered lists: candidate indices per read_id
    int n_molecules = 0;
};

// one read pair, nt4 (post-trim) sequences
struct PairIn { const uint8_t* r1; int l1; const uint8_t* r2; int l2; uint64_t name_seed; };

// GoBwaMemMateSW (gobwa.go:226-337): returns the final reg lists of both mates
void go_bwa_mem_mate_sw(const LariatOpts& o, const Index& idx, const PairIn& p, std::vector<AlnReg>& r1, std::vector<AlnReg>& r2, Counters* cn);

// DoRFAForOneBarcode (lariat.go:461-547) without DumpToBams
void do_rfa_for_one_barcode(const LariatOpts& o, const Index& idx, const std::vector<PairIn>& pairs, bool worth_running_rfa,
                            const int64_t* cen_start, const int64_t* cen_end, BarcodeResult& out, Counters* cn);

// Go 1.9 sort.Sort (quickSort + shell pass + insertion sort + heapSort): lariat.go:1546, split.go:108
template <class Less, class Swap> void go19_sort(int n, Less less, Swap swp);

// The jitter stream of tagBestAlignments (lariat.go:1486 rand.New(rand.NewSource(seed)), :1499,:1510 random.Float64()/2.0):
// Go's math/rand source restated — rng.go: additive lagged Fibonacci x[n] = x[n-607] + x[n-273] mod 2^64, seeded by a
// Lehmer stream XORed with the 607-entry table rngCooked.  The table (go_rng_cooked.inc) is derived from its published
// definition by tools/gen_go_rng_cooked.py and pinned by Go's well-known Seed(1) value stream (tests/test_go_rng.py).
struct GoRand {
    uint64_t vec[607];
    int tap, feed;
    explicit GoRand(int64_t seed);   // rngSource.Seed
    uint64_t uint64();               // rngSource.Uint64
    int64_t int63() { return (int64_t)(uint64() & 0x7fffffffffffffffull); }
    double float64();                // Rand.Float64: float64(Int63()) / (1<<63), redrawn when it rounds to 1
};

}  // namespace orc

#include "gosort_impl.h"

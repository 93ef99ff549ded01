// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// Restatement of lariat's per-barcode align loop.  Each function cites the Go source it follows
// (paths relative to /root/reference/go/src).
#include "lariat_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>

namespace orc {

// ---- Go's math/rand source (see header) ------------------------------------------------------
static const uint64_t kGoRngCooked[607] = {
#include "go_rng_cooked.inc"
};
static inline int32_t go_seedrand(int32_t x) {   // rng.go seedrand: x[n+1] = 48271 * x[n] mod (2^31 - 1)
    int32_t hi = x / 44488, lo = x % 44488;
    x = 48271 * lo - 3399 * hi;
    if (x < 0) x += 2147483647;
    return x;
}
GoRand::GoRand(int64_t seed) {
    tap = 0; feed = 607 - 273;
    seed = seed % 2147483647;
    if (seed < 0) seed += 2147483647;
    if (seed == 0) seed = 89482311;
    int32_t x = (int32_t)seed;
    for (int i = -20; i < 607; ++i) {
        x = go_seedrand(x);
        if (i >= 0) {
            int64_t u = (int64_t)x << 40;
            x = go_seedrand(x);
            u ^= (int64_t)x << 20;
            x = go_seedrand(x);
            u ^= (int64_t)x;
            vec[i] = (uint64_t)u ^ kGoRngCooked[i];
        }
    }
}
uint64_t GoRand::uint64() {
    if (--tap < 0) tap += 607;
    if (--feed < 0) feed += 607;
    uint64_t x = vec[feed] + vec[tap];
    vec[feed] = x;
    return x;
}
double GoRand::float64() {
    for (;;) {
        double f = (double)int63() / 9223372036854775808.0;
        if (f != 1.0) return f;
    }
}

// ---- gobwa.go:226-337 GoBwaMemMateSW --------------------------------------------------------
void go_bwa_mem_mate_sw(const LariatOpts& o, const Index& idx, const PairIn& p, std::vector<AlnReg>& r1, std::vector<AlnReg>& r2, Counters* cn) {
    PeStat pes[4];
    pes[0].failed = 1;
    pes[1].low = o.pes_low; pes[1].high = o.pes_high; pes[1].failed = 0; pes[1].avg = 200.0; pes[1].std = 100.0;
    pes[2].failed = 1; pes[3].failed = 1;
    r1.clear(); r2.clear();
    if (p.l1 > 0) r1 = mem_align1_core(o.mem, idx, p.l1, p.r1, cn);   // gobwa.go:243-251
    if (p.l2 > 0) r2 = mem_align1_core(o.mem, idx, p.l2, p.r2, cn);   // gobwa.go:252-260
    int best1 = 0, best2 = 0;                                         // gobwa.go:264-283
    for (const AlnReg& a : r1) if (a.score > best1) best1 = a.score;
    for (const AlnReg& a : r2) if (a.score > best2) best2 = a.score;
    // rescue read1 from read2's (pre-rescue) hits: gobwa.go:286-301.  The Go code walks a snapshot of read2's regs.
    {
        std::vector<AlnReg> snap = r2;
        int num = 0;
        for (size_t i = 0; i < snap.size() && num < o.rescue_max_hits && p.l1 > 0; ++i)
            if (snap[i].score >= best2 - o.rescue_score_delta) {
                ++num;
                mem_matesw(o.mem, idx, pes, snap[i], p.l1, p.r1, r1, cn);
            }
    }
    // rescue read2 from read1's post-rescue hits, threshold from the PRE-rescue best: gobwa.go:309-325
    {
        std::vector<AlnReg> snap = r1;
        int num = 0;
        for (size_t i = 0; i < snap.size() && num < o.rescue_max_hits && p.l2 > 0; ++i)
            if (snap[i].score >= best1 - o.rescue_score_delta) {
                ++num;
                mem_matesw(o.mem, idx, pes, snap[i], p.l2, p.r2, r2, cn);
            }
    }
}

// ---- ordered_alignment_map.go ----------------------------------------------------------------
struct OrderedAlignmentMap {
    std::vector<int> index;          // key -> slot (-1 absent); keys are read ids
    std::vector<int> reverse_index;  // slot -> key
    std::vector<int> store;          // slot -> candidate index
    explicit OrderedAlignmentMap(int nkeys = 0) : index(nkeys, -1) {}
    int Get(int key) const { int i = index[key]; return i >= 0 ? store[i] : -1; }
    void Set(int key, int val) {
        int i = index[key];
        if (i >= 0) store[i] = val;
        else { index[key] = (int)store.size(); reverse_index.push_back(key); store.push_back(val); }
    }
    void Delete(int key) {   // ordered_alignment_map.go:39-51: move the last element into the hole
        int i = index[key];
        if (i < 0) return;
        if (store.size() > 1) {
            store[i] = store.back();
            index[reverse_index[store.size() - 1]] = i;
            reverse_index[i] = reverse_index.back();
        }
        store.pop_back(); reverse_index.pop_back();
        index[key] = -1;
    }
    int Len() const { return (int)reverse_index.size(); }
};

// CandidateMolecule (lariat.go:181-196)
struct Molecule {
    int id = 0, rid = -1;
    int64_t start = 0, stop = 0;
    // alignments: OrderedMap read_id -> OrderedMap(aln id -> *Alignment), insertion ordered
    std::vector<int> aln_reads;                  // read ids in insertion order
    std::vector<std::vector<int>> aln_lists;     // per inserted read: candidate indices in insertion order
    std::vector<int> aln_slot;                   // read_id -> slot in aln_reads (-1)
    OrderedAlignmentMap best_alignment_for_read, active_alignments;
    bool active_molecule = false;
    double molecule_confidence = 1.0, differences = 0;
    int soft_clipped = 0;
};

struct Ctx {
    const LariatOpts& o;
    std::vector<Cand>& c;
    double improper;
    Ctx(const LariatOpts& oo, std::vector<Cand>& cc) : o(oo), c(cc), improper(oo.improper_pair_penalty) {}
};

// lariat.go:1102-1133
static bool isPair(const Cand& r1, const Cand& r2) {
    if (r1.reversed == r2.reversed || r1.rid != r2.rid) return false;
    const Cand *forward, *reverse;
    if (r1.reversed) { forward = &r2; reverse = &r1; } else { forward = &r1; reverse = &r2; }
    int64_t dist = reverse->pos - forward->pos;
    return dist >= -35 && dist < 750;
}

// lariat.go:599-624
static double scoreAlignment(const Ctx& x, const Cand* aln, const Cand* mate, double log_molecule_penalty) {
    double score = 0.0;
    if (aln) {
        score += (double)(aln->mismatches * -2 + aln->indels * -3);
        if (aln->soft_clipped > 0) {
            score -= 5.0 * (double)aln->soft_clipped;
            score -= (double)aln->soft_clipped_length * 0.5;
        }
    }
    if (mate) {
        score += (double)(mate->mismatches * -2 + mate->indels * -3);
        if (mate->soft_clipped > 0) {
            score -= 5.0 * (double)mate->soft_clipped;
            score -= (double)mate->soft_clipped_length * 0.5;
        }
    }
    if (!mate || !aln || !isPair(*aln, *mate)) score += x.improper;
    if (aln && !aln->active_molecule) score += log_molecule_penalty;
    return score;
}

// lariat.go:590-597
static double psuedoCountAlignmentScore(const Cand& aln, double log_molecule_penalty) {
    double score = 0.0;
    score -= 10.0;
    score -= ((double)aln.read_len - 25.0) * 0.5;
    score += log_molecule_penalty;
    return score;
}

// lariat.go:1309-1319
static bool isActiveMolecule(const Molecule& m, int read_change) {
    double active = (double)(m.active_alignments.Len() + read_change);
    double potential = (double)m.best_alignment_for_read.Len();
    if (active <= 4) return false;
    if (active / potential < 0.1) return false;
    return true;
}

struct Move { double score_change = -DBL_MAX; int source = -1, sink = -1; std::vector<int> toDelete, toSet; int num_moved = 0; };

// lariat.go:1179-1307 (mismatch-locus bookkeeping only validates invariants / prints: it cannot change a score)
static double fastScore(const Ctx& x, const std::vector<Molecule>& mols, int si, int ti, double lup, Move* mv) {
    const Molecule &S = mols[si], &T = mols[ti];
    double change = 0, alignment_change = 0;
    int num = 0;
    if (mv) { mv->toDelete.clear(); mv->toSet.clear(); }
    for (int sidx : S.active_alignments.store) {
        const Cand& sa = x.c[sidx];
        int read_id = sa.read_id;
        int tidx = T.best_alignment_for_read.Get(read_id);
        if (tidx >= 0) {
            const Cand& ta = x.c[tidx];
            int mate_id = sa.mate_id;
            int sm = S.active_alignments.Get(mate_id);
            bool source_has_mate = sm >= 0;
            bool source_has_mate_pair = source_has_mate && isPair(sa, x.c[sm]);
            int tm = T.best_alignment_for_read.Get(mate_id);
            bool sink_has_mate_pair = tm >= 0 && isPair(ta, x.c[tm]) && source_has_mate;
            if (!source_has_mate_pair || (source_has_mate && sink_has_mate_pair)) {
                if (mv) { mv->toDelete.push_back(read_id); mv->toSet.push_back(tidx); }
            }
            alignment_change += ta.log_alignment_probability - sa.log_alignment_probability;
            if (source_has_mate_pair && !sink_has_mate_pair && S.id != T.id) alignment_change += lup / 2.0;
            else if (!source_has_mate_pair && sink_has_mate_pair && S.id != T.id) alignment_change -= lup / 2.0;
            num++;
        }
    }
    bool source_active_before = isActiveMolecule(S, 0), source_active_after = isActiveMolecule(S, -num);
    if (!source_active_after && source_active_before && S.id != T.id) change -= (double)S.best_alignment_for_read.Len() * -0.5;
    bool sink_active_before = isActiveMolecule(T, 0), sink_active_after = isActiveMolecule(T, num);
    if (sink_active_after && !sink_active_before && S.id != T.id) change += (double)T.best_alignment_for_read.Len() * -0.5;
    if (S.active_alignments.Len() - num == 0 && num > 0 && S.id != T.id) change -= -3.0;
    if (T.active_alignments.Len() == 0 && num > 0 && S.id != T.id) change += -3.0;
    change += alignment_change;
    if (mv) { mv->source = si; mv->sink = ti; mv->num_moved = num; mv->score_change = change; }
    return change;
}

// lariat.go:1331-1368
static void acceptMove(Ctx& x, std::vector<Molecule>& mols, const Move& mv) {
    Molecule &S = mols[mv.source], &T = mols[mv.sink];
    for (size_t i = 0; i < mv.toDelete.size(); ++i) {
        int read_id = mv.toDelete[i], sinkAln = mv.toSet[i];
        int sourceAln = S.active_alignments.Get(read_id);
        S.active_alignments.Delete(read_id);
        T.active_alignments.Set(read_id, sinkAln);
        x.c[sourceAln].active = false;
        x.c[sinkAln].active = true;
    }
}

// lariat.go:570-588 (setBad=false)
static void setMoleculeDifferences(Ctx& x, std::vector<Molecule>& mols) {
    for (Molecule& m : mols) {
        int differences = 0;
        for (int a : m.active_alignments.store) differences += x.c[a].mismatches;
        m.differences = (double)differences / (double)m.active_alignments.Len();
        for (int a : m.active_alignments.store) x.c[a].molecule_difference = m.differences;
    }
}

// split.go:29-139
static int GetSplitAlignment(Ctx& x, int primary, const std::vector<int>& alignments, const int64_t* cen_start, const int64_t* cen_end, double* second_best_out) {
    Cand& P = x.c[primary];
    *second_best_out = 0.0;
    if (P.pos == -1) return -1;
    int Ps = P.readmap_s, Pe = P.readmap_e;
    if (Ps > Pe) std::swap(Ps, Pe);
    if ((Pe - Ps) > P.read_len - 15) return -1;
    std::vector<std::pair<int, double>> cands;   // SplitScoring{alignment, score}
    for (int sc : alignments) {
        Cand& S = x.c[sc];
        if (S.active) continue;
        if (S.pos == -1) continue;
        int Ss = S.readmap_s, Se = S.readmap_e, overlap;
        if (Ss > Se) std::swap(Ss, Se);
        if ((Ps < Ss && Pe > Se) || (Ss < Ps && Se > Pe)) continue;
        else if (Ps < Ss) overlap = Pe - Ss;
        else overlap = Se - Ps;
        if (overlap < (Se - Ss) / 2) {
            S.is_proper = P.mate_alignment >= 0 ? isPair(S, x.c[P.mate_alignment]) : false;
            if (S.score >= 36 || S.is_proper) cands.push_back({sc, (double)S.score});
        }
    }
    if (cands.empty()) return -1;
    go19_sort((int)cands.size(), [&](int i, int j) { return cands[i].second > cands[j].second; },
              [&](int i, int j) { std::swap(cands[i], cands[j]); });
    int c = cands[0].first;
    double mapq;
    double second_best = scoreAlignment(x, &P, nullptr, 0.0) + psuedoCountAlignmentScore(x.c[c], 0.0);
    if (cands.size() > 1) {
        mapq = cands[0].second - cands[1].second;
        second_best = scoreAlignment(x, &P, &x.c[cands[1].first], 0.0);
    } else mapq = cands[0].second;
    int64_t start = -1, end = -1;
    if (cen_start && x.c[c].rid >= 0 && cen_start[x.c[c].rid] >= 0) { start = cen_start[x.c[c].rid]; end = cen_end[x.c[c].rid]; }
    if (x.c[c].pos > start && x.c[c].pos <= end) mapq = 0.0;
    if (mapq > 60) mapq = 60;
    x.c[c].mapq = (int)mapq;
    *second_best_out = second_best;
    return c;
}

// lariat.go:867-992
static void estimateMapQualities(Ctx& x, std::vector<std::vector<int>>& alignments, std::vector<Molecule>* mols, double lup,
                                 const int64_t* cen_start, const int64_t* cen_end) {
    // moleculeMapqProbabilitySums, lariat.go:767-790
    if (mols) {
        for (size_t s = 0; s < mols->size(); ++s)
            for (size_t t = 0; t < mols->size(); ++t) {
                if (s == t) continue;
                std::vector<int> src;
                for (int a : (*mols)[s].active_alignments.store)
                    if ((*mols)[t].best_alignment_for_read.Get(x.c[a].read_id) >= 0) src.push_back(a);
                double ch = fastScore(x, *mols, (int)s, (int)t, lup, nullptr);
                double p = std::pow(10.0, ch);
                for (int a : src) x.c[a].sum_move_probability_change += p;
            }
    }
    // updateAlignmentsMoleculeStatus, lariat.go:687-719
    std::vector<int> copies_in(alignments.size(), 0), copies_out(alignments.size(), 0);
    std::vector<std::vector<int>> unique_active(alignments.size());
    if (mols) {
        for (Molecule& m : *mols) {   // setMoleculeConfidences, lariat.go:1048-1059
            m.molecule_confidence = (double)m.active_alignments.Len() / (double)m.best_alignment_for_read.Len();
            for (int a : m.active_alignments.store) {
                if (x.c[a].soft_clipped > 0) m.soft_clipped++;
                x.c[a].molecule_confidence = m.molecule_confidence;
            }
        }
        setMoleculeDifferences(x, *mols);
        for (size_t read_id = 0; read_id < alignments.size(); ++read_id)
            for (int a : alignments[read_id]) {
                Cand& A = x.c[a];
                bool act = false;
                if (A.molecule_id != -1) {
                    Molecule& m = (*mols)[A.molecule_id];
                    act = m.active_alignments.Len() - m.soft_clipped > 4 && m.molecule_confidence > 0.1;
                    A.active_molecule = act;
                }
                if (act) {
                    (*mols)[A.molecule_id].active_molecule = true;
                    copies_in[read_id]++;
                    bool has = false;
                    for (int m : unique_active[read_id]) has = has || m == A.molecule_id;
                    if (!has) unique_active[read_id].push_back(A.molecule_id);
                } else copies_out[read_id]++;
                if (A.molecule_id != -1) A.md_reads_in_molecule = (*mols)[A.molecule_id].active_alignments.Len();
            }
    }
    // calculateLogMoleculePenalty, lariat.go:792-825
    double lmp = 0.0;
    if (mols && !mols->empty()) {
        double dnaLength = 1000.0;
        for (Molecule& m : *mols) {
            if (m.active_molecule) {
                int64_t smallest = INT64_MAX, biggest = -1;
                for (int a : m.active_alignments.store) {
                    if (x.c[a].pos > biggest) biggest = x.c[a].pos;
                    if (x.c[a].pos < smallest) smallest = x.c[a].pos;
                }
                if (biggest >= smallest) dnaLength += (double)(biggest - smallest) + 1000.0;
            } else {
                for (int a : m.active_alignments.store) dnaLength += (double)(x.c[a].aend - x.c[a].pos) * 2.0;
            }
        }
        lmp = std::log10(dnaLength / x.o.genome_length * 0.05);
    }
    for (size_t read_id = 0; read_id < alignments.size(); ++read_id) {
        std::vector<int>& arr = alignments[read_id];
        std::vector<double> scores;
        // appendPsuedocountAlignmentScore, lariat.go:721-739
        if (!arr.empty()) {
            std::vector<int>& mateArray = alignments[x.c[arr[0]].mate_id];
            double best = -DBL_MAX;
            for (int m : mateArray) {
                double s = scoreAlignment(x, nullptr, &x.c[m], lmp);
                if (s > best) best = s;
            }
            if (!mateArray.empty()) scores.push_back(best + psuedoCountAlignmentScore(x.c[arr[0]], lmp));
            else scores.push_back(psuedoCountAlignmentScore(x.c[arr[0]], lmp));
        }
        for (int a : arr)   // lariat.go:892-900
            for (int m : alignments[x.c[a].mate_id])
                if (x.c[a].active && x.c[m].active) { x.c[a].mate_alignment = m; x.c[m].mate_alignment = a; }
        for (int a : arr) {   // lariat.go:902-915
            std::vector<int>& mateArray = alignments[x.c[a].mate_id];
            double best = -DBL_MAX;
            for (int m : mateArray) {
                double s = scoreAlignment(x, &x.c[a], &x.c[m], lmp);
                if (s > best) best = s;
            }
            if (mateArray.empty()) best = scoreAlignment(x, &x.c[a], nullptr, lmp);
            scores.push_back(best);
        }
        // second best, lariat.go:917-943
        double second_best_raw_score = scores.empty() ? 0.0 : scores[0];
        double second_best_log_probability = -1000.0;
        int second_best_alignment = -1;
        bool second_best_proper_pair = false;
        int second_best_molecule_reads = -1;
        double second_best_molecule_confidence = -1.0;
        for (int a : arr)
            for (int m : alignments[x.c[a].mate_id]) {
                double s = scoreAlignment(x, &x.c[a], &x.c[m], lmp);
                if (!x.c[a].active && s > second_best_log_probability) {
                    second_best_log_probability = s;
                    second_best_raw_score = scoreAlignment(x, &x.c[a], &x.c[m], 0.0);
                    second_best_alignment = a;
                    x.c[a].mate_alignment = m;
                    second_best_proper_pair = x.c[a].is_proper;
                    if (x.c[a].molecule_id != -1) {
                        Molecule& alt = (*mols)[x.c[a].molecule_id];
                        second_best_molecule_confidence = alt.molecule_confidence;
                        second_best_molecule_reads = alt.active_alignments.Len();
                    }
                }
            }
        for (int a : arr)   // lariat.go:946-961
            if (x.c[a].active) {
                x.c[a].md_sb_proper = second_best_proper_pair;
                x.c[a].md_sb_molecule_confidence = second_best_molecule_confidence;
                x.c[a].md_sb_molecule_reads = second_best_molecule_reads;
                x.c[a].md_copies = (int)arr.size();
                x.c[a].md_copies_in_active = copies_in[x.c[a].read_id];
                x.c[a].md_copies_outside = copies_out[read_id];
                x.c[a].md_unique_active = (int)unique_active[read_id].size();
                x.c[a].second_best = second_best_alignment;
                x.c[a].second_best_score = second_best_raw_score;
                x.c[a].md_score = scoreAlignment(x, &x.c[a], x.c[a].mate_alignment >= 0 ? &x.c[x.c[a].mate_alignment] : nullptr, 0.0);
            }
        // lariat.go:963-968
        std::sort(scores.begin(), scores.end());
        double total_probability = 0;
        for (int i = (int)scores.size() - 1; i >= 0 && (int)scores.size() - i <= 15; i--) total_probability += std::pow(10.0, scores[i]);
        for (int a : arr) {   // lariat.go:971-989
            Cand& A = x.c[a];
            double score = scoreAlignment(x, &A, A.mate_alignment >= 0 ? &x.c[A.mate_alignment] : nullptr, lmp);
            double mapq = -10.0 * std::log10(1.0 - std::pow(10.0, score) / total_probability);
            double moleculeMapq = -10.0 * std::log10(1.0 - (1.0 / A.sum_move_probability_change));
            // Go math.Min propagates NaN
            mapq = (std::isnan(mapq) || std::isnan(moleculeMapq)) ? NAN : std::min(mapq, moleculeMapq);
            mapq = std::isnan(mapq) ? NAN : std::min(60.0, mapq);
            int64_t start = -1, end = -1;
            if (cen_start && A.rid >= 0 && cen_start[A.rid] >= 0) { start = cen_start[A.rid]; end = cen_end[A.rid]; }
            if (A.pos > start && A.pos <= end) mapq = 0.0;
            A.mapq = std::isnan(mapq) ? INT32_MIN : (int)mapq;   // Go int(NaN) on amd64 = MinInt64
        }
    }
}

// lariat.go:655-685
static void markDuplicates(Ctx& x, const std::vector<std::vector<int>>& alignments) {
    std::map<std::tuple<bool, bool, int, int64_t, int, int64_t>, bool> seen;
    for (const auto& arr : alignments)
        for (int a : arr) {
            Cand& A = x.c[a];
            if (!A.active) continue;
            const Cand& M = x.c[A.mate_alignment];
            auto key = std::make_tuple(A.read1, A.reversed, A.rid, A.pos, M.rid, M.pos);
            if (seen.count(key)) A.duplicate = true;
            else seen[key] = true;
        }
}

// ---- lariat.go:461-547 ------------------------------------------------------------------------
void do_rfa_for_one_barcode(const LariatOpts& o, const Index& idx, const std::vector<PairIn>& pairs, bool worth_running_rfa,
                            const int64_t* cen_start, const int64_t* cen_end, BarcodeResult& out, Counters* cn) {
    out = BarcodeResult();
    std::vector<Cand>& C = out.cands;
    Ctx x(o, C);
    int n_reads = (int)pairs.size() * 2;
    out.cand_off.assign(n_reads + 1, 0);
    out.alignments.assign(n_reads, {});
    // GetChains (lariat.go:1706-1788) + GetAlignments (lariat.go:1552-1704)
    int hit_num = 0;
    for (size_t i = 0; i < pairs.size(); ++i) {
        std::vector<AlnReg> regs[2];
        go_bwa_mem_mate_sw(o, idx, pairs[i], regs[0], regs[1], cn);
        for (int m = 0; m < 2; ++m) {
            int read_id = (int)i * 2 + m, mate_id = (int)i * 2 + (1 - m);
            const uint8_t* rseq = m == 0 ? pairs[i].r1 : pairs[i].r2;
            int rlen = m == 0 ? pairs[i].l1 : pairs[i].l2;
            out.cand_off[read_id] = (int)C.size();
            int bestScore = 0;
            for (const AlnReg& r : regs[m]) if (r.score > bestScore) bestScore = r.score;
            if (regs[m].empty()) {   // placeholder ChainedHit: lariat.go:1737-1750 / 1773-1785
                Cand a;
                a.id = m == 0 ? 0 : hit_num;   // the read-1 placeholder omits hit_id (lariat.go:1738-1748)
                hit_num++;
                a.read_id = read_id; a.mate_id = mate_id; a.read1 = (m == 0);
                a.rid = -1; a.pos = -1; a.aend = 0; a.rb = a.re = -1;
                a.read_len = rlen;
                a.score = 0;
                a.log_alignment_probability = scoreAlignment(x, &a, nullptr, 0.0) - x.improper;
                a.in_filtered = (a.score >= bestScore - o.aln_score_delta);
                C.push_back(a);
                if (cn) ++cn->n_cand;
                continue;
            }
            for (const AlnReg& r : regs[m]) {
                Cand a;
                a.id = hit_num++;
                a.read_id = read_id; a.mate_id = mate_id; a.read1 = (m == 0);
                a.read_len = rlen;
                a.rb = r.rb; a.re = r.re;
                // InterpretAlign, gobwa.go:339-371
                const Contig& ctg = idx.contigs[r.rid];
                int64_t Offset, End;
                if (r.rb < idx.l_pac) Offset = r.rb - ctg.offset; else Offset = idx.l_pac * 2 - 1 - r.rb - ctg.offset;
                if (r.re < idx.l_pac) End = r.re - ctg.offset; else End = idx.l_pac * 2 - 1 - r.re - ctg.offset;
                a.score = r.score; a.readmap_s = r.qb; a.readmap_e = r.qe;
                // GoBwaSmithWaterman -> mem_reg2aln, gobwa.go:400-415,449-488
                Aln al = mem_reg2aln(o.mem, idx, rlen, rseq, r, cn);
                a.rid = al.rid; a.reversed = al.is_rev != 0; a.nm = al.NM; a.cigar = al.cigar;
                // lariat.go:1573-1637: CIGAR walk against GetSeq(refStart, refEnd)
                int64_t refStart = Offset, refEnd = End;
                if (a.reversed) { refStart = End + 1; refEnd = Offset + 1; }
                // GetSeq (gobwa.go:50-80): slice clamped to the contig, reverse-complemented when reversed
                std::vector<uint8_t> refSeq((size_t)std::max<int64_t>(0, refEnd - refStart), 0);
                {
                    int64_t cb = refStart + ctg.offset, ce = refEnd + ctg.offset, mid = (cb + ce) >> 1;
                    int rid2;
                    std::vector<uint8_t> raw = bns_fetch_seq(idx, &cb, mid, &ce, &rid2);
                    int64_t n = ce - cb;
                    // stored as nt4 here instead of ASCII; unfilled tail stays 0 ('\0' upstream, never equal to a base) -> use 255
                    std::fill(refSeq.begin(), refSeq.end(), 255);
                    if (a.reversed) { for (int64_t k = 0; k < n && n - k - 1 < (int64_t)refSeq.size(); ++k) refSeq[n - k - 1] = 3 - raw[k]; }
                    else { for (int64_t k = 0; k < n && k < (int64_t)refSeq.size(); ++k) refSeq[k] = raw[k]; }
                }
                int matches = 0, indels = 0, indel_length = 0, soft_clipping = 0, soft_clipping_length = 0;
                int refSeqOffset = 0, readOffset = 0;
                int ncig = (int)a.cigar.size();
                int k0 = a.reversed ? ncig - 1 : 0, kinc = a.reversed ? -1 : 1;
                for (int k = k0; k < ncig && k >= 0; k += kinc) {
                    int op = a.cigar[k] & 0xf, len = (int)(a.cigar[k] >> 4);
                    if (op == 0) {
                        matches += len;
                        for (int t = 0; t < len; ++t) {
                            if (refSeqOffset + t >= (int)refSeq.size()) continue;
                            if (readOffset + t >= rlen) break;   // upstream panics here
                            // the read holds nt4 codes (N = 4); the reference slice holds ACGT only
                            if (refSeq[refSeqOffset + t] != rseq[readOffset + t]) {
                                if (a.reversed) a.mismatchLocs.push_back((int)refEnd - (refSeqOffset + t));
                                else a.mismatchLocs.push_back(refSeqOffset + (int)refStart + t);
                                a.mismatchReadLocs.push_back(readOffset + t);
                            }
                        }
                        refSeqOffset += len; readOffset += len;
                    } else if (op == 1) { indels += 1; indel_length += len; readOffset += len; }
                    else if (op == 2) { indels += 1; indel_length += len; refSeqOffset += len; }
                    else if (op == 3) { soft_clipping += 1; soft_clipping_length += len; readOffset += len; }
                }
                int mismatches = a.nm - indel_length;
                matches -= mismatches;
                if (mismatches < 0) mismatches = 0;
                a.matches = matches; a.mismatches = mismatches; a.indels = indels;
                a.soft_clipped = soft_clipping; a.soft_clipped_length = soft_clipping_length;
                a.pos = Offset; a.aend = End;
                if (a.pos != -1 && a.reversed) { a.pos = End + 1; a.aend = Offset + 1; }   // lariat.go:1645-1650
                a.log_alignment_probability = scoreAlignment(x, &a, nullptr, 0.0) - x.improper;   // lariat.go:1691
                a.in_filtered = (a.score >= bestScore - o.aln_score_delta);                        // lariat.go:1698
                C.push_back(a);
                if (cn) ++cn->n_cand;
            }
        }
    }
    out.cand_off[n_reads] = (int)C.size();
    for (int r = 0; r < n_reads; ++r)
        for (int a = out.cand_off[r]; a < out.cand_off[r + 1]; ++a)
            if (C[a].in_filtered) out.alignments[r].push_back(a);
    if (!o.run_inference) return;
    std::vector<std::vector<int>>& alignments = out.alignments;

    // tagBestAlignments (lariat.go:1466-1549)
    std::vector<std::vector<int>> positions;
    std::map<int, int> contigs;   // contig -> index into positions, first-seen order
    std::vector<bool> touched(n_reads, false);
    for (int read_id = 0; read_id < n_reads; ++read_id) {
        const std::vector<int>& arr = alignments[read_id];
        bool was_touched = touched[read_id];
        double bestScore = -DBL_MAX;
        int bestAlignment = -1, bestMate = -1;
        GoRand rng((int64_t)pairs[read_id >> 1].name_seed);
        for (int a : arr) {
            const std::vector<int>& mates = alignments[C[a].mate_id];
            for (int m : mates) {
                double total = scoreAlignment(x, &C[a], &C[m], 0.0) + (rng.float64() / 2.0);
                if (total > bestScore) { bestScore = total; bestAlignment = a; bestMate = m; }
            }
            if (mates.empty()) {
                double s = (double)C[a].score + rng.float64() / 2.0;
                if (s > bestScore) { bestScore = s; bestAlignment = a; }
            }
            auto it = contigs.find(C[a].rid);
            if (it != contigs.end()) positions[it->second].push_back(a);
            else { contigs[C[a].rid] = (int)positions.size(); positions.push_back({a}); }
        }
        if (!was_touched) {
            C[bestAlignment].active = true; C[bestAlignment].bwa_pick = true;
            if (bestMate >= 0) {
                if (isPair(C[bestAlignment], C[bestMate])) { C[bestAlignment].is_proper = true; C[bestMate].is_proper = true; }
                C[bestMate].active = true; C[bestMate].bwa_pick = true;
                touched[C[bestMate].read_id] = true;
            }
        }
    }
    for (auto& pl : positions)
        go19_sort((int)pl.size(), [&](int i, int j) { return C[pl[i]].pos < C[pl[j]].pos; }, [&](int i, int j) { std::swap(pl[i], pl[j]); });

    if (!worth_running_rfa) {   // lariat.go:489-496
        estimateMapQualities(x, alignments, nullptr, x.improper, cen_start, cen_end);
        markDuplicates(x, alignments);
    } else {
        // inferMolecules (lariat.go:1370-1408)
        std::vector<Molecule> mols;
        for (auto& pl : positions) {
            for (size_t i = 0; i < pl.size(); ++i) {
                Cand& A = C[pl[i]];
                if (i == 0 || A.pos - C[pl[i - 1]].pos > 50000) {
                    if (i > 0) mols.back().stop = C[pl[i - 1]].pos;
                    Molecule m;
                    m.rid = A.rid; m.start = A.pos; m.id = (int)mols.size();
                    m.aln_slot.assign(n_reads, -1);
                    m.best_alignment_for_read = OrderedAlignmentMap(n_reads);
                    m.active_alignments = OrderedAlignmentMap(n_reads);
                    mols.push_back(std::move(m));
                }
                Molecule& m = mols.back();
                int slot = m.aln_slot[A.read_id];
                if (slot < 0) { slot = (int)m.aln_reads.size(); m.aln_slot[A.read_id] = slot; m.aln_reads.push_back(A.read_id); m.aln_lists.push_back({}); }
                // OrderedMap.Set(id, aln): overwrite when the id already exists (only possible for id collisions)
                bool found = false;
                for (int& e : m.aln_lists[slot]) if (C[e].id == A.id) { e = pl[i]; found = true; break; }
                if (!found) m.aln_lists[slot].push_back(pl[i]);
            }
            if (!pl.empty()) mols.back().stop = C[pl.back()].pos;
        }
        // markBestAlignmentForReadInMolecule (lariat.go:1410-1463)
        for (Molecule& m : mols) {
            for (size_t s = 0; s < m.aln_reads.size(); ++s) {
                int read_id = m.aln_reads[s];
                double best_score = -DBL_MAX;
                int best_alignment = -1;
                for (int a : m.aln_lists[s]) {
                    int ms = m.aln_slot[C[a].mate_id];
                    if (ms >= 0 && !m.aln_lists[ms].empty()) {
                        for (int ma : m.aln_lists[ms]) {
                            double sc = scoreAlignment(x, &C[a], &C[ma], 0.0);
                            if (sc > best_score) { best_score = sc; best_alignment = a; }
                        }
                    } else if (C[a].log_alignment_probability > best_score) { best_score = C[a].log_alignment_probability; best_alignment = a; }
                    if (C[a].active) m.active_alignments.Set(read_id, a);
                }
                if (C[best_alignment].active) m.active_alignments.Set(read_id, best_alignment);
                m.best_alignment_for_read.Set(read_id, best_alignment);
            }
        }
        // scrapMolecules (lariat.go:1061-1086)
        {
            std::vector<Molecule> kept;
            int count = 0;
            for (Molecule& m : mols) {
                bool keep = m.active_alignments.Len() > 0;
                for (auto& l : m.aln_lists) for (int a : l) C[a].molecule_id = keep ? count : -1;
                if (keep) { kept.push_back(std::move(m)); count++; }
            }
            mols.swap(kept);
        }
        setMoleculeDifferences(x, mols);
        // optimizer.Optimize(opt, 1, 2, 4*M) (optimizer.go:15-27) with GenerateMove (lariat.go:1135-1167)
        int M = (int)mols.size(), source = 0;
        for (int temp = 0; temp < 2; ++temp)
            for (int step = 0; step < 4 * M; ++step) {
                if (mols[source].active_alignments.Len() == 0) { source = (source + 1) % M; continue; }
                Move best, mv;
                for (int i = 0; i < M; ++i) {
                    if (i == source) continue;
                    double score = fastScore(x, mols, source, i, x.improper, &mv);
                    if ((score > best.score_change || (score == best.score_change && best.sink >= 0 &&
                                                       mols[mv.sink].active_alignments.Len() > mols[best.sink].active_alignments.Len())) &&
                        mv.num_moved > 0)
                        best = mv;
                }
                if (best.score_change > 0 || (best.score_change == 0 && best.sink >= 0 &&
                                              mols[best.sink].active_alignments.Len() > mols[source].active_alignments.Len()))
                    acceptMove(x, mols, best);
                source = (source + 1) % M;
            }
        out.n_molecules = M;
        estimateMapQualities(x, alignments, &mols, x.improper, cen_start, cen_end);
        markDuplicates(x, alignments);
    }
    // CheckSplitReads over `full` (split.go:142-158)
    for (int r = 0; r < n_reads; ++r) {
        std::vector<int> full;
        for (int a = out.cand_off[r]; a < out.cand_off[r + 1]; ++a) full.push_back(a);
        int active = -1;
        for (int a : full) if (C[a].active) { active = a; break; }
        if (active < 0) continue;
        double second_best;
        int split = GetSplitAlignment(x, active, full, cen_start, cen_end, &second_best);
        C[active].secondary = split;
        if (split >= 0) {
            C[split].has_split_md = true;   // a new MapQData with the two scores: the other fields are zero again
            C[split].md_copies = C[split].md_copies_in_active = C[split].md_unique_active = C[split].md_copies_outside = C[split].md_reads_in_molecule = 0;
            C[split].split_second_best = second_best;
            C[split].split_score = scoreAlignment(x, &C[split], C[active].mate_alignment >= 0 ? &C[C[active].mate_alignment] : nullptr, 0.0);
            C[split].primary = active;
        }
    }
}

}  // namespace orc

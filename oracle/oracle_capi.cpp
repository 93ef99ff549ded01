// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// C entry points (ctypes) that run the CPU restatement and fill the SAME result structs the product's
// C-ABI returns (include/lariat_hip.h), so tests diff field by field.  Also the `cpu_baseline` leg of bench.py
// (threaded over barcodes like the reference's worker pool, lariat.go:348-350).
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/lariat_hip.h"
#include "lariat_oracle.h"

#include "gosort_impl.h"
#include "ksort_impl.h"
using namespace orc;

namespace {
thread_local std::string g_err;

struct ResultArena {
    lh_result r;
    std::vector<int64_t> cand_off, pos, aend, rb, re, cigar_off, mm_off, mate_idx, active_idx, second_best_idx, split_idx;
    std::vector<int32_t> rid, score, qb, qe, nm, matches, mismatches, indels, soft_clipped, soft_clipped_length, mm_ref, mm_read, molecule_id, mapq, split_mapq;
    std::vector<uint8_t> reversed, in_filtered, active, is_proper, bwa_pick, active_molecule, duplicate;
    std::vector<uint32_t> cigar;
    std::vector<double> lap, mol_diff, mol_conf, sum_move, second_best_score, as_score, split_second_best, split_score;
    std::vector<int32_t> md_int;   // 7 per candidate: copies, in active, unique active, outside, reads in molecule, second best: molecule reads, proper
    std::vector<double> md_sb_conf;
};

struct DumpArena {
    lh_stage_dump d;
    std::vector<int64_t> intv_off, seed_off, seed_rbeg, chain_off, chain_pos, reg_off, reg_rb, reg_re;
    std::vector<uint64_t> intv;
    std::vector<int32_t> seed_qbeg, seed_len, seed_rid, chain_nseeds, chain_rid, chain_w, chain_kept, reg_qb, reg_qe, reg_rid, reg_score, reg_truesc, reg_w,
        reg_seedcov, reg_seedlen0, reg_csub, reg_secondary;
};

LariatOpts to_opts(const lh_opts* o) {
    LariatOpts L;
    if (!o) return L;
    MemOpt& m = L.mem;
    m.a = o->a; m.b = o->b; m.o_del = o->o_del; m.e_del = o->e_del; m.o_ins = o->o_ins; m.e_ins = o->e_ins;
    m.pen_unpaired = o->pen_unpaired; m.pen_clip5 = o->pen_clip5; m.pen_clip3 = o->pen_clip3;
    m.w = o->w; m.zdrop = o->zdrop; m.T = o->T;
    m.min_seed_len = o->min_seed_len; m.min_chain_weight = o->min_chain_weight; m.max_chain_extend = o->max_chain_extend;
    m.split_factor = o->split_factor; m.split_width = o->split_width; m.max_occ = o->max_occ; m.max_chain_gap = o->max_chain_gap; m.max_ins = o->max_ins;
    m.mask_level = o->mask_level; m.drop_ratio = o->drop_ratio; m.XA_drop_ratio = o->XA_drop_ratio; m.mask_level_redun = o->mask_level_redun;
    m.mapQ_coef_len = o->mapQ_coef_len; m.max_mem_intv = o->max_mem_intv; m.max_matesw = o->max_matesw;
    int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) m.mat[k++] = i == j ? m.a : -m.b; m.mat[k++] = -1; }
    for (int j = 0; j < 5; ++j) m.mat[k++] = -1;
    L.pes_low = o->pes_low; L.pes_high = o->pes_high;
    L.rescue_score_delta = o->rescue_score_delta; L.rescue_max_hits = o->rescue_max_hits; L.aln_score_delta = o->aln_score_delta;
    L.improper_pair_penalty = o->improper_pair_penalty; L.genome_length = o->genome_length; L.run_inference = o->run_inference != 0;
    return L;
}
}  // namespace

extern "C" {

const char* lo_last_error() { return g_err.c_str(); }

// the math/rand value stream of rand.New(rand.NewSource(seed)): n draws of Int63 and of Float64 (tests/test_go_rng.py)
void lo_go_rand_stream(int64_t seed, int32_t n, int64_t* int63_out, double* float64_out) {
    if (int63_out) { orc::GoRand g(seed); for (int i = 0; i < n; ++i) int63_out[i] = g.int63(); }
    if (float64_out) { orc::GoRand g(seed); for (int i = 0; i < n; ++i) float64_out[i] = g.float64(); }
}

// Go 1.9's sort.Sort over `n_sorts` index spaces [first[k], first[k + 1]) of keys[]: perm[] starts as the identity and is swapped along
// with the keys (tests/test_sort.py checks the device's serial and wave-wide restatements against it)
void lo_gosort(int32_t n_sorts, const int32_t* first, int64_t* keys, int32_t* perm) {
    for (int k = 0; k < n_sorts; ++k) {
        int64_t* kp = keys + first[k];
        int32_t* ip = perm + first[k];
        orc::go19_sort(first[k + 1] - first[k], [&](int i, int j) { return kp[i] < kp[j]; },
                       [&](int i, int j) { std::swap(kp[i], kp[j]); std::swap(ip[i], ip[j]); });
    }
}

// klib's ks_introsort over `n_sorts` index spaces of keys[]: (key, index) pairs compared by key alone; perm[] = the index each position ends up holding
// (tests/test_sort.py checks the device's one-lane and wave-wide restatements against it)
void lo_ks_introsort(int32_t n_sorts, const int32_t* first, const int64_t* keys, int32_t* perm) {
    struct KI { int64_t k; int32_t i; };
    for (int c = 0; c < n_sorts; ++c) {
        const int n = first[c + 1] - first[c];
        std::vector<KI> v((size_t)n);
        for (int i = 0; i < n; ++i) { v[(size_t)i].k = keys[first[c] + i]; v[(size_t)i].i = i; }
        orc::ks_introsort((size_t)n, v.data(), [](const KI& a, const KI& b) { return a.k < b.k; });
        for (int i = 0; i < n; ++i) perm[first[c] + i] = v[(size_t)i].i;
    }
}

void lo_opts_init(lh_opts* o) {
    memset(o, 0, sizeof *o);
    MemOpt m;
    o->abi_version = LH_ABI_VERSION;
    o->a = m.a; o->b = m.b; o->o_del = m.o_del; o->e_del = m.e_del; o->o_ins = m.o_ins; o->e_ins = m.e_ins;
    o->pen_unpaired = m.pen_unpaired; o->pen_clip5 = m.pen_clip5; o->pen_clip3 = m.pen_clip3;
    o->w = m.w; o->zdrop = m.zdrop; o->T = m.T;
    o->min_seed_len = m.min_seed_len; o->min_chain_weight = m.min_chain_weight; o->max_chain_extend = m.max_chain_extend;
    o->split_factor = m.split_factor; o->split_width = m.split_width; o->max_occ = m.max_occ; o->max_chain_gap = m.max_chain_gap; o->max_ins = m.max_ins;
    o->mask_level = m.mask_level; o->drop_ratio = m.drop_ratio; o->XA_drop_ratio = m.XA_drop_ratio; o->mask_level_redun = m.mask_level_redun;
    o->mapQ_coef_len = m.mapQ_coef_len; o->max_mem_intv = m.max_mem_intv; o->max_matesw = m.max_matesw;
    o->pes_low = -35; o->pes_high = 500;
    o->rescue_score_delta = 25; o->rescue_max_hits = 50; o->aln_score_delta = 17;
    o->improper_pair_penalty = -4.0; o->genome_length = 3200000000.0; o->run_inference = 1;
}

int lo_index_load(const char* prefix, Index** out) {
    auto idx = std::make_unique<Index>();
    std::string err;
    if (!index_load(prefix, *idx, &err)) { g_err = err; return LH_E_IO; }
    *out = idx.release();
    return LH_OK;
}

// an index from in-memory images in the files' layout (what lh_index_export returns): the oracle for device-built indexes
int lo_index_from_arrays(uint64_t primary, const uint64_t L2[5], const uint32_t* bwt, uint64_t bwt_words, int32_t sa_intv, const uint64_t* sa, uint64_t n_sa,
                         const uint8_t* pac, int64_t l_pac, int32_t n_contigs, const int64_t* contig_off, const int32_t* contig_len, const char* const* contig_name,
                         Index** out) {
    auto idx = std::make_unique<Index>();
    idx->primary = primary;
    for (int i = 0; i < 5; ++i) idx->L2[i] = L2[i];
    idx->seq_len = L2[4]; idx->bwt_size = bwt_words;
    idx->bwt.assign(bwt, bwt + bwt_words);
    idx->sa_intv = sa_intv; idx->n_sa = n_sa;
    idx->sa.assign(sa, sa + n_sa);
    idx->l_pac = l_pac;
    idx->pac.assign(pac, pac + (l_pac / 4 + 1));
    for (int i = 0; i < n_contigs; ++i) {
        Contig c;
        c.name = contig_name[i]; c.offset = contig_off[i]; c.len = contig_len[i];
        idx->contigs.push_back(c);
    }
    *out = idx.release();
    return LH_OK;
}

// naive builder (small genomes): nt4 contigs -> index in memory
int lo_index_build_naive(int32_t n_contigs, const char* const* names, const uint8_t* const* nt4, const int64_t* lens, Index** out) {
    std::vector<std::string> nm;
    std::vector<std::vector<uint8_t>> sq;
    for (int i = 0; i < n_contigs; ++i) { nm.push_back(names[i]); sq.emplace_back(nt4[i], nt4[i] + lens[i]); }
    auto idx = std::make_unique<Index>();
    index_build_naive(nm, sq, *idx);
    *out = idx.release();
    return LH_OK;
}

void lo_index_free(Index* idx) { delete idx; }
// bntann1_t.is_alt per contig (what <prefix>.alt sets on load), for indexes made from arrays
void lo_index_set_alt(Index* idx, const uint8_t* is_alt) { for (size_t i = 0; i < idx->contigs.size(); ++i) idx->contigs[i].is_alt = is_alt[i] ? 1 : 0; }
int32_t lo_index_contig_alt(const Index* idx, int i) { return idx->contigs[i].is_alt; }
int64_t lo_index_l_pac(const Index* idx) { return idx->l_pac; }
int32_t lo_index_n_contigs(const Index* idx) { return (int32_t)idx->contigs.size(); }
const char* lo_index_contig_name(const Index* idx, int i) { return idx->contigs[i].name.c_str(); }
int64_t lo_index_contig_len(const Index* idx, int i) { return idx->contigs[i].len; }
int64_t lo_index_contig_offset(const Index* idx, int i) { return idx->contigs[i].offset; }

// raw arrays (so the product's lh_index_from_arrays can be fed from an oracle-built index in tests)
uint64_t lo_index_primary(const Index* idx) { return idx->primary; }
const uint64_t* lo_index_L2(const Index* idx) { return idx->L2; }
const uint32_t* lo_index_bwt(const Index* idx, uint64_t* n_words) { *n_words = idx->bwt_size; return idx->bwt.data(); }
const uint64_t* lo_index_sa(const Index* idx, uint64_t* n_sa, int32_t* sa_intv) { *n_sa = idx->n_sa; *sa_intv = idx->sa_intv; return idx->sa.data(); }
const uint8_t* lo_index_pac(const Index* idx) { return idx->pac.data(); }

// file images: which = 0 bwt, 1 sa, 2 pac, 3 ann, 4 amb.  Returns size; copies into buf if buf != NULL.
int64_t lo_index_image(const Index* idx, int which, uint8_t* buf) {
    std::vector<uint8_t> v;
    std::string s;
    switch (which) {
        case 0: v = image_bwt(*idx); break;
        case 1: v = image_sa(*idx); break;
        case 2: v = image_pac(*idx); break;
        case 3: s = image_ann(*idx); v.assign(s.begin(), s.end()); break;
        case 4: s = image_amb(*idx); v.assign(s.begin(), s.end()); break;
        default: return -1;
    }
    if (buf) memcpy(buf, v.data(), v.size());
    return (int64_t)v.size();
}

// GetSeq (gobwa.go:50-80)
int lo_get_seq(const Index* idx, int32_t rid, int64_t start, int64_t end, int32_t reversed, char* out) {
    if (rid < 0 || rid >= (int)idx->contigs.size() || end < start) { g_err = "bad GetSeq args"; return LH_E_ARG; }
    int64_t off = idx->contigs[rid].offset, cb = start + off, ce = end + off;
    int r2;
    memset(out, 0, end - start);
    std::vector<uint8_t> raw = bns_fetch_seq(*idx, &cb, (cb + ce) >> 1, &ce, &r2);
    int64_t n = ce - cb;
    for (int64_t i = 0; i < n; ++i) {
        if (reversed) { if (n - i - 1 < end - start) out[n - i - 1] = "TGCA"[raw[i]]; }
        else if (i < end - start) out[i] = "ACGT"[raw[i]];
    }
    return LH_OK;
}

// the hot path on the CPU, threaded over barcodes
int lo_align_barcodes(const Index* idx, const lh_opts* opts, const lh_batch* b, int32_t threads, lh_result** out) {
    LariatOpts L = to_opts(opts);
    int nb = b->n_barcodes;
    std::vector<BarcodeResult> res(nb);
    std::vector<Counters> cnts(threads > 0 ? threads : 1);
    std::atomic<int> next(0);
    auto work = [&](int tid) {
        for (;;) {
            int bc = next.fetch_add(1);
            if (bc >= nb) break;
            std::vector<PairIn> pairs;
            for (int p = b->bc_pair_off[bc]; p < b->bc_pair_off[bc + 1]; ++p) {
                PairIn pi;
                pi.r1 = b->seq + b->seq_off[2 * p]; pi.l1 = (int)(b->seq_off[2 * p + 1] - b->seq_off[2 * p]);
                pi.r2 = b->seq + b->seq_off[2 * p + 1]; pi.l2 = (int)(b->seq_off[2 * p + 2] - b->seq_off[2 * p + 1]);
                pi.name_seed = b->name_seed ? b->name_seed[p] : 1;
                pairs.push_back(pi);
            }
            do_rfa_for_one_barcode(L, *idx, pairs, b->bc_do_rfa ? b->bc_do_rfa[bc] != 0 : true, b->cen_start, b->cen_end, res[bc], &cnts[tid]);
        }
    };
    if (threads <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (!out) return LH_OK;
    auto A = new ResultArena();
    int64_t n_reads = 2 * (int64_t)b->n_pairs;
    A->cand_off.assign(n_reads + 1, 0);
    A->active_idx.assign(n_reads, -1); A->second_best_idx.assign(n_reads, -1); A->split_idx.assign(n_reads, -1);
    A->second_best_score.assign(n_reads, 0); A->as_score.assign(n_reads, 0); A->split_mapq.assign(n_reads, 0);
    A->split_second_best.assign(n_reads, 0); A->split_score.assign(n_reads, 0);
    A->cigar_off.push_back(0); A->mm_off.push_back(0);
    int64_t base = 0;
    for (int bc = 0; bc < nb; ++bc) {
        const BarcodeResult& R = res[bc];
        int64_t r0 = 2 * (int64_t)b->bc_pair_off[bc];
        int nr = (int)R.cand_off.size() - 1;
        for (int r = 0; r < nr; ++r) A->cand_off[r0 + r] = base + R.cand_off[r];
        for (size_t i = 0; i < R.cands.size(); ++i) {
            const Cand& c = R.cands[i];
            A->rid.push_back(c.rid); A->pos.push_back(c.pos); A->aend.push_back(c.aend); A->rb.push_back(c.rb); A->re.push_back(c.re);
            A->reversed.push_back(c.reversed); A->score.push_back(c.score); A->qb.push_back(c.readmap_s); A->qe.push_back(c.readmap_e);
            A->nm.push_back(c.nm); A->matches.push_back(c.matches); A->mismatches.push_back(c.mismatches); A->indels.push_back(c.indels);
            A->soft_clipped.push_back(c.soft_clipped); A->soft_clipped_length.push_back(c.soft_clipped_length);
            A->in_filtered.push_back(c.in_filtered);
            A->cigar.insert(A->cigar.end(), c.cigar.begin(), c.cigar.end()); A->cigar_off.push_back((int64_t)A->cigar.size());
            A->mm_ref.insert(A->mm_ref.end(), c.mismatchLocs.begin(), c.mismatchLocs.end());
            A->mm_read.insert(A->mm_read.end(), c.mismatchReadLocs.begin(), c.mismatchReadLocs.end()); A->mm_off.push_back((int64_t)A->mm_ref.size());
            A->lap.push_back(c.log_alignment_probability);
            A->active.push_back(c.active); A->is_proper.push_back(c.is_proper); A->bwa_pick.push_back(c.bwa_pick);
            A->active_molecule.push_back(c.active_molecule); A->duplicate.push_back(c.duplicate);
            A->molecule_id.push_back(c.molecule_id); A->mapq.push_back(c.mapq);
            A->mol_diff.push_back(c.molecule_difference); A->mol_conf.push_back(c.molecule_confidence); A->sum_move.push_back(c.sum_move_probability_change);
            A->mate_idx.push_back(c.mate_alignment >= 0 ? base + c.mate_alignment : -1);
            for (int v : {c.md_copies, c.md_copies_in_active, c.md_unique_active, c.md_copies_outside, c.md_reads_in_molecule, c.md_sb_molecule_reads, (int)c.md_sb_proper}) A->md_int.push_back(v);
            A->md_sb_conf.push_back(c.md_sb_molecule_confidence);
            if (c.active) {
                int64_t r = r0 + c.read_id;
                A->active_idx[r] = base + (int64_t)i;
                A->second_best_idx[r] = c.second_best >= 0 ? base + c.second_best : -1;
                A->second_best_score[r] = c.second_best_score; A->as_score[r] = c.md_score;
                if (c.secondary >= 0) {
                    const Cand& s = R.cands[c.secondary];
                    A->split_idx[r] = base + c.secondary; A->split_mapq[r] = s.mapq;
                    A->split_second_best[r] = s.split_second_best; A->split_score[r] = s.split_score;
                }
            }
        }
        base += (int64_t)R.cands.size();
    }
    // barcodes are contiguous pair ranges; reads of pairs not covered by any barcode stay empty
    A->cand_off[n_reads] = base;
    for (int64_t r = n_reads - 1; r >= 0; --r) if (A->cand_off[r] == 0 && r > 0 && A->cand_off[r + 1] != 0 && false) {}
    lh_result& r = A->r;
    memset(&r, 0, sizeof r);
    r.abi_version = LH_ABI_VERSION; r.n_reads = (int32_t)n_reads; r.n_cand = base;
    r.cand_off = A->cand_off.data(); r.rid = A->rid.data(); r.pos = A->pos.data(); r.aend = A->aend.data(); r.rb = A->rb.data(); r.re = A->re.data();
    r.reversed = A->reversed.data(); r.score = A->score.data(); r.qb = A->qb.data(); r.qe = A->qe.data(); r.nm = A->nm.data();
    r.matches = A->matches.data(); r.mismatches = A->mismatches.data(); r.indels = A->indels.data(); r.soft_clipped = A->soft_clipped.data();
    r.soft_clipped_length = A->soft_clipped_length.data(); r.in_filtered = A->in_filtered.data();
    r.cigar_off = A->cigar_off.data(); r.cigar = A->cigar.data(); r.mm_off = A->mm_off.data(); r.mm_ref_loc = A->mm_ref.data(); r.mm_read_loc = A->mm_read.data();
    r.log_alignment_probability = A->lap.data();
    r.active = A->active.data(); r.is_proper = A->is_proper.data(); r.bwa_pick = A->bwa_pick.data(); r.active_molecule = A->active_molecule.data();
    r.duplicate = A->duplicate.data(); r.molecule_id = A->molecule_id.data(); r.mapq = A->mapq.data();
    r.molecule_difference = A->mol_diff.data(); r.molecule_confidence = A->mol_conf.data(); r.sum_move_probability_change = A->sum_move.data();
    r.mate_idx = A->mate_idx.data();
    r.active_idx = A->active_idx.data(); r.second_best_idx = A->second_best_idx.data(); r.second_best_score = A->second_best_score.data();
    r.as_score = A->as_score.data(); r.split_idx = A->split_idx.data(); r.split_mapq = A->split_mapq.data();
    r.split_second_best = A->split_second_best.data(); r.split_score = A->split_score.data();
    Counters tot;
    for (auto& c : cnts) tot.add(c);
    r.n_ext = tot.n_ext; r.n_lf = tot.n_lf; r.n_sa = tot.n_sa; r.win_bases = tot.win_bases; r.n_chain_ext = tot.n_chain_ext;
    r.ext_cells = tot.ext_cells; r.glob_cells = tot.glob_cells; r.n_rescue = tot.n_rescue; r.rescue_cells = tot.rescue_cells; r.rescue_cells_exec = tot.rescue_cells; r.n_glob_listed = r.n_glob_exec = tot.n_glob;   // (the oracle runs every cell)
    r.arena_ = A;
    *out = &A->r;
    return LH_OK;
}

// MapQData as the oracle's molecules left it (the fields -debugBamTags prints), for a result of lo_align_barcodes: md_int = 7 values per
// candidate (copies, copies_in_active_molecules, unique_molecules_active, copies_outside_active_molecules, reads_in_molecule,
// second_best_molecule_reads, second_best_proper_pair), md_sb_conf = second_best_molecule_confidence
int lo_result_mapq_data(const lh_result* r, const int32_t** md_int, const double** md_sb_conf) {
    if (!r || !r->arena_ || !md_int || !md_sb_conf) { g_err = "lo_result_mapq_data: bad argument"; return LH_E_ARG; }
    ResultArena* A = (ResultArena*)r->arena_;
    *md_int = A->md_int.data(); *md_sb_conf = A->md_sb_conf.data();
    return LH_OK;
}

void lo_result_free(lh_result* r) { if (r) delete (ResultArena*)r->arena_; }

// stage dump of mem_align1_core for every read of the batch (single end, before rescue)
int lo_stage_dump(const Index* idx, const lh_opts* opts, const lh_batch* b, lh_stage_dump** out) {
    LariatOpts L = to_opts(opts);
    auto A = new DumpArena();
    int64_t n_reads = 2 * (int64_t)b->n_pairs;
    A->intv_off.push_back(0); A->seed_off.push_back(0); A->chain_off.push_back(0); A->reg_off.push_back(0);
    for (int64_t r = 0; r < n_reads; ++r) {
        const uint8_t* s = b->seq + b->seq_off[r];
        int l = (int)(b->seq_off[r + 1] - b->seq_off[r]);
        std::vector<Intv> intv;
        std::vector<Seed> seeds;
        std::vector<Chain> chn;
        std::vector<AlnReg> regs;
        if (l > 0) {
            chn = mem_chain(L.mem, *idx, l, s, nullptr, &intv, &seeds);
            mem_chain_flt(L.mem, chn);
            for (const Chain& c : chn) mem_chain2aln(L.mem, *idx, l, s, c, regs, nullptr);
            mem_sort_dedup_patch(L.mem, idx, s, regs, nullptr);
        }
        for (const Intv& p : intv) { A->intv.push_back(p.x[0]); A->intv.push_back(p.x[1]); A->intv.push_back(p.x[2]); A->intv.push_back(p.info); }
        A->intv_off.push_back((int64_t)A->intv.size() / 4);
        for (const Seed& sd : seeds) { A->seed_rbeg.push_back(sd.rbeg); A->seed_qbeg.push_back(sd.qbeg); A->seed_len.push_back(sd.len); A->seed_rid.push_back(sd.score); }
        A->seed_off.push_back((int64_t)A->seed_rbeg.size());
        for (const Chain& c : chn) {
            A->chain_nseeds.push_back((int)c.seeds.size()); A->chain_rid.push_back(c.rid); A->chain_w.push_back((int)c.w); A->chain_kept.push_back(c.kept);
            A->chain_pos.push_back(c.pos);
        }
        A->chain_off.push_back((int64_t)A->chain_rid.size());
        for (const AlnReg& g : regs) {
            A->reg_rb.push_back(g.rb); A->reg_re.push_back(g.re); A->reg_qb.push_back(g.qb); A->reg_qe.push_back(g.qe); A->reg_rid.push_back(g.rid);
            A->reg_score.push_back(g.score); A->reg_truesc.push_back(g.truesc); A->reg_w.push_back(g.w); A->reg_seedcov.push_back(g.seedcov);
            A->reg_seedlen0.push_back(g.seedlen0); A->reg_csub.push_back(g.csub); A->reg_secondary.push_back(g.secondary);
        }
        A->reg_off.push_back((int64_t)A->reg_rb.size());
    }
    lh_stage_dump& d = A->d;
    memset(&d, 0, sizeof d);
    d.n_reads = (int32_t)n_reads;
    d.intv_off = A->intv_off.data(); d.intv = A->intv.data();
    d.seed_off = A->seed_off.data(); d.seed_rbeg = A->seed_rbeg.data(); d.seed_qbeg = A->seed_qbeg.data(); d.seed_len = A->seed_len.data(); d.seed_rid = A->seed_rid.data();
    d.chain_off = A->chain_off.data(); d.chain_nseeds = A->chain_nseeds.data(); d.chain_rid = A->chain_rid.data(); d.chain_w = A->chain_w.data();
    d.chain_kept = A->chain_kept.data(); d.chain_pos = A->chain_pos.data();
    d.reg_off = A->reg_off.data(); d.reg_rb = A->reg_rb.data(); d.reg_re = A->reg_re.data(); d.reg_qb = A->reg_qb.data(); d.reg_qe = A->reg_qe.data();
    d.reg_rid = A->reg_rid.data(); d.reg_score = A->reg_score.data(); d.reg_truesc = A->reg_truesc.data(); d.reg_w = A->reg_w.data();
    d.reg_seedcov = A->reg_seedcov.data(); d.reg_seedlen0 = A->reg_seedlen0.data(); d.reg_csub = A->reg_csub.data(); d.reg_secondary = A->reg_secondary.data();
    d.arena_ = A;
    *out = &A->d;
    return LH_OK;
}
void lo_stage_dump_free(lh_stage_dump* d) { if (d) delete (DumpArena*)d->arena_; }

// tests/test_oracle_middle.py: mem_matesw's Smith-Waterman as mem_matesw calls it (ksw_align2 with KSW_XSUBO | KSW_XSTART | KSW_XBYTE | min_seed_len, the
// scoring of mem_opt_init) on a caller-supplied query and window (nt4 bytes); out = score, te, qe, tb, qb, score2, te2
void lo_ksw_align2(int32_t qlen, const uint8_t* query, int32_t tlen, const uint8_t* target, int32_t* out) {
    orc::MemOpt m;
    std::vector<uint8_t> q(query, query + qlen), t(target, target + tlen);
    const int xtra = 0x40000 | 0x80000 | (qlen * m.a < 250 ? 0x10000 : 0) | (m.min_seed_len * m.a);
    orc::Kswr r = orc::ksw_align2(qlen, q.data(), tlen, t.data(), 5, m.mat, m.o_del, m.e_del, m.o_ins, m.e_ins, xtra, nullptr);
    out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.tb; out[4] = r.qb; out[5] = r.score2; out[6] = r.te2;
}

// tools/rescue_probe.py: switch the per-attempt probe of mem_matesw's Smith-Waterman on (clearing its counters) or off; out (64 words) receives the counters
void lo_rescue_probe(int on, uint64_t* out) {
    if (out) for (int i = 0; i < 64; ++i) out[i] = orc::g_rescue_probe[i].load();
    if (on) for (int i = 0; i < 64; ++i) orc::g_rescue_probe[i] = 0;
    orc::g_rescue_probe_on = on;
}

}  // extern "C"

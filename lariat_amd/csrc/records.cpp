// records.cpp — N1, first half: the content of the BAM records (host only).
// Follows go/src/inference/bamwriter.go: DoDumpToBam (:634-657), AppendBam (:286-568), HardClip (:663-688), fixCigar /
// cigartable (:254-279), reverseComp / reverseQual / reverseCigar (:575-612), and lariat.go:1102-1133 (isPair).
// AppendBam edits the alignment it is given (pos = -1, mapq = 0 for an improper low-score alignment); later records of the
// same pair read those edited values, so the edits are kept in per-batch copies of pos[] and mapq[] and the records are
// produced in the reference's order.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/lariat_hip.h"
#include "records_internal.h"

extern "C" int lh_set_error_(int code, const char* msg);

namespace {

struct Ctx {
    const lh_result* r;
    const lh_ingest_batch* in;
    int32_t n_contigs;
    const char* const* names;
    std::vector<int64_t> pos;     // Alignment.pos, edited by AppendBam
    std::vector<int32_t> mapq;    // Alignment.mapq, edited by AppendBam
    std::string out;
};

std::string col(const char* base, const int64_t* off, int64_t i) { return std::string(base + off[i], (size_t)(off[i + 1] - off[i])); }

bool is_pair(const Ctx& c, int64_t a, int64_t b) {   // lariat.go:1102-1133, on the (possibly edited) positions
    const lh_result* r = c.r;
    if (r->reversed[a] == r->reversed[b] || r->rid[a] != r->rid[b]) return false;
    int64_t fwd = r->reversed[a] ? b : a, rev = r->reversed[a] ? a : b;
    int64_t dist = c.pos[rev] - c.pos[fwd];
    return dist >= -35 && dist < 750;
}

const char* contig(const Ctx& c, int64_t a) {
    int32_t rid = c.r->rid[a];
    return (rid >= 0 && rid < c.n_contigs) ? c.names[rid] : nullptr;
}

void put_int(std::string& s, long long v) {   // decimal, without the cost of snprintf (a record holds ~60 numbers)
    char b[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[n++] = '-';
    while (n) s += b[--n];
}

// what -debugBamTags reads from the molecules of a barcode (MapQData fields filled by updateAlignmentsMoleculeStatus, lariat.go:687-719,
// and estimateMapQualities, lariat.go:917-958): a molecule's number of active alignments and its confidence, taken from its active
// alignments (a molecule the optimizer emptied has none: 0 reads, and moleculeConfidence of 0 active alignments is 0)
struct BcMolecules {
    int32_t set = -1;
    bool ran = false;   // worthRunningRFA: candidate_molecules != nil
    std::vector<int32_t> reads; std::vector<double> conf;
    void load(const Ctx& c, int32_t s) {
        set = s;
        const lh_result* r = c.r;
        const lh_batch& b = c.in->batch;
        ran = b.bc_do_rfa ? b.bc_do_rfa[s] != 0 : true;
        reads.clear(); conf.clear();
        for (int64_t a = r->cand_off[2 * (int64_t)b.bc_pair_off[s]]; a < r->cand_off[2 * (int64_t)b.bc_pair_off[s + 1]]; ++a) {
            const int32_t m = r->molecule_id[a];
            if (m < 0 || !r->in_filtered[a]) continue;
            if ((size_t)m >= reads.size()) { reads.resize((size_t)m + 1, 0); conf.resize((size_t)m + 1, 0.0); }
            if (r->active[a]) { reads[(size_t)m]++; conf[(size_t)m] = r->molecule_confidence[a]; }
        }
    }
};

void put_f6(std::string& s, double v) { char b[400]; snprintf(b, sizeof b, "%.6f", v); s += b; }   // strconv.FormatFloat(v, 'f', 6, 64)
std::string op_counts(const lh_result* r, int64_t a) {
    std::string s = "Match:"; put_int(s, r->matches[a]); s += ":Mismatches:"; put_int(s, r->mismatches[a]); s += ":Indels:"; put_int(s, r->indels[a]);
    s += ":soft_clipped:"; put_int(s, r->soft_clipped[a]);
    return s;
}

// AppendBam(aln, primary, debugTags, attach_bx): fills R
void append_bam(Ctx& c, LhRec& R, int64_t read, int64_t aln, int64_t primary, bool attach_bx, const BcMolecules* dbg) {
    const lh_result* r = c.r;
    const lh_ingest_batch* in = c.in;
    const int64_t pair = read >> 1;
    const bool read1 = (read & 1) == 0;
    const char* ref = contig(c, aln);
    int flags = 0;
    if (!r->is_proper[aln] && r->score[aln] - 17 < 19) { c.pos[aln] = -1; c.mapq[aln] = 0; }
    const int64_t pm = r->mate_idx[primary];   // primary.mate_alignment
    const char* mate_ref = nullptr;
    long long mate_pos = -1, tlen = 0;
    if (pm >= 0) {   // mate_id >= 0: every read of a pair has a mate
        flags |= 1;
        if (r->is_proper[aln]) {
            if (aln == primary) flags |= 0x2;
            else if (is_pair(c, aln, pm)) flags |= 0x2;
        }
        if (c.pos[pm] == -1 || (!r->is_proper[primary] && r->score[pm] - 17 < 19)) { flags |= 0x8; mate_pos = -1; mate_ref = nullptr; }
        else {
            if (r->reversed[pm]) flags |= 0x20;
            mate_ref = contig(c, pm);
            mate_pos = c.pos[pm];
        }
        flags |= read1 ? 0x40 : 0x80;
        if (r->duplicate[aln]) flags |= 0x400;
        if (c.pos[pm] == -1) { mate_ref = nullptr; tlen = 0; }
        else if (aln == primary) {
            const int64_t ma = r->mate_idx[aln];
            if (ma >= 0 && r->rid[aln] == r->rid[ma] && (r->is_proper[primary] || r->score[pm] - 17 >= 19))
                tlen = r->reversed[aln] ? -(long long)(r->aend[aln] - c.pos[ma]) : (long long)(r->aend[ma] - c.pos[aln]);
            else tlen = 0;
        } else tlen = 0;
    }
    if (aln != primary) flags |= 256;
    int mq = c.mapq[aln] & 0xff;   // byte(aln.mapq)
    if (c.pos[aln] == -1) { flags |= 0x4; mq = 0; ref = nullptr; }
    if (r->reversed[aln]) flags |= 0x10;
    // SEQ / QUAL in read orientation, then reverse-complemented for a reversed alignment
    const int64_t* so = in->batch.seq_off;
    const uint8_t* sq = in->batch.seq + so[read];
    const int64_t slen = so[read + 1] - so[read];
    std::string seq((size_t)slen, 'N'), qual = read1 ? col(in->qual1, in->qual1_off, pair) : col(in->qual2, in->qual2_off, pair);
    for (int64_t i = 0; i < slen; ++i) seq[(size_t)i] = "ACGTN"[sq[i] > 4 ? 4 : sq[i]];
    if (r->reversed[aln]) {
        std::string t(seq.size(), 'N');
        for (size_t i = 0; i < seq.size(); ++i) { char ch = seq[seq.size() - 1 - i]; t[i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : 'N'; }
        seq.swap(t);
        std::string q2(qual.rbegin(), qual.rend());
        qual.swap(q2);
    }
    // cigar: lariat's ops 0..4 = M I D S H (fixCigar maps them to BAM's 0 1 2 4 5); HardClip for the split record
    std::vector<uint32_t> cg(r->cigar + r->cigar_off[aln], r->cigar + r->cigar_off[aln + 1]);
    std::vector<char> opch(cg.size());
    for (size_t i = 0; i < cg.size(); ++i) opch[i] = "MIDSH"[(cg[i] & 0xf) > 4 ? 4 : (cg[i] & 0xf)];
    if (primary != aln) {
        size_t start = 0, end = seq.size();
        if (cg.size() >= 1 && opch[0] == 'S') { start = cg[0] >> 4; opch[0] = 'H'; }
        if (cg.size() >= 2 && opch[cg.size() - 1] == 'S') { end -= cg[cg.size() - 1] >> 4; opch[cg.size() - 1] = 'H'; }
        if (start > seq.size()) start = seq.size();
        if (end > seq.size() || end < start) end = start;   // the reference would panic on an inconsistent cigar
        size_t qs = start < qual.size() ? start : qual.size(), qe = end < qual.size() ? end : qual.size();
        seq = seq.substr(start, end - start);
        qual = qual.substr(qs, qe > qs ? qe - qs : 0);
    }
    R.name = in->name + in->name_off[pair]; R.name_len = (size_t)(in->name_off[pair + 1] - in->name_off[pair]);
    R.flags = flags; R.mapq = mq;
    R.rid = ref ? r->rid[aln] : -1;
    R.pos = c.pos[aln];
    R.cig_len.clear(); R.cig_op.clear();
    for (size_t i = 0; i < cg.size(); ++i) { R.cig_len.push_back(cg[i] >> 4); R.cig_op.push_back(opch[i]); }
    R.mrid = mate_ref ? r->rid[pm] : -1;
    R.mpos = mate_pos; R.tlen = tlen;
    R.seq.swap(seq); R.qual.swap(qual);
    R.n_tags = 0;
    // ---- tags, in the reference's order ----
    auto tagz = [&](const char* t, const std::string& v) { R.tag(t, 'Z').z = v; };
    auto tagi = [&](const char* t, long long v) { R.tag(t, 'i').i = (int32_t)v; };
    tagz("RX", col(in->rawbc, in->rawbc_off, pair));
    tagz("QX", col(in->bcqual, in->bcqual_off, pair));
    if (read1) { tagz("TR", col(in->trim_bases, in->trim_off, pair)); tagz("TQ", col(in->trim_quals, in->trim_off, pair)); }
    std::string si = col(in->si, in->si_off, pair);
    if (si.size() > 1) { tagz("BC", si); tagz("QT", col(in->siqual, in->siqual_off, pair)); }
    std::string rg = col(in->rgid, in->rgid_off, pair);
    if (!rg.empty()) tagz("RG", rg);
    // mapq_data: the active alignment's comes from estimateMapQualities (lariat.go:948-958), a split's from split.go:154
    const bool is_split = aln != primary;
    const double xs = is_split ? r->split_second_best[read] : r->second_best_score[read];
    const double as = is_split ? r->split_score[read] : r->as_score[read];
    const int64_t sb = is_split ? -1 : r->second_best_idx[read];
    tagi("XS", (long long)xs);
    std::string xc, ac;
    if (sb >= 0)
        for (int64_t k = r->mm_off[sb]; k < r->mm_off[sb + 1]; ++k) { put_int(xc, r->mm_ref_loc[k]); xc += ','; put_int(xc, r->mm_read_loc[k]); xc += ",1;"; }
    tagz("XC", xc);
    for (int64_t k = r->mm_off[aln]; k < r->mm_off[aln + 1]; ++k) { put_int(ac, r->mm_ref_loc[k]); ac += ','; put_int(ac, r->mm_read_loc[k]); ac += ",1;"; }
    tagz("AC", ac);
    tagi("AS", (long long)as);
    tagz("XM", (sb >= 0 && r->active_molecule[sb]) ? "1" : "0");
    tagz("AM", r->active_molecule[aln] ? "1" : "0");
    tagi("XT", (sb >= 0 && r->molecule_id[aln] == r->molecule_id[sb]) ? 1 : 0);
    // SA: the split as seen from the primary, or the primary as seen from the split
    const int64_t other = is_split ? primary : r->split_idx[read];
    if (other >= 0 && c.pos[other] > -1) {
        std::vector<uint32_t> oc(r->cigar + r->cigar_off[other], r->cigar + r->cigar_off[other + 1]);
        if (r->reversed[other]) { std::vector<uint32_t> t(oc.rbegin(), oc.rend()); oc.swap(t); }
        std::string cs;
        long long indel = 0;
        for (uint32_t v : oc) {
            uint32_t op = v & 0xf, len = v >> 4;
            const char* ch = (op == 3 && !is_split) ? "H" : (op == 0 ? "M" : op == 1 ? "I" : op == 2 ? "D" : "S");
            if (op == 1 || op == 2) indel += len;
            put_int(cs, len); cs += ch;
        }
        std::string sa = contig(c, other) ? contig(c, other) : "";
        sa += ','; put_int(sa, c.pos[other]); sa += ','; sa += r->reversed[other] ? '-' : '+'; sa += ','; sa += cs; sa += ',';
        put_int(sa, c.mapq[other]); sa += ','; put_int(sa, (r->mm_off[other + 1] - r->mm_off[other]) + indel); sa += ';';
        tagz("SA", sa);
    }
    if (dbg) {   // bamwriter.go:498-558, in the order the tags are appended there (AC and XC appear a second time, as in the reference)
        auto zi = [&](const char* t, long long v) { std::string s; put_int(s, v); tagz(t, s); };
        auto zf = [&](const char* t, double v) { std::string s; put_f6(s, v); tagz(t, s); };
        long long copies = 0, in_act = 0, out_act = 0, uniq = 0, rd = 0;
        if (!is_split) {   // a split's MapQData holds the two scores only (split.go:154)
            std::vector<int32_t> seen;
            for (int64_t a = r->cand_off[read]; a < r->cand_off[read + 1]; ++a) {
                if (!r->in_filtered[a]) continue;
                ++copies;
                if (!dbg->ran) continue;
                if (r->active_molecule[a]) {
                    ++in_act;
                    bool has = false;
                    for (int32_t m : seen) has = has || m == r->molecule_id[a];
                    if (!has) seen.push_back(r->molecule_id[a]);
                } else ++out_act;
            }
            uniq = (long long)seen.size();
            if (dbg->ran && r->molecule_id[aln] >= 0) rd = dbg->reads[(size_t)r->molecule_id[aln]];
        }
        if (sb >= 0) {
            const int64_t sm = r->mate_idx[sb];
            if (sm >= 0) { zf("XM", r->log_alignment_probability[sm]); tagz("XZ", op_counts(r, sm)); }
            tagz("XX", op_counts(r, sb));
            zf("XL", r->log_alignment_probability[sb]);
            tagz("XP", r->is_proper[sb] ? "true" : "false");
            const int32_t m = r->molecule_id[sb];
            zi("XR", m >= 0 ? dbg->reads[(size_t)m] : -1);
            zf("XC", m >= 0 ? dbg->conf[(size_t)m] : -1.0);
        }
        tagz("AA", "");   // active_alignments_in_molecules is only filled under -debug (lariat.go:995)
        zi("CP", copies); zi("CM", in_act); zi("CU", uniq); zi("CS", out_act); zi("RD", rd);
        zf("MS", r->sum_move_probability_change[aln]);
        zf("MC", r->molecule_confidence[aln]);
        tagz("PP", r->is_proper[aln] ? "true" : "false");
        if (pm >= 0) { zi("PS", r->score[pm]); zf("PL", r->log_alignment_probability[pm]); }   // (the reference dereferences primary.mate_alignment unconditionally)
        tagz("AC", op_counts(r, aln));
        if (pm >= 0) tagz("PC", op_counts(r, pm));
    }
    std::string bc = col(in->bc, in->bc_off, pair);
    if (bc.find('-') != std::string::npos && attach_bx) {
        tagz("BX", bc);
        if (r->active_molecule[aln]) { char b[64]; snprintf(b, sizeof b, "%.6f", r->molecule_difference[aln]); tagz("DM", b); }
    }
}

// one tab-separated line per record with BAM-native values (lh_records_text)
void emit_text(const Ctx& c, const LhRec& R, std::string& o) {
    o.append(R.name, R.name_len); o += '\t';
    put_int(o, R.flags); o += '\t';
    o += R.rid >= 0 ? c.names[R.rid] : "*"; o += '\t';
    put_int(o, R.pos); o += '\t';
    put_int(o, R.mapq); o += '\t';
    if (R.cig_len.empty()) o += '*';
    for (size_t i = 0; i < R.cig_len.size(); ++i) { put_int(o, R.cig_len[i]); o += R.cig_op[i]; }
    o += '\t';
    o += R.mrid >= 0 ? c.names[R.mrid] : "*"; o += '\t';
    put_int(o, R.mpos); o += '\t';
    put_int(o, R.tlen); o += '\t';
    o += R.seq.empty() ? "*" : R.seq; o += '\t';
    o += R.qual.empty() ? "*" : R.qual;
    for (size_t k = 0; k < R.n_tags; ++k) {
        const LhRecTag& g = R.tags[k];
        o += '\t'; o += g.tag[0]; o += g.tag[1];
        if (g.type == 'i') { o += ":i:"; put_int(o, g.i); }
        else { o += ":Z:"; o += g.z; }
    }
    o += '\n';
}

}   // namespace

int lh_records_visit_(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, int32_t flags, int* n_threads,
                      const std::function<void(int)>& begin_thread, const std::function<void(int, const LhRec&)>& sink) {
    if (!res || !in) return lh_set_error_(LH_E_ARG, "lh_records_text: null argument");
    if (res->n_reads != 2 * in->batch.n_pairs) return lh_set_error_(LH_E_ARG, "lh_records_text: result and batch describe different reads");
    Ctx c;
    c.r = res; c.in = in; c.n_contigs = n_contigs; c.names = contig_names;
    c.pos.assign(res->pos, res->pos + res->n_cand);
    c.mapq.assign(res->mapq, res->mapq + res->n_cand);
    for (int64_t read = 0; read < res->n_reads; ++read)
        if (res->active_idx[read] < 0) return lh_set_error_(LH_E_ARG, "lh_records_text: a read has no active alignment (inference was not run?)");
    // AppendBam's edits stay inside a pair (an alignment, its mate, their splits), so pairs are independent: ranges of pairs
    // are rendered by several host threads, in order inside a range.
    const int64_t n_pairs = in->batch.n_pairs;
    int nt = n_threads && *n_threads > 0 ? *n_threads : (int)std::thread::hardware_concurrency();   // the caller's budget (one rank of several per node: cpu_count / world)
    if (nt < 1) nt = 1;
    if ((int64_t)nt > (n_pairs + 255) / 256) nt = (int)((n_pairs + 255) / 256);
    if (nt < 1) nt = 1;
    if (n_threads) *n_threads = nt;
    for (int t = 0; t < nt; ++t) begin_thread(t);
    auto work = [&](int t) {
        const int64_t p0 = n_pairs * t / nt, p1 = n_pairs * (t + 1) / nt;
        LhRec R;
        BcMolecules mol;
        const bool debug = (flags & LH_REC_DEBUG_TAGS) != 0;
        int32_t set = 0;
        for (int64_t read = 2 * p0; read < 2 * p1; ++read) {   // DoDumpToBam: reads in read_id order, the active alignment then its split
            while (set + 1 < in->n_sets && (read >> 1) >= in->batch.bc_pair_off[set + 1]) ++set;
            const bool attach_bx = in->set_complete[set] != 0;   // Data.attach_bx = WorkUnit.unique_barcode (lariat.go:493,546)
            const int64_t a = res->active_idx[read];
            if (debug && mol.set != set) mol.load(c, set);
            append_bam(c, R, read, a, a, attach_bx, debug ? &mol : nullptr);
            sink(t, R);
            if (res->split_idx[read] >= 0) { append_bam(c, R, read, res->split_idx[read], a, attach_bx, debug ? &mol : nullptr); sink(t, R); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    return LH_OK;
}

// the records of a batch as one text block per range of pairs (joined by lh_records_text)
int lh_records_parts_(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, int32_t flags, std::vector<std::string>& part) {
    Ctx names_only;
    names_only.names = contig_names;
    return lh_records_visit_(res, in, n_contigs, contig_names, flags, nullptr,
                             [&](int t) { if ((size_t)t >= part.size()) part.resize((size_t)t + 1); part[(size_t)t].reserve(1 << 20); },
                             [&](int t, const LhRec& R) { emit_text(names_only, R, part[(size_t)t]); });
}

extern "C" int lh_records_text(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, char** text, int64_t* text_len) {
    return lh_records_text_ex(res, in, n_contigs, contig_names, 0, text, text_len);
}

extern "C" int lh_records_text_ex(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, int32_t flags, char** text, int64_t* text_len) {
    if (!text || !text_len) return lh_set_error_(LH_E_ARG, "lh_records_text: null argument");
    std::vector<std::string> part;
    int rc = lh_records_parts_(res, in, n_contigs, contig_names, flags, part);
    if (rc) return rc;
    size_t total = 0;
    for (auto& o : part) total += o.size();
    char* buf = (char*)malloc(total + 1);
    if (!buf) return lh_set_error_(LH_E_ARG, "out of memory");
    size_t at = 0;
    for (auto& o : part) { memcpy(buf + at, o.data(), o.size()); at += o.size(); }
    buf[total] = 0;
    *text = buf; *text_len = (int64_t)total;
    return LH_OK;
}

extern "C" void lh_records_free(char* text) { free(text); }

// bamfile.cpp — N1, second half: BAM records and BGZF files (host only).
// Layout of the outputs follows go/src/inference/bamwriter.go: CreateBAM (:46-130 header: references, read groups, program
// line, @CO comments on first chunks), CreateBAMs (:139-191 file names and the packing of short contigs), AppendBams
// (:281-284: every record goes to bc_sorted_bam.bam and to one position bucket, ZZZ_unmapped for IsUnmapped records).
// The record fields come from records.cpp's LhRec — the same structure lh_records_text renders as text — so the text and the
// binary form cannot drift apart; the encoding itself is the SAM/BAM specification's (little-endian record, 4-bit bases, reg2bin, BGZF blocks of <= 0xff00
// bytes with the BC extra field, empty end-of-file block).
#include <zlib.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "../../include/lariat_hip.h"

extern "C" int lh_set_error_(int code, const char* msg);
#include "records_internal.h"

namespace {

const size_t BGZF_DATA = 0xff00;

void put32(std::string& s, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; s.append(b, 4); }
void put16(std::string& s, uint16_t v) { char b[2] = {(char)v, (char)(v >> 8)}; s.append(b, 2); }

// one BGZF block (a gzip member with the BC extra subfield) around `n` bytes
struct Deflater {   // one per compressing thread: deflateInit2 allocates ~256 KB, too much to repeat per 64-KB block
    z_stream zs;
    bool ok;
    explicit Deflater(int level) { memset(&zs, 0, sizeof zs); ok = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK; }
    ~Deflater() { if (ok) deflateEnd(&zs); }
};

bool bgzf_block(Deflater& d, const char* data, size_t n, std::string& out) {
    uint8_t buf[0x10000 + 64];
    if (!d.ok || deflateReset(&d.zs) != Z_OK) return false;
    z_stream& zs = d.zs;
    zs.next_in = (Bytef*)data; zs.avail_in = (uInt)n;
    zs.next_out = buf; zs.avail_out = sizeof buf;
    int rc = deflate(&zs, Z_FINISH);
    size_t clen = sizeof buf - zs.avail_out;
    if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
    const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    out.append((const char*)hdr, 12);
    out += 'B'; out += 'C'; put16(out, 2); put16(out, (uint16_t)(clen + 25));
    out.append((const char*)buf, clen);
    put32(out, (uint32_t)crc32(crc32(0, nullptr, 0), (const Bytef*)data, (uInt)n));
    put32(out, (uint32_t)n);
    return true;
}

struct Out {
    std::string path;
    FILE* f = nullptr;
    std::string pending;   // uncompressed BAM bytes not yet written
};

int reg2bin(int64_t beg, int64_t end) {   // SAM spec section 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

}   // namespace

struct lh_bam_writer {
    int32_t rec_flags = 0;   // LH_REC_*
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    std::map<std::string, int> rid_of;
    std::vector<Out> outs;                       // 0 = bc_sorted, last = unmapped
    std::vector<std::vector<int>> bucket;        // [contig][chunk] -> outs index
    int64_t chunk = 40000000;
    int level = Z_DEFAULT_COMPRESSION, threads = 1;
    double t_records = 0, t_join = 0, t_write = 0;   // the last lh_bam_append's phases, seconds
    bool failed = false;
};

namespace {

std::string header_bytes(const lh_bam_writer* w, const char* read_groups, bool comments, const char* cl) {
    std::string text = "@HD\tVN:1.5\tSO:unknown\n";
    for (size_t i = 0; i < w->names.size(); ++i) text += "@SQ\tSN:" + w->names[i] + "\tLN:" + std::to_string(w->lens[i]) + "\tAS:" + w->names[i] + "\tSP:human\n";
    std::string rgs = read_groups ? read_groups : "";
    size_t p = 0;
    while (p <= rgs.size()) {   // strings.Split(read_groups, ","): one @RG per id with >= 5 ':' fields (bamwriter.go:77-104)
        size_t q = rgs.find(',', p);
        if (q == std::string::npos) q = rgs.size();
        std::string id = rgs.substr(p, q - p);
        std::vector<std::string> f;
        size_t a = 0;
        while (a <= id.size()) { size_t b = id.find(':', a); if (b == std::string::npos) b = id.size(); f.push_back(id.substr(a, b - a)); a = b + 1; }
        if (!id.empty() && f.size() >= 5) text += "@RG\tID:" + id + "\tLB:" + f[1] + "." + f[2] + "\tPL:ILLUMINA\tPU:" + id + "\tSM:" + f[0] + "\n";
        p = q + 1;
    }
    text += std::string("@PG\tID:lariat\tPN:longranger.lariat\tCL:") + (cl ? cl : "") + "\tVN:lariat_amd\n";
    if (comments) text += "@CO\t10x_bam_to_fastq:R1(RX:QX,TR:TQ,SEQ:QUAL)\n@CO\t10x_bam_to_fastq:R2(SEQ:QUAL)\n@CO\t10x_bam_to_fastq:I1(BC:QT)\n";
    std::string h = "BAM\1";
    put32(h, (uint32_t)text.size());
    h += text;
    put32(h, (uint32_t)w->names.size());
    for (size_t i = 0; i < w->names.size(); ++i) {
        put32(h, (uint32_t)w->names[i].size() + 1);
        h += w->names[i]; h += '\0';
        put32(h, (uint32_t)w->lens[i]);
    }
    return h;
}

// one record (records.cpp) -> its BAM encoding appended to `rec`; returns the position bucket (outs index)
int encode(const lh_bam_writer* w, const LhRec& R, std::string& rec) {
    int64_t reflen = 0;
    for (size_t i = 0; i < R.cig_len.size(); ++i)
        if (R.cig_op[i] == 'M' || R.cig_op[i] == 'D' || R.cig_op[i] == 'N') reflen += R.cig_len[i];
    const size_t at = rec.size();
    put32(rec, 0);   // block_size, patched below
    put32(rec, (uint32_t)R.rid);
    put32(rec, (uint32_t)(int32_t)R.pos);
    const uint32_t lname = (uint32_t)R.name_len + 1;
    const int bin = R.pos < 0 ? 4680 : reg2bin(R.pos, R.pos + (reflen > 0 ? reflen : 1));
    put32(rec, (uint32_t)bin << 16 | (uint32_t)(R.mapq & 0xff) << 8 | (lname & 0xff));
    put32(rec, (uint32_t)R.flags << 16 | (uint32_t)(R.cig_len.size() & 0xffff));
    put32(rec, (uint32_t)R.seq.size());
    put32(rec, (uint32_t)R.mrid);
    put32(rec, (uint32_t)(int32_t)R.mpos);
    put32(rec, (uint32_t)(int32_t)R.tlen);
    rec.append(R.name, R.name_len); rec += '\0';
    for (size_t i = 0; i < R.cig_len.size(); ++i) {
        const char ch = R.cig_op[i];
        const uint32_t op = ch == 'M' ? 0 : ch == 'I' ? 1 : ch == 'D' ? 2 : ch == 'N' ? 3 : ch == 'S' ? 4 : ch == 'H' ? 5 : 6;
        put32(rec, R.cig_len[i] << 4 | op);
    }
    auto nyb = [](char ch) { switch (ch) { case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5;
                                           case 'S': return 6; case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11;
                                           case 'K': return 12; case 'D': return 13; case 'B': return 14; default: return 15; } };
    for (size_t i = 0; i < R.seq.size(); i += 2) rec += (char)(nyb(R.seq[i]) << 4 | (i + 1 < R.seq.size() ? nyb(R.seq[i + 1]) : 0));
    if (R.qual.size() == R.seq.size()) for (char ch : R.qual) rec += (char)(ch - 33);   // fixQual
    else rec.append(R.seq.size(), (char)0xff);
    for (size_t k = 0; k < R.n_tags; ++k) {
        const LhRecTag& g = R.tags[k];
        rec += g.tag[0]; rec += g.tag[1];
        if (g.type == 'i') { rec += 'i'; put32(rec, (uint32_t)g.i); }
        else { rec += 'Z'; rec += g.z; rec += '\0'; }
    }
    const uint32_t body = (uint32_t)(rec.size() - at - 4);
    rec[at] = (char)body; rec[at + 1] = (char)(body >> 8); rec[at + 2] = (char)(body >> 16); rec[at + 3] = (char)(body >> 24);
    // position bucket (AppendBams): IsUnmapped records carry pos -1 after AppendBam's edit
    if (R.pos < 0 || R.rid < 0) return (int)w->outs.size() - 1;
    size_t ch = (size_t)(R.pos / w->chunk);
    const std::vector<int>& b = w->bucket[(size_t)R.rid];
    return b[ch < b.size() ? ch : b.size() - 1];
}

// compresses and writes every complete block of every file (all of it when `all`), blocks in parallel
bool flush(lh_bam_writer* w, bool all) {
    struct Job { int out; size_t off, n; std::string z; bool ok = true; };
    std::vector<Job> jobs;
    for (size_t o = 0; o < w->outs.size(); ++o) {
        const std::string& s = w->outs[o].pending;
        size_t off = 0;
        while (s.size() - off >= BGZF_DATA || (all && off < s.size())) {
            size_t n = s.size() - off < BGZF_DATA ? s.size() - off : BGZF_DATA;
            Job j; j.out = (int)o; j.off = off; j.n = n;
            jobs.push_back(std::move(j));
            off += n;
        }
    }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Deflater d(w->level);
        for (size_t i = next++; i < jobs.size(); i = next++) jobs[i].ok = bgzf_block(d, w->outs[(size_t)jobs[i].out].pending.data() + jobs[i].off, jobs[i].n, jobs[i].z);
    };
    int nt = w->threads < 1 ? 1 : w->threads;
    if ((size_t)nt > jobs.size()) nt = (int)jobs.size();
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    // the compressed blocks go out file by file, the files side by side (jobs are grouped by file, in order): the first file, bc_sorted_bam.bam,
    // holds half of the bytes, the position buckets share the rest
    std::vector<size_t> done(w->outs.size(), 0), first_job(w->outs.size() + 1, jobs.size());
    for (size_t i = jobs.size(); i-- > 0;) first_job[(size_t)jobs[i].out] = i;
    for (size_t o = w->outs.size(); o-- > 0;) if (first_job[o] == jobs.size()) first_job[o] = first_job[o + 1];
    std::atomic<size_t> next_out{0};
    std::atomic<bool> all_ok{true};
    auto write_out = [&]() {
        for (size_t o = next_out++; o < w->outs.size(); o = next_out++) {
            Out& out = w->outs[o];
            for (size_t i = first_job[o]; i < jobs.size() && (size_t)jobs[i].out == o; ++i) {
                Job& j = jobs[i];
                if (!j.ok || fwrite(j.z.data(), 1, j.z.size(), out.f) != j.z.size()) { all_ok = false; return; }
                done[o] = j.off + j.n;
                std::string().swap(j.z);
            }
        }
    };
    {
        int wt = w->threads < 1 ? 1 : (w->threads > 16 ? 16 : w->threads);
        if ((size_t)wt > w->outs.size()) wt = (int)w->outs.size();
        std::vector<std::thread> wth;
        for (int t = 1; t < wt; ++t) wth.emplace_back(write_out);
        write_out();
        for (auto& t : wth) t.join();
    }
    if (!all_ok) return false;
    for (size_t o = 0; o < w->outs.size(); ++o) if (done[o]) w->outs[o].pending.erase(0, done[o]);
    return true;
}

}   // namespace

extern "C" int lh_bam_open(const char* dir, int32_t n_contigs, const char* const* contig_names, const int64_t* contig_lens, const char* read_groups,
                           int32_t position_chunk_size, int32_t first_chunk, const char* command_line, int32_t threads, lh_bam_writer** out) {
    if (!dir || !contig_names || !contig_lens || !out || n_contigs <= 0) return lh_set_error_(LH_E_ARG, "lh_bam_open: bad argument");
    lh_bam_writer* w = new lh_bam_writer();
    for (int i = 0; i < n_contigs; ++i) { w->names.push_back(contig_names[i]); w->lens.push_back(contig_lens[i]); w->rid_of[contig_names[i]] = i; }
    if (position_chunk_size > 0) w->chunk = position_chunk_size;
    w->threads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    std::vector<bool> co;   // which files carry the @CO lines
    auto add = [&](const std::string& name, bool comments) { Out o; o.path = std::string(dir) + "/" + name; w->outs.push_back(o); co.push_back(comments); return (int)w->outs.size() - 1; };
    add("bc_sorted_bam.bam", first_chunk != 0);
    // CreateBAMs (bamwriter.go:139-191)
    bool chr_first = first_chunk != 0;
    int last = -1;
    int64_t running = 0;
    w->bucket.resize((size_t)n_contigs);
    for (int i = 0; i < n_contigs; ++i) {
        int64_t size = w->lens[(size_t)i];
        int nchunks = (int)std::ceil((double)size / (double)w->chunk);
        char idx[16];
        snprintf(idx, sizeof idx, "%06d", i);
        if (nchunks > 1) {
            for (int c = 0; c < nchunks; ++c) {
                char offs[24];
                snprintf(offs, sizeof offs, "%010lld", (long long)c * (long long)w->chunk);
                w->bucket[(size_t)i].push_back(add(std::string(idx) + "-" + w->names[(size_t)i] + "_" + offs + "_pos_bucketed.bam", chr_first));
                chr_first = false;
            }
        } else {
            if (running == 0 || running + size > w->chunk) {
                last = add(std::string(idx) + "-" + w->names[(size_t)i] + "_0000000000_pos_bucketed.bam", chr_first);
                chr_first = false;
                running = size;
            } else running += size;
            w->bucket[(size_t)i].push_back(last);
        }
    }
    add("ZZZ_unmapped_pos_bucketed.bam", first_chunk != 0);
    for (size_t o = 0; o < w->outs.size(); ++o) {
        w->outs[o].f = fopen(w->outs[o].path.c_str(), "wb");
        if (!w->outs[o].f) {
            std::string m = "cannot create " + w->outs[o].path;
            for (Out& x : w->outs) if (x.f) fclose(x.f);
            delete w;
            return lh_set_error_(LH_E_IO, m.c_str());
        }
        w->outs[o].pending = header_bytes(w, read_groups, co[o], command_line);
    }
    *out = w;
    return LH_OK;
}

extern "C" int lh_bam_set_level(lh_bam_writer* w, int32_t level) {
    if (!w || level < -1 || level > 9) return lh_set_error_(LH_E_ARG, "lh_bam_set_level: level must be -1 (zlib's default) .. 9");
    w->level = level;
    return LH_OK;
}
extern "C" int lh_bam_timings(const lh_bam_writer* w, double* records_s, double* join_s, double* write_s) {
    if (!w) return lh_set_error_(LH_E_ARG, "lh_bam_timings: null writer");
    if (records_s) *records_s = w->t_records;
    if (join_s) *join_s = w->t_join;
    if (write_s) *write_s = w->t_write;
    return LH_OK;
}
extern "C" int lh_bam_append(lh_bam_writer* w, const lh_result* res, const lh_ingest_batch* in) {
    if (!w || !res || !in) return lh_set_error_(LH_E_ARG, "lh_bam_append: null argument");
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    std::vector<const char*> names;
    for (auto& s : w->names) names.push_back(s.c_str());
    // every range of pairs is rendered and encoded by its own host thread, straight from the result's arrays into BAM records
    // (no text form in between); per-file byte strings are joined in order
    std::vector<std::vector<std::string>> local;
    int nt = w->threads;   // in: at most this many host threads (lh_bam_open); out: how many were used
    std::atomic<int64_t> bad{-1};   // a record BAM cannot hold (the reference's writer rejects it too): the first one is reported, none is written wrong
    int rc = lh_records_visit_(res, in, (int32_t)names.size(), names.data(), w->rec_flags, &nt,
                               [&](int t) { if ((size_t)t >= local.size()) local.resize((size_t)t + 1); local[(size_t)t].assign(w->outs.size(), std::string()); },
                               [&](int t, const LhRec& R) {
                                   if (R.name_len > 254 || R.cig_len.size() > 65535) { int64_t none = -1; bad.compare_exchange_strong(none, (int64_t)R.name_len << 32 | (int64_t)(R.cig_len.size() & 0xffffffff)); return; }
                                   std::string& bc = local[(size_t)t][0];          // BarcodeSortedBam
                                   const size_t at = bc.size();
                                   const int b = encode(w, R, bc);
                                   local[(size_t)t][(size_t)b].append(bc, at, std::string::npos);   // its position bucket
                               });
    if (rc) return rc;
    if (bad.load() >= 0) {
        const std::string m = "lh_bam_append: a record does not fit the BAM format (read name of " + std::to_string(bad.load() >> 32) + " bytes: at most 254; " +
                              std::to_string(bad.load() & 0xffffffff) + " CIGAR operations: at most 65535); nothing was appended";
        return lh_set_error_(LH_E_LIMIT, m.c_str());
    }
    double t1 = now();
    double t2 = now();
    {   // per-file byte strings are joined in block order; the files are independent, so one host thread per file
        std::atomic<size_t> next_out{0};
        auto join = [&]() {
            for (size_t o = next_out++; o < w->outs.size(); o = next_out++) {
                size_t add = 0;
                for (int t = 0; t < nt; ++t) add += local[(size_t)t][o].size();
                if (!add) continue;
                std::string& pend = w->outs[o].pending;
                pend.reserve(pend.size() + add);
                for (int t = 0; t < nt; ++t) pend += local[(size_t)t][o];
            }
        };
        int jt = (int)w->outs.size() < w->threads ? (int)w->outs.size() : w->threads;
        if (jt > 16) jt = 16;
        std::vector<std::thread> th;
        for (int t = 1; t < jt; ++t) th.emplace_back(join);
        join();
        for (auto& t : th) t.join();
    }
    double t3 = now();
    if (!flush(w, false)) { w->failed = true; return lh_set_error_(LH_E_IO, "lh_bam_append: compression or write failed"); }
    w->t_records = t1 - t0; w->t_join = t3 - t2; w->t_write = now() - t3;   // lh_bam_timings
    return LH_OK;
}

extern "C" int lh_bam_set_flags(lh_bam_writer* w, int32_t flags) {
    if (!w) return lh_set_error_(LH_E_ARG, "lh_bam_set_flags: null writer");
    w->rec_flags = flags;
    return LH_OK;
}

extern "C" int lh_bam_close(lh_bam_writer* w) {
    if (!w) return LH_OK;
    bool ok = !w->failed && flush(w, true);
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (Out& o : w->outs)
        if (o.f) {
            if (fwrite(eof_block, 1, 28, o.f) != 28) ok = false;
            if (fclose(o.f)) ok = false;
        }
    delete w;
    return ok ? LH_OK : lh_set_error_(LH_E_IO, "lh_bam_close: write failed");
}

// ------------------------------------------------------------------------------------------------------------------
// lh_bam_concat — the one host-side step of the multi-GPU path (BASELINE.json north_star: barcodes partition across the GPUs by
// barcode range, "only a host-side concat of per-GPU BAM shards"; reference: one process writes the file set of
// bamwriter.go:139-191).  Every rank writes its own file set with lh_bam_open/_append/_close; because the input is
// barcode-sorted and the ranks own contiguous barcode ranges, concatenating bc_sorted_bam.bam in rank order IS the
// single-process file, and a position bucket of the whole job is the concatenation of the ranks' buckets (records inside a
// bucket are unordered until the downstream sort, as upstream).  BGZF makes this a block copy: shard 0 is copied whole,
// later shards lose their header (the block the header ends in is re-compressed from the first record on), every
// end-of-file block but the last is dropped.
namespace {
struct BgzfBlock { size_t off, size, isize; };

bool bgzf_scan(const std::string& file, std::vector<BgzfBlock>& blocks) {
    size_t p = 0;
    while (p < file.size()) {
        if (file.size() - p < 18) return false;
        const uint8_t* h = (const uint8_t*)file.data() + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        size_t xlen = h[10] | (size_t)h[11] << 8;
        if (file.size() - p < 12 + xlen) return false;
        size_t bsize = 0;
        for (size_t q = 12; q + 4 <= 12 + xlen;) {
            size_t slen = h[q + 2] | (size_t)h[q + 3] << 8;
            if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) bsize = (h[q + 4] | (size_t)h[q + 5] << 8) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || file.size() - p < bsize) return false;
        const uint8_t* t = h + bsize - 4;
        blocks.push_back(BgzfBlock{p, bsize, (size_t)t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24});
        p += bsize;
    }
    return true;
}

bool bgzf_inflate(const std::string& file, const BgzfBlock& b, std::string& out) {
    const uint8_t* h = (const uint8_t*)file.data() + b.off;
    size_t xlen = h[10] | (size_t)h[11] << 8;
    out.resize(b.isize);
    if (!b.isize) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = (uInt)(b.size - 12 - xlen - 8);
    zs.next_out = (Bytef*)&out[0]; zs.avail_out = (uInt)b.isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

bool slurp_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    bool ok = !ferror(f);
    fclose(f);
    return ok;
}

// length of the BAM header (magic, text, reference table) at the start of an uncompressed stream; 0 if `s` does not hold all of it yet
size_t bam_header_len(const std::string& s, uint32_t* n_ref_out) {
    if (s.size() < 12 || memcmp(s.data(), "BAM\1", 4)) return 0;
    auto u32 = [&](size_t o) { const uint8_t* q = (const uint8_t*)s.data() + o; return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; };
    size_t p = 8 + u32(4);
    if (s.size() < p + 4) return 0;
    uint32_t n_ref = u32(p);
    p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (s.size() < p + 4) return 0;
        p += 4 + u32(p) + 4;
        if (s.size() < p) return 0;
    }
    if (n_ref_out) *n_ref_out = n_ref;
    return p;
}
}   // namespace

#include <dirent.h>

extern "C" int lh_bam_concat(int32_t n_shards, const char* const* shard_dirs, const char* out_dir) {
    if (n_shards <= 0 || !shard_dirs || !out_dir) return lh_set_error_(LH_E_ARG, "lh_bam_concat: bad argument");
    std::vector<std::string> files;
    {
        DIR* d = opendir(shard_dirs[0]);
        if (!d) return lh_set_error_(LH_E_IO, (std::string("lh_bam_concat: cannot list ") + shard_dirs[0]).c_str());
        while (dirent* e = readdir(d)) {
            std::string n = e->d_name;
            if (n.size() > 4 && n.compare(n.size() - 4, 4, ".bam") == 0) files.push_back(n);
        }
        closedir(d);
    }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (const std::string& name : files) {
        FILE* out = fopen((std::string(out_dir) + "/" + name).c_str(), "wb");
        if (!out) return lh_set_error_(LH_E_IO, ("lh_bam_concat: cannot create " + std::string(out_dir) + "/" + name).c_str());
        uint32_t n_ref0 = 0;
        bool ok = true;
        std::string err;
        for (int s = 0; s < n_shards && ok; ++s) {
            std::string file, path = std::string(shard_dirs[s]) + "/" + name;
            std::vector<BgzfBlock> blocks;
            if (!slurp_file(path, file) || !bgzf_scan(file, blocks)) { ok = false; err = "cannot read BGZF file " + path; break; }
            while (!blocks.empty() && blocks.back().isize == 0) blocks.pop_back();   // end-of-file marker(s)
            // the header: inflate blocks until it is complete
            std::string head, blk;
            size_t k = 0, hlen = 0;
            uint32_t n_ref = 0;
            while (k < blocks.size() && !(hlen = bam_header_len(head, &n_ref))) {
                if (!bgzf_inflate(file, blocks[k], blk)) { ok = false; err = "corrupt BGZF block in " + path; break; }
                head += blk;
                ++k;
            }
            if (!ok) break;
            if (!hlen && !(hlen = bam_header_len(head, &n_ref))) { ok = false; err = "no BAM header in " + path; break; }
            if (s == 0) n_ref0 = n_ref;
            else if (n_ref != n_ref0) { ok = false; err = "shards disagree on the reference table: " + path; break; }
            if (s == 0) {   // whole, as written
                size_t end = blocks.empty() ? 0 : blocks.back().off + blocks.back().size;
                if (end && fwrite(file.data(), 1, end, out) != end) { ok = false; err = "write failed"; }
            } else {
                if (head.size() > hlen) {   // records that share the header's last block: their own block now
                    Deflater d(Z_DEFAULT_COMPRESSION);
                    for (size_t o = hlen; o < head.size() && ok; o += BGZF_DATA) {
                        std::string z;
                        size_t n = head.size() - o < BGZF_DATA ? head.size() - o : BGZF_DATA;
                        if (!bgzf_block(d, head.data() + o, n, z) || fwrite(z.data(), 1, z.size(), out) != z.size()) { ok = false; err = "compression or write failed"; }
                    }
                }
                if (ok && k < blocks.size()) {
                    size_t b0 = blocks[k].off, b1 = blocks.back().off + blocks.back().size;
                    if (fwrite(file.data() + b0, 1, b1 - b0, out) != b1 - b0) { ok = false; err = "write failed"; }
                }
            }
        }
        if (ok && fwrite(eof_block, 1, 28, out) != 28) { ok = false; err = "write failed"; }
        if (fclose(out)) { ok = false; err = "write failed"; }
        if (!ok) return lh_set_error_(LH_E_IO, ("lh_bam_concat: " + err).c_str());
    }
    return LH_OK;
}

// ingest.cpp — N2: the 9-line barcode-sorted FASTQ reader (host only).
// Follows go/src/fastqreader/reader.go: ReadOneLine (:91-147) and ReadBarcodeSet (:173-260), zipread.go:62-85 (the gunzip
// pipe), and the use the driver makes of a set: lariat.go:353-375 (read loop), :1088-1100 (worthRunningRFA),
// :1483-1484 (tie-break seed = LE u64 of md5(read name)[0:8]), gobwa.go:159-167 (SequenceConvert).
//
// One pass over the bytes: the file (or the gunzip pipe) is read in 4 MiB blocks, lines are located with memchr and
// copied once into per-batch arenas in the structure-of-arrays shape lh_batch wants, so a batch can go straight to
// lh_batch_upload.  The set boundaries reproduce ReadBarcodeSet turn by turn, including its deferred record, its
// deferred error and the off-by-one it applies to the record that triggered a break.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lariat_hip.h"

extern "C" int lh_set_error_(int code, const char* msg);

namespace {

// ---- md5 (RFC 1321), written out for the 8 bytes lariat takes from it ----
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    static uint32_t rol(uint32_t x, int s) { return x << s | x >> (32 - s); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            uint32_t t = D;
            D = C; C = B;
            B = B + rol(A + f + K[i] + w[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    uint64_t first8(const uint8_t* msg, size_t n) {
        size_t i = 0;
        for (; i + 64 <= n; i += 64) block(msg + i);
        uint8_t tail[128];
        size_t r = n - i;
        memcpy(tail, msg + i, r);
        tail[r++] = 0x80;
        size_t pad = r <= 56 ? 56 : 120;
        memset(tail + r, 0, pad - r);
        uint64_t bits = (uint64_t)n * 8;
        for (int k = 0; k < 8; ++k) tail[pad + k] = (uint8_t)(bits >> (8 * k));
        block(tail);
        if (pad == 120) block(tail + 64);
        return (uint64_t)a | (uint64_t)b << 32;   // digest bytes 0..7, little endian
    }
};

struct Nt4 {
    uint8_t t[256];
    Nt4() {
        memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
    }
};
const Nt4 g_nt4;

struct Text {   // one text column of a batch: bytes + offsets
    std::vector<char> bytes;
    std::vector<int64_t> off{0};
    void add(const char* p, size_t n) { bytes.insert(bytes.end(), p, p + n); off.push_back((int64_t)bytes.size()); }
    void pop() { off.pop_back(); bytes.resize((size_t)off.back()); }
    void clear() { bytes.clear(); off.assign(1, 0); }
};

struct Arena;
struct Pool {   // arenas are recycled with their capacity: growing fresh 100-MB vectors costs more (page faults) than parsing
    std::mutex mu;
    std::vector<Arena*> free_;
    ~Pool();
};

struct Arena {   // owns everything an lh_ingest_batch points to
    lh_ingest_batch b;
    std::shared_ptr<Pool> pool;
    std::vector<int32_t> bc_pair_off{0};
    std::vector<uint8_t> do_rfa, complete, seq;
    std::vector<int64_t> seq_off{0};
    std::vector<uint64_t> name_seed;
    Text name, rgid, qual1, qual2, trimb, trimq, bc, rawbc, bcqual, si, siqual;
    int64_t n_pairs() const { return (int64_t)name_seed.size(); }
    void add_seq(const char* p, size_t n) {
        size_t o = seq.size();
        seq.resize(o + n);
        for (size_t i = 0; i < n; ++i) seq[o + i] = g_nt4.t[(uint8_t)p[i]];
        seq_off.push_back((int64_t)seq.size());
    }
    void reset() {
        bc_pair_off.assign(1, 0); do_rfa.clear(); complete.clear(); seq.clear(); seq_off.assign(1, 0); name_seed.clear();
        Text* t[] = {&name, &rgid, &qual1, &qual2, &trimb, &trimq, &bc, &rawbc, &bcqual, &si, &siqual};
        for (Text* x : t) x->clear();
    }
    void pop_pair() {   // removes the record appended last (ReadBarcodeSet's `end -= 1`)
        seq_off.pop_back(); seq_off.pop_back(); seq.resize((size_t)seq_off.back());
        name_seed.pop_back();
        name.pop(); rgid.pop(); qual1.pop(); qual2.pop(); trimb.pop(); trimq.pop(); bc.pop(); rawbc.pop(); bcqual.pop(); si.pop(); siqual.pop();
    }
};

struct View {
    const char* p = nullptr;
    size_t n = 0;
    bool operator==(const View& o) const { return n == o.n && (n == 0 || memcmp(p, o.p, n) == 0); }
    bool operator!=(const View& o) const { return !(*this == o); }
};
Pool::~Pool() { for (Arena* a : free_) delete a; }

Arena* arena_get(const std::shared_ptr<Pool>& pool) {
    Arena* a = nullptr;
    {
        std::lock_guard<std::mutex> g(pool->mu);
        if (!pool->free_.empty()) { a = pool->free_.back(); pool->free_.pop_back(); }
    }
    if (!a) a = new Arena();
    a->pool = pool;
    return a;
}
void arena_put(Arena* a) {
    std::shared_ptr<Pool> pool = a->pool;
    a->pool.reset();
    if (!pool) { delete a; return; }
    a->reset();
    std::lock_guard<std::mutex> g(pool->mu);
    if (pool->free_.size() < 8) pool->free_.push_back(a); else delete a;
}

struct Record {   // views into the read buffer (valid until the next refill) or, for the deferred record, into `own`
    View name, rgid, r1, q1, r2, q2, tb, tq, bc, rawbc, bcq, si, siq;
    std::string own;
    void keep() {   // ReadBarcodeSet's Pending outlives the buffer contents: give it its own bytes
        View* f[] = {&name, &rgid, &r1, &q1, &r2, &q2, &tb, &tq, &bc, &rawbc, &bcq, &si, &siq};
        std::string o;
        size_t off[13];
        for (int i = 0; i < 13; ++i) { off[i] = o.size(); o.append(f[i]->p ? f[i]->p : "", f[i]->n); }
        own.swap(o);
        for (int i = 0; i < 13; ++i) f[i]->p = own.data() + off[i];
    }
};

}   // namespace

struct lh_ingest {
    FILE* f = nullptr;
    bool piped = false, pipe_failed = false;
    int trim = 0, cap = 30000, chunk = 200;
    std::vector<char> buf;
    size_t pos = 0, lim = 0;
    bool eof = false;
    int64_t line_no = 0, sets_done = 0;
    // ReadBarcodeSet state
    bool have_pending = false;
    Record pending;
    int deferred = 0;   // 0 none, 1 io.EOF, 2 other error
    bool have_last = false;
    std::string last_bc;
    // a whole set parsed ahead of a batch boundary
    Arena* carry = nullptr;
    std::shared_ptr<Pool> pool = std::make_shared<Pool>();

    // bufio.Reader.ReadString('\n') as a view: the bytes up to and including '\n'; at end of input the remainder comes back
    // with EOF (the callers treat that as an error and drop it).  Returns 1 = line with '\n', 0 = EOF (partial line in out),
    // -1 = read error.  Views stay valid until the next call that has to refill, which is why a record is parsed from
    // `mark`: everything from `mark` on is kept (moved to the front) when the buffer is refilled.
    size_t mark = 0;
    bool moved = false;
    int read_line(View& out) {
        for (;;) {
            if (pos < lim) {
                const char* p = buf.data() + pos;
                const char* nl = (const char*)memchr(p, '\n', lim - pos);
                if (nl) { out.p = p; out.n = (size_t)(nl - p) + 1; pos += out.n; return 1; }
            }
            if (eof) { out.p = buf.data() + pos; out.n = lim - pos; pos = lim; return 0; }
            // refill, keeping [mark, lim)
            size_t keep = lim - mark;
            if (buf.size() < keep + (4u << 20)) buf.resize(keep + (4u << 20));
            if (mark) memmove(buf.data(), buf.data() + mark, keep);
            pos -= mark; lim = keep; mark = 0; moved = true;
            size_t n = fread(buf.data() + lim, 1, buf.size() - lim, f);
            lim += n;
            if (n == 0) {
                eof = true;
                if (ferror(f)) return -1;
                if (piped) {   // a corrupt or truncated .gz makes gunzip exit non-zero: not a clean end of input
                    int st = pclose(f);
                    f = nullptr;
                    if (st != 0) { pipe_failed = true; return -1; }
                }
            }
        }
    }
};

namespace {

// ReadOneLine: 0 ok, 1 io.EOF, 2 other error
int read_one(lh_ingest* in, Record& r) {
    View line;
    for (;;) {   // search for the next start-of-record
        in->line_no++;
        in->mark = in->pos;
        int st = in->read_line(line);
        if (st <= 0) return st == 0 ? 1 : 2;
        if (line.p[0] == '@') break;
        // "Bad line": skipped
    }
    for (;;) {   // the header and the 8 lines after it, as views into one buffer generation
        in->moved = false;
        size_t start = in->mark;
        in->pos = start;
        View got[9];
        int st = 1, i = 0;
        for (; i < 9; ++i) {
            st = in->read_line(got[i]);
            if (in->moved) break;   // the buffer was refilled: earlier views are stale, parse the record again from its start
            if (i > 0) {
                if (got[i].n == 0) return st == 0 ? 1 : 2;   // line[0:len-1] of an empty slice panics upstream
                got[i].n--;                                   // stuff_to_get[i] = line[0 : len(line)-1], assigned before the error check
            }
            if (st <= 0) return st == 0 ? 1 : 2;
        }
        if (i < 9) continue;
        // strings.Fields(line[1 : len-1]): split on white space; ReadInfo = first field, ReadGroupId = last if >= 2
        const char* p = got[0].p + 1;
        const char* e = got[0].p + got[0].n - 1;
        View first, last;
        int nf = 0;
        while (p < e) {
            while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) ++p;
            if (p >= e) break;
            const char* s0 = p;
            while (p < e && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) ++p;
            last.p = s0; last.n = (size_t)(p - s0);
            if (nf++ == 0) first = last;
        }
        if (nf == 0) return 2;   // the reference indexes fields[0] and panics; treated as a read error
        r.name = first;
        r.rgid = nf < 2 ? View() : last;
        const View* g = got + 1;
        size_t to_trim = g[0].n < (size_t)in->trim ? g[0].n : (size_t)in->trim;
        // the quality line is sliced with the same count (reader.go:135-139); a shorter quality line makes the reference panic
        // (slice bounds): a read error here, so that trim_bases and trim_quals always share their offsets
        if (g[1].n < to_trim) return 2;
        size_t tq = to_trim;
        r.r1 = View{g[0].p + to_trim, g[0].n - to_trim};
        r.q1 = View{g[1].p + tq, g[1].n - tq};
        r.tb = View{g[0].p, to_trim};
        r.tq = View{g[1].p, tq};
        r.r2 = g[2]; r.q2 = g[3];
        const char* c0 = (const char*)memchr(g[4].p, ',', g[4].n);
        if (!c0) { r.bc = g[4]; r.rawbc = g[4]; }
        else {
            const char* cl = g[4].p + g[4].n;
            while (cl > g[4].p && cl[-1] != ',') --cl;
            r.bc = View{g[4].p, (size_t)(c0 - g[4].p)};
            r.rawbc = View{cl, (size_t)(g[4].p + g[4].n - cl)};
        }
        r.bcq = g[5]; r.si = g[6]; r.siq = g[7];
        in->mark = in->pos;
        return 0;
    }
}

void append(Arena& A, const Record& r) {
    A.add_seq(r.r1.p, r.r1.n);
    A.add_seq(r.r2.p, r.r2.n);
    Md5 m;
    A.name_seed.push_back(m.first8((const uint8_t*)r.name.p, r.name.n));
    A.name.add(r.name.p, r.name.n); A.rgid.add(r.rgid.p, r.rgid.n);
    A.qual1.add(r.q1.p, r.q1.n); A.qual2.add(r.q2.p, r.q2.n);
    A.trimb.add(r.tb.p, r.tb.n); A.trimq.add(r.tq.p, r.tq.n);
    A.bc.add(r.bc.p, r.bc.n); A.rawbc.add(r.rawbc.p, r.rawbc.n); A.bcqual.add(r.bcq.p, r.bcq.n);
    A.si.add(r.si.p, r.si.n); A.siqual.add(r.siq.p, r.siq.n);
}

// ReadBarcodeSet: appends one set to A.  Returns 0 = a set was appended, 1 = io.EOF, 2 = error (nothing appended).
int read_set(lh_ingest* in, Arena& A) {
    if (in->deferred) return in->deferred;
    const int64_t first_pair = A.n_pairs();
    bool new_barcode = false;
    int index = 0;
    if (in->have_pending) { append(A, in->pending); in->have_pending = false; index++; }
    Record rec;
    std::string first_bc;
    bool first_wl = true;
    auto first_barcode = [&]() {   // record_array[0].Barcode10X and NotWhitelist(&record_array[0])
        int64_t o0 = A.bc.off[(size_t)first_pair], o1 = A.bc.off[(size_t)first_pair + 1];
        first_bc.assign(A.bc.bytes.data() + o0, (size_t)(o1 - o0));
        first_wl = first_bc.find('-') != std::string::npos;
    };
    if (index == 1) first_barcode();
    for (; index < in->cap; index++) {
        int err = read_one(in, rec);
        // the reference appends an empty record BEFORE reading into it: on error it is part of the array until the truncation below
        if (err) {
            if (index == 0) return err;
            in->deferred = err;
            // placeholder for the ill-defined record, removed (io.EOF) or kept (other errors) by the truncation rule
            Record empty;
            append(A, empty);
            break;
        }
        append(A, rec);
        if (index == 0) first_barcode();
        bool different = !(rec.bc.n == first_bc.size() && memcmp(rec.bc.p, first_bc.data(), rec.bc.n) == 0);
        if (different || (!first_wl && index >= in->chunk)) {
            in->pending = rec; in->pending.keep(); in->have_pending = true;   // belongs to the next set
            new_barcode = true;
            break;
        } else if (in->have_last && first_bc == in->last_bc && index >= in->chunk) {
            new_barcode = false;   // "abnormal break": continuation of a barcode that was cut before
            break;
        }
    }
    if (A.n_pairs() > first_pair) { in->last_bc = first_bc; in->have_last = true; }
    bool complete;
    if (new_barcode || in->deferred == 1) { A.pop_pair(); complete = true; }
    else complete = false;
    int64_t n = A.n_pairs() - first_pair;
    A.bc_pair_off.push_back((int32_t)A.n_pairs());
    A.complete.push_back(complete ? 1 : 0);
    // worthRunningRFA (lariat.go:1088-1100): a complete barcode with a '-' and at least 5 pairs
    A.do_rfa.push_back((complete && n >= 5 && first_wl) ? 1 : 0);
    in->sets_done++;
    return 0;
}

void publish(Arena* A, int64_t first_set, bool at_eof) {
    lh_ingest_batch& b = A->b;
    memset(&b, 0, sizeof b);
    b.batch.n_barcodes = (int32_t)A->do_rfa.size();
    b.batch.n_pairs = (int32_t)A->n_pairs();
    b.batch.bc_pair_off = A->bc_pair_off.data(); b.batch.bc_do_rfa = A->do_rfa.data();
    b.batch.seq_off = A->seq_off.data();
    if (A->seq.empty()) A->seq.push_back(0);
    b.batch.seq = A->seq.data(); b.batch.name_seed = A->name_seed.data();
    b.n_sets = b.batch.n_barcodes; b.set_complete = A->complete.data();
#define LH_TXT(field, col) b.field##_off = A->col.off.data(); b.field = A->col.bytes.data()
    LH_TXT(name, name); LH_TXT(rgid, rgid); LH_TXT(qual1, qual1); LH_TXT(qual2, qual2); LH_TXT(bc, bc); LH_TXT(rawbc, rawbc); LH_TXT(bcqual, bcqual);
    LH_TXT(si, si); LH_TXT(siqual, siqual);
#undef LH_TXT
    b.trim_off = A->trimb.off.data(); b.trim_bases = A->trimb.bytes.data(); b.trim_quals = A->trimq.bytes.data();
    b.first_set_index = first_set; b.at_eof = at_eof ? 1 : 0;
    b.arena_ = A;
}

}   // namespace

extern "C" uint64_t lh_name_seed(const char* name, int64_t n) {
    Md5 m;
    return m.first8((const uint8_t*)name, (size_t)(n < 0 ? 0 : n));
}

extern "C" int lh_ingest_open(const char* path, int32_t trim, int32_t cap, int32_t chunk, lh_ingest** out) {
    if (!path || !out) return lh_set_error_(LH_E_ARG, "lh_ingest_open: null argument");
    FILE* probe = fopen(path, "rb");
    if (!probe) return lh_set_error_(LH_E_IO, (std::string("cannot open ") + path).c_str());
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, probe);
    lh_ingest* in = new lh_ingest();
    in->trim = trim < 0 ? 0 : trim;
    if (cap > 0) in->cap = cap;
    if (chunk > 0) in->chunk = chunk;
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {   // zipread.go:62-85: the system gunzip is the decompressor
        fclose(probe);
        std::string q = "gunzip -c '";
        for (const char* p = path; *p; ++p) { if (*p == '\'') q += "'\\''"; else q += *p; }
        q += "'";
        in->f = popen(q.c_str(), "r");
        in->piped = true;
        if (!in->f) { delete in; return lh_set_error_(LH_E_IO, "cannot start gunzip"); }
    } else {
        rewind(probe);
        in->f = probe;
    }
    *out = in;
    return LH_OK;
}

extern "C" int lh_ingest_next(lh_ingest* in, int64_t max_pairs, lh_ingest_batch** out) {
    if (!in || !out) return lh_set_error_(LH_E_ARG, "lh_ingest_next: null argument");
    Arena* A = in->carry ? in->carry : arena_get(in->pool);
    in->carry = nullptr;
    int64_t first_set = in->sets_done - (int64_t)A->do_rfa.size();
    bool at_eof = false;
    for (;;) {
        if (A->n_pairs() >= max_pairs && !A->do_rfa.empty()) break;
        // parse the next set into its own arena when the batch already holds something, so that it can be carried over whole
        Arena* T = A->do_rfa.empty() ? A : arena_get(in->pool);
        int rc = read_set(in, *T);
        if (rc) {
            if (T != A) arena_put(T);
            if (in->pipe_failed) { arena_put(A); return lh_set_error_(LH_E_IO, "gunzip failed: the compressed FASTQ is corrupt or truncated"); }
            if (rc == 2 && A->do_rfa.empty()) { arena_put(A); return lh_set_error_(LH_E_IO, "read error in the FASTQ stream"); }
            at_eof = true;
            break;
        }
        if (T == A) continue;
        if (A->n_pairs() + T->n_pairs() > max_pairs) { in->carry = T; break; }
        // merge T (exactly one set) into A
        const int64_t base_pairs = A->n_pairs();
        const int64_t sbase = (int64_t)A->seq.size();
        A->seq.insert(A->seq.end(), T->seq.begin(), T->seq.end());
        for (size_t i = 1; i < T->seq_off.size(); ++i) A->seq_off.push_back(sbase + T->seq_off[i]);
        A->name_seed.insert(A->name_seed.end(), T->name_seed.begin(), T->name_seed.end());
        A->bc_pair_off.push_back((int32_t)(base_pairs + T->n_pairs()));
        A->do_rfa.push_back(T->do_rfa[0]); A->complete.push_back(T->complete[0]);
        Text* dst[] = {&A->name, &A->rgid, &A->qual1, &A->qual2, &A->trimb, &A->trimq, &A->bc, &A->rawbc, &A->bcqual, &A->si, &A->siqual};
        Text* src[] = {&T->name, &T->rgid, &T->qual1, &T->qual2, &T->trimb, &T->trimq, &T->bc, &T->rawbc, &T->bcqual, &T->si, &T->siqual};
        for (int k = 0; k < 11; ++k) {
            int64_t b0 = (int64_t)dst[k]->bytes.size();
            dst[k]->bytes.insert(dst[k]->bytes.end(), src[k]->bytes.begin(), src[k]->bytes.end());
            for (size_t i = 1; i < src[k]->off.size(); ++i) dst[k]->off.push_back(b0 + src[k]->off[i]);
        }
        arena_put(T);
    }
    publish(A, first_set, at_eof && !in->carry);
    *out = &A->b;
    return LH_OK;
}

extern "C" void lh_ingest_batch_free(lh_ingest_batch* b) {
    if (b) arena_put((Arena*)b->arena_);
}

extern "C" void lh_ingest_close(lh_ingest* in) {
    if (!in) return;
    if (in->f) { if (in->piped) pclose(in->f); else fclose(in->f); }
    if (in->carry) { in->carry->pool.reset(); delete in->carry; }
    delete in;
}

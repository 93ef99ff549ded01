// ORACLE — TEST INFRASTRUCTURE ONLY (see bwa_oracle.h).
// Restatement of BWA-MEM 0.7.17's single-end candidate generation, mate rescue and
// reg->aln conversion: the routines lariat reaches through cgo at
//   go/src/gobwa/gobwa.go:181,244,253 (mem_align1_core), :291,315 (mem_matesw),
//   :404 (mem_reg2aln), :59 (bns_fetch_seq); prototypes go/src/gobwa/bwa_bridge.h:35-39.
// Source of truth is absent from /root/reference (empty submodule go/src/gobwa/bwa);
// function names below are upstream's (bwt.c, bwamem.c, bwa.c, ksw.c).
#include "bwa_oracle.h"
#include <atomic>
#include <cstdlib>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace orc {

const uint8_t nst_nt4_table[256] = {
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5 /*'-'*/, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

MemOpt::MemOpt() {   // mem_opt_init + bwa_fill_scmat
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b;
        mat[k++] = -1;   // ambiguous base
    }
    for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

// ---------------------------------------------------------------- bwt.c
static inline uint32_t occ_aux4(uint32_t w) {   // packed per-base counts of the 16 symbols in w: A|C<<8|G<<16|T<<24
    uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
    uint32_t t = __builtin_popcount(hi & lo), g = __builtin_popcount(hi & ~lo), cc = __builtin_popcount(~hi & lo);
    uint32_t a = 16 - t - g - cc;
    return a | cc << 8 | g << 16 | t << 24;
}

void bwt_occ4(const Index& b, bwtint_t k, bwtint_t cnt[4]) {
    if (k == (bwtint_t)(-1)) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= b.primary);   // because $ is not in bwt
    const uint32_t* p = b.bwt.data() + ((k >> 7) << 4);
    memcpy(cnt, p, 32);
    p += 8;
    const uint32_t* end = p + ((k >> 4) - ((k & ~(bwtint_t)127) >> 4));
    bwtint_t x = 0;
    for (; p < end; ++p) x += occ_aux4(*p);
    uint32_t tmp = *p & ~((1U << ((~k & 15) << 1)) - 1);
    x += occ_aux4(tmp) - (~k & 15);
    cnt[0] += x & 0xff; cnt[1] += x >> 8 & 0xff; cnt[2] += x >> 16 & 0xff; cnt[3] += x >> 24;
}

void bwt_2occ4(const Index& b, bwtint_t k, bwtint_t l, bwtint_t cntk[4], bwtint_t cntl[4]) {
    bwt_occ4(b, k, cntk);   // upstream has a same-block fast path; results are identical
    bwt_occ4(b, l, cntl);
}

static inline bwtint_t bwt_occ(const Index& b, bwtint_t k, int c) {
    bwtint_t cnt[4];
    bwt_occ4(b, k, cnt);
    return cnt[c];
}

void bwt_extend(const Index& b, const Intv& ik, Intv ok[4], int is_back, Counters* c) {
    bwtint_t tk[4], tl[4];
    if (c) ++c->n_ext;
    bwt_2occ4(b, ik.x[!is_back] - 1, ik.x[!is_back] - 1 + ik.x[2], tk, tl);
    for (int i = 0; i != 4; ++i) {
        ok[i].x[!is_back] = b.L2[i] + 1 + tk[i];
        ok[i].x[2] = tl[i] - tk[i];
    }
    ok[3].x[is_back] = ik.x[is_back] + (ik.x[!is_back] <= b.primary && ik.x[!is_back] + ik.x[2] - 1 >= b.primary);
    ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
    ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
    ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

static inline bwtint_t bwt_invPsi(const Index& b, bwtint_t k) {
    bwtint_t x = k - (k > b.primary);
    x = b.bwt[((x >> 7) << 4) + 8 + ((x & 0x7f) >> 4)] >> ((~x & 0xf) << 1) & 3;   // bwt_B0
    x = b.L2[x] + bwt_occ(b, k, (int)x);
    return k == b.primary ? 0 : x;
}

bwtint_t bwt_sa(const Index& b, bwtint_t k, Counters* c) {
    bwtint_t sa = 0, mask = b.sa_intv - 1;
    while (k & mask) {
        ++sa;
        k = bwt_invPsi(b, k);
        if (c) ++c->n_lf;
    }
    if (c) ++c->n_sa;
    return sa + b.sa[k / b.sa_intv];
}

static inline void bwt_set_intv(const Index& b, int c, Intv& ik) {
    ik.x[0] = b.L2[c] + 1; ik.x[2] = b.L2[c + 1] - b.L2[c]; ik.x[1] = b.L2[3 - c] + 1; ik.info = 0;
}

// bwt_smem1a with max_intv = 0 (the only way mem_collect_intv calls it)
int bwt_smem1(const Index& b, int len, const uint8_t* q, int x, int min_intv, std::vector<Intv>& mem, Counters* cn) {
    int i, c, ret;
    Intv ik, ok[4];
    std::vector<Intv> va, vb;
    std::vector<Intv>*prev = &va, *curr = &vb;
    mem.clear();
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    bwt_set_intv(b, q[x], ik);
    ik.info = x + 1;
    for (i = x + 1, curr->clear(); i < len; ++i) {   // forward search
        if (q[i] < 4) {
            c = 3 - q[i];
            bwt_extend(b, ik, ok, 0, cn);
            if (ok[c].x[2] != ik.x[2]) {   // change of the interval size
                curr->push_back(ik);
                if (ok[c].x[2] < (bwtint_t)min_intv) break;   // too small to be extended further
            }
            ik = ok[c]; ik.info = i + 1;
        } else {   // an ambiguous base
            curr->push_back(ik);
            break;
        }
    }
    if (i == len) curr->push_back(ik);
    std::reverse(curr->begin(), curr->end());   // longer matches first
    ret = (int)(*curr)[0].info;
    std::swap(curr, prev);
    for (i = x - 1; i >= -1; --i) {   // backward search for MEMs
        c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
        curr->clear();
        for (size_t j = 0; j < prev->size(); ++j) {
            const Intv* p = &(*prev)[j];
            if (c >= 0) bwt_extend(b, *p, ok, 1, cn);
            if (c < 0 || ok[c].x[2] < (bwtint_t)min_intv) {   // cannot be extended
                if (curr->empty()) {   // no longer match survived this round
                    if (mem.empty() || (uint64_t)(i + 1) < mem.back().info >> 32) {   // skip contained matches
                        ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
                        mem.push_back(ik);
                    }
                }
            } else if (curr->empty() || ok[c].x[2] != curr->back().x[2]) {
                ok[c].info = p->info;
                curr->push_back(ok[c]);
            }
        }
        if (curr->empty()) break;
        std::swap(curr, prev);
    }
    std::reverse(mem.begin(), mem.end());   // sorted by the start coordinate
    return ret;
}

int bwt_seed_strategy1(const Index& b, int len, const uint8_t* q, int x, int min_len, int max_intv, Intv* mem, Counters* cn) {
    int i, c;
    Intv ik, ok[4];
    memset(mem, 0, sizeof(Intv));
    if (q[x] > 3) return x + 1;
    bwt_set_intv(b, q[x], ik);
    for (i = x + 1; i < len; ++i) {
        if (q[i] < 4) {
            c = 3 - q[i];
            bwt_extend(b, ik, ok, 0, cn);
            if (ok[c].x[2] < (bwtint_t)max_intv && i - x >= min_len) {
                *mem = ok[c];
                mem->info = (uint64_t)x << 32 | (i + 1);
                return i + 1;
            }
            ik = ok[c];
        } else return i + 1;
    }
    return len;
}

// ---------------------------------------------------------------- bwamem.c: seeding / chaining
void mem_collect_intv(const MemOpt& o, const Index& b, int len, const uint8_t* seq, std::vector<Intv>& mem, Counters* cn) {
    int x = 0;
    int split_len = (int)(o.min_seed_len * o.split_factor + .499);
    std::vector<Intv> mem1;
    mem.clear();
    while (x < len) {   // first pass: all SMEMs
        if (seq[x] < 4) {
            x = bwt_smem1(b, len, seq, x, 1, mem1, cn);
            for (const Intv& p : mem1)
                if ((int)((uint32_t)p.info - (p.info >> 32)) >= o.min_seed_len) mem.push_back(p);
        } else ++x;
    }
    size_t old_n = mem.size();   // second pass: MEMs inside a long SMEM
    for (size_t k = 0; k < old_n; ++k) {
        Intv p = mem[k];
        int start = (int)(p.info >> 32), end = (int32_t)p.info;
        if (end - start < split_len || p.x[2] > (bwtint_t)o.split_width) continue;
        bwt_smem1(b, len, seq, (start + end) >> 1, (int)p.x[2] + 1, mem1, cn);
        for (const Intv& m : mem1)
            if ((int)((uint32_t)m.info - (m.info >> 32)) >= o.min_seed_len) mem.push_back(m);
    }
    if (o.max_mem_intv > 0) {   // third pass: LAST-like
        x = 0;
        while (x < len) {
            if (seq[x] < 4) {
                Intv m;
                x = bwt_seed_strategy1(b, len, seq, x, o.min_seed_len, o.max_mem_intv, &m, cn);
                if (m.x[2] > 0) mem.push_back(m);
            } else ++x;
        }
    }
    ks_introsort(mem.size(), mem.data(), [](const Intv& a, const Intv& c) { return a.info < c.info; });
}

static int test_and_merge(const MemOpt& o, int64_t l_pac, Chain* c, const Seed* p, int seed_rid) {
    int64_t qend, rend, x, y;
    const Seed* last = &c->seeds.back();
    qend = last->qbeg + last->len;
    rend = last->rbeg + last->len;
    if (seed_rid != c->rid) return 0;   // different chr; request a new chain
    if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
        return 1;   // contained seed; do nothing
    if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;   // don't chain if on different strand
    x = p->qbeg - last->qbeg;   // always non-negative
    y = p->rbeg - last->rbeg;
    if (y >= 0 && x - y <= o.w && y - x <= o.w && x - last->len < o.max_chain_gap && y - last->len < o.max_chain_gap) {
        c->seeds.push_back(*p);
        return 1;
    }
    return 0;   // request to add a new chain
}

std::vector<Chain> mem_chain(const MemOpt& o, const Index& b, int len, const uint8_t* seq, Counters* cn, std::vector<Intv>* intv_out, std::vector<Seed>* seeds_out) {
    std::vector<Chain> chain;
    if (len < o.min_seed_len) return chain;
    // kbtree(chn) keyed by pos.  Restated as a vector kept sorted by pos; equal keys are
    // inserted after existing ones and `lower` is the right-most chain with pos <= key.
    // (Equal-pos behaviour of upstream's kbtree is not pinned by any fixture.)
    std::vector<Chain*> tree;
    std::vector<Intv> mem;
    mem_collect_intv(o, b, len, seq, mem, cn);
    if (intv_out) *intv_out = mem;
    int bq = 0, e = 0, l_rep = 0;
    for (const Intv& p : mem) {   // compute frac_rep
        int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
        if (p.x[2] <= (bwtint_t)o.max_occ) continue;
        if (sb > e) { l_rep += e - bq; bq = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - bq;
    for (const Intv& p : mem) {
        int step, count, slen = (int)((uint32_t)p.info - (p.info >> 32));
        int64_t k;
        step = p.x[2] > (bwtint_t)o.max_occ ? (int)(p.x[2] / o.max_occ) : 1;
        for (k = count = 0; k < (int64_t)p.x[2] && count < o.max_occ; k += step, ++count) {
            Seed s;
            int rid, to_add = 0;
            s.rbeg = (int64_t)bwt_sa(b, p.x[0] + k, cn);
            s.qbeg = (int)(p.info >> 32);
            s.score = s.len = slen;
            rid = bns_intv2rid(b, s.rbeg, s.rbeg + s.len);
            if (seeds_out) { Seed t = s; t.score = rid; seeds_out->push_back(t); }
            if (rid < 0) continue;   // bridging multiple reference sequences or the forward-reverse boundary
            if (!tree.empty()) {
                // lower = right-most chain with pos <= s.rbeg
                size_t lo = 0, hi = tree.size();
                while (lo < hi) { size_t m = (lo + hi) >> 1; if (tree[m]->pos <= s.rbeg) lo = m + 1; else hi = m; }
                Chain* lower = lo ? tree[lo - 1] : nullptr;
                if (!lower || !test_and_merge(o, b.l_pac, lower, &s, rid)) to_add = 1;
                if (to_add) {
                    Chain* c = new Chain();
                    c->pos = s.rbeg; c->seeds.push_back(s); c->rid = rid; c->is_alt = !!b.contigs[rid].is_alt;
                    tree.insert(tree.begin() + lo, c);
                }
            } else {
                Chain* c = new Chain();
                c->pos = s.rbeg; c->seeds.push_back(s); c->rid = rid; c->is_alt = !!b.contigs[rid].is_alt;
                tree.push_back(c);
            }
        }
    }
    chain.reserve(tree.size());
    for (Chain* c : tree) { c->frac_rep = (float)l_rep / len; chain.push_back(*c); delete c; }
    return chain;
}

static int mem_chain_weight(const Chain& c) {
    int64_t end;
    int w = 0, tmp;
    end = 0;
    for (const Seed& s : c.seeds) {
        if (s.qbeg >= end) w += s.len;
        else if (s.qbeg + s.len > end) w += s.qbeg + s.len - (int)end;
        end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
    }
    tmp = w; w = 0; end = 0;
    for (const Seed& s : c.seeds) {
        if (s.rbeg >= end) w += s.len;
        else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
        end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}

#define chn_beg(ch) ((ch).seeds.front().qbeg)
#define chn_end(ch) ((ch).seeds.back().qbeg + (ch).seeds.back().len)

int mem_chain_flt(const MemOpt& o, std::vector<Chain>& a) {
    int i, k, n_chn = (int)a.size();
    std::vector<int> chains;
    if (n_chn == 0) return 0;
    for (i = k = 0; i < n_chn; ++i) {
        Chain& c = a[i];
        c.first = -1; c.kept = 0;
        c.w = mem_chain_weight(c);
        if ((int)c.w < o.min_chain_weight) continue;
        if (k != i) a[k] = a[i];
        ++k;
    }
    n_chn = k;
    a.resize(n_chn);
    ks_introsort(a.size(), a.data(), [](const Chain& x, const Chain& y) { return x.w > y.w; });
    a[0].kept = 3;
    chains.push_back(0);
    for (i = 1; i < n_chn; ++i) {
        int large_ovlp = 0;
        for (k = 0; k < (int)chains.size(); ++k) {
            int j = chains[k];
            int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
            int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
            if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {   // have overlap
                int li = chn_end(a[i]) - chn_beg(a[i]);
                int lj = chn_end(a[j]) - chn_beg(a[j]);
                int min_l = li < lj ? li : lj;
                if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {   // significant overlap
                    large_ovlp = 1;
                    if (a[j].first < 0) a[j].first = i;   // keep the first shadowed hit s.t. mapq can be more accurate
                    if (a[i].w < a[j].w * o.drop_ratio && (int)a[j].w - (int)a[i].w >= o.min_seed_len << 1) break;
                }
            }
        }
        if (k == (int)chains.size()) {
            chains.push_back(i);
            a[i].kept = large_ovlp ? 2 : 3;
        }
    }
    for (i = 0; i < (int)chains.size(); ++i) {
        Chain& c = a[chains[i]];
        if (c.first >= 0) a[c.first].kept = 1;
    }
    for (i = k = 0; i < n_chn; ++i) {   // don't extend more than max_chain_extend .kept=1/2 chains
        if (a[i].kept == 0 || a[i].kept == 3) continue;
        if (++k >= o.max_chain_extend) break;
    }
    for (; i < n_chn; ++i)
        if (a[i].kept < 3) a[i].kept = 0;
    for (i = k = 0; i < n_chn; ++i) {
        if (a[i].kept == 0) continue;
        if (k != i) a[k] = a[i];
        ++k;
    }
    a.resize(k);
    return k;
}

// ---------------------------------------------------------------- ksw.c
struct eh_t { int32_t h, e; };

int ksw_extend2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins,
                int w, int end_bonus, int zdrop, int h0, int* _qle, int* _tle, int* _gtle, int* _gscore, int* _max_off, Counters* cn) {
    int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    assert(h0 > 0);
    std::vector<int8_t> qp((size_t)qlen * m);
    std::vector<eh_t> eh(qlen + 1, eh_t{0, 0});
    for (k = i = 0; k < m; ++k) {   // query profile
        const int8_t* p = &mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }
    // fill the first row
    eh[0].h = h0;
    if (qlen >= 1) eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j) eh[j].h = eh[j - 1].h - e_ins;
    // adjust $w if it is too large
    k = m * m;
    for (i = 0, max = 0; i < k; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    // DP loop
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1;
    max_off = 0;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, mm = 0, mj = -1;
        const int8_t* q = &qp[(size_t)target[i] * qlen];
        // apply the band and the constraint (if provided)
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        // compute the first column
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            if (h1 < 0) h1 = 0;
        } else h1 = 0;
        for (j = beg; j < end; ++j) {
            // At the beginning of the loop: eh[j] = { H(i-1,j-1), E(i,j) }, f = F(i,j) and h1 = H(i,j-1)
            eh_t* p = &eh[j];
            int h, M = p->h, e = p->e;
            p->h = h1;
            M = M ? M + q[j] : 0;   // separating H and M to disallow a cigar like "100M3I3D20M"
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = mm > h ? mj : j;   // record the position where max score is achieved
            mm = mm > h ? mm : h;
            t = M - oe_del;
            t = t > 0 ? t : 0;
            e -= e_del;
            e = e > t ? e : t;   // computed E(i+1,j)
            p->e = e;
            t = M - oe_ins;
            t = t > 0 ? t : 0;
            f -= e_ins;
            f = f > t ? f : t;   // computed F(i,j+1)
        }
        if (cn) cn->ext_cells += end > beg ? end - beg : 0;
        eh[end].h = h1; eh[end].e = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (mm == 0) break;
        if (mm > max) {
            max = mm; max_i = i; max_j = mj;
            max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        // update beg and end for the next round
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    if (_gtle) *_gtle = max_ie + 1;
    if (_gscore) *_gscore = gscore;
    if (_max_off) *_max_off = max_off;
    return max;
}

#define MINUS_INF -0x40000000

static inline void push_cigar(std::vector<uint32_t>& cigar, int op, int len) {
    if (cigar.empty() || op != (int)(cigar.back() & 0xf)) cigar.push_back(len << 4 | op);
    else cigar.back() += len << 4;
}

int ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins,
                int w, int* n_cigar_, std::vector<uint32_t>* cigar_, Counters* cn) {
    int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
    bool bt = n_cigar_ && cigar_;
    if (n_cigar_) *n_cigar_ = 0;
    n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;   // maximum #columns of the backtrack matrix
    std::vector<uint8_t> z(bt ? (size_t)n_col * tlen : 0);
    std::vector<int8_t> qp((size_t)qlen * m);
    std::vector<eh_t> eh(qlen + 1, eh_t{0, 0});
    for (k = i = 0; k < m; ++k) {
        const int8_t* p = &mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }
    // fill the first row
    eh[0].h = 0; eh[0].e = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = MINUS_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;   // everything is -inf outside the band
    // DP loop
    for (i = 0; i < tlen; ++i) {   // target sequence is in the outer loop
        int32_t f = MINUS_INF, h1, beg, end, t;
        const int8_t* q = &qp[(size_t)target[i] * qlen];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;   // only loop through [beg,end) of the query sequence
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        uint8_t* zi = bt ? &z[(size_t)i * n_col] : nullptr;
        for (j = beg; j < end; ++j) {
            // At the beginning of the loop: eh[j] = { H(i-1,j-1), E(i,j) }, f = F(i,j) and h1 = H(i,j-1)
            eh_t* p = &eh[j];
            int32_t h, mm = p->h, e = p->e;
            uint8_t d;   // direction
            p->h = h1;
            mm += q[j];
            d = mm >= e ? 0 : 1;
            h = mm >= e ? mm : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = mm - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            p->e = e;
            t = mm - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            if (zi) zi[j - beg] = d;   // z[i,j] keeps h for the current cell and e/f for the next cell
        }
        if (cn) cn->glob_cells += end > beg ? end - beg : 0;
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    score = eh[qlen].h;
    if (bt) {   // backtrack
        int which = 0;
        std::vector<uint32_t> cigar;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;   // (i,k) points to the last cell
        while (i >= 0 && k >= 0) {
            which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { push_cigar(cigar, 0, 1); --i; --k; }
            else if (which == 1) { push_cigar(cigar, 2, 1); --i; }
            else { push_cigar(cigar, 1, 1); --k; }
        }
        if (i >= 0) push_cigar(cigar, 2, i + 1);
        if (k >= 0) push_cigar(cigar, 1, k + 1);
        std::reverse(cigar.begin(), cigar.end());
        *n_cigar_ = (int)cigar.size();
        *cigar_ = cigar;
    }
    return score;
}

// ---- ksw_align2: emulation of the SSE2 striped kernels ksw_u8 / ksw_i16 -------------------
// The striped layout matters: the lazy-F loop does not refresh E, so results are defined by
// this exact evaluation order, not by the textbook recurrences.  Vectors are emulated lane by lane.
#define KSW_XBYTE 0x10000
#define KSW_XSTOP 0x20000
#define KSW_XSUBO 0x40000
#define KSW_XSTART 0x80000

namespace {
struct KswQ {
    int qlen, slen, p, size;   // p = lanes per vector (16 for u8, 8 for i16)
    int shift, mdiff, max;
    std::vector<int> qp;       // [m][slen][p]
};

KswQ ksw_qinit(int size, int qlen, const uint8_t* query, int m, const int8_t* mat) {
    KswQ q;
    size = size > 1 ? 2 : 1;
    q.size = size; q.qlen = qlen;
    q.p = 8 * (3 - size);   // values per __m128i
    q.slen = (qlen + q.p - 1) / q.p;
    int tmp = m * m;
    q.shift = 127; q.mdiff = 0;
    for (int a = 0; a < tmp; ++a) {   // find the minimum and maximum score
        if (mat[a] < (int8_t)q.shift) q.shift = mat[a];
        if (mat[a] > (int8_t)q.mdiff) q.mdiff = mat[a];
    }
    q.max = q.mdiff;
    q.shift = 256 - (q.shift & 0xff);   // NB: q->shift is uint8_t upstream
    q.shift &= 0xff;
    q.mdiff += q.shift;   // difference between the min and max scores
    q.qp.assign((size_t)m * q.slen * q.p, 0);
    size_t t = 0;
    int nlen = q.slen * q.p;
    for (int a = 0; a < m; ++a) {
        const int8_t* ma = mat + a * m;
        for (int i = 0; i < q.slen; ++i)
            for (int k = i; k < nlen; k += q.slen)   // p iterations
                q.qp[t++] = (k >= qlen ? 0 : ma[query[k]]) + (size == 1 ? q.shift : 0);
    }
    return q;
}

inline int sat_u8_add(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
inline int sat_u8_sub(int a, int b) { int s = a - b; return s < 0 ? 0 : s; }
inline int sat_i16_add(int a, int b) { int s = a + b; return s > 32767 ? 32767 : s < -32768 ? -32768 : s; }
inline int sat_u16_sub(int a, int b) { int s = a - b; return s < 0 ? 0 : s; }

Kswr ksw_u8(const KswQ& q, int tlen, const uint8_t* target, int o_del_, int e_del_, int o_ins_, int e_ins_, int xtra, Counters* cn) {
    const int P = 16;
    int slen = q.slen, i, te = -1, gmax = 0, minsc, endsc;
    std::vector<uint64_t> b;
    Kswr r;
    minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
    endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
    int oe_del = o_del_ + e_del_, e_del = e_del_, oe_ins = o_ins_ + e_ins_, e_ins = e_ins_, shift = q.shift;
    std::vector<int> H0((size_t)slen * P, 0), H1((size_t)slen * P, 0), E((size_t)slen * P, 0), Hmax((size_t)slen * P, 0);
    int h[P], e[P], f[P], mx[P];
    for (i = 0; i < tlen; ++i) {
        int j, k, imax;
        const int* S = &q.qp[(size_t)target[i] * slen * P];
        for (k = 0; k < P; ++k) { f[k] = 0; mx[k] = 0; }
        // h = H0[slen-1] shifted by one lane (slli_si128(h, 1))
        for (k = P - 1; k > 0; --k) h[k] = H0[(size_t)(slen - 1) * P + k - 1];
        h[0] = 0;
        for (j = 0; j < slen; ++j) {
            for (k = 0; k < P; ++k) {
                int hh = sat_u8_add(h[k], S[(size_t)j * P + k]);
                hh = sat_u8_sub(hh, shift);
                int ee = E[(size_t)j * P + k];
                hh = hh > ee ? hh : ee;
                hh = hh > f[k] ? hh : f[k];
                mx[k] = mx[k] > hh ? mx[k] : hh;
                H1[(size_t)j * P + k] = hh;
                ee = sat_u8_sub(ee, e_del);
                int t = sat_u8_sub(hh, oe_del);
                ee = ee > t ? ee : t;
                E[(size_t)j * P + k] = ee;
                f[k] = sat_u8_sub(f[k], e_ins);
                t = sat_u8_sub(hh, oe_ins);
                f[k] = f[k] > t ? f[k] : t;
                h[k] = H0[(size_t)j * P + k];
            }
        }
        if (cn) cn->rescue_cells += (uint64_t)slen * P;
        // lazy-F (mimics SWPS3); NB: E is not refreshed here
        for (k = 0; k < 16; ++k) {
            int done = 0;
            for (int l = P - 1; l > 0; --l) f[l] = f[l - 1];
            f[0] = 0;
            for (j = 0; j < slen; ++j) {
                int all = 1;
                for (int l = 0; l < P; ++l) {
                    int hh = H1[(size_t)j * P + l];
                    hh = hh > f[l] ? hh : f[l];
                    H1[(size_t)j * P + l] = hh;
                    hh = sat_u8_sub(hh, oe_ins);
                    f[l] = sat_u8_sub(f[l], e_ins);
                    if (sat_u8_sub(f[l], hh) != 0) all = 0;
                }
                if (all) { done = 1; break; }
            }
            if (done) break;
        }
        imax = 0;
        for (k = 0; k < P; ++k) imax = imax > mx[k] ? imax : mx[k];
        if (imax >= minsc) {   // write the b array
            if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | i);
            else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | i;   // modify the last
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            Hmax = H1;
            if (gmax + shift >= 255 || gmax >= endsc) break;
        }
        H0.swap(H1);
    }
    r.score = gmax + shift < 255 ? gmax : 255;
    r.te = te;
    if (r.score != 255) {   // get qe, the end of query match; find the 2nd best score
        int max = -1, tmp, low, high, qlen = slen * 16;
        for (i = 0; i < qlen; ++i) {
            int t = Hmax[i];
            if (t > max) { max = t; r.qe = i / 16 + i % 16 * slen; }
            else if (t == max && (tmp = i / 16 + i % 16 * slen) < r.qe) r.qe = tmp;
        }
        if (!b.empty()) {
            i = (r.score + q.max - 1) / q.max;
            low = te - i; high = te + i;
            for (size_t bi = 0; bi < b.size(); ++bi) {
                int e2 = (int32_t)b[bi];
                if ((e2 < low || e2 > high) && (int)(b[bi] >> 32) > r.score2) { r.score2 = (int)(b[bi] >> 32); r.te2 = e2; }
            }
        }
    }
    return r;
}

Kswr ksw_i16(const KswQ& q, int tlen, const uint8_t* target, int o_del_, int e_del_, int o_ins_, int e_ins_, int xtra, Counters* cn) {
    const int P = 8;
    int slen = q.slen, i, te = -1, gmax = 0, minsc, endsc;
    std::vector<uint64_t> b;
    Kswr r;
    minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
    endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
    int oe_del = o_del_ + e_del_, e_del = e_del_, oe_ins = o_ins_ + e_ins_, e_ins = e_ins_;
    std::vector<int> H0((size_t)slen * P, 0), H1((size_t)slen * P, 0), E((size_t)slen * P, 0), Hmax((size_t)slen * P, 0);
    int h[P], f[P], mx[P];
    for (i = 0; i < tlen; ++i) {
        int j, k, imax;
        const int* S = &q.qp[(size_t)target[i] * slen * P];
        for (k = 0; k < P; ++k) { f[k] = 0; mx[k] = 0; }
        for (k = P - 1; k > 0; --k) h[k] = H0[(size_t)(slen - 1) * P + k - 1];
        h[0] = 0;
        for (j = 0; j < slen; ++j) {
            for (k = 0; k < P; ++k) {
                int hh = sat_i16_add(h[k], S[(size_t)j * P + k]);
                int ee = E[(size_t)j * P + k];
                hh = hh > ee ? hh : ee;
                hh = hh > f[k] ? hh : f[k];
                mx[k] = mx[k] > hh ? mx[k] : hh;
                H1[(size_t)j * P + k] = hh;
                ee = sat_u16_sub(ee, e_del);
                int t = sat_u16_sub(hh, oe_del);
                ee = ee > t ? ee : t;
                E[(size_t)j * P + k] = ee;
                f[k] = sat_u16_sub(f[k], e_ins);
                t = sat_u16_sub(hh, oe_ins);
                f[k] = f[k] > t ? f[k] : t;
                h[k] = H0[(size_t)j * P + k];
            }
        }
        if (cn) cn->rescue_cells += (uint64_t)slen * P;
        for (k = 0; k < 16; ++k) {
            int done = 0;
            for (int l = P - 1; l > 0; --l) f[l] = f[l - 1];
            f[0] = 0;
            for (j = 0; j < slen; ++j) {
                int any_gt = 0;
                for (int l = 0; l < P; ++l) {
                    int hh = H1[(size_t)j * P + l];
                    hh = hh > f[l] ? hh : f[l];
                    H1[(size_t)j * P + l] = hh;
                    hh = sat_u16_sub(hh, oe_ins);
                    f[l] = sat_u16_sub(f[l], e_ins);
                    if (f[l] > hh) any_gt = 1;
                }
                if (!any_gt) { done = 1; break; }
            }
            if (done) break;
        }
        imax = 0;
        for (k = 0; k < P; ++k) imax = imax > mx[k] ? imax : mx[k];
        if (imax >= minsc) {
            if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | i);
            else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | i;
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            Hmax = H1;
            if (gmax >= endsc) break;
        }
        H0.swap(H1);
    }
    r.score = gmax; r.te = te;
    {
        int max = -1, tmp, low, high, qlen = slen * 8;
        for (i = 0, r.qe = -1; i < qlen; ++i) {
            int t = Hmax[i];
            if (t > max) { max = t; r.qe = i / 8 + i % 8 * slen; }
            else if (t == max && (tmp = i / 8 + i % 8 * slen) < r.qe) r.qe = tmp;
        }
        if (!b.empty()) {
            i = (r.score + q.max - 1) / q.max;
            low = te - i; high = te + i;
            for (size_t bi = 0; bi < b.size(); ++bi) {
                int e2 = (int32_t)b[bi];
                if ((e2 < low || e2 > high) && (int)(b[bi] >> 32) > r.score2) { r.score2 = (int)(b[bi] >> 32); r.te2 = e2; }
            }
        }
    }
    return r;
}

inline void revseq(int l, uint8_t* s) {
    for (int i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l - 1 - i]; s[l - 1 - i] = t; }
}
}  // namespace

Kswr ksw_align2(int qlen, uint8_t* query, int tlen, uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, Counters* cn) {
    int size = (xtra & KSW_XBYTE) ? 1 : 2;
    KswQ q = ksw_qinit(size, qlen, query, m, mat);
    Kswr r, rr;
    r = size == 2 ? ksw_i16(q, tlen, target, o_del, e_del, o_ins, e_ins, xtra, cn) : ksw_u8(q, tlen, target, o_del, e_del, o_ins, e_ins, xtra, cn);
    if (size == 1 && r.score == 255) {   // upstream: byte overflow -> caller of ksw_align2 does not retry; mem_matesw only sets XBYTE when l_ms*a < 250
    }
    if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
    revseq(r.qe + 1, query); revseq(r.te + 1, target);   // +1 because qe/te points to the exact end, not the position after the end
    KswQ q2 = ksw_qinit(size, r.qe + 1, query, m, mat);
    rr = size == 2 ? ksw_i16(q2, tlen, target, o_del, e_del, o_ins, e_ins, KSW_XSTOP | r.score, cn) : ksw_u8(q2, tlen, target, o_del, e_del, o_ins, e_ins, KSW_XSTOP | r.score, cn);
    revseq(r.qe + 1, query); revseq(r.te + 1, target);
    if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    return r;
}

// ---------------------------------------------------------------- bwa.c
bool bwa_gen_cigar2(const MemOpt& o, int w_, const Index& b, int l_query, uint8_t* query, int64_t rb, int64_t re,
                    int* score, std::vector<uint32_t>* cigar, int* NM, Counters* cn) {
    int i;
    int64_t l_pac = b.l_pac, rlen;
    if (cigar) cigar->clear();
    if (NM) *NM = -1;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;   // reject if negative length or bridging the forward and reverse strand
    std::vector<uint8_t> rseq = bns_get_seq(b, rb, re);
    rlen = (int64_t)rseq.size();
    if (re - rb != rlen) return false;   // possible if out of range
    if (rb >= l_pac) {   // then reverse both query and rseq; this is to ensure indels to be placed at the leftmost position
        std::reverse(query, query + l_query);
        std::reverse(rseq.begin(), rseq.end());
    }
    int n_cigar = 0;
    if (l_query == re - rb && w_ == 0) {   // no gap; no need to do DP
        if (cigar) { cigar->assign(1, (uint32_t)l_query << 4 | 0); n_cigar = 1; }
        for (i = 0, *score = 0; i < l_query; ++i) *score += o.mat[rseq[i] * 5 + query[i]];
    } else {
        int w, max_gap, max_ins, max_del, min_w;
        // set the band-width
        max_ins = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_ins) / o.e_ins + 1.);
        max_del = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_del) / o.e_del + 1.);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = abs((int)rlen - l_query) + 3;
        w = w > min_w ? w : min_w;
        // NW alignment
        *score = ksw_global2(l_query, query, (int)rlen, rseq.data(), 5, o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, w, cigar ? &n_cigar : nullptr, cigar, cn);
    }
    if (NM && cigar) {   // compute NM (MD string itself is not consumed by lariat)
        int k, x, y, n_mm = 0, n_gap = 0;
        for (k = 0, x = y = 0; k < n_cigar; ++k) {
            int op = (*cigar)[k] & 0xf, len = (*cigar)[k] >> 4;
            if (op == 0) {   // match
                for (i = 0; i < len; ++i)
                    if (query[x + i] != rseq[y + i]) ++n_mm;
                x += len; y += len;
            } else if (op == 2) {   // deletion
                if (k > 0 && k < n_cigar - 1) n_gap += len;   // don't count if D is the first or the last CIGAR
                y += len;
            } else if (op == 1) { x += len; n_gap += len; }   // insertion
        }
        *NM = n_mm + n_gap;
    }
    if (rb >= l_pac) std::reverse(query, query + l_query);   // reverse back query
    return true;
}

// ---------------------------------------------------------------- bwamem.c: extension
static inline int cal_max_gap(const MemOpt& o, int qlen) {
    int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}

#define MAX_BAND_TRY 2

void mem_chain2aln(const MemOpt& o, const Index& b, int l_query, const uint8_t* query, const Chain& c, std::vector<AlnReg>& av, Counters* cn) {
    int i, k, rid, max_off[2], aw[2];   // aw: actual bandwidth used in extension
    int64_t l_pac = b.l_pac, rmax[2], tmp, max = 0;
    int n = (int)c.seeds.size();
    if (n == 0) return;
    // get the max possible span
    rmax[0] = l_pac << 1; rmax[1] = 0;
    for (i = 0; i < n; ++i) {
        int64_t bb, e;
        const Seed* t = &c.seeds[i];
        bb = t->rbeg - (t->qbeg + cal_max_gap(o, t->qbeg));
        e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(o, l_query - t->qbeg - t->len));
        rmax[0] = rmax[0] < bb ? rmax[0] : bb;
        rmax[1] = rmax[1] > e ? rmax[1] : e;
        if (t->len > max) max = t->len;
    }
    rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
    rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
    if (rmax[0] < l_pac && l_pac < rmax[1]) {   // crossing the forward-reverse boundary; then choose one side
        if (c.seeds[0].rbeg < l_pac) rmax[1] = l_pac;   // this works because all seeds are guaranteed to be on the same strand
        else rmax[0] = l_pac;
    }
    // retrieve the reference sequence
    std::vector<uint8_t> rseq = bns_fetch_seq(b, &rmax[0], c.seeds[0].rbeg, &rmax[1], &rid);
    assert(c.rid == rid);
    if (cn) { cn->win_bases += rmax[1] - rmax[0]; ++cn->n_chain_ext; }

    std::vector<uint64_t> srt(n);
    for (i = 0; i < n; ++i) srt[i] = (uint64_t)c.seeds[i].score << 32 | i;
    ks_introsort(srt.size(), srt.data(), [](uint64_t x, uint64_t y) { return x < y; });

    for (k = n - 1; k >= 0; --k) {
        const Seed* s = &c.seeds[(uint32_t)srt[k]];
        for (i = 0; i < (int)av.size(); ++i) {   // test whether extension has been made before
            const AlnReg* p = &av[i];
            int64_t rd;
            int qd, w, max_gap;
            if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;   // not fully contained
            if (s->len - p->seedlen0 > .1 * l_query) continue;   // this seed may give a better alignment
            // qd: distance ahead of the seed on query; rd: on reference
            qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
            max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);   // the maximal gap allowed in regions ahead of the seed
            w = max_gap < p->w ? max_gap : p->w;   // bounded by the band width
            if (qd - rd < w && rd - qd < w) break;   // the seed is "around" a previous hit
            // similar to the previous four lines, but this time we look at the region behind
            qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
            max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);
            w = max_gap < p->w ? max_gap : p->w;
            if (qd - rd < w && rd - qd < w) break;
        }
        if (i < (int)av.size()) {   // the seed is (almost) contained in an existing alignment; further testing is needed
            for (i = k + 1; i < n; ++i) {   // check overlapping seeds in the same chain
                const Seed* t;
                if (srt[i] == 0) continue;
                t = &c.seeds[(uint32_t)srt[i]];
                if (t->len < s->len * .95) continue;   // only check overlapping if t is long enough
                if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
            }
            if (i == n) {   // no overlapping seeds; then skip extension
                srt[k] = 0;   // mark that seed extension has not been performed
                continue;
            }
        }

        av.emplace_back();
        AlnReg* a = &av.back();
        a->w = aw[0] = aw[1] = o.w;
        a->score = a->truesc = -1;
        a->rid = c.rid;

        if (s->qbeg) {   // left extension
            int qle, tle, gtle, gscore;
            std::vector<uint8_t> qs(s->qbeg);
            for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
            tmp = s->rbeg - rmax[0];
            std::vector<uint8_t> rs(tmp);
            for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
            for (i = 0; i < MAX_BAND_TRY; ++i) {
                int prev = a->score;
                aw[0] = o.w << i;
                a->score = ksw_extend2(s->qbeg, qs.data(), (int)tmp, rs.data(), 5, o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, aw[0], o.pen_clip5, o.zdrop,
                                       s->len * o.a, &qle, &tle, &gtle, &gscore, &max_off[0], cn);
                if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
            }
            // check whether we prefer to reach the end of the query
            if (gscore <= 0 || gscore <= a->score - o.pen_clip5) {   // local extension
                a->qb = s->qbeg - qle; a->rb = s->rbeg - tle;
                a->truesc = a->score;
            } else {   // to-end extension
                a->qb = 0; a->rb = s->rbeg - gtle;
                a->truesc = gscore;
            }
        } else { a->score = a->truesc = s->len * o.a; a->qb = 0; a->rb = s->rbeg; }

        if (s->qbeg + s->len != l_query) {   // right extension
            int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
            qe = s->qbeg + s->len;
            re = (int)(s->rbeg + s->len - rmax[0]);
            assert(re >= 0);
            for (i = 0; i < MAX_BAND_TRY; ++i) {
                int prev = a->score;
                aw[1] = o.w << i;
                a->score = ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq.data() + re, 5, o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, aw[1],
                                       o.pen_clip3, o.zdrop, sc0, &qle, &tle, &gtle, &gscore, &max_off[1], cn);
                if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
            }
            // similar to the above
            if (gscore <= 0 || gscore <= a->score - o.pen_clip3) {   // local extension
                a->qe = qe + qle; a->re = rmax[0] + re + tle;
                a->truesc += a->score - sc0;
            } else {   // to-end extension
                a->qe = l_query; a->re = rmax[0] + re + gtle;
                a->truesc += gscore - sc0;
            }
        } else { a->qe = l_query; a->re = s->rbeg + s->len; }

        // compute seedcov
        for (i = 0, a->seedcov = 0; i < n; ++i) {
            const Seed* t = &c.seeds[i];
            if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)   // seed fully contained
                a->seedcov += t->len;
        }
        a->w = aw[0] > aw[1] ? aw[0] : aw[1];
        a->seedlen0 = s->len;
        a->frac_rep = c.frac_rep;
    }
}

// ---------------------------------------------------------------- bwamem.c: dedup / patch
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

static int mem_patch_reg(const MemOpt& o, const Index* b, const uint8_t* query, const AlnReg* a, const AlnReg* bb, int* _w, Counters* cn) {
    int w, score, q_s, r_s;
    double r;
    if (b == nullptr || query == nullptr) return 0;
    assert(a->rid == bb->rid && a->rb <= bb->rb);
    if (a->rb < b->l_pac && bb->rb >= b->l_pac) return 0;   // on different strands
    if (a->qb >= bb->qb || a->qe >= bb->qe || a->re >= bb->re) return 0;   // not colinear
    w = (int)((a->re - bb->rb) - (a->qe - bb->qb));   // required bandwidth
    w = w > 0 ? w : -w;
    r = (double)(a->re - bb->rb) / (bb->re - a->rb) - (double)(a->qe - bb->qb) / (bb->qe - a->qb);   // relative bandwidth
    r = r > 0. ? r : -r;
    if (a->re < bb->rb || a->qe < bb->qb) {   // no overlap on query or on ref
        if (w > o.w << 1 || r >= PATCH_MAX_R_BW) return 0;   // the bandwidth or the relative bandwidth is too large
    } else if (w > o.w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;   // more permissive if overlapping on both ref and query
    // global alignment
    w += a->w + bb->w;
    w = w < o.w << 2 ? w : o.w << 2;
    std::vector<uint8_t> qcopy(query + a->qb, query + bb->qe);
    score = 0;
    bwa_gen_cigar2(o, w, *b, bb->qe - a->qb, qcopy.data(), a->rb, bb->re, &score, nullptr, nullptr, cn);
    q_s = (int)((double)(bb->qe - a->qb) / ((bb->qe - bb->qb) + (a->qe - a->qb)) * (bb->score + a->score) + .499);   // predicted score from query
    r_s = (int)((double)(bb->re - a->rb) / ((bb->re - bb->rb) + (a->re - a->rb)) * (bb->score + a->score) + .499);   // predicted score from ref
    if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
    *_w = w;
    return score;
}

int mem_sort_dedup_patch(const MemOpt& o, const Index* b, const uint8_t* query, std::vector<AlnReg>& a, Counters* cn) {
    int m, i, j, n = (int)a.size();
    if (n <= 1) return n;
    ks_introsort(a.size(), a.data(), [](const AlnReg& x, const AlnReg& y) { return x.re < y.re; });   // sort by the END position, not START!
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        AlnReg* p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + o.max_chain_gap) continue;   // then no need to go into the loop below
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + o.max_chain_gap; --j) {
            AlnReg* q = &a[j];
            int64_t orr, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;   // a[j] has been excluded
            orr = q->re - p->rb;   // overlap length on the reference
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;   // overlap length on the query
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;   // min ref len in alignment
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;   // min qry len in alignment
            if (orr > o.mask_level_redun * mr && oq > o.mask_level_redun * mq) {   // one of the hits is redundant
                if (p->score < q->score) {
                    p->qe = p->qb;
                    break;
                } else q->qe = q->qb;
            } else if (q->rb < p->rb && (score = mem_patch_reg(o, b, query, q, p, &w, cn)) > 0) {   // then merge q into p
                p->n_comp += q->n_comp + 1;
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)   // exclude identical hits
        if (a[i].qe > a[i].qb) {
            if (m != i) a[m++] = a[i];
            else ++m;
        }
    n = m;
    a.resize(n);
    ks_introsort(a.size(), a.data(), [](const AlnReg& x, const AlnReg& y) {
        return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb)));
    });
    for (i = 1; i < n; ++i)   // mark identical hits
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)   // exclude identical hits
        if (a[i].qe > a[i].qb) {
            if (m != i) a[m++] = a[i];
            else ++m;
        }
    a.resize(n ? m : 0);
    return (int)a.size();
}

std::vector<AlnReg> mem_align1_core(const MemOpt& o, const Index& b, int l_seq, const uint8_t* seq, Counters* cn) {
    std::vector<AlnReg> regs;
    if (cn) { ++cn->n_reads; cn->read_bases += l_seq; }
    std::vector<Chain> chn = mem_chain(o, b, l_seq, seq, cn);
    mem_chain_flt(o, chn);
    // mem_flt_chained_seeds: returns immediately unless l_seq is several hundred bp (5.5*ln(l) > 0.05*l ... see SURVEY Appendix A)
    for (const Chain& c : chn) mem_chain2aln(o, b, l_seq, seq, c, regs, cn);
    mem_sort_dedup_patch(o, &b, seq, regs, cn);
    for (AlnReg& p : regs)
        if (p.rid >= 0 && b.contigs[p.rid].is_alt) p.is_alt = 1;
    return regs;
}

// ---------------------------------------------------------------- bwamem_pair.c: mem_matesw
static inline int mem_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist) {
    int64_t p2;
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;   // p2 is the coordinate of read 2 on the read 1 strand
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// ---- rescue probe (tools/rescue_probe.py): what an exact no-DP / banded shortcut for ksw_align2 would need to be true, counted per mem_matesw
// attempt on the oracle's own job list.  For the window (tlen rows) and the query as aligned (qlen columns), with a = 1, b = 4, gap 6 + k:
//   K(d)   the best ungapped local segment of diagonal d = row - column (Kadane, floor 0)
//   V(d)   the most that disjoint segments of diagonal d can add to a path that pays one gap open + extension (7) for each: max sum of (gain - 7)+
// Any path's score is at most 7 + sum over its diagonals of V(d) when none of its pieces lies on d0, the result's diagonal.
std::atomic<uint64_t> g_rescue_probe[64];
int g_rescue_probe_on = 0;
// (r06) the certificate of the row-restricted forward pass (k_rescue3.h; DESIGN section 3, "K6: a window the size of its hit"), counted on every attempt:
//   K(d)      best ungapped segment of diagonal d;  d0 = the first diagonal with the largest K;  K0 = K(d0)
//   V6(d)     the most that disjoint segments of d add to a path that pays a gap open (6) for each:  max sum of (segment - 6)+
//   X8(d0)    the most that disjoint segments of d0 are worth to a path that leaves and returns between them (two gaps and two diagonals' worth of
//             extension for every return: 8 of it charged to d0):  max over segment sets of sum - 8 (n - 1)
//   any path with a cell more than w diagonals off d0 scores at most  max(6 + Vside, X8(d0) + Vside - (w + 1)),  Vside = sum of V6 over d != d0;
//   with w = X8 - K0 + Vside that is below K0, a score the rows [d0 - w, d0 + qlen + w) reach themselves: the full window's result is theirs.
static void rescue_probe_cert(const MemOpt& o, int qlen, const uint8_t* q, int tlen, const uint8_t* t, const Kswr& r) {
    auto& P = g_rescue_probe;
    const int nd = tlen + qlen - 1, go = o.o_del < o.o_ins ? o.o_del : o.o_ins;
    int K0 = 0, d0 = 0;
    long vall = 0;
    std::vector<int> V6(nd, 0);
    for (int di = 0; di < nd; ++di) {
        const int d = di - (qlen - 1);
        int k0 = d < 0 ? -d : 0, i0 = d < 0 ? 0 : d;
        int h = 0, best = 0, open = -1000000, vbest = 0;
        for (int k = k0, i = i0; k < qlen && i < tlen; ++k, ++i) {
            const int sc = (q[k] > 3 || t[i] > 3) ? -1 : (q[k] == t[i] ? o.a : -o.b);
            h = h + sc > 0 ? h + sc : 0;
            if (h > best) best = h;
            open = (open > vbest - go ? open : vbest - go) + sc;
            if (open > vbest) vbest = open;
        }
        V6[di] = vbest; vall += vbest;
        if (best > K0) { K0 = best; d0 = d; }
    }
    P[32]++;
    P[33] += (uint64_t)tlen * (uint64_t)qlen;
    if (go + vall < o.min_seed_len) {   // no path reaches min_seed_len: no region, no DP at all
        P[34]++;
        if (r.score >= o.min_seed_len) P[47]++;   // (must never happen)
        return;
    }
    const long vside = vall - V6[d0 + qlen - 1];
    int x8 = K0;
    {
        const int d = d0;
        int k0 = d < 0 ? -d : 0, i0 = d < 0 ? 0 : d;
        int open = -1000000, vbest = 0;
        for (int k = k0, i = i0; k < qlen && i < tlen; ++k, ++i) {
            const int sc = (q[k] > 3 || t[i] > 3) ? -1 : (q[k] == t[i] ? o.a : -o.b);
            open = (open > vbest - (go + 2) ? open : vbest - (go + 2)) + sc;
            if (open > vbest) vbest = open;
        }
        if (vbest + go + 2 > x8) x8 = vbest + go + 2;
    }
    if (go + vside >= K0) { P[35]++; P[36] += (uint64_t)tlen * (uint64_t)qlen; return; }   // another diagonal is too strong: the full window
    long w = x8 - K0 + vside;
    long r0 = d0 - w, r1 = d0 + qlen + w;
    if (r0 < 0) r0 = 0;
    if (r1 > tlen) r1 = tlen;
    P[37]++;
    P[36] += (uint64_t)(r1 - r0) * (uint64_t)qlen;
    P[38] += (uint64_t)w; if ((uint64_t)w > P[39]) P[39] = (uint64_t)w;
    if (x8 > K0) P[40]++;
    // what the certificate promises, checked against the full DP's result
    if (r.score >= o.min_seed_len) {
        if (r.te < r0 || r.te >= r1) P[47]++;
        if (r.score < K0) P[47]++;
        if (std::abs((r.te - r.qe) - d0) > w) P[46]++;   // the result's end is off the strip (allowed only if an equal path inside ends there too: counted, looked at)
    } else if (K0 >= o.min_seed_len) P[47]++;
    if (w <= 8) P[41]++;
    if (w <= 16) P[42]++;
    if (w <= 32) P[43]++;
}

// the certificate as k_resc_cert computes it (k_rescue3.h): h(d) = exact 5-mer matches on diagonal d, V(d) = (h - 2)+, d0 = the diagonal with the most hits.
// Class A (no DP at all): no path with a gap reaches K0 — 6 + Vside < K0 (paths off d0), Y2 < K0 (two or more pieces of d0: each return costs 8),
// and for every distance D the V within D of d0 sums to less than D (a chain of pieces off d0 pays its farthest piece's distance) — so the result is d0's best
// segment: first end of the maximum, the shortest segment that has it.  Class B: rows [d0 - w, d0 + qlen + w).  Class C: the whole window.  P[48..63].
static void rescue_probe_cert2(const MemOpt& o, int qlen, const uint8_t* q, int tlen, const uint8_t* t, const Kswr& r) {
    auto& P = g_rescue_probe;
    const int nd = tlen + qlen - 1;
    std::vector<int> H(nd, 0);
    for (int di = 0; di < nd; ++di) {
        const int d = di - (qlen - 1);
        int k0 = d < 0 ? -d : 0, i0 = d < 0 ? 0 : d, run = 0, h = 0;
        for (int k = k0, i = i0; k < qlen && i < tlen; ++k, ++i) {
            if (q[k] < 4 && q[k] == t[i]) { if (++run >= 5) ++h; } else run = 0;
        }
        H[di] = h;
    }
    int di0 = 0; long vall = 0;
    for (int di = 0; di < nd; ++di) { if (H[di] > H[di0]) di0 = di; vall += H[di] > 2 ? H[di] - 2 : 0; }
    const int d0 = di0 - (qlen - 1), h0 = H[di0];
    const uint64_t ref_fwd = (uint64_t)(16 * ((qlen + 15) / 16)) * (uint64_t)tlen;
    uint64_t ref_rev = 0;
    if (r.score >= o.min_seed_len && r.qe >= 0) ref_rev = (uint64_t)(16 * ((r.qe + 1 + 15) / 16)) * (uint64_t)(r.qb >= 0 ? r.te - r.tb + 1 : r.te + 1);
    P[48]++; P[49] += ref_fwd + ref_rev;
    if (6 + vall < o.min_seed_len) { P[50]++; if (r.score >= o.min_seed_len) P[63]++; return; }
    const long vside = vall - (h0 > 2 ? h0 - 2 : 0);
    // d0 exactly: K0 and where its first maximum ends, X8, Y2 (two or more segments)
    const int k0 = d0 < 0 ? -d0 : 0, i0 = d0 < 0 ? 0 : d0;
    int n = std::min(qlen - k0, tlen - i0);
    int K0 = 0, e0 = -1;
    {
        int h = 0;
        for (int c = 0; c < n; ++c) {
            const int sc = q[k0 + c] > 3 ? -1 : (q[k0 + c] == t[i0 + c] ? o.a : -o.b);
            h = h + sc > 0 ? h + sc : 0;
            if (h > K0) { K0 = h; e0 = c; }
        }
    }
    int open1 = -1000000, v1 = 0, open2 = -1000000, v2 = -1000000;   // v1: best sum of (seg - 8) over >= 1 segments so far (0: none yet allowed), v2: over >= 2 segments
    {
        // one-segment value without the charge: plain best segment so far = b1; sets of >= 2: second segment opens from b1 - 8
        int hh = 0, b1 = 0;
        for (int c = 0; c < n; ++c) {
            const int sc = q[k0 + c] > 3 ? -1 : (q[k0 + c] == t[i0 + c] ? o.a : -o.b);
            // segments after the first: open2 continues, or starts from the best value of the sets that end before c
            const int start2 = std::max(b1, v2) - 8;
            open2 = std::max(open2, start2) + sc;
            if (open2 > v2) v2 = open2;
            hh = hh + sc > 0 ? hh + sc : 0;   // (a first segment ending at c: Kadane)
            // b1 must only hold segments that ended strictly before the cell the next one starts at: update after use
            if (hh > b1) b1 = hh;
            (void)open1; (void)v1;
        }
    }
    const int Y2 = v2 < 0 ? 0 : v2;   // sum - 8 (n - 1), n >= 2 (b1 is uncharged, every later segment pays 8)
    int X8 = std::max(K0, Y2);
    bool A = 6 + vside < K0 && Y2 < K0 && K0 >= o.min_seed_len;
    if (A) {
        long cum = 0;
        for (int D = 1; D < nd && A; ++D) {
            const int a = di0 - D, b = di0 + D;
            if (a >= 0) cum += H[a] > 2 ? H[a] - 2 : 0;
            if (b < nd) cum += H[b] > 2 ? H[b] - 2 : 0;
            if (cum >= D) A = false;
            if (a < 0 && b >= nd) break;
        }
    }
    if (A) {
        int dmin = nd;
        for (int di = 0; di < nd; ++di) if (di != di0 && H[di] > 2) dmin = std::min(dmin, std::abs(di - di0));
        if (vside < dmin) P[56]++;   // the simpler sufficient condition: all of Vside is nearer than the nearest diagonal that has any
    }
    if (A) {
        P[51]++;
        // the predicted result
        const int te = i0 + e0, qe = k0 + e0;
        int rr = 0, cb = -1;
        for (int c = e0; c >= 0; --c) {
            const int sc = q[k0 + c] > 3 ? -1 : (q[k0 + c] == t[i0 + c] ? o.a : -o.b);
            rr = rr + sc > 0 ? rr + sc : 0;
            if (rr >= K0) { cb = c; break; }
        }
        if (r.score != K0 || r.te != te || r.qe != qe || cb < 0 || r.tb != i0 + cb || r.qb != k0 + cb) P[62]++;
        return;
    }
    if (6 + vside >= K0) { P[52]++; P[53] += ref_fwd + ref_rev; return; }
    long w = X8 - K0 + vside;
    long r0 = std::max(0L, (long)d0 - w), r1 = std::min((long)tlen, (long)d0 + qlen + w);
    P[54]++;
    P[53] += (uint64_t)(16 * ((qlen + 15) / 16)) * (uint64_t)(r1 - r0) + ref_rev;
    P[55] += (uint64_t)w;
    if (r.score >= o.min_seed_len && (r.te < r0 || r.te >= r1 || r.score < K0)) P[63]++;
    if (r.score < o.min_seed_len && K0 >= o.min_seed_len) P[63]++;
}

static void rescue_probe(const MemOpt& o, int qlen, const uint8_t* q, int tlen, const uint8_t* t, const Kswr& r) {
    auto& P = g_rescue_probe;
    rescue_probe_cert(o, qlen, q, tlen, t, r);
    rescue_probe_cert2(o, qlen, q, tlen, t, r);
    P[0]++;
    if (r.score < o.min_seed_len || r.qb < 0) { P[1]++; return; }   // no region comes of it
    P[2]++;
    const int S = r.score, d0 = r.te - r.qe;
    const int ungapped = (r.te - r.tb) == (r.qe - r.qb);
    const int nd = tlen + qlen - 1;
    std::vector<int> K(nd, 0), V(nd, 0);
    int n_act = 0, qstar = 0;
    long vsum[4] = {0, 0, 0, 0};   // V over |d - d0| > 0, 8, 16, 32
    for (int di = 0; di < nd; ++di) {
        const int d = di - (qlen - 1);
        int k0 = d < 0 ? -d : 0, i0 = d < 0 ? 0 : d;
        int h = 0, best = 0, open = -1000000, vbest = 0;
        for (int k = k0, i = i0; k < qlen && i < tlen; ++k, ++i) {
            const int sc = (q[k] > 3 || t[i] > 3) ? -1 : (q[k] == t[i] ? o.a : -o.b);
            h = h + sc > 0 ? h + sc : 0;
            if (h > best) best = h;
            const int no = (open > vbest - (o.o_ins + o.e_ins) ? open : vbest - (o.o_ins + o.e_ins)) + sc;   // a segment that has paid its gap
            open = no;
            if (open > vbest) vbest = open;
        }
        K[di] = best; V[di] = vbest;
        if (d != d0) {
            if (best >= 8) { n_act++; if (best > qstar) qstar = best; }
            const int dist = d > d0 ? d - d0 : d0 - d;
            if (dist > 0) vsum[0] += vbest;
            if (dist > 8) vsum[1] += vbest;
            if (dist > 16) vsum[2] += vbest;
            if (dist > 32) vsum[3] += vbest;
        }
    }
    // the result's own diagonal: is the result its best ungapped segment, first maximum, and does no detour around a bad stretch pay (a detour costs two gaps,
    // at least 14, and earns at most 7 on a quiet diagonal: net -7; counted as -6 so that a detour can never tie)
    int kad_ok = 0, skip_ok = 0;
    if (ungapped) {
        const int di0 = d0 + (qlen - 1);
        int k0 = d0 < 0 ? -d0 : 0, i0 = d0 < 0 ? 0 : d0;
        int h = 0, best = 0, bi = -1, g = 0, G2 = 0, G1 = 0, gbest = 0, gi = -1;   // G2: the best g two cells back
        for (int k = k0, i = i0; k < qlen && i < tlen; ++k, ++i) {
            const int sc = (q[k] > 3 || t[i] > 3) ? -1 : (q[k] == t[i] ? o.a : -o.b);
            h = h + sc > 0 ? h + sc : 0;
            if (h > best) { best = h; bi = i; }
            int ng = g + sc;
            if (G2 - 6 + sc > ng) ng = G2 - 6 + sc;
            if (ng < 0) ng = 0;
            G2 = G1; if (g > G1) G1 = g;
            g = ng;
            if (g > gbest) { gbest = g; gi = i; }
        }
        kad_ok = best == S && bi == r.te && K[di0] == S;
        skip_ok = kad_ok && gbest == S && gi == r.te;
    }
    if (ungapped) P[3]++;
    if (kad_ok) P[4]++;
    if (skip_ok) P[5]++;
    if (n_act == 0) P[6]++;
    if (skip_ok && n_act == 0) P[7]++;                       // the no-DP proof as it stands: every other diagonal quiet
    for (int w = 0; w < 4; ++w) {
        if (7 + vsum[w] < S) P[8 + w]++;                     // no path that avoids the band reaches the score
        if (7 + vsum[w] < 19) P[12 + w]++;                   // ... nor min_seed_len
        if (skip_ok && vsum[w] == 0) P[16 + w]++;
    }
    P[20] += (uint64_t)n_act; if ((uint64_t)qstar > P[21]) P[21] = (uint64_t)qstar;
    if (S == qlen * o.a) { P[22]++; P[23] += (uint64_t)(tlen - 1 - r.te); }   // a perfect score: no later row can beat it, the forward pass could stop at te
    P[24] += (uint64_t)tlen; P[25] += (uint64_t)S;
    {   // how far the alignment strays from its end diagonal (the band a banded DP would need): from the start cell's diagonal
        const int dev = std::abs((r.tb - r.qb) - d0);
        if (dev <= 8) P[26]++;
        if (dev <= 16) P[27]++;
    }
    if (n_act <= 3 && qstar <= 12) P[28]++;
}

int mem_matesw(const MemOpt& o, const Index& b, const PeStat pes[4], const AlnReg& a, int l_ms, const uint8_t* ms, std::vector<AlnReg>& ma, Counters* cn) {
    int64_t l_pac = b.l_pac;
    int i, r, skip[4], n = 0, rid = -1;
    for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
    for (i = 0; i < (int)ma.size(); ++i) {   // check which orientation has been found
        int64_t dist;
        r = mem_infer_dir(l_pac, a.rb, ma[i].rb, &dist);
        if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
    }
    if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;   // consistent pair exist; no need to perform SW
    for (r = 0; r < 4; ++r) {
        int is_rev, is_larger;
        std::vector<uint8_t> rev, ref;
        const uint8_t* seq;
        int64_t rb, re;
        if (skip[r]) continue;
        is_rev = (r >> 1 != (r & 1));   // whether to reverse complement the mate
        is_larger = !(r >> 1);          // whether the mate has larger coordinate
        if (is_rev) {
            rev.resize(l_ms);
            for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
            seq = rev.data();
        } else seq = ms;
        if (!is_rev) {
            rb = is_larger ? a.rb + pes[r].low : a.rb - pes[r].high;
            re = (is_larger ? a.rb + pes[r].high : a.rb - pes[r].low) + l_ms;   // if on the same strand, end position should be larger to make room for the seq length
        } else {
            rb = (is_larger ? a.rb + pes[r].low : a.rb - pes[r].high) - l_ms;   // similarly on opposite strands
            re = is_larger ? a.rb + pes[r].high : a.rb - pes[r].low;
        }
        if (rb < 0) rb = 0;
        if (re > l_pac << 1) re = l_pac << 1;
        rid = -1;
        if (rb < re) ref = bns_fetch_seq(b, &rb, (rb + re) >> 1, &re, &rid);
        if (a.rid == rid && re - rb >= o.min_seed_len) {   // no funny things happening
            Kswr aln;
            AlnReg bb;
            int tmp, xtra = KSW_XSUBO | KSW_XSTART | (l_ms * o.a < 250 ? KSW_XBYTE : 0) | (o.min_seed_len * o.a);
            std::vector<uint8_t> qcopy(seq, seq + l_ms);
            if (cn) ++cn->n_rescue;
            aln = ksw_align2(l_ms, qcopy.data(), (int)(re - rb), ref.data(), 5, o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, xtra, cn);
            if (g_rescue_probe_on) rescue_probe(o, l_ms, qcopy.data(), (int)(re - rb), ref.data(), aln);
            if (aln.score >= o.min_seed_len && aln.qb >= 0) {   // something goes wrong if aln.qb < 0
                bb.rid = a.rid;
                bb.is_alt = a.is_alt;
                bb.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
                bb.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
                bb.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
                bb.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
                bb.score = aln.score;
                bb.csub = aln.score2;
                bb.secondary = -1;
                bb.seedcov = (int)((bb.re - bb.rb < bb.qe - bb.qb ? bb.re - bb.rb : bb.qe - bb.qb) >> 1);
                ma.push_back(bb);   // make room for a new element
                // move b s.t. ma is sorted
                for (i = 0; i < (int)ma.size() - 1; ++i)   // find the insertion point
                    if (ma[i].score < bb.score) break;
                tmp = i;
                for (i = (int)ma.size() - 1; i > tmp; --i) ma[i] = ma[i - 1];
                ma[i] = bb;
            }
            ++n;
        }
        if (n) mem_sort_dedup_patch(o, nullptr, nullptr, ma, cn);
    }
    return n;
}

// ---------------------------------------------------------------- bwamem.c: mem_reg2aln
static inline int infer_bw(int l1, int l2, int score, int a, int q, int r) {
    int w;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;   // to get equal alignment length, we need at least two gaps
    w = (int)(((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.));
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}

Aln mem_reg2aln(const MemOpt& o, const Index& b, int l_query, const uint8_t* query_, const AlnReg& ar, Counters* cn) {
    Aln a;
    int i, w2, tmp, qb, qe, NM = 0, score = 0, is_rev, last_sc = -(1 << 30);
    int64_t pos, rb, re;
    if (ar.rb < 0 || ar.re < 0) {   // generate an unmapped record
        a.rid = -1; a.pos = -1; a.flag |= 0x4;
        return a;
    }
    qb = ar.qb; qe = ar.qe;
    rb = ar.rb; re = ar.re;
    std::vector<uint8_t> query(query_, query_ + l_query);
    if (ar.secondary >= 0) a.flag |= 0x100;   // secondary alignment (mapq itself is never consumed by lariat, gobwa.go:475)
    tmp = infer_bw(qe - qb, (int)(re - rb), ar.truesc, o.a, o.o_del, o.e_del);
    w2 = infer_bw(qe - qb, (int)(re - rb), ar.truesc, o.a, o.o_ins, o.e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > o.w) w2 = w2 < ar.w ? w2 : ar.w;
    i = 0;
    if (cn && !(qe - qb == re - rb && (w2 < o.w << 2 ? w2 : o.w << 2) == 0)) cn->n_glob++;
    do {
        w2 = w2 < o.w << 2 ? w2 : o.w << 2;
        bwa_gen_cigar2(o, w2, b, qe - qb, &query[qb], rb, re, &score, &a.cigar, &NM, cn);
        if (score == last_sc || w2 == o.w << 2) break;   // it is possible that global alignment and local alignment give different scores
        last_sc = score;
        w2 <<= 1;
    } while (++i < 3 && score < ar.truesc - o.a);
    a.NM = NM;
    pos = bns_depos(b, rb < b.l_pac ? rb : re - 1, &is_rev);
    a.is_rev = is_rev;
    if (!a.cigar.empty()) {   // squeeze out leading or trailing deletions
        if ((a.cigar[0] & 0xf) == 2) {
            pos += a.cigar[0] >> 4;
            a.cigar.erase(a.cigar.begin());
        } else if ((a.cigar.back() & 0xf) == 2) {
            a.cigar.pop_back();
        }
    }
    if (qb != 0 || qe != l_query) {   // add clipping to CIGAR
        int clip5, clip3;
        clip5 = is_rev ? l_query - qe : qb;
        clip3 = is_rev ? qb : l_query - qe;
        if (clip5) a.cigar.insert(a.cigar.begin(), (uint32_t)clip5 << 4 | 3);
        if (clip3) a.cigar.push_back((uint32_t)clip3 << 4 | 3);
    }
    a.rid = bns_pos2rid(b, pos);
    a.pos = pos - b.contigs[a.rid].offset;
    a.score = ar.score; a.sub = ar.sub > ar.csub ? ar.sub : ar.csub;
    a.is_alt = ar.is_alt; a.alt_sc = ar.alt_sc;
    return a;
}

}  // namespace orc

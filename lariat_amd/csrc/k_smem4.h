// k_smem4.h — K1: SMEM seeding with PERSISTENT lanes (one read per lane, every lane always busy), one launch per pass.
// mem_collect_intv (bwt_smem1a passes 1 + 2 and bwt_seed_strategy1), reached from go/src/gobwa/gobwa.go:244,253.
//
// Measured on MI355X (profiles/r01_*): with one lane per read but BWA's loop nest kept as written, the lanes of a wave drift
// into different loops and the hardware serialises them: 13 of 64 lanes are active on average.  Here a pass is a per-lane STATE
// MACHINE around one shared program point: in every turn of the main loop each lane that has a bwt_extend pending performs it (all
// lanes together: their table reads in flight at the same time), then each lane does the bookkeeping of the loop it is in (forward
// extension / backward sweep / pass 3) and prepares its next request.  A lane that finishes its read takes the next one from the
// wave's chunk (chunks of 64 reads come from one device counter), so no lane waits for the slowest read of a batch.  The arithmetic
// and the order of list operations per read are exactly those of bwt_smem1a / bwt_seed_strategy1: only the interleaving across reads
// differs — and work of the reference that is PROVABLY without effect is left out:
//
//  - UNIQUE RUNS: once a match has a single occurrence (interval size 1), extending it is comparing the read with the text at that
//    occurrence.  With a fully resident suffix array the lane looks the position up once (sa[x0]) and then advances sixteen bases per
//    turn by XOR-ing 4-bit packed read and text words, instead of one base per turn through two occurrence records.
//  - POSITIONS INSTEAD OF ROWS (new in r03): an interval that a run produced has one occurrence, and everything downstream (pass 2's
//    probe, pass 3's text walks, K2's seeds) wants that occurrence's text position — which the run knows — not its suffix-array row,
//    which used to cost an inverse-suffix-array read per run end (and K2 a suffix-array read to undo it).  Such an interval is stored
//    as x0 = LH_POSF | position of its first base, x1 = 0, x2 = 1; k_intv_rows turns it back into rows for the stage dump.  What the
//    sweep needs to know about a unique match's neighbourhood comes from the PLCP array (DIndex::plcp, indexed by text position).
//  - CALLS BY TEXT (new in r03): after a read's first unique run the lane knows a locus P where the read (mostly) lies.  A later
//    bwt_smem1a call from x is decided from the text at P alone when that is provable (see START_SMEM1): two PLCP bytes and the
//    comparison with the text replace the ~18 tree / occurrence steps of the forward walk, their filter reads and the sweep.
//  - SWEEP FILTER, K-MER TREE TABLE, COLLAPSED SWEEPS, FORWARD JUMP, PASS-2 PROBE, PASS 3 BY TEXT: see the blocks below.
//
//  - query: 4-bit packed in LDS (8 bases per word, word w of lane L at qn[w*64+L]), staged by the whole wave
//  - prev/curr interval lists: 16-B packed entries in an HBM slab interleaved by thread; the entry the next row starts with stays in
//    registers (and is never written to the slab: most backward rows have a single survivor) and the following one is prefetched
//    while the current extension is in flight
#pragma once
#ifndef LH_K1_SLAB_CHUNK
#define LH_K1_SLAB_CHUNK 4   // entries of a lane's interval list that share a 64-B line of the slab (1: the r02-r05 layout); must divide LH_MAXLEN + 2
#endif
#include "lh_dev.h"
static_assert((LH_MAXLEN + 2) % LH_K1_SLAB_CHUNK == 0, "a lane's list of LH_MAXLEN + 2 entries is a whole number of chunks");

struct __attribute__((aligned(16))) PEnt { u64 lo, hi; };   // x0:40 | x2[0..23]  /  x1:40 | x2[24..32] | info:15
#define LH_M40 0xffffffffffull
__device__ __forceinline__ PEnt pe_pack(u64 x0, u64 x1, u64 x2, int info) {
    PEnt e;
    e.lo = x0 | (x2 << 40);
    e.hi = x1 | ((x2 >> 24) << 40) | ((u64)(uint32_t)info << 49);
    return e;
}
#define PE_X0(e) ((e).lo & LH_M40)
#define PE_X1(e) ((e).hi & LH_M40)
#define PE_X2(e) (((e).lo >> 40) | ((((e).hi >> 40) & 0x1ffull) << 24))
#define PE_INFO(e) ((int)((e).hi >> 49))
#define LH_POSF (1ull << 62)   // DIntv::x0 of an interval with ONE occurrence, stored by text position: x0 = LH_POSF | position of its first base

#define S4_FETCH 0
#define S4_DONE 1
#define S4_REQ_FWD 2
#define S4_REQ_BWD 3
#define S4_REQ_P3 4
#define S4_REQ_FRUN 5    // unique run, forward / backward: active states that need no bwt_extend
#define S4_REQ_BRUN 6
#define S4_BWD_EMIT 8    // states >= 8 are transitions handled without an extension (section B of the main loop)
#define S4_BWD_INIT 9
#define S4_BWD_EMIT0 10
#define S4_SMEM_DONE 11
#define S4_P1_SCAN 12
#define S4_P2_NEXT 13
#define S4_P3_SCAN 14
#define S4_READ_DONE 15
#define S4_FRUN_INIT 16  // unique runs: each global read is issued in one turn and used in the next (its latency hides behind the turn's extensions)
#define S4_FRUN_INIT2 17
#define S4_FRUN_END 18
#define S4_FRUN_END2 19
#define S4_BRUN_INIT 20
#define S4_BRUN_INIT2 21
#define S4_BRUN_END 22
#define S4_TRI_C1 23     // the collapsed sweep did not apply: the longest entry's second row bound (not kept by a run) is read back before the sweep as written
#define S4_P3_JUMP 24    // pass 3: the walk's first bases come from the k-mer tree table
#define S4_P3_JUMP2 25
#define S4_TRI_C1B 26
#define S4_TRI_LCP2 27   // the sweep of a forward list whose longest entry is unique, decided from the PLCP array (see BWD_ROW_BODY)
#define S4_FJUMP 28      // forward extension: the first levels of the walk in one read of the k-mer tree (see START_SMEM1)
#define S4_FJUMP2 29
#define S4_P2_PROBE 30   // pass 2: can the re-seeding inside this SMEM yield a seed at all? (DIndex::rep_t, see S4_P2_NEXT)
#define S4_P2_PROBE2 31
#define S4_P2_PROBE3 32
#define S4_P3_PREP 35    // pass 3 by text (see P3TEXT): positions of the read's unique SMEMs, then a batch of walks = one PLCP byte each
#define S4_P3_PREP2 36
#define S4_P3_PREP3 37
#define S4_P3_T0 38
#define S4_P3_T1 39
#define S4_BT_INIT 40    // pass 1, a bwt_smem1a call decided from the text at the read's known locus (see START_SMEM1)
#define S4_BT_B 41
#define S4_PENDING 34    // one of the shared blocks below runs for this lane before the turn's extensions (todo says which)
#define TD_ROW 1          // BWD_ROW_BODY
#define TD_SMEM 2         // START_SMEM1
#define TD_FADV 4         // BLOOM_ISSUE + FWD_ADVANCE
#define TD_KEY 8          // ... after the filter key has been taken from the read again (a jump moved the interval's end)
#define LH_KMER 12

#ifndef LH_SLOW_BATCH
#define LH_SLOW_BATCH 4       // pass 1
#endif
#ifndef LH_SLOW_BATCH_P2
#define LH_SLOW_BATCH_P2 16   // the pass-2 and pass-3 kernels: their lanes spend a larger share of their turns in the batched states, and a batched
#endif                        // state costs the wave a memory latency; measured at hg38 scale with 2 / 4 / 8 / 16 / 32 / 48 lanes:
#ifndef LH_SLOW_BATCH_P3      // pass 1 6.61 / 6.68 / 6.67 / 7.07 ms, pass 2 2.87 / 2.84 / 2.75 / 2.66 / 2.74 / 3.09, pass 3 3.00 / 2.79 / 2.70 / 2.61 / 2.61 / 2.53
#define LH_SLOW_BATCH_P3 48
#endif
#ifndef LH_SMEM4_WAVES
#define LH_SMEM4_WAVES 4   // waves per SIMD the register budget is sized for
#endif
// THE FIRST CALL OF EVERY READ, IN LOCKSTEP (new in r03).  Measured (profiles/r03_*): pass 1 is bound by instruction issue — every turn of
// the state machine runs the code of every state some lane of the wave is in (16 of 64 lanes active) — and most of its turns belong to the
// read's first bwt_smem1a call, which is the same short program for every read: the k-mer tree's entry for the first ktree_levels bases,
// a few occurrence-table steps until one occurrence is left, the suffix array, the comparison with the text.  k_smem_first runs exactly
// that, one THREAD per read, all threads of a wave in the same loop at the same time; nothing is pushed to a forward list before the
// interval's end reaches position LH_BLOOM_K (the sweep filter's rule for a call from position 0: see START_SMEM1), and with nothing
// before position 0 the sweep emits the longest entry, so the call's result is [0, b) at one occurrence.  A read this does not settle
// (a base that differs from the text at b < len; more than one occurrence left at LH_BLOOM_K bases; non-bases early on; no dense suffix
// array or no sweep filter) is LISTED for the state machine, which takes it up where this kernel left it (K1Resume).
struct K1Resume { i64 Pk; int32_t x, pk; };   // the state machine starts its calls at x; Pk: text position of read base 0 at the locus of a unique match (-1: none); pk = that match's length | (start of the previous call + 2) << 8 | (bt_skip + 1) << 16
__global__ void __launch_bounds__(256) k_smem_first(DIndex ix, DOpts o, int n_reads, const uint32_t* __restrict__ q4, const i64* __restrict__ seq_off, DIntv* __restrict__ intv_out,
                                                     int32_t* __restrict__ n_intv, int32_t* __restrict__ status, K1Resume* __restrict__ resume, int32_t* __restrict__ todo,
                                                     int32_t* __restrict__ todo_count, DCounters* __restrict__ ctr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    const PEnt* const kt = (const PEnt*)ix.ktree;
    const int ktl = kt ? ix.ktree_levels : 0;
    const bool can = ix.isa && ix.tn && q4 && ix.bloom1 && ktl >= 2 && ktl <= 16 && ktl < LH_BLOOM_K && o.min_seed_len >= LH_BLOOM_K;
    unsigned n_ext = 0, n_exec = 0, n_kt = 0;
    int listed = 0;
    if (r < n_reads) {
        const i64 off = seq_off[r];
        const int len = (int)(seq_off[r + 1] - off);
        K1Resume rs; rs.Pk = -1; rs.x = 0; rs.pk = 0;
        int on = -1;   // -1: not settled here (the general kernel makes every call)
        if (can && len >= o.min_seed_len && len <= LH_MAXLEN) {
            const uint32_t w0 = dev_nib8(q4, off), w1 = dev_nib8(q4, off + 8);
            const uint32_t a = w0 & 0x44444444u, b = w1 & 0x44444444u;
            const int valid = a ? (__ffs((int)a) - 1) >> 2 : 8 + (b ? (__ffs((int)b) - 1) >> 2 : 8);
            if (valid >= ktl) {   // the forward jump of START_SMEM1 at x = 0: J = min(tree depth, valid bases, LH_BLOOM_K) = the tree's depth
                uint32_t ca = w0 & 0x33333333u, cb = w1 & 0x33333333u;
                ca = (ca | ca >> 2) & 0x0f0f0f0fu; ca = (ca | ca >> 4) & 0x00ff00ffu; ca = (ca | ca >> 8) & 0xffffu;
                cb = (cb | cb >> 2) & 0x0f0f0f0fu; cb = (cb | cb >> 4) & 0x00ff00ffu; cb = (cb | cb >> 8) & 0xffffu;
                const uint32_t code = (ca | cb << 16) & (ktl >= 16 ? 0xffffffffu : (1u << (2 * ktl)) - 1u);
                const PEnt te = kt[(((1ull << (2 * ktl)) - 4) / 3) + code];
                DIntv c; c.x0 = PE_X0(te); c.x1 = PE_X1(te); c.x2 = PE_X2(te); c.info = 0;
                unsigned e1 = (unsigned)(ktl - 1), e2 = 0;
                int i = ktl, state = c.x2 >= 1 ? 1 : 0;   // 0: give up, 1: walking, 2: the match ended with several occurrences left, 3: one occurrence left
                while (state == 1 && c.x2 > 1) {
                    if (i >= len || i >= LH_BLOOM_K) { state = 0; break; }   // the list closes at the read's end, or entries may be pushed from here on: the state machine's business
                    const int bq = (int)((dev_nib8(q4, off + i)) & 0xf);
                    if (bq > 3) { state = 0; break; }
                    const DIntv ok = dev_extend_c(ix, c, 3 - bq, 0);
                    ++e1; ++e2;
                    if (ok.x2 < 1) { state = 2; break; }   // [0, i) is the whole list and shorter than a seed: the call yields nothing and returns i
                    c = ok; ++i;
                }
                if (state == 1 && c.x2 == 1) state = 3;
                if (state == 2) { on = 0; rs.x = i; rs.pk = 2 << 8 | (i + 1) << 16; n_ext = e1; n_exec = e2; n_kt = 1; }
                if (state == 3) {   // the unique run of the state machine (S4_FRUN_*), start to end in one loop
                    const i64 P = (i64)ix.sa[c.x0];
                    int bnd = len;
                    for (int k = i; k < len; k += 8) {
                        const uint32_t qw = dev_nib8(q4, off + k), tw = dev_nib8(ix.tn, P + k);
                        uint32_t x = qw ^ tw;
                        if (len - k < 8) x &= (1u << (4 * (len - k))) - 1u;
                        if (x) { bnd = k + ((__ffs((int)x) - 1) >> 2); break; }
                    }
                    e1 += (unsigned)(bnd - i);
                    if (bnd < len && (int)(dev_nib8(q4, off + bnd) & 0xf) <= 3) ++e1;   // the bwt_extend that returned an empty interval
                    on = 0;
                    if (bnd >= o.min_seed_len) {
                        DIntv m; m.x0 = LH_POSF | (u64)P; m.x1 = 0; m.x2 = 1; m.info = (u64)(uint32_t)bnd;
                        intv_out[(size_t)r * LH_MAX_INTV] = m;
                        on = 1;
                    }
                    rs.x = bnd; rs.Pk = P; rs.pk = bnd | 2 << 8 | (bnd + 1) << 16;
                    n_ext = e1; n_exec = e2; n_kt = 1;
                }
            }
        }
        if (on >= 0) n_intv[r] = on;
        if (on >= 0 && rs.x >= len) status[r] = 0;   // the read is finished: pass 1 of the state machine does not see it
        else { resume[r] = rs; listed = 1; }
    }
    const u64 lm = __ballot(listed);
    if (lm) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(todo_count, (int32_t)__popcll(lm));
        basep = wave_readlane(basep, 0);
        if (listed) todo[basep + lanes_below(lm, lane)] = r;
    }
    if (ctr) {
        unsigned t1 = (unsigned)wave_sum_i32((int)n_ext), t2 = (unsigned)wave_sum_i32((int)n_exec), t3 = (unsigned)wave_sum_i32((int)n_kt);
        if (lane == 0 && t1) { atomicAdd(&LH_CTR(ctr)->n_ext, (u64)t1); atomicAdd(&LH_CTR(ctr)->n_ext_exec[0], (u64)t2); atomicAdd(&LH_CTR(ctr)->n_ktree[0], (u64)t3); }
    }
}

// (Measured and dropped in r03: the REST of pass 1 as a plain per-thread program as well — bwt_smem1a with the same filters and shortcuts,
// forward list in LDS, one call per read and launch, four launches before the state machine took the remainder.  Bit-exact, but 13 of 64
// lanes active (the calls after the first are not one program: calls by text, walks with and without pushes, three kinds of sweep) and
// latency-bound at 4 waves per SIMD: 13.0 ms against the 10.5 ms of k_smem_first + the state machine.  Measured again in r04 with ONE, two and three
// such launches before the state machine (a read with a single difference from its locus is finished by one): 10.83 / 11.27 / 12.40 ms against
// 10.63 without: the state machine's share does not shrink by what the launch costs.  See the git history (4478681 of r03, bbe4586 of r04).)
// (Also measured and dropped: the sweep filter's answers for a walk's first 16 levels read ahead, four per turn, so that a walk from
// beyond position LH_BLOOM_K can jump to the tree's depth — 28 M fewer tree reads per launch, but the extra state costs every turn: 9.5 ms
// against 8.8; and the reads with work in pass 2 listed by a thread-per-read kernel first: the state machine saves 0.5 ms, the list costs 0.8.)
// PASS 1: all SMEMs of the read (bwt_smem1a with min_intv 1 from every position the previous call returned).  PASS 2: re-seeding inside
// the long SMEMs pass 1 left (recognised from the stored intervals).  PASS 3: bwt_seed_strategy1, appended to the intervals the earlier
// launches left (it depends on the read alone, and the intervals are sorted afterwards).  Each launch carries only its own states:
// fewer instructions per turn, fewer registers.
// BIG: the second chance of the reads whose intervals outgrew their LH_MAX_INTV regular slots (listed by k_big_collect): the same
// passes again, over the list, into slab slots of LH_BIG_INTV intervals (BWA's interval vector grows; no read is refused for it).
struct K1Big { const int32_t* list; const int32_t* count; const int32_t* slot; DIntv* slab; const K1Resume* resume; };   // slot[r]: the read's big-slab slot, -1: none; slab: 2 x LH_BIG_INTV per slot (unsorted | sorted); !BIG pass 1: list / count / resume = the reads k_smem_first left, and where
// ---- request trace of pass 1 (development aid, -DLH_K1_TRACE: tools/k1_trace.py).  Every memory request of the state machine — table, address, bytes — is
// appended to the lane's own sequence (entry k of lane t at trace[k * T + t]); k_k1_replay then issues the same sequences with the same launch geometry and
// nothing else: each lane one request per turn, the next one only when the last has landed (what the state machine's turn structure does), no bookkeeping.
// Its time is the floor of this request stream on this chip; the per-table counts say what the stream is made of.
#define K1T_OCC 0
#define K1T_TREE 1
#define K1T_BLOOM1 2
#define K1T_BLOOM2 3
#define K1T_REPT 4
#define K1T_PLCP 5
#define K1T_TEXT 6
#define K1T_SA 7
#define K1T_ISA 8
#define K1T_SLAB_R 9
#define K1T_SLAB_W 10
#define K1T_INTV_W 11
#define K1T_READS 12
#define K1T_N 13
#ifdef LH_K1_TRACE
__device__ u64* lh_k1_trace;            // null: tracing off
__device__ uint32_t lh_k1_trace_cap;    // entries per lane
__device__ uint32_t* lh_k1_trace_n;     // [T] requests of lane t (may exceed the cap: the rest was not recorded)
#define K1_REQ(tab_, ptr_, bytes_)                                                                                                              \
    {                                                                                                                                            \
        if (PASS == 1 && !BIG && lh_k1_trace) {                                                                                                  \
            if (trace_k < lh_k1_trace_cap) lh_k1_trace[(size_t)trace_k * T + t] = (u64)(uintptr_t)(ptr_) | (u64)(bytes_) << 48 | (u64)(tab_) << 56; \
            ++trace_k;                                                                                                                           \
        }                                                                                                                                        \
    }
#else
#define K1_REQ(tab_, ptr_, bytes_)
#endif

template <int PASS, bool BIG>
__global__ void __launch_bounds__(64, PASS == 3 ? 8 : LH_SMEM4_WAVES) k_smem_pass(DIndex ix, DOpts o, int n_reads_all, const uint8_t* __restrict__ seq, const i64* __restrict__ seq_off,
                                               DIntv* __restrict__ intv_out, int32_t* __restrict__ n_intv, int32_t* __restrict__ status, PEnt* __restrict__ slab,
                                               int32_t* __restrict__ next_read, DCounters* __restrict__ ctr, K1Big big) {
    const int n_reads = (BIG || big.list) ? *big.count : n_reads_all;
    constexpr int ICAP = BIG ? LH_BIG_INTV : LH_MAX_INTV;
    __shared__ uint32_t qn[32 * 64];
    constexpr bool DO1 = PASS == 1, DO2 = PASS == 2, DO12 = DO1 || DO2, DO3 = PASS == 3, P3T = PASS == 3;
    // PASS 2 BY TASKS (r05).  A read's re-seeding calls — one bwt_smem1 from the middle of each long, rare SMEM — are independent of each other, and on repeat
    // families a single call sweeps a forward list of dozens of entries over a hundred rows: a read with ten of them kept its lane for tens of milliseconds
    // after every other lane had finished.  With a task list (k_p2_tasks: big.list / big.count) a read's calls are dealt out to up to LH_P2_SPLIT lanes; what
    // they emit goes to the read's interval slots through an atomic counter, in whatever order — k_smem_fin sorts the intervals by position, and two
    // intervals with equal positions are the same interval.  Reads without such an SMEM are not listed at all.
    const bool P2TASK = PASS == 2 && !BIG && big.list != nullptr;
    int task_k = 0;
    // PASS 3 BY TEXT (with a dense suffix array).  A walk of bwt_seed_strategy1 from x ends at the first length L >= min_seed_len + 1
    // where the match occurs fewer than max_mem_intv times.  If [x, x + Lw), Lw = min_seed_len + 1, lies inside a UNIQUE SMEM of the read
    // (pass 1 left it in the read's interval array, usually with its text position), the read equals the text there: the Lw-mer is unique
    // iff the suffix at its position shares fewer than Lw bases with every other suffix (one PLCP byte), and then the walk's result is
    // that one occurrence at length exactly Lw.  Consecutive walks of an SMEM start Lw apart: their bytes are read together.  Anything
    // else: the walk as written.
    const bool P3TEXT = P3T && ix.isa != nullptr && ix.plcp != nullptr && o.max_mem_intv > 1;
    const int Lw = o.min_seed_len + 1;
    uint32_t uspan = 0;   // the read's two longest unique SMEMs [us, ue) (reads have at most 250 bases: a byte each), at text positions up0 / up1
#define P3_NOTEXT (1 << 30)   // kept in rst (the read's status bits), cleared before they are stored: "this walk start was tried by text and is not unique"
    u64 up0 = 0, up1 = 0;
#define US0 ((int)(uspan & 0xff))
#define UE0 ((int)((uspan >> 8) & 0xff))
#define US1 ((int)((uspan >> 16) & 0xff))
#define UE1 ((int)(uspan >> 24))
    const int lane = LANE();
    const uint32_t T = gridDim.x * 64u, t = blockIdx.x * 64u + (uint32_t)lane;
#ifdef LH_K1_TRACE
    uint32_t trace_k = 0;
#endif
    // (r06) entry e of list A at LA[K1_SLAB_IX(e)]: LH_K1_SLAB_CHUNK consecutive entries of a lane share a 64-B line (the lane's next push and the sweep's next read hit the
    // line the last one touched), chunks interleaved by thread.  It was one entry per thread and row (LA[e * T]): with 17 of 64 lanes active, each on its own row, every
    // access was a line of its own — 28 % of pass 1's requests (profiles/r05_k1_request_floor.json)
    PEnt* const LA = slab + (size_t)LH_K1_SLAB_CHUNK * t;
    PEnt* const LB = slab + (size_t)(LH_MAXLEN + 2) * T + (size_t)LH_K1_SLAB_CHUNK * t;
    const int split_len = (int)(o.min_seed_len * o.split_factor + .499);
    int chunk_next = 0, chunk_end = 0;   // wave-uniform: the wave's current chunk of reads
    int st = S4_FETCH;
    int r = -1, len = 0, rst = 0, on = 0, ovf = 0;
    DIntv* out = intv_out;
    u64 c0 = 0, c1 = 0, c2 = 0, last_size = 0, p2mask = 0;   // c*: ik (forward, pass 3) or the prev entry p (backward); p2mask (pass 2): the read's long, rare SMEMs still to be re-seeded
    int cinfo = 0, ec = 0, i = 0, x = 0, j = 0, ncurr = 0, nprev = 0, min_intv = 1, last_mem_start = -1, ret = 0;
    PEnt ce, pn;
    ce.lo = ce.hi = pn.lo = pn.hi = 0;
    i64 run_p = 0;            // unique run: text position of the match's first base
    u64 ld64 = 0;             // a table value in flight
    uint32_t tw0 = 0, tw1 = 0, tw2 = 0, tw_sh = 0;   // the text words of the run's current 16-base window
    const bool runs = DO1 && ix.isa != nullptr;   // unique runs need min_intv == 1: pass 1 only
    // sweep filter (see FWD_PUSH_OK): the key of the LH_BLOOM_K read bases that end where the current forward interval ends,
    // the filter word read for it, the bits it must have; filt_from = first interval end for which that window is all bases
    const bool filt = DO12 && ix.bloom1 != nullptr && o.min_seed_len >= LH_BLOOM_K;
    u64 wkey = 0, bword = 0, bmask = 0;
    int filt_from = 0;
    unsigned n_ext_total = 0, n_exec_total = 0, n_ktree_total = 0, n_bt_total = 0;   // bwt_extend calls of the reference accounted for / really executed on the occurrence table / read from the k-mer tree
    // k-mer tree table (DIndex::ktree): a bwt_extend whose result is a match of at most ktl bases is ONE 16-B table read (the result
    // is a function of the matched string alone).  fcode = the bases of the current forward / pass-3 walk from x on (base t at
    // bits 2t), kept while the walk is that short; rcode = the 16 read bases from the current backward row's position i on.
    const PEnt* const kt = (const PEnt*)ix.ktree;
    const int ktl = kt ? ix.ktree_levels : 0;
    uint32_t fcode = 0, rcode = 0;
    // collapsed sweeps (pass 1): emin = end of the first entry pushed to the forward list (the shortest string), tri = the unique run in
    // progress stands for a list of several entries, tri_failed = it was tried for this SMEM and did not apply
    int emin = 0;
    uint32_t pbits = 0;   // pass 2, a call that passed the probe (S4_P2_PROBE): bit k = the LH_BLOOM_K-mer ending at read position x + 1 + k occurs again in the text
    // runs and calls by text (pass 1).  rflags: RF_RUNP = run_p is the text position of read base x in this call (a forward run or a call
    // by text set it: the backward run needs no suffix-array read), RF_BT = this call is being decided by text, RF_C1UNK = the current
    // entry's second row bound c1 is not known (a run made it), RF_PK = the read has a known locus (bits 8.. = the length of the unique
    // match that named it: a longer one replaces it — chance matches have short runs).  Pk = text position of read base 0 at that locus; bt_skip = a start position at which a call by text is known to fail (the base there differs from the text, or it was tried).
    const bool by_text = runs && filt && ix.plcp != nullptr;   // (like the sweep filter this leaves bwt_extend calls of the reference out of n_ext: off when the filter is off)
    int rflags = 0;   // + bits 16..23: bt_skip + 1, bits 24..31: x_prev + 2 = where the read's previous bwt_smem1a call started (see (B'))
#define BT_SKIP() (((rflags >> 16) & 0xff) - 1)
#define SET_BT_SKIP(v_) (rflags = (rflags & ~0xff0000) | (((v_) + 1) & 0xff) << 16)
#define X_PREV() ((int)((uint32_t)rflags >> 24) - 2)
#define SET_X_PREV(v_) (rflags = (rflags & 0xffffff) | (int)((uint32_t)((v_) + 2) << 24))
    i64 Pk = 0;
#define RF_RUNP 1
#define RF_BT 2
#define RF_C1UNK 4
#define RF_PK 8
#define RF_CURA 16     // the list being filled is list A
#define RF_REV 32      // prev is a forward list: walked from its end (longest match first)
#define RF_TRI 64      // the unique run in progress stands for a list of several entries (collapsed sweep)
#define RF_TRIF 128    // ... it was tried for this call and did not apply
    // The blocks that several states lead to — the start of a bwt_smem1a call, the start of a backward row, the next step of a
    // forward walk — exist ONCE, between the transitions and the extensions of a turn; a state that needs one sets its bit in todo and
    // parks the lane (S4_PENDING) instead of carrying a copy of the block (the compiler pays for every copy, and for every level of
    // nesting around it, with register moves on the paths of ALL lanes).
    int todo = 0;
    // sixteen read bases from s_ on, 2 bits each (base t at bits 2t; non-bases squeeze to arbitrary digits: callers mask)
#define CODE16(s_, out_)                                                                                     \
    {                                                                                                        \
        uint32_t a_, b_;                                                                                     \
        Q8(s_, a_) Q8((s_) + 8, b_)                                                                          \
        a_ &= 0x33333333u; b_ &= 0x33333333u;                                                                \
        a_ = (a_ | a_ >> 2) & 0x0f0f0f0fu; a_ = (a_ | a_ >> 4) & 0x00ff00ffu; a_ = (a_ | a_ >> 8) & 0xffffu; \
        b_ = (b_ | b_ >> 2) & 0x0f0f0f0fu; b_ = (b_ | b_ >> 4) & 0x00ff00ffu; b_ = (b_ | b_ >> 8) & 0xffffu; \
        out_ = a_ | b_ << 16;                                                                                \
    }
#define QB(i_) ((int)((qn[((i_) >> 3) * 64 + lane] >> (((i_) & 7) * 4)) & 0xF))
    // eight read bases from index s_ on (s_ may be negative or run past the read: those read as 4 = never equal to a text base)
#define Q8(s_, out_)                                                                                         \
    {                                                                                                        \
        int w_ = (s_) >> 3, sh_ = ((s_) & 7) * 4;                                                            \
        uint32_t lo_ = (w_ >= 0 && w_ < 32) ? qn[w_ * 64 + lane] : 0x44444444u;                              \
        uint32_t hi_ = (w_ + 1 >= 0 && w_ + 1 < 32) ? qn[(w_ + 1) * 64 + lane] : 0x44444444u;                \
        out_ = sh_ ? (lo_ >> sh_) | (hi_ << (32 - sh_)) : lo_;                                               \
    }
    // issue the reads of the sixteen text bases from position p_ on (p_ >= -16); T16_A() / T16_B() assemble the first / the second
    // eight once they have arrived
#define T16_LOAD(p_) { i64 w_ = (p_) >> 3; tw_sh = (uint32_t)((p_) & 7) * 4; tw0 = ix.tn[w_]; tw1 = ix.tn[w_ + 1]; tw2 = ix.tn[w_ + 2]; K1_REQ(K1T_TEXT, ix.tn + w_, 12) }
#define T16_A() (tw_sh ? (tw0 >> tw_sh) | (tw1 << (32 - tw_sh)) : tw0)
#define T16_B() (tw_sh ? (tw1 >> tw_sh) | (tw2 << (32 - tw_sh)) : tw1)
#define CURR ((rflags & RF_CURA) ? LA : LB)
#define PREV ((rflags & RF_CURA) ? LB : LA)
#define K1_SLAB_IX(e_) (((uint32_t)(e_) / LH_K1_SLAB_CHUNK) * (LH_K1_SLAB_CHUNK * T) + ((uint32_t)(e_) % LH_K1_SLAB_CHUNK))
#define PREV_AT(e_) PREV[K1_SLAB_IX(e_)]
    // forward extension: the next base decides between another bwt_extend and the end of the forward list
#define FWD_ADVANCE()                                                                                        \
    {                                                                                                        \
        int b_ = i < len ? QB(i) : 4;                                                                        \
        if (b_ > 3) {   /* end of read or ambiguous base: the current interval closes the forward list */    \
            ce = pe_pack(c0, c1, c2, cinfo);   /* the list's last entry is only ever read through ce */      \
            ncurr++;                                                                                         \
            st = S4_BWD_INIT;                                                                                \
        } else { ec = 3 - b_; st = S4_REQ_FWD; }                                                             \
    }
    // Sweep filter.  The backward sweep of bwt_smem1a extends EVERY interval of the forward list to the left until it dies,
    // but an interval [x, e) of the list can only give a MEM of min_seed_len bases or more if the LH_BLOOM_K-mer that ends
    // at e occurs min_intv times in the text: whatever the sweep emits for it contains that k-mer and has at least min_intv
    // occurrences.  If the filter (all k-mers for min_intv == 1, k-mers occurring twice or more otherwise) says it does not,
    // the interval is left out of the list: its rows would only have produced MEMs shorter than min_seed_len, which are
    // not kept, and the rows of the other intervals do not depend on it (an interval dies no later than the shorter ones
    // after it; only the first to die in a row is looked at; an interval that merged with a dropped one has the same rows
    // from there on).  A false positive of the filter only means the interval is swept as before.  The list's last
    // interval is always kept.  bword / bmask belong to the CURRENT interval; they are read one turn before they are used.
#define BLOOM_ISSUE()                                                                                        \
    {                                                                                                        \
        bword = 0; bmask = filt ? 1 : 0;   /* no filter: keep; window not all bases (or read start): drop */ \
        if (pbits && cinfo - x <= LH_BLOOM_K) bword = (pbits >> (cinfo - x - 1)) & 1;   /* pass 2 after the probe: the text's own bit for this window (exact: no filter read) */ \
        else if (filt && cinfo >= filt_from) {                                                               \
            uint32_t w_;                                                                                     \
            if (min_intv == 1) { dev_bloom_slot(wkey, ix.bloom1_words, &w_, &bmask); bword = ix.bloom1[w_]; K1_REQ(K1T_BLOOM1, ix.bloom1 + w_, 8) } \
            else { dev_bloom_slot(wkey, ix.bloom2_words, &w_, &bmask); bword = ix.bloom2[w_]; K1_REQ(K1T_BLOOM2, ix.bloom2 + w_, 8) }              \
        }                                                                                                    \
    }
#define FWD_PUSH_OK() ((bword & bmask) == bmask)
    // wkey for the LH_BLOOM_K bases that end at read position p_ (positions before the read count as non-bases); last_ = the last
    // non-base at or before p_ (p_ - LH_BLOOM_K if there is none in the window)
#define WKEY_AT(p_, last_)                                                                                   \
    {                                                                                                        \
        uint32_t w0_, w1_, w2_;                                                                              \
        Q8((p_) - 18, w0_) Q8((p_) - 10, w1_) Q8((p_) - 2, w2_)                                              \
        w2_ &= 0xfffu;                                                                                       \
        uint32_t n0_ = w0_ & 0x44444444u, n1_ = w1_ & 0x44444444u, n2_ = w2_ & 0x444u;                       \
        last_ = n2_ ? (p_) - 2 + ((31 - __clz((int)n2_)) >> 2) : n1_ ? (p_) - 10 + ((31 - __clz((int)n1_)) >> 2) \
              : n0_ ? (p_) - 18 + ((31 - __clz((int)n0_)) >> 2) : (p_) - 19;                                 \
        uint32_t t0_ = w0_ & 0x33333333u, t1_ = w1_ & 0x33333333u, t2_ = w2_ & 0x333u;                       \
        t0_ = (t0_ | t0_ >> 2) & 0x0f0f0f0fu; t0_ = (t0_ | t0_ >> 4) & 0x00ff00ffu; t0_ = (t0_ | t0_ >> 8) & 0xffffu; \
        t1_ = (t1_ | t1_ >> 2) & 0x0f0f0f0fu; t1_ = (t1_ | t1_ >> 4) & 0x00ff00ffu; t1_ = (t1_ | t1_ >> 8) & 0xffffu; \
        t2_ = (t2_ | t2_ >> 2) & 0x0f0fu; t2_ = (t2_ | t2_ >> 4) & 0x3fu;                                    \
        wkey = (u64)t0_ | (u64)t1_ << 16 | (u64)t2_ << 32;                                                   \
    }
    // START OF A bwt_smem1a CALL.
    // CALL BY TEXT (pass 1, with the read's locus P known from an earlier unique run of this read).  Let b be the end of the match of
    // read[x ..] with the text at P + x (found by comparing: the forward run), u the start of the match of read[.. x) with the text at
    // P going left (the backward run).  (A) If the suffix of the text at P + x shares fewer than b - x bases with every other suffix
    // (one PLCP byte), read[x, b) occurs only at P and read[x, b] nowhere: the forward walk of bwt_smem1a runs to b exactly (its return
    // value), and the forward list is [x, e) for some ends e < b plus the unique [x, b).  In the sweep, every row down to u leaves the
    // longest entry alive (it occurs at P), so nothing is emitted there; at row u - 1 it dies and becomes the MEM [u, b).  The other
    // entries [u, e) can only emit something more if one of them survives row u - 1, i.e. if read[u - 1, e) occurs in the text — not at
    // P (the base before u differs there), so read[u, e) would have to occur somewhere else as well.  (B) If the suffix at P + u shares at
    // most x - u bases with every other suffix, no [u, e) with e > x does: the call yields exactly [u, b) at one occurrence, whose text
    // position is P + u.  When the left comparison stops at the start of the read or at a non-base, bwt_smem1a emits only the longest
    // entry anyway (its c < 0 rule) and (B) is not needed.  (B') Nor is it when u - 1 is where the read's PREVIOUS call started: that call
    // returned the end of the longest match of read[u - 1 ..] anywhere in the text, which is <= x (this call starts where it returned), so
    // read[u - 1, e) with e > x occurs nowhere — the usual case: a substitution at u - 1, the call from there runs into a chance match, the
    // next one is this.  If (A) or (B) does not hold, the call is made as written, from x.
    // FORWARD JUMP.  While the walk's interval ends before filt_from nothing can be pushed to the forward list (the window of the
    // sweep filter is not all bases: the start of the read, usually), so the steps up to level J = min(tree depth, filt_from - x,
    // valid bases from x) only matter through the interval they arrive at: the tree's entry for the J bases, one read.  Sizes do
    // not grow along the walk, so "size at level J >= min_intv" is bwt_smem1a's loop condition for every skipped step; otherwise
    // the walk ends inside the skipped levels and is made step by step.  (A size of 1 reached inside them: the unique run that
    // would have started there finds the same end from level J.)
#define START_SMEM1()                                                                                        \
    {                                                                                                        \
        int s_ = QB(x);                                                                                      \
        c0 = ix.L2[s_] + 1; c2 = ix.L2[s_ + 1] - ix.L2[s_]; c1 = ix.L2[3 - s_] + 1; cinfo = x + 1;          \
        ncurr = 0; i = x + 1; fcode = (uint32_t)s_;                                                          \
        rflags = (rflags & ~(RF_RUNP | RF_BT | RF_C1UNK | RF_TRI | RF_TRIF)) | RF_CURA;                      \
        if (DO1 && by_text && (rflags & RF_PK) && x != BT_SKIP() && Pk + x >= 0 && (u64)(Pk + x) < ix.seq_len) st = S4_BT_INIT; \
        else {                                                                                               \
        if (filt) { int last_; WKEY_AT(x, last_) filt_from = last_ + 1 + LH_BLOOM_K; }                       \
        int J_ = 0;                                                                                          \
        if (filt && ktl > 1) {                                                                               \
            uint32_t a_, b_;                                                                                 \
            Q8(x, a_) Q8(x + 8, b_)                                                                          \
            a_ &= 0x44444444u; b_ &= 0x44444444u;                                                            \
            int v_ = a_ ? (__ffs((int)a_) - 1) >> 2 : 8 + (b_ ? (__ffs((int)b_) - 1) >> 2 : 8);   /* valid bases from x on */ \
            J_ = ktl < v_ ? ktl : v_;                                                                        \
            const int lim_ = pbits ? __ffs((int)pbits) : filt_from - x;   /* the first level whose interval could be pushed */ \
            if (lim_ < J_) J_ = lim_;                                                                        \
        }                                                                                                    \
        if (J_ >= 2) {                                                                                       \
            uint32_t cd_;                                                                                    \
            CODE16(x, cd_)                                                                                   \
            cd_ &= (1u << (2 * J_)) - 1u;                                                                    \
            ld64 = (((1ull << (2 * J_)) - 4) / 3) + cd_; j = J_; rcode = cd_;                                \
            st = S4_FJUMP;                                                                                   \
        } else todo |= TD_FADV;                                                                              \
        }                                                                                                    \
    }
    // pass 3: next base of the forward-only walk
#define P3_ADVANCE()                                                                                         \
    {                                                                                                        \
        if (i >= len) st = S4_READ_DONE;                                                                     \
        else {                                                                                               \
            int b_ = QB(i);                                                                                  \
            if (b_ > 3) { x = i + 1; st = S4_P3_SCAN; }                                                      \
            else { ec = 3 - b_; st = S4_REQ_P3; }                                                            \
        }                                                                                                    \
    }
    // start of a backward row at read position i: the first prev entry is in registers (ce)
#define BWD_ROW_BODY()                                                                                       \
    {                                                                                                        \
        int c_ = i < 0 ? 4 : QB(i);                                                                          \
        ncurr = 0; last_size = 0; j = 0;                                                                     \
        c0 = PE_X0(ce); c1 = PE_X1(ce); c2 = PE_X2(ce); cinfo = PE_INFO(ce);                                 \
        if (c_ > 3) {   /* nothing extends: only the first (longest) entry can be a new MEM */               \
            if ((rflags & (RF_C1UNK | RF_RUNP)) == (RF_C1UNK | RF_RUNP)) { c0 = LH_POSF | (u64)run_p; c1 = 0; }   /* a run's entry, first row of the sweep: by its position */ \
            st = S4_BWD_EMIT0;                                                                               \
        }                                                                                                    \
        else if (runs && nprev == 1 && c2 == 1 && min_intv == 1) { rflags &= ~RF_TRI; st = S4_BRUN_INIT; }   /* one unique match left */ \
        /* Several entries, the longest of them unique (the usual forward list: [x, e) for growing e until one occurrence is  \
           left).  Rows down to u, the position where the unique match ends on the left, cannot emit anything: every entry     \
           still matches at that occurrence.  If the SHORTEST entry's string [u, emin) is unique as well — the suffix of the    \
           text at u's position shares fewer than emin - u bases with every other suffix (PLCP array) — every entry has shrunk  \
           to that one occurrence by row u: equal sizes of nested occurrence sets are equal sets, bwt_smem1a keeps one interval \
           per size, so the list has collapsed into its longest entry, whose failure at u - 1 is the only thing the sweep       \
           reports.  That is the unique run below; otherwise the sweep is done row by row as written (tri_failed).  Like the    \
           sweep filter, this leaves bwt_extend calls of the reference out (no n_ext for them): off when the filter is off. */  \
        else if (by_text && (rflags & (RF_REV | RF_TRIF)) == RF_REV && nprev > 1 && c2 == 1 && min_intv == 1) { rflags |= RF_TRI; st = S4_BRUN_INIT; } \
        else {                                                                                               \
            ec = c_; st = S4_REQ_BWD;                                                                        \
            if (nprev > 1) { pn = PREV_AT((rflags & RF_REV) ? nprev - 2 : 1); K1_REQ(K1T_SLAB_R, &PREV_AT((rflags & RF_REV) ? nprev - 2 : 1), 16) }                                                \
            if (ktl) CODE16(i, rcode)                                                                        \
        }                                                                                                    \
    }
    // after prev entry j: the next entry of the row (prefetched), or the next row, or the end of this bwt_smem1a
#define BWD_ADVANCE()                                                                                        \
    {                                                                                                        \
        ++j;                                                                                                 \
        if (j < nprev) {                                                                                     \
            c0 = PE_X0(pn); c1 = PE_X1(pn); c2 = PE_X2(pn); cinfo = PE_INFO(pn);                             \
            st = S4_REQ_BWD;                                                                                 \
            if (j + 1 < nprev) { pn = PREV_AT((rflags & RF_REV) ? nprev - 2 - j : j + 1); K1_REQ(K1T_SLAB_R, &PREV_AT((rflags & RF_REV) ? nprev - 2 - j : j + 1), 16) }                                    \
        } else if (ncurr == 0) st = S4_SMEM_DONE;                                                            \
        else {                                                                                               \
            nprev = ncurr; --i;                                                                              \
            rflags = (rflags ^ RF_CURA) & ~(RF_RUNP | RF_C1UNK | RF_REV);   /* (rows made by bwt_extend carry both bounds; run_p no longer belongs to them) */ \
            if (i < -1) st = S4_SMEM_DONE;                                                                   \
            else { todo |= TD_ROW; st = S4_PENDING; }                                                        \
        }                                                                                                    \
    }
    // a backward-sweep interval that cannot be extended and is not contained in the previous MEM becomes a MEM
#define EMIT_MEM()                                                                                           \
    {                                                                                                        \
        if (cinfo - (i + 1) >= o.min_seed_len) {                                                             \
            DIntv m_; m_.x0 = c0; m_.x1 = c1; m_.x2 = c2; m_.info = (u64)(uint32_t)cinfo | (u64)(i + 1) << 32;  \
            if (P2TASK) {   /* the read's other calls run on other lanes: a slot from the read's counter */     \
                const int slot_ = atomicAdd(&n_intv[r], 1);                                                  \
                if (slot_ >= ICAP) ovf = 1; else out[slot_] = m_;                                            \
            } else if (on >= ICAP) ovf = 1;                                                                  \
            else { out[on] = m_; on++; K1_REQ(K1T_INTV_W, out + on - 1, 32) }                                \
        }                                                                                                    \
        last_mem_start = i + 1;                                                                              \
    }
    for (;;) {
        // ---- A. lanes without a read take the next ones of the wave's chunk; the wave stages their bases in LDS ----
        // Lanes that left the extension loops wait until LH_SLOW_BATCH of them have gathered (or nothing else is in flight):
        // the divergent blocks of A and B then run for many lanes at once instead of for one or two in every turn.
        const bool slow_turn = __popcll(__ballot(st == S4_FETCH || (st >= 8 && st < S4_FRUN_INIT))) >= (PASS == 3 ? LH_SLOW_BATCH_P3 : PASS == 2 ? LH_SLOW_BATCH_P2 : LH_SLOW_BATCH) ||
                               !__any((st >= S4_REQ_FWD && st < 8) || st >= S4_FRUN_INIT);
        u64 need = slow_turn ? __ballot(st == S4_FETCH) : 0;
        if (need) {
            int cnt = __popcll(need), newbase = 0;
            if (chunk_next + cnt > chunk_end) {
                int nb = 0;
                if (lane == 0) nb = atomicAdd(next_read, 64);
                newbase = wave_readlane(nb, 0);
            }
            i64 off = 0;
            int ln = 0, rr = -1;
            if (st == S4_FETCH) {
                int idx = chunk_next + lanes_below(need, lane);
                rr = idx < chunk_end ? idx : newbase + (idx - chunk_end);
                if (rr >= n_reads) st = S4_DONE;
                else {
                    if (BIG || big.list) rr = big.list[rr];
                    if (P2TASK) { task_k = rr & 15; rr >>= 4; }   // pass 2 by tasks: (read, j << 2 | m - 1): every m-th of the read's re-seeding calls, from the j-th on
                    off = seq_off[rr]; ln = (int)(seq_off[rr + 1] - off);
                }
            }
            if (chunk_next + cnt > chunk_end) { chunk_next = newbase + (chunk_next + cnt - chunk_end); chunk_end = newbase + 64; }
            else chunk_next += cnt;
            u64 got = __ballot(st == S4_FETCH);
            while (got) {   // four reads per round: their words are requested together, one memory latency per round
                int Lk[4], lnk[4];
                uint32_t wk[4];
                for (int k = 0; k < 4; ++k) {
                    Lk[k] = -1; lnk[k] = 0; wk[k] = 0;
                    if (got) {
                        Lk[k] = __ffsll((unsigned long long)got) - 1;
                        got &= got - 1;
                        const i64 offL = shfl_i64(off, Lk[k]);
                        lnk[k] = wave_readlane(ln, Lk[k]);
                        if (lnk[k] > LH_MAXLEN) lnk[k] = 0;
                        if (4 * lane < lnk[k]) __builtin_memcpy(&wk[k], seq + offL + 4 * lane, 4);   // the batch buffer is padded: a read's tail word is readable
                    }
                }
                for (int k = 0; k < 4; ++k) {
                    if (Lk[k] < 0) break;
                    uint32_t nb16 = 0;
                    for (int b = 0; b < 4; ++b) {
                        uint32_t v = (wk[k] >> (8 * b)) & 0xff;
                        v = (4 * lane + b < lnk[k] && v < 4) ? v : 4;
                        nb16 |= v << (4 * b);
                    }
                    uint32_t other = __shfl_xor(nb16, 1);
                    if (!(lane & 1)) qn[(lane >> 1) * 64 + Lk[k]] = nb16 | other << 16;
                }
            }
            if (st == S4_FETCH) {
                r = rr; len = ln; rst = 0; on = 0; ovf = 0; p2mask = 0; rflags = 0;
                K1_REQ(K1T_READS, seq + off, 64)   // (the read's bases, staged by the wave; its offsets and its resume record)
                K1_REQ(K1T_READS, seq_off + rr, 16)
                if (len > LH_MAXLEN) { rst |= LH_ST_TOO_LONG; len = 0; }
                out = BIG ? big.slab + (size_t)big.slot[r] * (2 * LH_BIG_INTV) : intv_out + (size_t)r * LH_MAX_INTV;
                if (!DO1) { on = n_intv[r]; rst |= status[r]; }   // continue behind the intervals of the earlier passes
                if (len >= o.min_seed_len) { x = 0; st = DO1 ? S4_P1_SCAN : DO2 ? S4_P2_NEXT : S4_P3_SCAN; }
                else st = S4_READ_DONE;
                if (DO1 && !BIG && big.resume) {   // k_smem_first made the read's first call: its interval (if any) is stored, the calls go on from x
                    const K1Resume rs = big.resume[r];
                    if (rs.x > 0 && st == S4_P1_SCAN) {
                        x = rs.x; on = n_intv[r]; SET_X_PREV(((rs.pk >> 8) & 0xff) - 2); SET_BT_SKIP(((rs.pk >> 16) & 0xff) - 1);
                        if (rs.pk & 0xff) { Pk = rs.Pk; rflags |= RF_PK | (rs.pk & 0xff) << 8; }
                    }
                }
                if (P3TEXT && st == S4_P3_SCAN) {   // the two longest unique SMEMs among the read's intervals
                    uspan = 0;
                    for (int k = 0; k < on; ++k) {
                        DIntv p = out[k];
                        const int ps = (int)(p.info >> 32), pe = (int)(uint32_t)p.info;
                        if (p.x2 != 1 || pe - ps < Lw) continue;
                        if (pe - ps > UE0 - US0) { uspan = (uspan << 16) | (uint32_t)ps | (uint32_t)pe << 8; up1 = up0; up0 = p.x0; }
                        else if (pe - ps > UE1 - US1) { uspan = (uspan & 0xffffu) | (uint32_t)ps << 16 | (uint32_t)pe << 24; up1 = p.x0; }
                    }
                    if (UE0 > US0) st = S4_P3_PREP;
                }
                if (DO2 && st == S4_P2_NEXT) {   // the long, rare SMEMs among pass 1's intervals (EMIT_MEM's test)
                    const int on1 = P2TASK ? big.slot[r] : on;   // (by tasks: the read's other calls may have emitted already: the SMEMs to re-seed inside are PASS 1's — k_p2_tasks noted how many there are)
                    int ord = 0;
                    for (int k = 0; k < on1; ++k) {
                        DIntv p = out[k];
                        if ((int)(uint32_t)p.info - (int)(p.info >> 32) >= split_len && p.x2 <= (u64)o.split_width) {
                            if (!P2TASK || ord % ((task_k & 3) + 1) == (task_k >> 2)) p2mask |= 1ull << k;   // this lane's share of the read's calls
                            ++ord;
                        }
                    }
                }
            }
        }
        // ---- B. transitions between the loops of mem_collect_intv (rare per lane; the blocks are ordered so that the
        //         usual chains finish in one pass) ----
        if (st >= S4_FRUN_INIT) {   // ONE step of the load / use chains per turn: a value read here is used in the next turn
            if (DO1 && st == S4_FRUN_INIT) { ld64 = ix.sa[c0]; K1_REQ(K1T_SA, ix.sa + c0, 8) st = S4_FRUN_INIT2; }
            else if (DO1 && st == S4_FRUN_INIT2) {
                run_p = (i64)ld64; T16_LOAD(run_p + (i - x))
                rflags |= RF_RUNP;
                st = S4_REQ_FRUN;
            }
            else if (DO1 && st == S4_FRUN_END) {
                // The unique interval closes the forward list (it is the list's last entry: kept in ce only).  Its second row bound is the
                // row of the reverse strand's copy of the match: only a sweep made by bwt_extend needs it — never when the run's interval
                // is the whole list or the list can collapse (BWD_ROW_BODY); it is read back if that does not work out (S4_TRI_C1).  Without the
                // PLCP array the sweep reports rows: both bounds, as before.
                if (by_text) {
                    rflags |= RF_C1UNK;
                    ce = pe_pack(c0, 0, c2, cinfo);
                    ncurr++;
                    st = S4_BWD_INIT;
                } else { ld64 = ix.isa[(i64)ix.seq_len - (run_p + (i - x))]; K1_REQ(K1T_ISA, ix.isa + ((i64)ix.seq_len - (run_p + (i - x))), 8) st = S4_FRUN_END2; }
            }
            else if (DO1 && st == S4_FRUN_END2) {
                c1 = ld64;
                ce = pe_pack(c0, c1, c2, cinfo);
                ncurr++;
                st = S4_BWD_INIT;
            }
            else if (DO1 && st == S4_BT_INIT) {   // a call by text: the comparison starts at x, the PLCP byte of (A) is read beside the first text words
                run_p = Pk + x;
                T16_LOAD(run_p)
                ec = ix.plcp[run_p]; K1_REQ(K1T_PLCP, ix.plcp + run_p, 1)
                i = x; last_mem_start = -1; rflags |= RF_BT | RF_RUNP;
                st = S4_REQ_FRUN;
            }
            else if (DO1 && st == S4_BT_B) {   // (B): ec = the PLCP byte at u's position; i + 1 = u
                if (ec <= x - (i + 1)) { c0 = LH_POSF | (u64)run_p; c1 = 0; c2 = 1; rflags &= ~RF_BT; n_bt_total++; st = S4_BWD_EMIT0; }
                else { rflags &= ~(RF_RUNP | RF_BT | RF_C1UNK); SET_BT_SKIP(x); todo |= TD_SMEM; st = S4_PENDING; }   // not provable: the call as written
            }
            else if (DO12 && st == S4_FJUMP) { pn = kt[ld64]; K1_REQ(K1T_TREE, kt + ld64, 16) st = S4_FJUMP2; }
            else if (DO12 && st == S4_FJUMP2) {
                if (PE_X2(pn) >= (u64)min_intv) {   // as if the bwt_extend steps up to level j had been made
                    c0 = PE_X0(pn); c1 = PE_X1(pn); c2 = PE_X2(pn);
                    n_ext_total += j - 1; n_ktree_total++;
                    i = x + j; cinfo = i; fcode = rcode;
                    if (runs && c2 == 1 && min_intv == 1) st = S4_FRUN_INIT;
                    else { todo |= TD_FADV | TD_KEY; st = S4_PENDING; }
                } else { todo |= TD_FADV; st = S4_PENDING; }   // the walk ends inside the skipped levels: step by step from level 1
            }
            else if (DO2 && st == S4_P2_PROBE) {
                if (ld64 & LH_POSF) ld64 &= ~LH_POSF;   // pass 1 stored the SMEM by its position
                else ld64 = ix.sa[ld64];
                st = S4_P2_PROBE2;
            }
            else if (DO2 && st == S4_P2_PROBE2) {   // the bits of the K windows' first positions: text position of the SMEM's start + i on
                const u64 t0 = ld64 + (u64)i;
                pn.lo = ix.rep_t[t0 >> 6]; pn.hi = ix.rep_t[(t0 >> 6) + 1]; tw_sh = (uint32_t)(t0 & 63);
                st = S4_P2_PROBE3;
            }
            else if (DO2 && st == S4_P2_PROBE3) {
                const u64 bits = tw_sh ? (pn.lo >> tw_sh) | (pn.hi << (64 - tw_sh)) : pn.lo;
                pbits = (uint32_t)(bits & ((1ull << LH_BLOOM_K) - 1));   // the walk's first LH_BLOOM_K filter decisions, and how far it may jump
                if (pbits) { todo |= TD_SMEM; st = S4_PENDING; }
                else st = S4_P2_NEXT;
            }
            else if (P3T && st == S4_P3_PREP) {
                if (up0 & LH_POSF) ld64 = up0 & ~LH_POSF;
                else ld64 = ix.sa[up0];
                st = S4_P3_PREP2;
            }
            else if (P3T && st == S4_P3_PREP2) {
                up0 = ld64;
                if (UE1 > US1) {
                    if (up1 & LH_POSF) ld64 = up1 & ~LH_POSF;
                    else ld64 = ix.sa[up1];
                    st = S4_P3_PREP3;
                }
                else st = S4_P3_SCAN;
            }
            else if (P3T && st == S4_P3_PREP3) { up1 = ld64; st = S4_P3_SCAN; }
            else if (P3T && st == S4_P3_T0) {   // up to four walks from run_p on, Lw apart, as long as they lie inside the SMEM (j of them): their PLCP bytes
                const uint8_t* pl = ix.plcp + run_p;
                uint32_t v = pl[0];
                if (j > 1) v |= (uint32_t)pl[Lw] << 8;
                if (j > 2) v |= (uint32_t)pl[2 * Lw] << 16;
                if (j > 3) v |= (uint32_t)pl[3 * Lw] << 24;
                tw0 = v;
                st = S4_P3_T1;
            }
            else if (P3T && st == S4_P3_T1) {
                st = S4_P3_SCAN;
                for (int k = 0; k < j; ++k) {
                    if ((int)((tw0 >> (8 * k)) & 0xff) < Lw) {   // unique: the walk ends here with one occurrence
                        if (on >= ICAP) ovf = 1;
                        else { DIntv m; m.x0 = LH_POSF | (u64)(run_p + (i64)k * Lw); m.x1 = 0; m.x2 = 1; m.info = (u64)x << 32 | (u64)(x + Lw); out[on++] = m; }
                        n_ext_total += (unsigned)(Lw - 1);
                        x += Lw;
                    } else { rst |= P3_NOTEXT; break; }   // it occurs again: the walk as written, from x
                }
                if (!(rst & P3_NOTEXT)) {   // the next walks of the SMEM (no non-base inside one): straight on
                    const int in0 = x >= US0 && x + Lw <= UE0, in1 = x >= US1 && x + Lw <= UE1;
                    if (in0 || in1) {
                        run_p = in0 ? (i64)up0 + (x - US0) : (i64)up1 + (x - US1);
                        const int room = ((in0 ? UE0 : UE1) - x) / Lw;
                        j = room < 4 ? room : 4;
                        st = S4_P3_T0;
                    }
                }
            }
            else if (DO3 && st == S4_P3_JUMP) { pn = kt ? kt[ld64] : ((const PEnt*)ix.kmer12)[ld64]; st = S4_P3_JUMP2; }
            else if (DO3 && st == S4_P3_JUMP2) {   // as if the bwt_extend steps after the first base had been made (none of them can end the walk: i - x < min_seed_len)
                const int jl = kt ? ktl : LH_KMER;
                c0 = PE_X0(pn); c1 = PE_X1(pn); c2 = PE_X2(pn);
                n_ext_total += jl - 1;
                i = x + jl;
                P3_ADVANCE()
            }
            else if (DO1 && st == S4_BRUN_INIT) {
                if (rflags & RF_RUNP) { T16_LOAD(run_p - 16) st = S4_REQ_BRUN; }   // the forward run (or the call by text) left the position of read base x
                else { ld64 = ix.sa[c0]; K1_REQ(K1T_SA, ix.sa + c0, 8) st = S4_BRUN_INIT2; }
            }
            else if (DO1 && st == S4_BRUN_INIT2) { run_p = (i64)ld64; T16_LOAD(run_p - 16) st = S4_REQ_BRUN; }
            else if (DO1 && st == S4_BRUN_END) {   // the row at i could not extend the one interval left: it is a MEM unless contained in the previous one
                const bool edge = i < 0 || QB(i) > 3;   // stopped by the start of the read or a non-base: bwt_smem1a emits the longest entry only
                if (rflags & RF_BT) {
                    if (edge || i == X_PREV()) { c0 = LH_POSF | (u64)run_p; c1 = 0; c2 = 1; rflags &= ~RF_BT; n_bt_total++; st = S4_BWD_EMIT0; }   // (B') see START_SMEM1
                    else { ec = ix.plcp[run_p]; K1_REQ(K1T_PLCP, ix.plcp + run_p, 1) st = S4_BT_B; }
                } else if ((rflags & RF_TRI) && !edge) { ec = ix.plcp[run_p]; K1_REQ(K1T_PLCP, ix.plcp + run_p, 1) st = S4_TRI_LCP2; }
                else if (by_text) { c0 = LH_POSF | (u64)run_p; c1 = 0; st = S4_BWD_EMIT0; }   // one occurrence, at run_p
                else { ld64 = ix.isa[run_p]; K1_REQ(K1T_ISA, ix.isa + run_p, 8) st = S4_TRI_C1B; }   // (no PLCP array: the row, as before)
            }
            else if (DO1 && st == S4_TRI_LCP2) {   // ec = bases the suffix at u's position shares with another suffix (u = i + 1)
                if (ec < emin - (i + 1)) {   // the shortest entry is unique from u on: the list is its longest entry
                    c0 = LH_POSF | (u64)run_p; c1 = 0;
                    st = S4_BWD_EMIT0;   // a MEM unless contained in the previous one, then the call is over
                } else {
                    // Not the whole list, but its tail: an entry [x, e) with e - u > ec is unique from row u on, so by then it has merged
                    // into the entry before it (equal sizes of nested occurrence sets are equal sets, and stay equal in later rows); an
                    // entry that is NOT unique at row u was never merged into one that is.  Down to row u nothing is emitted, so the
                    // sweep over the longest entry and the entries that may still be distinct at row u leaves the same list there as
                    // the sweep over all of them.  The list's ends are distinct and ascending from emin: at most u + ec - emin + 1 of
                    // them are <= u + ec, the first ones of the forward list.  The sweep as written over those, from its first row —
                    // with the longest entry's second row bound, which the run did not keep: the row of the reverse strand's copy.
                    const int m_ = i + 1 + ec - emin + 1;
                    if (m_ < nprev - 1) nprev = m_ + 1;
                    ld64 = ix.isa[(i64)ix.seq_len - (run_p + (PE_INFO(ce) - (i + 1)))]; K1_REQ(K1T_ISA, ix.isa + ((i64)ix.seq_len - (run_p + (PE_INFO(ce) - (i + 1)))), 8)
                    st = S4_TRI_C1;
                }
            }
            else if (DO1 && st == S4_TRI_C1) {
                ce = pe_pack(PE_X0(ce), ld64, PE_X2(ce), PE_INFO(ce));
                rflags = (rflags | RF_TRIF) & ~(RF_TRI | RF_RUNP | RF_C1UNK);
                i = x - 1;
                todo |= TD_ROW; st = S4_PENDING;
            }
            else if (DO1 && st == S4_TRI_C1B) { c0 = ld64; st = S4_BWD_EMIT0; }
        }
        while (slow_turn && __any(st >= 8 && st < S4_FRUN_INIT)) {
            if (DO12 && st == S4_BWD_EMIT) {
                EMIT_MEM()
                BWD_ADVANCE()
            }
            if (DO12 && st == S4_BWD_INIT) {   // the forward list becomes prev and is walked from its end (longest match first)
                ret = cinfo; rflags = (rflags ^ RF_CURA) | RF_REV; nprev = ncurr; last_mem_start = -1; i = x - 1;
                todo |= TD_ROW; st = S4_PENDING;
            }
            if (DO12 && st == S4_BWD_EMIT0) {
                if (last_mem_start < 0 || i + 1 < last_mem_start) EMIT_MEM()
                st = S4_SMEM_DONE;
            }
            if (DO12 && st == S4_SMEM_DONE) {
                if (DO1) { SET_X_PREV(x); x = ret; st = S4_P1_SCAN; }
                else st = S4_P2_NEXT;
            }
            if (DO1 && st == S4_P1_SCAN) {   // first pass: all SMEMs
                while (x < len && QB(x) > 3) ++x;
                if (x >= len) st = S4_READ_DONE;
                else { min_intv = 1; pbits = 0; todo |= TD_SMEM; st = S4_PENDING; }
            }
            if (DO2 && st == S4_P2_NEXT) {   // second pass: re-seed inside long, rare SMEMs of the first pass
                st = S4_READ_DONE; x = 0;
                while (p2mask) {
                    int k = __ffsll((unsigned long long)p2mask) - 1;
                    p2mask &= p2mask - 1;
                    DIntv p = out[k];
                    int xm = ((int)(p.info >> 32) + (int)(uint32_t)p.info) >> 1;
                    if (QB(xm) > 3) continue;   // bwt_smem1a returns at once on an ambiguous base
                    x = xm; min_intv = (int)p.x2 + 1;
                    // Every MEM this bwt_smem1a call can contribute covers xm, is min_seed_len >= LH_BLOOM_K bases long and occurs
                    // min_intv >= 2 times: it contains one of the LH_BLOOM_K-mers [w, w + K) with xm - K < w <= xm, which then occurs twice
                    // or more.  If those windows all lie inside the SMEM, the read equals the text there (at any occurrence: the first
                    // row's), and the text's bits say whether a k-mer occurs again.  None does (the usual case in unique sequence): the
                    // call cannot contribute and is left out, like the intervals the sweep filter drops (off when the filter is off).
                    if (filt && ix.rep_t && xm - (LH_BLOOM_K - 1) >= (int)(p.info >> 32) && xm + LH_BLOOM_K <= (int)(uint32_t)p.info) {
                        ld64 = p.x0; i = xm - (LH_BLOOM_K - 1) - (int)(p.info >> 32);
                        st = S4_P2_PROBE;
                    } else { pbits = 0; todo |= TD_SMEM; st = S4_PENDING; }
                    break;
                }
            }
            while (DO3 && st == S4_P3_SCAN) {   // third pass: LAST-like forward-only seeds (bwt_seed_strategy1)
                while (x < len && QB(x) > 3) ++x;
                if (x >= len || o.max_mem_intv <= 0) st = S4_READ_DONE;
                else if (P3TEXT && !(rst & P3_NOTEXT) && ((x >= US0 && x + Lw <= UE0) || (x >= US1 && x + Lw <= UE1))) {
                    const int in0 = x >= US0 && x + Lw <= UE0;
                    run_p = in0 ? (i64)up0 + (x - US0) : (i64)up1 + (x - US1);   // the text position the walk starts at
                    const int room = ((in0 ? UE0 : UE1) - x) / Lw;                // walks that fit into the SMEM from here
                    j = room < 4 ? room : 4;
                    st = S4_P3_T0;
                }
                else {
                    rst &= ~P3_NOTEXT;
                    int b = QB(x);
                    c0 = ix.L2[b] + 1; c2 = ix.L2[b + 1] - ix.L2[b]; c1 = ix.L2[3 - b] + 1;
                    i = x + 1;
                    int jumped = 0;
                    fcode = (uint32_t)b;
                    if (ktl > 1 && x + ktl <= len && o.min_seed_len >= ktl) {   // the walk's first ktl bases from the tree's deepest level
                        uint32_t w0, w1;
                        Q8(x, w0)
                        Q8(x + 8, w1)
                        if (ktl < 8) { w0 &= (1u << (4 * ktl)) - 1u; w1 = 0; }
                        else if (ktl < 16) w1 &= (1u << (4 * (ktl - 8))) - 1u;
                        if (!((w0 | w1) & 0x44444444u)) {
                            CODE16(x, fcode)
                            fcode &= (1u << (2 * ktl)) - 1u;
                            ld64 = (((1ull << (2 * ktl)) - 4) / 3) + fcode;
                            st = S4_P3_JUMP;
                            jumped = 1;
                        }
                    } else
                    if (ix.kmer12 && x + LH_KMER <= len && o.min_seed_len >= LH_KMER) {
                        uint32_t w0, w1;
                        Q8(x, w0)
                        Q8(x + 8, w1)
                        w1 &= 0xffffu;   // bases x+8 .. x+11
                        if (!((w0 | w1) & 0x44444444u)) {   // no ambiguous base among the twelve
                            // 4-bit -> 2-bit per base (base k of the 12-mer ends up at bits 2k..2k+1)
                            uint32_t t0 = w0 & 0x33333333u, t1 = w1 & 0x3333u;
                            t0 = (t0 | t0 >> 2) & 0x0f0f0f0fu; t0 = (t0 | t0 >> 4) & 0x00ff00ffu; t0 = (t0 | t0 >> 8) & 0xffffu;
                            t1 = (t1 | t1 >> 2) & 0x0f0fu; t1 = (t1 | t1 >> 4) & 0xffu;
                            ld64 = t0 | t1 << 16;
                            st = S4_P3_JUMP;
                            jumped = 1;
                        }
                    }
                    if (!jumped) P3_ADVANCE()
                }
            }
            if (st == S4_READ_DONE) {
                rst &= ~P3_NOTEXT;
                if (ovf) rst |= LH_ST_INTV_OVERFLOW;
                if (P2TASK) { if (rst) atomicOr(&status[r], rst); }   // (the counter is the calls' own; k_p2_clamp brings it back to the slots there are)
                else { n_intv[r] = on; status[r] = rst; }
                st = S4_FETCH;
            }
        }
        // ---- B'. the shared blocks (see todo) ----
        if (DO12 && (todo & TD_SMEM)) START_SMEM1()
        if (DO12 && (todo & TD_ROW)) BWD_ROW_BODY()
        if (DO12 && (todo & TD_FADV)) {
            if (todo & TD_KEY) { int last_; WKEY_AT(i - 1, last_) (void)last_; }
            BLOOM_ISSUE()
            FWD_ADVANCE()
        }
        todo = 0;
        // ---- C. anything left to extend? ----
        if (!__any(st >= S4_REQ_FWD && st < 8)) {   // parked lanes advance in A / B of the next turn
            if (!__any(st != S4_DONE)) break;
            continue;
        }
        // ---- D. the shared program point: one bwt_extend per requesting lane ----
        DIntv ok;
        ok.x0 = ok.x1 = ok.x2 = ok.info = 0;
        // Both kinds of lanes ISSUE their reads before either kind uses them: as an if / else the tree lanes' read would have to land
        // before the occurrence records of the other lanes are even requested (two memory latencies per turn instead of one).
        const bool req = st >= S4_REQ_FWD && st <= S4_REQ_P3;
        const int lnew = st == S4_REQ_BWD ? cinfo - i : i + 1 - x;   // bases of the match this bwt_extend yields: [i, cinfo) or [x, i]
        const bool by_tree = req && lnew <= ktl, by_occ = req && !by_tree;
        PEnt te; te.lo = te.hi = 0;
        uint4 hk = {0, 0, 0, 0}, dk = hk, hl = hk, dl = hk;
        u64 k2 = 0, l2 = 0;
        const u64 xa = st == S4_REQ_BWD ? c0 : c1;   // x[!is_back]
        if (by_tree) {
            uint32_t code;
            if (st == S4_REQ_BWD) code = rcode & ((1u << (2 * lnew)) - 1u);
            else { fcode |= (uint32_t)(3 - ec) << (2 * (lnew - 1)); code = fcode; }   // forward steps complement the base (ec = 3 - base)
            te = kt[(((1ull << (2 * lnew)) - 4) / 3) + code]; K1_REQ(K1T_TREE, kt + ((((1ull << (2 * lnew)) - 4) / 3) + code), 16)
        }
        if (by_occ) {   // dev_2occ4(xa - 1, xa - 1 + c2): the records of the interval's two ends (one read if they share it)
            const u64 k = xa - 1, l = xa - 1 + c2;
            l2 = l - (l >= ix.primary);
            const uint4* pl = ix.occ + ((l2 >> 6) << 1);
            hl = pl[0]; dl = pl[1]; K1_REQ(K1T_OCC, pl, 32)
            if (k != (u64)-1) {
                k2 = k - (k >= ix.primary);
                hk = hl; dk = dl;
                if ((k2 >> 6) != (l2 >> 6)) { const uint4* pk = ix.occ + ((k2 >> 6) << 1); hk = pk[0]; dk = pk[1]; K1_REQ(K1T_OCC, pk, 32) }
            }
        }
        if (by_tree) {
            ok.x0 = PE_X0(te); ok.x1 = PE_X1(te); ok.x2 = PE_X2(te);
            n_ktree_total++;
        }
        if (by_occ) {   // dev_extend_c on the records read above
            u64 tk[4], tl[4];
            if (xa - 1 == (u64)-1) { tk[0] = tk[1] = tk[2] = tk[3] = 0; }
            else dev_occ4_rec(ix, k2, hk, dk, tk);
            dev_occ4_rec(ix, l2, hl, dl, tl);
            const int isb = st == S4_REQ_BWD;
            const u64 xb = isb ? c1 : c0;
            u64 s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
            u64 acc = xb + ((xa <= ix.primary && xa + c2 - 1 >= ix.primary) ? 1 : 0);
            u64 o3 = acc, o2 = o3 + s3, o1 = o2 + s2, o0 = o1 + s1;
            u64 na = ec == 0 ? ix.L2[0] + 1 + tk[0] : ec == 1 ? ix.L2[1] + 1 + tk[1] : ec == 2 ? ix.L2[2] + 1 + tk[2] : ix.L2[3] + 1 + tk[3];
            u64 nb = ec == 0 ? o0 : ec == 1 ? o1 : ec == 2 ? o2 : o3;
            ok.x2 = ec == 0 ? s0 : ec == 1 ? s1 : ec == 2 ? s2 : s3;
            if (isb) { ok.x0 = na; ok.x1 = nb; } else { ok.x1 = na; ok.x0 = nb; }
            n_exec_total++;
        }
        if (req) n_ext_total++;
        // ---- E. bookkeeping of the loop the lane is in, and its next request ----
        if (DO12 && st == S4_REQ_FWD) {
            if (ok.x2 != c2) {
                ce = pe_pack(c0, c1, c2, cinfo);
                if (ok.x2 < (u64)min_intv) { st = S4_BWD_INIT; ncurr++; }   // the interval is too small to be extended further: ce is the list's last entry
                else if (FWD_PUSH_OK()) { if (!ncurr) emin = cinfo; CURR[K1_SLAB_IX(ncurr)] = ce; K1_REQ(K1T_SLAB_W, &CURR[K1_SLAB_IX(ncurr)], 16) ncurr++; }
            }
            if (st == S4_REQ_FWD) {
                c0 = ok.x0; c1 = ok.x1; c2 = ok.x2; cinfo = i + 1; ++i;
                wkey = wkey >> 2 | (u64)(3 - ec) << (2 * LH_BLOOM_K - 2);   // the base just matched enters the window
                if (runs && c2 == 1 && min_intv == 1) st = S4_FRUN_INIT;   // a single occurrence: follow it through the text (it ends the list: no filter)
                else { todo |= TD_FADV; st = S4_PENDING; }
            }
        } else if (DO12 && st == S4_REQ_BWD) {
            if (ok.x2 < (u64)min_intv) {
                if (ncurr == 0 && (last_mem_start < 0 || i + 1 < last_mem_start)) st = S4_BWD_EMIT;   // no longer match survived, not contained in the previous MEM
            } else if (ncurr == 0 || ok.x2 != last_size) {
                PEnt e = pe_pack(ok.x0, ok.x1, ok.x2, cinfo);
                if (ncurr == 0) ce = e;   // a row's first entry is only ever read through ce
                else { CURR[K1_SLAB_IX(ncurr)] = e; K1_REQ(K1T_SLAB_W, &CURR[K1_SLAB_IX(ncurr)], 16) }
                ncurr++;
                last_size = ok.x2;
            }
            if (st == S4_REQ_BWD) BWD_ADVANCE()
        } else if (DO1 && st == S4_REQ_FRUN) {   // forward unique run: up to sixteen successful bwt_extend steps at once
            uint32_t qa, qb;
            Q8(i, qa)
            Q8(i + 8, qb)
            uint32_t xa = qa ^ T16_A(), xb = qb ^ T16_B();
            int n = xa ? (__ffs((int)xa) - 1) >> 2 : 8 + (xb ? (__ffs((int)xb) - 1) >> 2 : 8);
            if (!(rflags & RF_BT)) n_ext_total += (unsigned)n;
            i += n;
            if (n == 16) T16_LOAD(run_p + (i - x))
            else if (rflags & RF_BT) {   // the call by text: i = b, ec = the PLCP byte at x's position
                if (ec < i - x) {   // (A): read[x, b) occurs only here: the forward walk returns b, its last entry is this one occurrence
                    cinfo = i; ret = i; c2 = 1; SET_BT_SKIP(i);
                    if (x == 0 || QB(x - 1) > 3) { i = x - 1; c0 = LH_POSF | (u64)run_p; c1 = 0; rflags &= ~RF_BT; n_bt_total++; st = S4_BWD_EMIT0; }   // nothing before x: bwt_smem1a emits the longest entry
                    else { i = x - 1; T16_LOAD(run_p - 16) st = S4_REQ_BRUN; }
                } else { rflags &= ~(RF_RUNP | RF_BT | RF_C1UNK); SET_BT_SKIP(x); todo |= TD_SMEM; st = S4_PENDING; }   // not provable (or the read leaves the locus at x): the call as written
            }
            else {   // read exhausted, ambiguous base, text exhausted or a mismatch: the run's interval is the forward list's last entry
                if (i < len && QB(i) <= 3) n_ext_total++;   // the bwt_extend that returned an empty interval
                cinfo = i;
                if (by_text && i - x > ((rflags >> 8) & 0xff)) {   // the longest unique match so far names the read's locus: later calls may be decided by text there
                    Pk = run_p - x; rflags = (rflags & ~0xff00) | RF_PK | (i - x) << 8;
                    SET_BT_SKIP(i);   // (a call from there finds the base that ended this run)
                } else if ((rflags & RF_PK) && run_p - x == Pk) SET_BT_SKIP(i);
                st = S4_FRUN_END;
            }
        } else if (DO1 && st == S4_REQ_BRUN) {   // backward unique run: rows in which the one interval left survives
            uint32_t qa, qb;
            Q8(i - 15, qa)
            Q8(i - 7, qb)
            uint32_t xa = qa ^ T16_A(), xb = qb ^ T16_B();
            int n = xb ? __clz((int)xb) >> 2 : 8 + (xa ? __clz((int)xa) >> 2 : 8);
            if (!(rflags & RF_BT)) n_ext_total += (unsigned)n;
            i -= n; run_p -= n;
            if (n == 16) T16_LOAD(run_p - 16)
            else {
                if (!(rflags & RF_BT) && i >= 0 && QB(i) <= 3) n_ext_total++;   // the bwt_extend that returned an empty interval
                rflags &= ~RF_RUNP;   // (run_p now names the start of the extended match)
                st = S4_BRUN_END;
            }
        } else if (DO3 && st == S4_REQ_P3) {
            if (ok.x2 < (u64)o.max_mem_intv && i - x >= o.min_seed_len) {
                if (ok.x2 > 0) {
                    if (on >= ICAP) ovf = 1;
                    else { DIntv m = ok; m.info = (u64)x << 32 | (u64)(i + 1); out[on++] = m; }
                }
                x = i + 1; st = S4_P3_SCAN;
            } else {
                c0 = ok.x0; c1 = ok.x1; c2 = ok.x2; ++i;
                P3_ADVANCE()
            }
        }
    }
#undef QB
#undef Q8
#undef CODE16
#undef T16_LOAD
#undef T16_A
#undef T16_B
#undef CURR
#undef PREV
#undef PREV_AT
#undef K1_SLAB_IX
#undef P3_NOTEXT
#undef US0
#undef UE0
#undef US1
#undef UE1
#undef START_SMEM1
#undef WKEY_AT
#undef BLOOM_ISSUE
#undef FWD_PUSH_OK
#undef FWD_ADVANCE
#undef P3_ADVANCE
#undef BWD_ROW_BODY
#undef BWD_ADVANCE
#undef EMIT_MEM
#undef RF_RUNP
#undef RF_BT
#undef RF_C1UNK
#undef RF_PK
#undef RF_CURA
#undef RF_REV
#undef RF_TRI
#undef RF_TRIF
#undef BT_SKIP
#undef SET_BT_SKIP
#undef X_PREV
#undef SET_X_PREV
    if (ctr) {
        unsigned tot = (unsigned)wave_sum_i32((int)n_ext_total);
        if (lane == 0 && tot) atomicAdd(&LH_CTR(ctr)->n_ext, (u64)tot);
        unsigned ex = (unsigned)wave_sum_i32((int)n_exec_total);
        if (lane == 0 && ex) atomicAdd(&LH_CTR(ctr)->n_ext_exec[PASS - 1], (u64)ex);
        unsigned kx = (unsigned)wave_sum_i32((int)n_ktree_total);
        if (lane == 0 && kx) atomicAdd(&LH_CTR(ctr)->n_ktree[PASS - 1], (u64)kx);
        unsigned bx = (unsigned)wave_sum_i32((int)n_bt_total);
        if (lane == 0 && bx) atomicAdd(&LH_CTR(ctr)->n_bt, (u64)bx);
    }
#ifdef LH_K1_TRACE
    if (PASS == 1 && !BIG && lh_k1_trace) lh_k1_trace_n[t] = trace_k;
#endif
}

#ifdef LH_K1_TRACE
// the recorded sequences again, with pass 1's launch geometry: lane t issues its k-th request when its (k - 1)-th has landed (the address is made to depend on
// the value read), the wave goes on when all of its lanes' requests have — a turn of the state machine without anything between the requests
__global__ void __launch_bounds__(64, LH_SMEM4_WAVES) k_k1_replay(const u64* __restrict__ trace, const uint32_t* __restrict__ n_req, uint32_t cap, u64* __restrict__ sink) {
    const uint32_t T = gridDim.x * 64u, t = blockIdx.x * 64u + (uint32_t)LANE();
    uint32_t n = n_req[t];
    if (n > cap) n = cap;
    uint32_t nmax = n;
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o_ = __shfl_xor(nmax, m); nmax = nmax > o_ ? nmax : o_; }
    u64 acc = 0;
    for (uint32_t k = 0; k < nmax; ++k) {
        if (k < n) {
            const u64 e = trace[(size_t)k * T + t];
            const int tab = (int)(e >> 56), bytes = (int)((e >> 48) & 0xff);
            const uintptr_t a = (uintptr_t)(e & 0xffffffffffffull) + (acc == 0x9e3779b97f4a7c15ull ? 64 : 0);   // (never true: but the address now waits for the last value)
            if (tab == K1T_SLAB_W) {   // the state machine's list stores: the same bytes to the same place (scratch that pass 1 has finished with; the
                uint4 v = {(uint32_t)acc, 1, 2, 3};   // interval array it leaves to the later passes is read instead of written)
                *(uint4*)a = v;
            } else if (bytes >= 32) { const uint4 v0 = ((const uint4*)a)[0], v1 = ((const uint4*)a)[1]; acc += v0.x + v1.w; }
            else if (bytes >= 16) { const uint4 v = *(const uint4*)a; acc += v.x + v.w; }
            else if (bytes == 12) { const uint32_t* q = (const uint32_t*)a; acc += q[0] + q[1] + q[2]; }
            else if (bytes == 8) acc += *(const u64*)a;
            else acc += *(const uint8_t*)a;
        }
    }
    if (acc == 0x9e3779b97f4a7c15ull) *sink = acc;
}
// requests and bytes per table
__global__ void __launch_bounds__(256) k_k1_trace_hist(const u64* __restrict__ trace, const uint32_t* __restrict__ n_req, uint32_t cap, uint32_t T, unsigned long long* __restrict__ hist) {
    __shared__ unsigned long long h[2 * K1T_N + 2];
    if (threadIdx.x < 2 * K1T_N + 2) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        uint32_t n = n_req[t];
        if (n > cap) { atomicAdd(&h[2 * K1T_N], (unsigned long long)(n - cap)); n = cap; }
        for (uint32_t k = 0; k < n; ++k) {
            const u64 e = trace[(size_t)k * T + t];
            const int tab = (int)(e >> 56);
            if (tab < K1T_N) { atomicAdd(&h[2 * tab], 1ull); atomicAdd(&h[2 * tab + 1], (unsigned long long)((e >> 48) & 0xff)); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * K1T_N + 2 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
#endif

// pass 2's tasks (see P2TASK): a read with c long, rare SMEMs among pass 1's intervals (mem_collect_intv's test: length >= min_seed_len * split_factor, at most
// split_width occurrences) is listed m = min(c, LH_P2_SPLIT) times, entry read << 4 | j << 2 | m - 1: the lane that takes it makes the read's j-th, (j + m)-th
// ... re-seeding call.  One thread per read; at most LH_P2_SPLIT entries per read: the list never overflows its LH_P2_SPLIT x n_reads slots.
#define LH_P2_SPLIT 4
__global__ void __launch_bounds__(256) k_p2_tasks(DOpts o, int n_reads, const DIntv* __restrict__ intv, const int32_t* __restrict__ n_intv, int32_t* __restrict__ tasks,
                                                   int32_t* __restrict__ n_tasks, int32_t* __restrict__ n_pass1) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    int m = 0;
    if (r < n_reads) {
        const int split_len = (int)(o.min_seed_len * o.split_factor + .499);
        const int on = n_intv[r] < LH_MAX_INTV ? n_intv[r] : LH_MAX_INTV;
        n_pass1[r] = on;
        for (int k = 0; k < on; ++k) {
            const DIntv p = intv[(size_t)r * LH_MAX_INTV + k];
            m += (int)(uint32_t)p.info - (int)(p.info >> 32) >= split_len && p.x2 <= (u64)o.split_width;
        }
        m = m < LH_P2_SPLIT ? m : LH_P2_SPLIT;
    }
    const int incl = wave_scan_add_i32(m);
    const int tot = wave_readlane(incl, 63);
    int basep = 0;
    if (lane == 0 && tot) basep = atomicAdd(n_tasks, tot);
    basep = wave_readlane(basep, 0);
    for (int j = 0; j < m; ++j) tasks[basep + incl - m + j] = r << 4 | j << 2 | (m - 1);
}
__global__ void __launch_bounds__(256) k_p2_clamp(int n_reads, int32_t* __restrict__ n_intv) {   // (a read whose calls asked for more slots than it has: flagged by the call that was refused)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads && n_intv[r] > LH_MAX_INTV) n_intv[r] = LH_MAX_INTV;
}

// a read's interval array: its regular slots, or the sorted half of its big-slab slot
__device__ __forceinline__ DIntv* dev_intv_of(DIntv* intv, const K1Big& big, int r, int sorted) {
    const int s = big.slot ? big.slot[r] : -1;
    return s < 0 ? intv + (size_t)r * LH_MAX_INTV : big.slab + ((size_t)s * 2 + (sorted ? 1 : 0)) * LH_BIG_INTV;
}
// seed count of a read (mem_chain: at most max_occ sampled occurrences per interval) and l_rep from its sorted intervals
__device__ __forceinline__ int dev_l_rep(const DOpts& o, const DIntv* a, int n) {
    int b = 0, e = 0, l_rep = 0;
    for (int u = 0; u < n; ++u) {
        DIntv p = a[u];
        if (p.x2 <= (u64)o.max_occ) continue;
        int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    return l_rep + e - b;
}
__device__ __forceinline__ int dev_seed_count(const DOpts& o, u64 s) {
    if (s <= (u64)o.max_occ) return (int)s;   // (no 64-bit division for the usual interval)
    u64 step = s / (u64)o.max_occ;
    u64 c = (s + step - 1) / step;
    return (int)(c < (u64)o.max_occ ? c : (u64)o.max_occ);
}
// PASS 3 IN LOCKSTEP, one thread per read (new in r03).  bwt_seed_strategy1 is a forward-only loop — no interval lists, no sweeps — so
// it needs no state machine: every thread walks its own read, the threads of a wave in the same short loop.  Same shortcuts as the
// pass-3 state machine (k_smem_pass<3>): a walk that starts inside one of the read's unique SMEMs is one PLCP byte (P3TEXT), any other
// takes its first bases from the k-mer tree table and goes on through the occurrence table.  Measured: 0.1 G wave-instructions at
// ~40 active lanes for what cost the state machine 1.2 G at 24.
__global__ void __launch_bounds__(256) k_smem_p3_lock(DIndex ix, DOpts o, int n_reads, const uint32_t* __restrict__ q4, const i64* __restrict__ seq_off, DIntv* __restrict__ intv_out,
                                                       int32_t* __restrict__ n_intv, int32_t* __restrict__ status, DCounters* __restrict__ ctr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = LANE();
    const PEnt* const kt = (const PEnt*)ix.ktree;
    const int ktl = kt ? ix.ktree_levels : 0;
    const int Lw = o.min_seed_len + 1;
    const bool text_ok = ix.isa != nullptr && ix.plcp != nullptr && o.max_mem_intv > 1;
    unsigned n_ext = 0, n_exec = 0, n_kt = 0;
    if (r < n_reads) {
        const i64 off = seq_off[r];
        int len = (int)(seq_off[r + 1] - off);
        int rst = status[r], on = n_intv[r], ovf = 0;
        if (len > LH_MAXLEN) len = 0;
        DIntv* out = intv_out + (size_t)r * LH_MAX_INTV;
#define QN(i_) ((int)(dev_nib8(q4, off + (i_)) & 0xf))
        if (len >= o.min_seed_len && o.max_mem_intv > 0) {
            // the read's two longest unique SMEMs [us, ue) and their text positions (pass 1 stored them by position; a row is one suffix-array read)
            int us0 = 0, ue0 = 0, us1 = 0, ue1 = 0;
            u64 up0 = 0, up1 = 0;
            if (text_ok)
                for (int k = 0; k < on; ++k) {
                    const DIntv p = out[k];
                    const int ps = (int)(p.info >> 32), pe = (int)(uint32_t)p.info;
                    if (p.x2 != 1 || pe - ps < Lw) continue;
                    if (pe - ps > ue0 - us0) { us1 = us0; ue1 = ue0; up1 = up0; us0 = ps; ue0 = pe; up0 = p.x0; }
                    else if (pe - ps > ue1 - us1) { us1 = ps; ue1 = pe; up1 = p.x0; }
                }
            if (ue0 > us0) up0 = (up0 & LH_POSF) ? up0 & ~LH_POSF : ix.sa[up0];
            if (ue1 > us1) up1 = (up1 & LH_POSF) ? up1 & ~LH_POSF : ix.sa[up1];
            int x = 0, notext = 0;
            while (x < len) {
                if (QN(x) > 3) { ++x; continue; }
                const int in0 = x >= us0 && x + Lw <= ue0, in1 = x >= us1 && x + Lw <= ue1;
                if (text_ok && !notext && (in0 || in1)) {   // by text: unique iff the suffix there shares fewer than Lw bases with every other one
                    const i64 tp = in0 ? (i64)up0 + (x - us0) : (i64)up1 + (x - us1);
                    if ((int)ix.plcp[tp] < Lw) {
                        if (on >= LH_MAX_INTV) ovf = 1;
                        else { DIntv m; m.x0 = LH_POSF | (u64)tp; m.x1 = 0; m.x2 = 1; m.info = (u64)x << 32 | (u64)(x + Lw); out[on++] = m; }
                        n_ext += (unsigned)(Lw - 1);
                        x += Lw;
                        continue;
                    }
                    notext = 1;   // it occurs again: the walk as written, from x
                }
                notext = 0;
                // bwt_seed_strategy1 from x
                DIntv c = dev_set_intv(ix, QN(x));
                int i = x + 1;
                uint32_t fcode = (uint32_t)QN(x);
                if (ktl > 1 && x + ktl <= len && o.min_seed_len >= ktl) {   // the walk's first ktl bases from the tree's deepest level (none of those steps can end it)
                    uint32_t w0 = dev_nib8(q4, off + x), w1 = dev_nib8(q4, off + x + 8);
                    uint32_t m0 = w0, m1 = w1;
                    if (ktl < 8) { m0 &= (1u << (4 * ktl)) - 1u; m1 = 0; }
                    else if (ktl < 16) m1 &= (1u << (4 * (ktl - 8))) - 1u;
                    if (!((m0 | m1) & 0x44444444u)) {
                        uint32_t ca = w0 & 0x33333333u, cb = w1 & 0x33333333u;
                        ca = (ca | ca >> 2) & 0x0f0f0f0fu; ca = (ca | ca >> 4) & 0x00ff00ffu; ca = (ca | ca >> 8) & 0xffffu;
                        cb = (cb | cb >> 2) & 0x0f0f0f0fu; cb = (cb | cb >> 4) & 0x00ff00ffu; cb = (cb | cb >> 8) & 0xffffu;
                        const uint32_t code = (ca | cb << 16) & (ktl >= 16 ? 0xffffffffu : (1u << (2 * ktl)) - 1u);
                        const PEnt te = kt[(((1ull << (2 * ktl)) - 4) / 3) + code];
                        c.x0 = PE_X0(te); c.x1 = PE_X1(te); c.x2 = PE_X2(te);
                        n_ext += (unsigned)(ktl - 1);
                        i = x + ktl;
                    }
                }
                int nx = len;   // where the next walk starts
                for (; i < len; ++i) {
                    const int b = QN(i);
                    if (b > 3) { nx = i + 1; break; }
                    DIntv ok;
                    const int lnew = i + 1 - x;
                    if (lnew <= ktl) {   // a short match: its interval is the tree's entry
                        fcode |= (uint32_t)b << (2 * (lnew - 1));
                        const PEnt te = kt[(((1ull << (2 * lnew)) - 4) / 3) + fcode];
                        ok.x0 = PE_X0(te); ok.x1 = PE_X1(te); ok.x2 = PE_X2(te); ok.info = 0;
                        ++n_kt;
                    } else { ok = dev_extend_c(ix, c, 3 - b, 0); ++n_exec; }
                    ++n_ext;
                    if (ok.x2 < (u64)o.max_mem_intv && i - x >= o.min_seed_len) {
                        if (ok.x2 > 0) {
                            if (on >= LH_MAX_INTV) ovf = 1;
                            else { ok.info = (u64)x << 32 | (u64)(i + 1); out[on++] = ok; }
                        }
                        nx = i + 1;
                        break;
                    }
                    c = ok;
                }
                x = nx;
            }
        }
#undef QN
        if (ovf) rst |= LH_ST_INTV_OVERFLOW;
        n_intv[r] = on;
        status[r] = rst;
    }
    if (ctr) {
        unsigned t1 = (unsigned)wave_sum_i32((int)n_ext), t2 = (unsigned)wave_sum_i32((int)n_exec), t3 = (unsigned)wave_sum_i32((int)n_kt);
        if (lane == 0 && t1) { atomicAdd(&LH_CTR(ctr)->n_ext, (u64)t1); atomicAdd(&LH_CTR(ctr)->n_ext_exec[2], (u64)t2); atomicAdd(&LH_CTR(ctr)->n_ktree[2], (u64)t3); }
    }
}

// sort each read's intervals by info (rank sort; equal keys are identical intervals), seed counts, l_rep.  16 lanes per read
// (reads in the big slab: k_smem_fin_big).
__global__ void __launch_bounds__(256) k_smem_fin(DOpts o, int n_reads, DIntv* __restrict__ intv, const int32_t* __restrict__ n_intv, int32_t* __restrict__ seed_cnt,
                                                   int32_t* __restrict__ l_rep_out, const int32_t* __restrict__ big_slot) {
    int gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    int r = gid < n_reads ? gid : n_reads - 1;
    int live = gid < n_reads && !(big_slot && big_slot[r] >= 0);
    DIntv* a = intv + (size_t)r * LH_MAX_INTV;
    int n = live ? n_intv[r] : 0;
    DIntv mine[4];
    int rank[4];
    for (int t = 0; t < 4; ++t) {
        int e = sub + 16 * t;
        mine[t].x0 = mine[t].x1 = mine[t].x2 = 0; mine[t].info = ~0ull;
        if (e < n) mine[t] = a[e];
        rank[t] = 0;
    }
    const int nt = (n + 15) >> 4;   // slots in use (one for the usual dozen intervals: the others' compares are skipped by the whole wave)
    for (int u = 0; u < n; ++u) {
        u64 oi = a[u].info;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < nt) { int e = sub + 16 * t; rank[t] += (oi < mine[t].info) || (oi == mine[t].info && u < e); }
    }
    __syncthreads();   // every lane holds its entries before any is overwritten
    int cnt = 0, rep = 0;   // rep: intervals with more than max_occ occurrences (none, usually: l_rep is 0 without another walk over the intervals)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int e = sub + 16 * t;
        if (t < nt && e < n && live) {
            a[rank[t]] = mine[t];
            cnt += dev_seed_count(o, mine[t].x2);
            rep += mine[t].x2 > (u64)o.max_occ;
        }
    }
    cnt += (int)dpp_xor1((uint32_t)cnt); cnt += (int)dpp_xor2((uint32_t)cnt); cnt += (int)dpp_half_mirror((uint32_t)cnt); cnt += (int)dpp_ror8((uint32_t)cnt);
    rep += (int)dpp_xor1((uint32_t)rep); rep += (int)dpp_xor2((uint32_t)rep); rep += (int)dpp_half_mirror((uint32_t)rep); rep += (int)dpp_ror8((uint32_t)rep);
    __syncthreads();
    if (sub == 0 && live) { seed_cnt[r] = cnt; l_rep_out[r] = rep ? dev_l_rep(o, a, n) : 0; }
}
// the same for a read in the big slab, one wave per read: ranks against the unsorted half, written to the sorted half
__global__ void __launch_bounds__(64) k_smem_fin_big(DOpts o, K1Big big, const int32_t* __restrict__ n_intv, int32_t* __restrict__ seed_cnt, int32_t* __restrict__ l_rep_out) {
    const int lane = LANE(), n_big = *big.count;
    for (int item = blockIdx.x; item < n_big; item += gridDim.x) {
        const int r = big.list[item];
        const DIntv* a = big.slab + (size_t)big.slot[r] * (2 * LH_BIG_INTV);
        DIntv* b = big.slab + ((size_t)big.slot[r] * 2 + 1) * LH_BIG_INTV;
        const int n = n_intv[r];
        int cnt = 0;
        for (int e = lane; e < n; e += 64) {
            const DIntv m = a[e];
            int rank = 0;
            for (int u = 0; u < n; ++u) { const u64 oi = a[u].info; rank += (oi < m.info) || (oi == m.info && u < e); }
            b[rank] = m;
            cnt += dev_seed_count(o, m.x2);
        }
        cnt = wave_sum_i32(cnt);
        WAVE_SYNC();
        if (lane == 0) { seed_cnt[r] = cnt; l_rep_out[r] = dev_l_rep(o, b, n); }
        WAVE_SYNC();
    }
}
// after the three passes: the reads whose intervals did not fit their regular slots get a slot of the big slab and are listed for the
// second chance (their flag is cleared; it is set again if LH_BIG_INTV do not hold them either, or the slab has no slot left)
// base > 0: a further round after the slab has grown (the host saw more reads asking than there were slots): only the reads the round before
// left without a slot are listed, for slots base .. base + cap - 1
__global__ void __launch_bounds__(256) k_big_collect(int n_reads, int32_t* __restrict__ status, int32_t* __restrict__ n_intv, int32_t* __restrict__ slot, int32_t* __restrict__ list,
                                                      int32_t* __restrict__ count, int base, int cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (base > 0 && slot[r] >= 0) return;   // listed by an earlier round (if its flag is set again, LH_BIG_INTV slots did not hold it either)
    int s = -1;
    if (status[r] & LH_ST_INTV_OVERFLOW) {
        s = atomicAdd(count + 1, 1);   // count[1]: slots asked for; count[0]: reads listed
        if (s < cap) { list[atomicAdd(count, 1)] = r; status[r] &= ~LH_ST_INTV_OVERFLOW; n_intv[r] = 0; s += base; }
        else s = -1;
    }
    slot[r] = s;
}

// intervals stored by text position (LH_POSF) -> suffix-array rows, as bwt_smem1a reports them: x0 = the row of the match, x1 = the row of
// its reverse complement (the stage dump, which the tests compare with the oracle's intervals)
__global__ void __launch_bounds__(256) k_intv_rows(DIndex ix, int n_reads, DIntv* __restrict__ intv, const int32_t* __restrict__ n_intv, K1Big big) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads || !ix.isa) return;
    DIntv* a = dev_intv_of(intv, big, r, 1);
    const int n = n_intv[r];
    for (int k = 0; k < n; ++k) {
        DIntv p = a[k];
        if (!(p.x0 & LH_POSF)) continue;
        const u64 pos = p.x0 & ~LH_POSF;
        const int l = (int)(uint32_t)p.info - (int)(p.info >> 32);
        p.x0 = ix.isa[pos]; p.x1 = ix.isa[ix.seq_len - (pos + (u64)l)];
        a[k] = p;
    }
}

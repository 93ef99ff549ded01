// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <mutex>

#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

namespace emu {
thread_local Tls tls;

__asm__(
    ".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

constexpr size_t STACK = 256 * 1024;

static thread_local unsigned long g_progress = 0;
static void release_wave_if_ready(Wave& w) {
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; g_progress++; }
}
static void release_block_if_ready(Block& b) {
    if (b.blk_live > 0 && b.blk_arrived == b.blk_live) { b.blk_arrived = 0; b.blk_gen++; g_progress++; }
}

static void fiber_entry() {
    Block* b = tls.blk;
    Fiber* f = b->cur;
    (*b->body)();
    f->done = true;
    g_progress++;
    Wave& w = b->waves[f->tid / WAVE];
    w.live--;
    w.live_mask &= ~(1ull << (f->tid % WAVE));
    release_wave_if_ready(w);
    b->blk_live--;
    release_block_if_ready(*b);
    void* dummy;
    emu_switch(&dummy, b->main_sp);
    abort();
}

void note(const char* file, int line) { Fiber* f = tls.blk->cur; f->file = file; f->line = line; f->hist[f->nh++ & 15] = line; }

void yield() {
    Block* b = tls.blk;
    Fiber* f = b->cur;
    emu_switch(&f->sp, b->main_sp);
}

void wave_barrier() {
    Wave& w = my_wave();
    uint64_t g = w.gen;
    if (++w.arrived == w.live) { w.arrived = 0; w.gen++; g_progress++; }
    else while (w.gen == g) yield();
}

void block_barrier() {
    Block& b = *tls.blk;
    uint64_t g = b.blk_gen;
    if (++b.blk_arrived == b.blk_live) { b.blk_arrived = 0; b.blk_gen++; g_progress++; }
    else while (b.blk_gen == g) yield();
}

uint64_t exchange(uint64_t v, int src_lane) {
    Wave& w = my_wave();
    int par = (int)(w.gen & 1);
    w.exch[par][my_lane()] = v;
    wave_barrier();
    return w.exch[par][src_lane & (WAVE - 1)];
}

uint64_t ballot(bool p) {
    Wave& w = my_wave();
    int par = (int)(w.gen & 1);
    w.exch[par][my_lane()] = p ? 1 : 0;
    uint64_t lm = w.live_mask;
    wave_barrier();
    uint64_t m = 0;
    for (int l = 0; l < WAVE; ++l)
        if ((lm >> l & 1) && w.exch[par][l]) m |= 1ull << l;
    return m;
}

static void run_block(Block& b, dim3 grid, dim3 block, dim3 bidx, const std::function<void()>& body) {
    unsigned nt = block.x;
    b.body = &body;
    b.blk_live = (int)nt; b.blk_arrived = 0;
    unsigned nw = (nt + WAVE - 1) / WAVE;
    b.waves.assign(nw, Wave());
    if (b.fibers.size() < nt) b.fibers.resize(nt);
    for (unsigned t = 0; t < nt; ++t) {
        Fiber& f = b.fibers[t];
        if (!f.stack) f.stack = (char*)aligned_alloc(64, STACK);
        f.done = false; f.tid = t; f.nh = 0;
        Wave& w = b.waves[t / WAVE];
        w.live++; w.live_mask |= 1ull << (t % WAVE);
        uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);   // fake caller slot keeps the ABI alignment at entry
        *--sp = (void*)&fiber_entry;     // return address popped by emu_switch's ret
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        f.sp = sp;
    }
    tls.blk = &b;
    tls.blockIdx = bidx; tls.blockDim = block; tls.gridDim = grid;
    int stuck = 0;
    for (;;) {
        bool any = false;
        unsigned long p0 = g_progress;
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = b.fibers[t];
            if (f.done) continue;
            any = true;
            b.cur = &f;
            tls.threadIdx = dim3(t, 0, 0);
            emu_switch(&b.main_sp, f.sp);
        }
        if (!any) break;
        if (g_progress == p0) {
            if (++stuck > 4) {   // every live fiber is parked at a rendezvous that can never complete: divergent cross-lane op
                fprintf(stderr, "hip_emu: DEADLOCK in block (%u,%u): lanes reached different cross-lane ops\n", bidx.x, bidx.y);
                for (unsigned w = 0; w < nw; ++w)
                    fprintf(stderr, "  wave %u: live=%d arrived(wave)=%d block_arrived=%d/%d\n", w, b.waves[w].live, b.waves[w].arrived, b.blk_arrived, b.blk_live);
                for (unsigned t = 0; t < nt; ++t) if (!b.fibers[t].done) { fprintf(stderr, "    lane %u last sync site %s:%d  hist:", t, b.fibers[t].file, b.fibers[t].line); for (int h = 0; h < 16; ++h) fprintf(stderr, " %d", b.fibers[t].hist[(b.fibers[t].nh + h) & 15]); fprintf(stderr, " (n=%d)\n", b.fibers[t].nh); }
                abort();
            }
        } else stuck = 0;
    }
}

static int emu_threads() {
    const char* e = getenv("LH_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    size_t nblk = (size_t)grid.x * grid.y;
    int nthr = emu_threads();
    if ((size_t)nthr > nblk) nthr = (int)nblk;
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        Block b;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblk) break;
            run_block(b, grid, block, dim3((unsigned)(i % grid.x), (unsigned)(i / grid.x), 0), body);
        }
        for (Fiber& f : b.fibers) free(f.stack);
    };
    if (nthr <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
}
}  // namespace emu

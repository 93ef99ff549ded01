// hip_emu.h — TEST INFRASTRUCTURE ONLY.  A tiny SPMD emulator that runs the product's HIP kernel SOURCES on the
// host CPU so that `pytest -m "not gpu"` can exercise kernel logic in a container without a GPU.
// It is NOT a fallback: nothing under lariat_amd/ includes or loads it; the shipped library is HIP only.
//
// Model: one workgroup at a time per OS thread; every GPU thread is a fiber (own stack, hand-written context
// switch); cross-lane operations (__shfl*, __ballot, __syncthreads) are rendezvous points at which the fibers
// of a 64-lane wave exchange values.  Kernels must reach every cross-lane op with all live lanes of the wave
// (which is also what the real hardware needs for defined results).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace emu {
constexpr int WAVE = 64;
struct Wave {
    int live = 0, arrived = 0;
    uint64_t gen = 0, live_mask = 0;
    uint64_t exch[2][WAVE];
};
struct Fiber {
    const char* file = "";
    int line = 0;
    int hist[16] = {0};
    int nh = 0;
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    unsigned tid = 0;
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    void* main_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    int blk_live = 0, blk_arrived = 0;
    uint64_t blk_gen = 0;
};
struct Tls {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    Block* blk = nullptr;
};
extern thread_local Tls tls;
extern "C" void emu_switch(void** save_sp, void* load_sp);
void yield();
void note(const char* file, int line);
void wave_barrier();
void block_barrier();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
inline Wave& my_wave() { return tls.blk->waves[tls.threadIdx.x / WAVE]; }
inline int my_lane() { return tls.threadIdx.x % WAVE; }
uint64_t exchange(uint64_t v, int src_lane);   // returns value published by src_lane
uint64_t ballot(bool p);
}  // namespace emu

#define threadIdx (emu::tls.threadIdx)
#define blockIdx (emu::tls.blockIdx)
#define blockDim (emu::tls.blockDim)
#define gridDim (emu::tls.gridDim)

// ---- device intrinsics used by the kernels ----------------------------------------------------
static inline void __syncthreads() { emu::block_barrier(); }
static inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
static inline int __any(int p) { return emu::ballot(p != 0) != 0; }
static inline int __all(int p) { return emu::ballot(p != 0) == emu::my_wave().live_mask; }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    int lane = emu::my_lane();
    int s = (lane & ~(width - 1)) | (src & (width - 1));
    u = emu::exchange(u, s);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = emu::my_lane();
    int s = (lane & (width - 1)) < (int)d ? lane : lane - (int)d;
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = emu::exchange(u, s);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = emu::my_lane();
    int s = (lane & (width - 1)) + (int)d >= width ? lane : lane + (int)d;
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = emu::exchange(u, s);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
    int lane = emu::my_lane();
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = emu::exchange(u, lane ^ m);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned long long __brevll(unsigned long long x);
static inline unsigned __brev(unsigned x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
}
static inline unsigned long long __brevll(unsigned long long x) { return ((unsigned long long)__brev((unsigned)x) << 32) | __brev((unsigned)(x >> 32)); }
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T atomicMin(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline unsigned long long wall_clock64() { return 0; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- mini runtime --------------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = n ? aligned_alloc(256, (n + 255) / 256 * 256) : nullptr; return (n && !*p) ? 2 : 0; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
#define hipHostMallocDefault 0
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
#define hipStreamNonBlocking 1
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent(); return 0; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new EmuEvent(); return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}

// k_diag.h — diagnostic microbenchmark: rate of independent random reads of `granule`-byte blocks (64 B = one BWA
// occurrence block) from a table in HBM.  SURVEY.md §8d: the 64-B-granule random rate is the practical ceiling of the
// FM-index walks (K1/K2) and must be measured, not assumed from the streaming figure.
#pragma once
#include "lh_dev.h"

__global__ void __launch_bounds__(256) k_diag_random_read(const uint4* __restrict__ table, u64 n_blocks, int vec_per_block, int per_thread, u64 seed,
                                                          uint32_t* __restrict__ sink) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x = seed + t * 0x9e3779b97f4a7c15ull;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;   // xorshift64*
        u64 b = (x * 0x2545F4914F6CDD1Dull) % n_blocks;
        const uint4* p = table + b * (u64)vec_per_block;
        for (int v = 0; v < vec_per_block; ++v) { uint4 d = p[v]; acc ^= d.x ^ d.y ^ d.z ^ d.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;   // keeps the loads alive
}

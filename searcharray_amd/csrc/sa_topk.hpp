// sa_topk.hpp -- block-cooperative top-k selection primitives (device).
//
// Ranking key: a 64-bit composite  score_bits << 32 | tie , where score_bits is the IEEE
// pattern of a non-negative fp32 score (monotone as an unsigned integer) and `tie` grows as the
// doc id shrinks, so "larger key" == "higher score, then smaller doc id".  Keys are unique per
// doc, which makes the k-th largest key a clean threshold: exactly k keys are >= it.  Key 0
// means "no candidate" (score 0 never ranks; the reference's argpartition would return
// arbitrary zero-score docs there, reference utils/sort.py:24).
#pragma once
#include "sa_common.hpp"

template <int NW>
__device__ __forceinline__ u64 sa_block_max64(u64 v, u64* red64) {
    v = sa_wave_max64(v);
    if (sa_lane() == 0) red64[sa_wave_id()] = v;
    __syncthreads();
    u64 m = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { u64 x = red64[w]; m = x > m ? x : m; }
    __syncthreads();
    return m;
}

// Largest value v such that at least k of the block's keys are >= v (0 when fewer than k keys
// are non-zero).  Each thread contributes EPT keys held in registers.  MSB-first bisection,
// two bits per step: three counts per step travel packed in one 64-bit block reduction.
//
// `key(e)` yields the e-th key of the calling thread (e in [0, EPT)); it is re-evaluated every
// step, so composite keys can be formed on the fly from narrower registers.
template <int EPT, int NW, class KeyFn>
__device__ __forceinline__ u64 sa_block_kth_largest(KeyFn key, u32 k, u64* red64) {
    u64 m = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) { const u64 x = key(e); m = x > m ? x : m; }
    m = sa_block_max64<NW>(m, red64);
    if (m == 0 || k == 0) return 0;
    int top = 63 - __clzll((long long)m);
    if ((top & 1) == 0) top++;            // pairs of bits: (top, top-1)
    u64 prefix = 0;
    for (int bit = top; bit >= 1; bit -= 2) {
        const u64 c1 = prefix | (1ull << (bit - 1));
        const u64 c2 = prefix | (2ull << (bit - 1));
        const u64 c3 = prefix | (3ull << (bit - 1));
        u64 packed = 0;                    // 21 bits per count (block holds < 2^21 keys)
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const u64 x = key(e);
            packed += (x >= c1 ? 1ull : 0ull) + (x >= c2 ? (1ull << 21) : 0ull) + (x >= c3 ? (1ull << 42) : 0ull);
        }
        // block sum of a u64
#pragma unroll
        for (int o = SA_WAVE / 2; o > 0; o >>= 1) packed += __shfl_xor(packed, o, SA_WAVE);
        if (sa_lane() == 0) red64[sa_wave_id()] = packed;
        __syncthreads();
        u64 tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) tot += red64[w];
        __syncthreads();
        const u32 n1 = (u32)(tot & 0x1FFFFF), n2 = (u32)((tot >> 21) & 0x1FFFFF), n3 = (u32)((tot >> 42) & 0x1FFFFF);
        if (n3 >= k) prefix = c3;
        else if (n2 >= k) prefix = c2;
        else if (n1 >= k) prefix = c1;
    }
    return prefix;
}

// In-LDS bitonic sort, descending, of n_pow2 (power of two, <= 2 * blockDim) u64 keys.
__device__ __forceinline__ void sa_block_bitonic_desc(u64* a, u32 n_pow2) {
    for (u32 size = 2; size <= n_pow2; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (u32 t = threadIdx.x; t < (n_pow2 >> 1); t += blockDim.x) {
                const u32 lo = 2 * t - (t & (stride - 1));
                const u32 hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const u64 x = a[lo], y = a[hi];
                if (desc ? (x < y) : (x > y)) { a[lo] = y; a[hi] = x; }
            }
        }
    }
    __syncthreads();
}

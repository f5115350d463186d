// sa_common.hpp -- shared definitions for the gfx950 kernels and the C-ABI host code.
//
// Layout constants are the reference's roaringish wire format
// (reference searcharray/roaringish/roaringish.py:30-35):
//   [63:36] doc id (28 b) | [35:18] position // 18 | [17:0] bitmap of position % 18
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <string>

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;

#define SA_KEY_SHIFT 36
#define SA_LSB_BITS 18
#define SA_LSB_MASK 0x3FFFFull
#define SA_HEADER_MASK 0xFFFFFFFFFFFC0000ull
#define SA_KEY_MASK 0xFFFFFFF000000000ull
#define SA_UPPER_BIT (1ull << 17)
#define SA_NO_TERM 0xFFFFFFFFu
#define SA_NO_DOC 0xFFFFFFFFFFFFFFFFull
#define SA_WAVE 64

// ---- error reporting (thread-local message, negative status codes) ----
enum { SA_OK = 0, SA_ERR_HIP = -1, SA_ERR_ARG = -2, SA_ERR_NOMEM = -3, SA_ERR_STATE = -4,
       SA_ERR_UNSUPPORTED = -5, SA_ERR_COMM = -6, SA_ERR_IO = -7 };

void sa_set_error(const char* fmt, ...);

#define SA_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            sa_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,              \
                         hipGetErrorString(e_));                                         \
            return SA_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

#define SA_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != SA_OK) return rc_; \
    } while (0)

#define SA_ARG(cond, msg)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            sa_set_error("invalid argument: %s", msg);      \
            return SA_ERR_ARG;                              \
        }                                                   \
    } while (0)

static inline u32 sa_div_up(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ---- device helpers ----
__device__ __forceinline__ int sa_lane() { return threadIdx.x & (SA_WAVE - 1); }
// A pointer a kernel has READ from memory (a field of a job struct) is a flat pointer to the compiler: its loads are flat_load, which
// count against the LDS counter as well as the vector-memory one -- every wait for an LDS read then waits for the loads in flight.
// sa_glob says "this is global memory": the loads through the pointer it returns are global_load.
#ifndef SA_AS_GLOBAL
#define SA_AS_GLOBAL __attribute__((address_space(1)))
#endif
template <typename T> __device__ __forceinline__ T SA_AS_GLOBAL* sa_glob(T* p) { return (T SA_AS_GLOBAL*)p; }
__device__ __forceinline__ int sa_wave_id() { return threadIdx.x / SA_WAVE; }

__device__ __forceinline__ u32 sa_wave_sum(u32 v) {
#pragma unroll
    for (int o = SA_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SA_WAVE);
    return v;
}
__device__ __forceinline__ u32 sa_wave_max(u32 v) {
#pragma unroll
    for (int o = SA_WAVE / 2; o > 0; o >>= 1) {
        u32 w = __shfl_xor(v, o, SA_WAVE);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ u64 sa_wave_max64(u64 v) {
#pragma unroll
    for (int o = SA_WAVE / 2; o > 0; o >>= 1) {
        u64 w = __shfl_xor(v, o, SA_WAVE);
        v = w > v ? w : v;
    }
    return v;
}

// Full-wave (64 lanes, all active) unsigned max / min through the DPP data path: a row scan by
// row_shr 1/2/4/8, then row_bcast:15 and row_bcast:31 fold the four rows into lane 63.  Pure
// VALU (no LDS crossbar round trips like ds_bpermute-based __shfl), result is wave-uniform.
#define SA_DPP_ROW_SHR(n) (0x110 + (n))
#define SA_DPP_ROW_BCAST15 0x142
#define SA_DPP_ROW_BCAST31 0x143

__device__ __forceinline__ u32 sa_wave_max_u32(u32 v) {
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(1), 0xf, 0xf, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(2), 0xf, 0xf, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(4), 0xf, 0xf, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(8), 0xf, 0xf, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST15, 0xa, 0xf, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST31, 0xc, 0xf, false); v = t > v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ u32 sa_wave_min_u32(u32 v) {
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_SHR(1), 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_SHR(2), 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_SHR(4), 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_SHR(8), 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_BCAST15, 0xa, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, SA_DPP_ROW_BCAST31, 0xc, 0xf, false); v = t < v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// Block-wide sum of a u32 for blocks of NWAVES waves.  `red` is LDS scratch of >= NWAVES+1
// words; the result is returned to every thread.  Contains two barriers; safe to call in a loop.
template <int NWAVES>
__device__ __forceinline__ u32 sa_block_sum(u32 v, u32* red) {
    v = sa_wave_sum(v);
    if (sa_lane() == 0) red[sa_wave_id()] = v;
    __syncthreads();
    u32 s = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) s += red[w];
    __syncthreads();
    return s;
}

// Block-wide exclusive scan of one u32 per thread (thread order); returns the exclusive
// prefix and writes the block total to *total.  `red` is LDS scratch of >= NWAVES words.
template <int NWAVES>
__device__ __forceinline__ u32 sa_block_excl_scan(u32 v, u32* red, u32* total) {
    const int lane = sa_lane();
    u32 incl = v;
#pragma unroll
    for (int o = 1; o < SA_WAVE; o <<= 1) {
        u32 t = __shfl_up(incl, o, SA_WAVE);
        if (lane >= o) incl += t;
    }
    if (lane == SA_WAVE - 1) red[sa_wave_id()] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) {
        u32 s = red[w];
        if (w < sa_wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// Lower bound over a sorted u64 array on (x & mask): first index in [lo, hi) whose masked
// value is >= key.
__device__ __forceinline__ u32 sa_lower_bound(const u64* __restrict__ a, u32 lo, u32 hi, u64 key, u64 mask) {
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        if ((a[mid] & mask) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

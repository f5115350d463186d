// sa_scan.hpp -- stable stream compaction with device-resident lengths.
//
// Every variable-length step of the roaringish set algebra (intersect, merge, group-by-doc,
// filter) is "flag + exclusive scan + scatter".  The element count of each array lives in
// device memory (a u32), so a whole chain of dependent steps is enqueued on one stream with
// fixed worst-case grids and NO host round trip between steps.
//
// A functor F supplies
//     __device__ bool flag(u32 i) const;            // keep element i ?
//     __device__ void emit(u32 i, u32 pos) const;   // element i is the pos-th kept element
// Flags are recomputed in the emit pass instead of being stored (halves the traffic).
//
// Three launches: count per 2048-element chunk -> single-block scan of chunk counts (also
// publishes the total to a device counter) -> emit.  Output order is input order (stable).
#pragma once
#include "sa_common.hpp"

#define SA_CT 256                 // threads per block (4 waves)
#define SA_CI 8                   // rounds per chunk
#define SA_CHUNK (SA_CT * SA_CI)  // elements per chunk
#define SA_CW (SA_CT / SA_WAVE)

template <class F>
__global__ void __launch_bounds__(SA_CT)
sa_k_compact_count(F f, const u32* __restrict__ n_dev, u32 n_max, u32* __restrict__ chunk_counts) {
    __shared__ u32 red[SA_CW + 1];
    u32 n = n_dev ? *n_dev : n_max;
    if (n > n_max) n = n_max;
    const u32 nchunks = (n_max + SA_CHUNK - 1) / SA_CHUNK;
    for (u32 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u32 base = c * SA_CHUNK;
        u32 cnt = 0;
        if (base < n) {
#pragma unroll
            for (int j = 0; j < SA_CI; j++) {
                const u32 i = base + j * SA_CT + threadIdx.x;
                if (i < n && f.flag(i)) cnt++;
            }
        }
        const u32 tot = sa_block_sum<SA_CW>(cnt, red);
        if (threadIdx.x == 0) chunk_counts[c] = tot;
    }
}

// In-place exclusive scan of chunk counts by ONE block of 1024 threads; total -> *total_out.
__global__ void __launch_bounds__(1024)
sa_k_scan_chunks(u32* __restrict__ counts, u32 nchunks, u32* __restrict__ total_out);

template <class F>
__global__ void __launch_bounds__(SA_CT)
sa_k_compact_emit(F f, const u32* __restrict__ n_dev, u32 n_max, const u32* __restrict__ chunk_off) {
    __shared__ u32 wc[SA_CI][SA_CW];
    u32 n = n_dev ? *n_dev : n_max;
    if (n > n_max) n = n_max;
    const u32 nchunks = (n_max + SA_CHUNK - 1) / SA_CHUNK;
    const int lane = sa_lane(), wave = sa_wave_id();
    const u64 lt = (1ull << lane) - 1ull;
    for (u32 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u32 base = c * SA_CHUNK;
        if (base >= n) continue;                      // uniform per block
        u32 rank[SA_CI];
        u32 mine = 0;
#pragma unroll
        for (int j = 0; j < SA_CI; j++) {
            const u32 i = base + j * SA_CT + threadIdx.x;
            const bool fl = (i < n) && f.flag(i);
            const u64 b = __ballot(fl);
            rank[j] = (u32)__popcll(b & lt);
            if (lane == 0) wc[j][wave] = (u32)__popcll(b);
            mine |= (fl ? 1u : 0u) << j;
        }
        __syncthreads();
        u32 off = chunk_off[c];
#pragma unroll
        for (int j = 0; j < SA_CI; j++) {
#pragma unroll
            for (int w = 0; w < SA_CW; w++) {
                if (w == wave && ((mine >> j) & 1u)) f.emit(base + j * SA_CT + threadIdx.x, off + rank[j]);
                off += wc[j][w];
            }
        }
        __syncthreads();
    }
}

// Workspace for one compaction: chunk counters sized for n_max elements.
static inline u32 sa_compact_chunks(u32 n_max) { return (n_max + SA_CHUNK - 1) / SA_CHUNK; }

static inline u32 sa_compact_grid(u32 n_max) {
    u32 c = sa_compact_chunks(n_max);
    if (c < 1) c = 1;
    return c < 4096 ? c : 4096;
}

// Enqueue a full compaction on `stream`.  chunk_ws: u32[sa_compact_chunks(n_max)] scratch;
// total_out: device u32 receiving the number of kept elements.
template <class F>
static inline void sa_compact(F f, const u32* n_dev, u32 n_max, u32* chunk_ws, u32* total_out,
                              hipStream_t stream) {
    const u32 nchunks = sa_compact_chunks(n_max);
    if (nchunks == 0) {
        hipMemsetAsync(total_out, 0, sizeof(u32), stream);
        return;
    }
    const u32 grid = sa_compact_grid(n_max);
    hipLaunchKernelGGL((sa_k_compact_count<F>), dim3(grid), dim3(SA_CT), 0, stream, f, n_dev, n_max, chunk_ws);
    hipLaunchKernelGGL(sa_k_scan_chunks, dim3(1), dim3(1024), 0, stream, chunk_ws, nchunks, total_out);
    hipLaunchKernelGGL((sa_k_compact_emit<F>), dim3(grid), dim3(SA_CT), 0, stream, f, n_dev, n_max, chunk_ws);
}

// sa_ops.hip -- kernel-level mirrors of the reference's native entry points (Part 1 of the
// C ABI): host buffers in, host buffers out, same results as the Cython functions.  They exist
// so the reference's own call sites can be rebound one function at a time and
// so every device primitive has a parity test against the oracle.
#include "sa_common.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"

// RAII-less helper: a bundle of device allocations freed on scope exit
struct DevBufs {
    void* ptrs[12];
    int n = 0;
    ~DevBufs() { for (int i = 0; i < n; i++) hipFree(ptrs[i]); }
    template <class T> int alloc(T** p, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) { sa_set_error("hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e)); return SA_ERR_HIP; }
        ptrs[n++] = q;
        *p = (T*)q;
        return SA_OK;
    }
};

// ---- bm25_score: reference bm25.pyx:11-25 ---------------------------------------------
__global__ void sa_k_bm25_score(float* __restrict__ tf, const float* __restrict__ dl, float avgdl, float idf,
                                float k1, float b, u64 n) {
    const float one_minus_b = 1.0f - b;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const float t = tf[i];
        const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl[i], avgdl))));
        tf[i] = __fmul_rn(__fdiv_rn(t, __fadd_rn(t, norm)), idf);
    }
}

extern "C" int sa_bm25_score(float* term_freqs, const float* doc_lens, float avg_doc_lens, float idf,
                             float k1, float b, int64_t n) {
    SA_ARG(n >= 0, "n < 0");
    if (n == 0) return SA_OK;
    SA_ARG(term_freqs && doc_lens, "null argument");
    DevBufs bufs;
    float *d_tf, *d_dl;
    SA_TRY(bufs.alloc(&d_tf, (size_t)n));
    SA_TRY(bufs.alloc(&d_dl, (size_t)n));
    SA_HIP(hipMemcpy(d_tf, term_freqs, (size_t)n * 4, hipMemcpyHostToDevice));
    SA_HIP(hipMemcpy(d_dl, doc_lens, (size_t)n * 4, hipMemcpyHostToDevice));
    const u32 grid = sa_div_up((u64)n, 256) < 8192 ? sa_div_up((u64)n, 256) : 8192;
    hipLaunchKernelGGL(sa_k_bm25_score, dim3(grid), dim3(256), 0, 0, d_tf, d_dl, avg_doc_lens, idf, k1, b, (u64)n);
    SA_HIP(hipGetLastError());
    SA_HIP(hipMemcpy(term_freqs, d_tf, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SA_OK;
}

// ---- as_dense: reference roaringish_ops.pyx:84-98 --------------------------------------
__global__ void sa_k_scatter(const u64* __restrict__ idx, const float* __restrict__ val, u64 n, float* __restrict__ out,
                             u64 size, u32* __restrict__ err) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 j = idx[i];
        if (j < size) out[j] = val[i]; else *err = 1u;
    }
}

extern "C" int sa_as_dense(const uint64_t* indices, const float* values, int64_t n, float* out, int64_t size) {
    SA_ARG(n >= 0 && size >= 0, "negative size");
    if (size == 0) return SA_OK;
    SA_ARG(out, "out is null");
    SA_ARG(n == 0 || (indices && values), "null argument");
    DevBufs bufs;
    u64* d_idx; float *d_val, *d_out; u32* d_err;
    SA_TRY(bufs.alloc(&d_idx, (size_t)n));
    SA_TRY(bufs.alloc(&d_val, (size_t)n));
    SA_TRY(bufs.alloc(&d_out, (size_t)size));
    SA_TRY(bufs.alloc(&d_err, 1));
    SA_HIP(hipMemset(d_out, 0, (size_t)size * 4));
    SA_HIP(hipMemset(d_err, 0, 4));
    if (n) {
        SA_HIP(hipMemcpy(d_idx, indices, (size_t)n * 8, hipMemcpyHostToDevice));
        SA_HIP(hipMemcpy(d_val, values, (size_t)n * 4, hipMemcpyHostToDevice));
        const u32 grid = sa_div_up((u64)n, 256) < 8192 ? sa_div_up((u64)n, 256) : 8192;
        hipLaunchKernelGGL(sa_k_scatter, dim3(grid), dim3(256), 0, 0, d_idx, d_val, (u64)n, d_out, (u64)size, d_err);
        SA_HIP(hipGetLastError());
    }
    u32 err = 0;
    SA_HIP(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    if (err) { sa_set_error("as_dense: index out of range"); return SA_ERR_ARG; }
    SA_HIP(hipMemcpy(out, d_out, (size_t)size * 4, hipMemcpyDeviceToHost));
    return SA_OK;
}

// ---- popcount64_reduce / unique: run heads of a sorted array ----------------------------
struct KeyRuns {
    const u64* arr;
    u32 n;
    u64 shift;
    u64 value_mask;
    u64* keys_out;
    float* counts_out;     // null for unique()
    __device__ __forceinline__ bool flag(u32 i) const {
        return i == 0 || (arr[i] >> shift) != (arr[i - 1] >> shift);
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 key = arr[i] >> shift;
        keys_out[pos] = key;
        if (counts_out) {
            // reference accumulates in a float (popcount.pyx:228); the sums are small integers
            float c = (float)__popcll(arr[i] & value_mask);
            for (u32 j = i + 1; j < n && (arr[j] >> shift) == key; j++) c += (float)__popcll(arr[j] & value_mask);
            counts_out[pos] = c;
        }
    }
};

static int sa_key_runs(const uint64_t* arr, int64_t n, uint64_t shift, uint64_t value_mask,
                       uint64_t* keys_out, float* counts_out, int64_t* n_out) {
    SA_ARG(n >= 0 && n_out, "bad argument");
    *n_out = 0;
    if (n == 0) return SA_OK;
    SA_ARG(arr && keys_out, "null argument");
    SA_ARG(n < 0xFFFFF000ll, "array too long");
    SA_ARG(shift < 64, "shift must be < 64");
    DevBufs bufs;
    u64 *d_arr, *d_keys; float* d_counts = nullptr; u32 *d_chunks, *d_total;
    SA_TRY(bufs.alloc(&d_arr, (size_t)n));
    SA_TRY(bufs.alloc(&d_keys, (size_t)n));
    if (counts_out) SA_TRY(bufs.alloc(&d_counts, (size_t)n));
    SA_TRY(bufs.alloc(&d_chunks, sa_compact_chunks((u32)n) + 1));
    SA_TRY(bufs.alloc(&d_total, 1));
    SA_HIP(hipMemcpy(d_arr, arr, (size_t)n * 8, hipMemcpyHostToDevice));
    KeyRuns f;
    f.arr = d_arr; f.n = (u32)n; f.shift = shift; f.value_mask = value_mask; f.keys_out = d_keys; f.counts_out = d_counts;
    sa_compact(f, (const u32*)nullptr, (u32)n, d_chunks, d_total, (hipStream_t)0);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_total, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(keys_out, d_keys, (size_t)g * 8, hipMemcpyDeviceToHost));
    if (counts_out) SA_HIP(hipMemcpy(counts_out, d_counts, (size_t)g * 4, hipMemcpyDeviceToHost));
    *n_out = g;
    return SA_OK;
}

extern "C" int sa_popcount64_reduce(const uint64_t* arr, int64_t n, uint64_t key_shift, uint64_t value_mask,
                                    uint64_t* keys_out, float* counts_out, int64_t* n_out) {
    SA_ARG(n == 0 || counts_out, "counts_out is null");
    return sa_key_runs(arr, n, key_shift, value_mask, keys_out, counts_out, n_out);
}

extern "C" int sa_unique(const uint64_t* arr, int64_t n, uint64_t rshift, uint64_t* out, int64_t* n_out) {
    return sa_key_runs(arr, n, rshift, 0, out, nullptr, n_out);
}

// ---- popcount64: reference popcount.pyx:71-81 ------------------------------------------
__global__ void sa_k_popcount64(const u64* __restrict__ a, u64 n, u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        out[i] = (u64)__popcll(a[i]);
}

extern "C" int sa_popcount64(const uint64_t* arr, int64_t n, uint64_t* out) {
    SA_ARG(n >= 0, "n < 0");
    if (n == 0) return SA_OK;
    SA_ARG(arr && out, "null argument");
    DevBufs bufs;
    u64 *d_a, *d_o;
    SA_TRY(bufs.alloc(&d_a, (size_t)n));
    SA_TRY(bufs.alloc(&d_o, (size_t)n));
    SA_HIP(hipMemcpy(d_a, arr, (size_t)n * 8, hipMemcpyHostToDevice));
    const u32 grid = sa_div_up((u64)n, 256) < 8192 ? sa_div_up((u64)n, 256) : 8192;
    hipLaunchKernelGGL(sa_k_popcount64, dim3(grid), dim3(256), 0, 0, d_a, (u64)n, d_o);
    SA_HIP(hipGetLastError());
    SA_HIP(hipMemcpy(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SA_OK;
}

// ---- HBM read-bandwidth probe (calibration for the roofline figures, not a product path) ----
// Streams `bytes` of device memory with the same access shapes the scoring kernel can use
// (mode 0: one 8-byte load per lane, mode 1: one 16-byte load per lane) and reports GB/s.
__global__ void __launch_bounds__(256) sa_k_stream8(const u64* __restrict__ a, u64 n, u64* __restrict__ sink) {
    u64 acc = 0;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) acc += a[i] ^ a[i + stride] ^ a[i + 2 * stride] ^ a[i + 3 * stride];
    for (; i < n; i += stride) acc += a[i];
    if (acc == 0x123456789ull) sink[0] = acc;
}

struct alignas(16) sa_u64x2 { u64 x, y; };

__global__ void __launch_bounds__(256) sa_k_stream16(const sa_u64x2* __restrict__ a, u64 n, u64* __restrict__ sink) {
    u64 acc = 0;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const sa_u64x2 p = a[i], q = a[i + stride], r = a[i + 2 * stride], s = a[i + 3 * stride];
        acc += (p.x ^ p.y) + (q.x ^ q.y) + (r.x ^ r.y) + (s.x ^ s.y);
    }
    for (; i < n; i += stride) acc += a[i].x ^ a[i].y;
    if (acc == 0x123456789ull) sink[0] = acc;
}

extern "C" int sa_stream_probe(uint64_t bytes, int mode, int reps, double* gbps_out) {
    SA_ARG(gbps_out && bytes >= 4096 && reps > 0, "bad argument");
    DevBufs bufs;
    u64 *d_a, *d_sink;
    const u64 n8 = bytes / 8;
    SA_TRY(bufs.alloc(&d_a, n8 + 2));
    SA_TRY(bufs.alloc(&d_sink, 1));
    SA_HIP(hipMemset(d_a, 0x5A, n8 * 8));
    hipEvent_t e0, e1;
    SA_HIP(hipEventCreate(&e0));
    SA_HIP(hipEventCreate(&e1));
    const u32 grid = 256 * 16;
    double best = 0.0;
    for (int r = 0; r < reps + 1; r++) {
        SA_HIP(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(sa_k_stream8, dim3(grid), dim3(256), 0, 0, d_a, n8, d_sink);
        else hipLaunchKernelGGL(sa_k_stream16, dim3(grid), dim3(256), 0, 0, (const sa_u64x2*)d_a, n8 / 2, d_sink);
        SA_HIP(hipEventRecord(e1, 0));
        SA_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SA_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double g = (double)(n8 * 8) / (ms * 1e-3) / 1e9;
        if (r > 0 && g > best) best = g;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    SA_HIP(hipGetLastError());
    *gbps_out = best;
    return SA_OK;
}

// Page-locked host memory for the dense float32[N] results of the drop-in calls: the device copies
// into it at full PCIe rate, and a recycled buffer costs no page faults (a fresh 40 MB numpy array
// costs ~3 ms of first-touch faults, more than the copy itself).
extern "C" int sa_host_alloc(uint64_t bytes, void** out) {
    SA_ARG(out, "out is null");
    *out = nullptr;
    SA_HIP(hipHostMalloc(out, bytes ? (size_t)bytes : 1));
    return SA_OK;
}

extern "C" int sa_host_free(void* p) {
    if (p) SA_HIP(hipHostFree(p));
    return SA_OK;
}


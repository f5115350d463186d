// sa_setops.hip -- the reference's sorted-array primitives ("snp_ops") as data-parallel kernels,
// with kernel-level C-ABI mirrors (Part 1 of include/searcharray_hip.h).
//
//   intersect (drop / keep), adjacent, intersect_with_adjacents   reference intersect.pyx:32-390
//   merge (keep / drop duplicates), sort_merge_counts             reference merge.pyx:54-232
//   popcount_reduce_at, key_sum_over                              reference popcount.pyx:124-204
//   payload_slice                                                 reference roaringish_ops.pyx:46-68
//
// The reference walks two sorted arrays with a serial galloping two-pointer.  Here every element
// finds its partner with a lower-bound search and results are produced by stable stream
// compaction (sa_scan.hpp), which yields the same index pairs: for a value present on both sides
// the reference reports (first lhs index of the run, first rhs index of the run) in "drop" mode
// and every member of both runs in "keep" mode.
// The reference's drop variants start with `last = all ones` (intersect.pyx:39,146,224-225) and report a match
// only if `(last & mask) != (lhs & mask)`: a common value whose masked bits are ALL ones is dropped when it would be
// the FIRST pair reported -- being the largest value it is then the only one, so the list comes out empty; after any
// other match `last` has moved and it is reported like every value.  Reproduced (sa_pairs_drop, after the
// compaction) and pinned by tests/test_setops.py against outputs of the reference itself.
#include "sa_common.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"

#define SA_NONE 0xFFFFFFFFu

struct SetBufs {
    void* ptrs[24];
    int n = 0;
    ~SetBufs() { for (int i = 0; i < n; i++) hipFree(ptrs[i]); }
    template <class T> int alloc(T** p, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) { sa_set_error("hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e)); return SA_ERR_HIP; }
        ptrs[n++] = q;
        *p = (T*)q;
        return SA_OK;
    }
    template <class T> int upload(T** p, const T* host, size_t count) {
        SA_TRY(alloc(p, count));
        if (count) SA_HIP(hipMemcpy(*p, host, count * sizeof(T), hipMemcpyHostToDevice));
        return SA_OK;
    }
};

static inline u32 sa_grid(u64 n) {
    const u64 g = (n + 255) / 256;
    return (u32)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

__device__ __forceinline__ bool sa_run_head(const u64* a, u32 i, u64 mask) {
    return i == 0 || (a[i] & mask) != (a[i - 1] & mask);
}

// ---- intersect / adjacent ------------------------------------------------------------------
// drop mode: one pair per common value -- (first lhs of run, first rhs of run); `delta` != 0
// turns it into `adjacent`: lhs value + delta == rhs value (wrap to 0 never matches, which is the
// reference skipping leading rhs zeros, intersect.pyx:151-153).
struct PairDrop {
    const u64* lhs; const u64* rhs; u32 nr; u64 mask; u64 delta;
    u64* lhs_out; u64* rhs_out;
    __device__ __forceinline__ u32 partner(u32 i) const {
        if (!sa_run_head(lhs, i, mask)) return SA_NONE;
        const u64 key = (lhs[i] & mask) + delta;
        if (delta && (key & mask) == 0) return SA_NONE;
        const u32 j = sa_lower_bound(rhs, 0, nr, key & mask, mask);
        return (j < nr && (rhs[j] & mask) == (key & mask)) ? j : SA_NONE;
    }
    __device__ __forceinline__ bool flag(u32 i) const { return partner(i) != SA_NONE; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { lhs_out[pos] = i; rhs_out[pos] = partner(i); }
};

// keep mode: every index of `a` whose masked value occurs in `b`
struct MemberOf {
    const u64* a; const u64* b; u32 nb; u64 mask; u64* out;
    __device__ __forceinline__ bool flag(u32 i) const {
        const u64 key = a[i] & mask;
        const u32 j = sa_lower_bound(b, 0, nb, key, mask);
        return j < nb && (b[j] & mask) == key;
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = i; }
};

static int sa_pairs_drop(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, uint64_t mask, uint64_t delta,
                         uint64_t* lhs_idx, uint64_t* rhs_idx, int64_t* n_out) {
    SA_ARG(nl >= 0 && nr >= 0 && n_out, "bad argument");
    if (mask == 0) { sa_set_error("Mask cannot be zero"); return SA_ERR_ARG; }       // intersect.pyx:290-291
    *n_out = 0;
    if (nl == 0 || nr == 0) return SA_OK;
    SA_ARG(lhs && rhs && lhs_idx && rhs_idx, "null argument");
    SA_ARG(nl < 0xFFFFF000ll && nr < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_l, *d_r, *d_lo, *d_ro; u32 *d_chunks, *d_total;
    SA_TRY(b.upload(&d_l, lhs, (size_t)nl));
    SA_TRY(b.upload(&d_r, rhs, (size_t)nr));
    SA_TRY(b.alloc(&d_lo, (size_t)nl));
    SA_TRY(b.alloc(&d_ro, (size_t)nl));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)nl) + 1));
    SA_TRY(b.alloc(&d_total, 1));
    PairDrop f;
    f.lhs = d_l; f.rhs = d_r; f.nr = (u32)nr; f.mask = mask; f.delta = delta; f.lhs_out = d_lo; f.rhs_out = d_ro;
    sa_compact(f, (const u32*)nullptr, (u32)nl, d_chunks, d_total, (hipStream_t)0);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_total, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(lhs_idx, d_lo, (size_t)g * 8, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(rhs_idx, d_ro, (size_t)g * 8, hipMemcpyDeviceToHost));
    // the reference's `last = all ones` start value (see the header): a lone all-ones match is not reported
    if (g == 1 && (lhs[lhs_idx[0]] & mask) == mask) g = 0;
    *n_out = g;
    return SA_OK;
}

extern "C" int sa_intersect(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, uint64_t mask,
                            int drop_duplicates, uint64_t* lhs_idx, uint64_t* rhs_idx,
                            int64_t* n_lhs_out, int64_t* n_rhs_out) {
    SA_ARG(n_lhs_out && n_rhs_out, "null argument");
    if (drop_duplicates) {
        SA_TRY(sa_pairs_drop(lhs, nl, rhs, nr, mask, 0, lhs_idx, rhs_idx, n_lhs_out));
        *n_rhs_out = *n_lhs_out;
        return SA_OK;
    }
    if (mask == 0) { sa_set_error("Mask cannot be zero"); return SA_ERR_ARG; }
    *n_lhs_out = 0; *n_rhs_out = 0;
    if (nl <= 0 || nr <= 0) return SA_OK;
    SA_ARG(lhs && rhs && lhs_idx && rhs_idx, "null argument");
    SA_ARG(nl < 0xFFFFF000ll && nr < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_l, *d_r, *d_lo, *d_ro; u32 *d_chunks, *d_total;
    const size_t nmax = (size_t)(nl > nr ? nl : nr);
    SA_TRY(b.upload(&d_l, lhs, (size_t)nl));
    SA_TRY(b.upload(&d_r, rhs, (size_t)nr));
    SA_TRY(b.alloc(&d_lo, (size_t)nl));
    SA_TRY(b.alloc(&d_ro, (size_t)nr));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)nmax) + 1));
    SA_TRY(b.alloc(&d_total, 2));
    MemberOf fl; fl.a = d_l; fl.b = d_r; fl.nb = (u32)nr; fl.mask = mask; fl.out = d_lo;
    sa_compact(fl, (const u32*)nullptr, (u32)nl, d_chunks, d_total, (hipStream_t)0);
    MemberOf fr; fr.a = d_r; fr.b = d_l; fr.nb = (u32)nl; fr.mask = mask; fr.out = d_ro;
    sa_compact(fr, (const u32*)nullptr, (u32)nr, d_chunks, d_total + 1, (hipStream_t)0);
    SA_HIP(hipGetLastError());
    u32 g[2] = {0, 0};
    SA_HIP(hipMemcpy(g, d_total, 8, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(lhs_idx, d_lo, (size_t)g[0] * 8, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(rhs_idx, d_ro, (size_t)g[1] * 8, hipMemcpyDeviceToHost));
    *n_lhs_out = g[0]; *n_rhs_out = g[1];
    return SA_OK;
}

extern "C" int sa_adjacent(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, uint64_t mask,
                           uint64_t* lhs_idx, uint64_t* rhs_idx, int64_t* n_out) {
    if (mask == 0) { sa_set_error("Mask cannot be zero"); return SA_ERR_ARG; }
    return sa_pairs_drop(lhs, nl, rhs, nr, mask, mask & (~mask + 1), lhs_idx, rhs_idx, n_out);
}

extern "C" int sa_intersect_with_adjacents(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr,
                                           uint64_t mask, uint64_t* lhs_idx, uint64_t* rhs_idx, int64_t* n_out,
                                           uint64_t* adj_lhs_idx, uint64_t* adj_rhs_idx, int64_t* n_adj_out) {
    if (mask == 0) { sa_set_error("Mask cannot be zero"); return SA_ERR_ARG; }
    SA_TRY(sa_pairs_drop(lhs, nl, rhs, nr, mask, 0, lhs_idx, rhs_idx, n_out));
    return sa_pairs_drop(lhs, nl, rhs, nr, mask, mask & (~mask + 1), adj_lhs_idx, adj_rhs_idx, n_adj_out);
}

// ---- merge -----------------------------------------------------------------------------------
// keep: every element lands at (own index) + (rank in the other array), lhs first on ties.
__global__ void __launch_bounds__(256)
sa_k_merge_keep(const u64* __restrict__ a, u32 na, const u64* __restrict__ b, const u32* __restrict__ nb_dev, u32 nb_max,
                u64* __restrict__ out, u32* __restrict__ nout) {
    const u32 nb = nb_dev ? *nb_dev : nb_max;
    const u32 total = na + nb;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < na) {
            const u64 x = a[i];
            out[i + sa_lower_bound(b, 0, nb, x, ~0ull)] = x;            // # b strictly below x
        } else {
            const u32 j = i - na;
            const u64 x = b[j];
            // # a at or below x  (upper bound)
            u32 lo = 0, hi = na;
            while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (a[mid] <= x) lo = mid + 1; else hi = mid; }
            out[j + lo] = x;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *nout = total;
}

// drop: the k-th copy of a value in rhs survives only if lhs holds at most k copies of it
// (sequential pairing of equal elements, merge.pyx:95-132 -> multiplicity max(a, b)).
struct RhsSurvivor {
    const u64* lhs; u32 nl; const u64* rhs; u32 nr; u64* out;
    __device__ __forceinline__ bool flag(u32 j) const {
        const u64 x = rhs[j];
        const u32 k = j - sa_lower_bound(rhs, 0, nr, x, ~0ull);
        const u32 lo = sa_lower_bound(lhs, 0, nl, x, ~0ull);
        u32 a = lo, hi = nl;
        while (a < hi) { const u32 mid = a + ((hi - a) >> 1); if (lhs[mid] <= x) a = mid + 1; else hi = mid; }
        return k >= a - lo;
    }
    __device__ __forceinline__ void emit(u32 j, u32 pos) const { out[pos] = rhs[j]; }
};

extern "C" int sa_merge(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, int drop_duplicates,
                        uint64_t* out, int64_t* n_out) {
    SA_ARG(nl >= 0 && nr >= 0 && n_out, "bad argument");
    *n_out = 0;
    if (nl + nr == 0) return SA_OK;
    SA_ARG(out && (nl == 0 || lhs) && (nr == 0 || rhs), "null argument");
    SA_ARG(nl + nr < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_l, *d_r, *d_rk, *d_out; u32 *d_chunks, *d_cnt;
    SA_TRY(b.upload(&d_l, lhs, (size_t)nl));
    SA_TRY(b.upload(&d_r, rhs, (size_t)nr));
    SA_TRY(b.alloc(&d_rk, (size_t)nr));
    SA_TRY(b.alloc(&d_out, (size_t)(nl + nr)));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)(nr ? nr : 1)) + 1));
    SA_TRY(b.alloc(&d_cnt, 2));
    const u64* d_b = d_r;
    const u32* nb_dev = nullptr;
    if (drop_duplicates && nr > 0) {
        RhsSurvivor f; f.lhs = d_l; f.nl = (u32)nl; f.rhs = d_r; f.nr = (u32)nr; f.out = d_rk;
        sa_compact(f, (const u32*)nullptr, (u32)nr, d_chunks, d_cnt, (hipStream_t)0);
        d_b = d_rk;
        nb_dev = d_cnt;
    }
    hipLaunchKernelGGL(sa_k_merge_keep, dim3(sa_grid((u64)(nl + nr))), dim3(256), 0, 0, d_l, (u32)nl, d_b, nb_dev, (u32)nr,
                       d_out, d_cnt + 1);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_cnt + 1, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(out, d_out, (size_t)g * 8, hipMemcpyDeviceToHost));
    *n_out = g;
    return SA_OK;
}

// ---- sort_merge_counts: union of two id lists (ids unique within each), counts added on equal ids
__global__ void __launch_bounds__(256)
sa_k_add_matching(const u64* __restrict__ lids, float* __restrict__ lcnt, u32 nl, const u64* __restrict__ rids,
                  const float* __restrict__ rcnt, u32 nr, u32* __restrict__ absorbed) {
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < nr; j += gridDim.x * blockDim.x) {
        const u32 i = sa_lower_bound(lids, 0, nl, rids[j], ~0ull);
        if (i < nl && lids[i] == rids[j]) { lcnt[i] = lcnt[i] + rcnt[j]; absorbed[j] = 1u; }
        else absorbed[j] = 0u;
    }
}

struct KeepIdCount {
    const u64* ids; const float* cnt; const u32* absorbed; u64* ids_out; float* cnt_out;
    __device__ __forceinline__ bool flag(u32 j) const { return absorbed[j] == 0u; }
    __device__ __forceinline__ void emit(u32 j, u32 pos) const { ids_out[pos] = ids[j]; cnt_out[pos] = cnt[j]; }
};

__global__ void __launch_bounds__(256)
sa_k_merge_id_counts(const u64* __restrict__ a, const float* __restrict__ ac, u32 na, const u64* __restrict__ b,
                     const float* __restrict__ bc, const u32* __restrict__ nb_dev, u64* __restrict__ out,
                     float* __restrict__ outc, u32* __restrict__ nout) {
    const u32 nb = *nb_dev;
    const u32 total = na + nb;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < na) {
            const u32 p = i + sa_lower_bound(b, 0, nb, a[i], ~0ull);
            out[p] = a[i]; outc[p] = ac[i];
        } else {
            const u32 j = i - na;
            const u32 p = j + sa_lower_bound(a, 0, na, b[j], ~0ull);
            out[p] = b[j]; outc[p] = bc[j];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *nout = total;
}

extern "C" int sa_sort_merge_counts(const uint64_t* lhs_ids, const float* lhs_counts, int64_t nl,
                                    const uint64_t* rhs_ids, const float* rhs_counts, int64_t nr,
                                    uint64_t* ids_out, float* counts_out, int64_t* n_out) {
    SA_ARG(nl >= 0 && nr >= 0 && n_out, "bad argument");
    *n_out = 0;
    if (nl + nr == 0) return SA_OK;
    SA_ARG(ids_out && counts_out, "null output");
    SA_ARG(nl + nr < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_li, *d_ri, *d_rki, *d_oi; float *d_lc, *d_rc, *d_rkc, *d_oc; u32 *d_abs, *d_chunks, *d_cnt;
    SA_TRY(b.upload(&d_li, lhs_ids, (size_t)nl));
    SA_TRY(b.upload(&d_lc, lhs_counts, (size_t)nl));
    SA_TRY(b.upload(&d_ri, rhs_ids, (size_t)nr));
    SA_TRY(b.upload(&d_rc, rhs_counts, (size_t)nr));
    SA_TRY(b.alloc(&d_rki, (size_t)nr)); SA_TRY(b.alloc(&d_rkc, (size_t)nr)); SA_TRY(b.alloc(&d_abs, (size_t)nr));
    SA_TRY(b.alloc(&d_oi, (size_t)(nl + nr))); SA_TRY(b.alloc(&d_oc, (size_t)(nl + nr)));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)(nr ? nr : 1)) + 1));
    SA_TRY(b.alloc(&d_cnt, 2));
    SA_HIP(hipMemset(d_cnt, 0, 8));
    if (nr) {
        hipLaunchKernelGGL(sa_k_add_matching, dim3(sa_grid((u64)nr)), dim3(256), 0, 0, d_li, d_lc, (u32)nl, d_ri, d_rc, (u32)nr, d_abs);
        KeepIdCount f; f.ids = d_ri; f.cnt = d_rc; f.absorbed = d_abs; f.ids_out = d_rki; f.cnt_out = d_rkc;
        sa_compact(f, (const u32*)nullptr, (u32)nr, d_chunks, d_cnt, (hipStream_t)0);
    }
    hipLaunchKernelGGL(sa_k_merge_id_counts, dim3(sa_grid((u64)(nl + nr))), dim3(256), 0, 0, d_li, d_lc, (u32)nl, d_rki, d_rkc,
                       d_cnt, d_oi, d_oc, d_cnt + 1);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_cnt + 1, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(ids_out, d_oi, (size_t)g * 8, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(counts_out, d_oc, (size_t)g * 4, hipMemcpyDeviceToHost));
    *n_out = g;
    return SA_OK;
}

// ---- popcount_reduce_at / key_sum_over: groups of consecutive equal ids ----------------------
struct IdRuns {
    const u64* ids; const u64* vals; u32 n; int popcount; u64* ids_out; float* cnt_out;
    __device__ __forceinline__ bool flag(u32 i) const { return i == 0 || ids[i] != ids[i - 1]; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 id = ids[i];
        u64 sum = 0;                                          // u64 accumulator, stored as float (popcount.pyx:128,139)
        for (u32 j = i; j < n && ids[j] == id; j++) sum += popcount ? (u64)__popcll(vals[j]) : vals[j];
        ids_out[pos] = id;
        cnt_out[pos] = (float)sum;
    }
};

static int sa_id_runs(const uint64_t* ids, const uint64_t* vals, int64_t n, int popcount, uint64_t* ids_out,
                      float* counts_out, int64_t* n_out) {
    SA_ARG(n >= 0 && n_out, "bad argument");
    *n_out = 0;
    if (n == 0) return SA_OK;
    SA_ARG(ids && vals && ids_out && counts_out, "null argument");
    SA_ARG(n < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_i, *d_v, *d_oi; float* d_oc; u32 *d_chunks, *d_total;
    SA_TRY(b.upload(&d_i, ids, (size_t)n));
    SA_TRY(b.upload(&d_v, vals, (size_t)n));
    SA_TRY(b.alloc(&d_oi, (size_t)n)); SA_TRY(b.alloc(&d_oc, (size_t)n));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)n) + 1));
    SA_TRY(b.alloc(&d_total, 1));
    IdRuns f; f.ids = d_i; f.vals = d_v; f.n = (u32)n; f.popcount = popcount; f.ids_out = d_oi; f.cnt_out = d_oc;
    sa_compact(f, (const u32*)nullptr, (u32)n, d_chunks, d_total, (hipStream_t)0);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_total, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(ids_out, d_oi, (size_t)g * 8, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(counts_out, d_oc, (size_t)g * 4, hipMemcpyDeviceToHost));
    *n_out = g;
    return SA_OK;
}

extern "C" int sa_popcount_reduce_at(const uint64_t* ids, const uint64_t* payload, int64_t n, uint64_t* ids_out,
                                     float* counts_out, int64_t* n_out) {
    return sa_id_runs(ids, payload, n, 1, ids_out, counts_out, n_out);
}

extern "C" int sa_key_sum_over(const uint64_t* ids, const uint64_t* count, int64_t n, uint64_t* ids_out,
                               float* counts_out, int64_t* n_out) {
    return sa_id_runs(ids, count, n, 0, ids_out, counts_out, n_out);
}

// ---- payload_slice: reference roaringish_ops.pyx:46-60 (compares the UNSHIFTED masked msb) ---
struct PayloadSlice {
    const u64* arr; u64 msb_mask, lo, hi; u64* out;
    __device__ __forceinline__ bool flag(u32 i) const { const u64 v = arr[i] & msb_mask; return v >= lo && v <= hi; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = arr[i]; }
};

extern "C" int sa_payload_slice(const uint64_t* arr, int64_t n, uint64_t payload_msb_mask, uint64_t min_payload,
                                uint64_t max_payload, uint64_t* out, int64_t* n_out) {
    SA_ARG(n >= 0 && n_out, "bad argument");
    *n_out = 0;
    if (n == 0) return SA_OK;
    SA_ARG(arr && out, "null argument");
    SA_ARG(n < 0xFFFFF000ll, "array too long");
    SetBufs b;
    u64 *d_a, *d_o; u32 *d_chunks, *d_total;
    SA_TRY(b.upload(&d_a, arr, (size_t)n));
    SA_TRY(b.alloc(&d_o, (size_t)n));
    SA_TRY(b.alloc(&d_chunks, sa_compact_chunks((u32)n) + 1));
    SA_TRY(b.alloc(&d_total, 1));
    PayloadSlice f; f.arr = d_a; f.msb_mask = payload_msb_mask; f.lo = min_payload; f.hi = max_payload; f.out = d_o;
    sa_compact(f, (const u32*)nullptr, (u32)n, d_chunks, d_total, (hipStream_t)0);
    SA_HIP(hipGetLastError());
    u32 g = 0;
    SA_HIP(hipMemcpy(&g, d_total, 4, hipMemcpyDeviceToHost));
    SA_HIP(hipMemcpy(out, d_o, (size_t)g * 8, hipMemcpyDeviceToHost));
    *n_out = g;
    return SA_OK;
}

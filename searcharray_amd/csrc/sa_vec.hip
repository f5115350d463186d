// sa_vec.hip -- Part 4 of the C ABI: dense per-doc vectors that stay on the device, and the handful of
// elementwise combinations Solr-style multi-field queries need (reference searcharray/solr.py:112-355:
// edismax sums / maxes / masks the per-field, per-term score() vectors with numpy on the host -- T x F
// dense vectors of n_docs floats cross PCIe and are then combined at host memory speed).
//
// A vector is n float32 or float64 values in HBM.  Scores enter with sa_index_score_vec (the dense BM25
// of a term or phrase, written device-to-device instead of to the host); the combinations below follow
// numpy's arithmetic exactly -- float64 accumulators fed by float32 scores where the reference uses
// np.zeros(n) accumulators (term-centric), float32 throughout where it stacks float32 arrays
// (field-centric); every operation rounds once (no FMA contraction) -- so the result equals the host
// path bit for bit.  The API is synchronous: every call returns with its work done.
#include "sa_index.hpp"
#include "../../include/searcharray_hip.h"
#include <new>

struct sa_vec {
    int device = 0;
    u64 n = 0;
    int f64 = 0;
    void* d = nullptr;
};

static hipStream_t sa_vec_stream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    std::lock_guard<std::mutex> g(mu);
    if (device < 0 || device >= 64) return nullptr;
    if (!streams[device]) {
        hipSetDevice(device);
        hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
    }
    return streams[device];
}

static inline u32 sa_vgrid(u64 n) { return n / 256 + 1 < 65536 ? (u32)(n / 256 + 1) : 65536u; }
#define SA_VEC_LOOP(i, n) for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (u64)gridDim.x * blockDim.x)

extern "C" int sa_vec_create(int device, uint64_t n, int is_f64, sa_vec_t** out) {
    SA_ARG(out, "out is null");
    SA_HIP(hipSetDevice(device));
    sa_vec* v = new (std::nothrow) sa_vec();
    if (!v) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    v->device = device; v->n = n; v->f64 = is_f64 ? 1 : 0;
    const size_t bytes = (size_t)(n ? n : 1) * (is_f64 ? 8 : 4);
    if (hipMalloc(&v->d, bytes) != hipSuccess) { delete v; sa_set_error("hipMalloc failed (vector)"); return SA_ERR_HIP; }
    SA_HIP(hipMemset(v->d, 0, bytes));
    *out = v;
    return SA_OK;
}

extern "C" int sa_vec_destroy(sa_vec_t* v) {
    if (!v) return SA_OK;
    hipSetDevice(v->device);
    if (v->d) hipFree(v->d);
    delete v;
    return SA_OK;
}

extern "C" int sa_vec_zero(sa_vec_t* v) {
    SA_ARG(v, "null vector");
    SA_HIP(hipSetDevice(v->device));
    SA_HIP(hipMemset(v->d, 0, (size_t)v->n * (v->f64 ? 8 : 4)));
    return SA_OK;
}

extern "C" int sa_vec_copy(sa_vec_t* dst, const sa_vec_t* src) {
    SA_ARG(dst && src && dst->n == src->n && dst->f64 == src->f64 && dst->device == src->device, "vectors differ");
    SA_HIP(hipSetDevice(dst->device));
    if (dst->n) SA_HIP(hipMemcpy(dst->d, src->d, (size_t)dst->n * (dst->f64 ? 8 : 4), hipMemcpyDeviceToDevice));
    return SA_OK;
}

extern "C" int sa_vec_fetch(sa_vec_t* v, void* host_out) {
    SA_ARG(v && (host_out || v->n == 0), "null argument");
    SA_HIP(hipSetDevice(v->device));
    if (v->n) SA_HIP(hipMemcpy(host_out, v->d, (size_t)v->n * (v->f64 ? 8 : 4), hipMemcpyDeviceToHost));
    return SA_OK;
}

// ---- kernels -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sa_k_vec_scale_copy(const float* __restrict__ src, float boost, int has_boost, float* __restrict__ dst, u64 n) {
    SA_VEC_LOOP(i, n) dst[i] = has_boost ? __fmul_rn(src[i], boost) : src[i];
}
// term-centric inner step (solr.py:131-133): sum += s ; mx = maximum(mx, s)        (float64 <- float32)
__global__ void __launch_bounds__(256) sa_k_vec_dismax_acc(double* __restrict__ sum, double* __restrict__ mx, const float* __restrict__ s, u64 n) {
    SA_VEC_LOOP(i, n) { const double v = (double)s[i]; sum[i] = __dadd_rn(sum[i], v); mx[i] = v > mx[i] ? v : mx[i]; }
}
// clause = mx + (sum - mx) * tie ; total += clause ; cnt += clause > 0            (solr.py:135-143)
__global__ void __launch_bounds__(256) sa_k_vec_clause(const double* __restrict__ sum, const double* __restrict__ mx, double tie,
                                                       double* __restrict__ total, u32* __restrict__ cnt, u64 n) {
    SA_VEC_LOOP(i, n) {
        const double c = __dadd_rn(mx[i], __dmul_rn(__dsub_rn(sum[i], mx[i]), tie));
        total[i] = __dadd_rn(total[i], c);
        cnt[i] += c > 0.0 ? 1u : 0u;
    }
}
__global__ void __launch_bounds__(256) sa_k_vec_mask64(double* __restrict__ v, const u32* __restrict__ cnt, u32 need, u64 n) {
    SA_VEC_LOOP(i, n) if (cnt[i] < need) v[i] = 0.0;
}
// field-centric: sum32 += s ; cnt += s > 0                                          (solr.py:159-166)
__global__ void __launch_bounds__(256) sa_k_vec_sum_count32(float* __restrict__ sum, u32* __restrict__ cnt, const float* __restrict__ s, u64 n) {
    SA_VEC_LOOP(i, n) { sum[i] = __fadd_rn(sum[i], s[i]); cnt[i] += s[i] > 0.f ? 1u : 0u; }
}
// row = (cnt >= need ? sum : 0) * boost ; fsum += row ; fmax = max(fmax, row)        (solr.py:166-173)
__global__ void __launch_bounds__(256) sa_k_vec_field_row(const float* __restrict__ sum, const u32* __restrict__ cnt, u32 need, float boost,
                                                          int has_boost, int first, float* __restrict__ fsum, float* __restrict__ fmax, u64 n) {
    SA_VEC_LOOP(i, n) {
        float r = cnt[i] >= need ? sum[i] : 0.f;
        if (has_boost) r = __fmul_rn(r, boost);
        fsum[i] = first ? r : __fadd_rn(fsum[i], r);
        fmax[i] = first ? r : (r > fmax[i] ? r : fmax[i]);
    }
}
// out = fmax + (fsum - fmax) * tie                                                  (solr.py:174-175, float32)
__global__ void __launch_bounds__(256) sa_k_vec_field_finish(const float* __restrict__ fsum, const float* __restrict__ fmax, float tie, float* __restrict__ out, u64 n) {
    SA_VEC_LOOP(i, n) out[i] = __fadd_rn(fmax[i], __fmul_rn(__fsub_rn(fsum[i], fmax[i]), tie));
}
__global__ void __launch_bounds__(256) sa_k_vec_add32(float* __restrict__ dst, const float* __restrict__ src, int first, u64 n) {
    SA_VEC_LOOP(i, n) dst[i] = first ? src[i] : __fadd_rn(dst[i], src[i]);
}
// dst[i] += extra[i] where mask[i] != 0                                             (solr.py:337-351)
template <class D, class M>
__global__ void __launch_bounds__(256) sa_k_vec_add_where(D* __restrict__ dst, const float* __restrict__ extra, const M* __restrict__ mask, u64 n) {
    SA_VEC_LOOP(i, n) if (mask[i] != (M)0) dst[i] = (D)(dst[i] + (D)extra[i]);
}
template <class M>
__global__ void __launch_bounds__(256) sa_k_vec_count_where(const float* __restrict__ tf, const M* __restrict__ mask, unsigned long long* __restrict__ out, u64 n) {
    unsigned long long c = 0;
    SA_VEC_LOOP(i, n) c += (tf[i] > 0.f && mask[i] > (M)0) ? 1ull : 0ull;
    for (int o = SA_WAVE / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, SA_WAVE);
    if ((threadIdx.x & (SA_WAVE - 1)) == 0 && c) atomicAdd(out, c);
}

#define SA_VEC_SAME(a, b) SA_ARG((a) && (b) && (a)->n == (b)->n && (a)->device == (b)->device, "vectors differ in length or device")
#define SA_VEC_RUN(v, kernel, ...)                                                                    \
    do {                                                                                              \
        SA_HIP(hipSetDevice((v)->device));                                                            \
        hipStream_t st_ = sa_vec_stream((v)->device);                                                 \
        if ((v)->n) hipLaunchKernelGGL(kernel, dim3(sa_vgrid((v)->n)), dim3(256), 0, st_, __VA_ARGS__); \
        SA_HIP(hipStreamSynchronize(st_));                                                            \
        SA_HIP(hipGetLastError());                                                                    \
    } while (0)

extern "C" int sa_vec_dismax_acc(sa_vec_t* sum64, sa_vec_t* max64, const sa_vec_t* s32) {
    SA_VEC_SAME(sum64, max64); SA_VEC_SAME(sum64, s32);
    SA_ARG(sum64->f64 && max64->f64 && !s32->f64, "expected (f64, f64, f32)");
    SA_VEC_RUN(sum64, sa_k_vec_dismax_acc, (double*)sum64->d, (double*)max64->d, (const float*)s32->d, sum64->n);
    return SA_OK;
}

extern "C" int sa_vec_clause(const sa_vec_t* sum64, const sa_vec_t* max64, double tie, sa_vec_t* total64, sa_vec_t* cnt32) {
    SA_VEC_SAME(sum64, max64); SA_VEC_SAME(sum64, total64); SA_VEC_SAME(sum64, cnt32);
    SA_ARG(sum64->f64 && max64->f64 && total64->f64 && !cnt32->f64, "expected (f64, f64, f64, u32)");
    SA_VEC_RUN(sum64, sa_k_vec_clause, (const double*)sum64->d, (const double*)max64->d, tie, (double*)total64->d, (u32*)cnt32->d, sum64->n);
    return SA_OK;
}

extern "C" int sa_vec_mask_min_count(sa_vec_t* v64, const sa_vec_t* cnt32, uint32_t need) {
    SA_VEC_SAME(v64, cnt32);
    SA_ARG(v64->f64 && !cnt32->f64, "expected (f64, u32)");
    SA_VEC_RUN(v64, sa_k_vec_mask64, (double*)v64->d, (const u32*)cnt32->d, need, v64->n);
    return SA_OK;
}

extern "C" int sa_vec_sum_count32(sa_vec_t* sum32, sa_vec_t* cnt32, const sa_vec_t* s32) {
    SA_VEC_SAME(sum32, cnt32); SA_VEC_SAME(sum32, s32);
    SA_ARG(!sum32->f64 && !cnt32->f64 && !s32->f64, "expected 32-bit vectors");
    SA_VEC_RUN(sum32, sa_k_vec_sum_count32, (float*)sum32->d, (u32*)cnt32->d, (const float*)s32->d, sum32->n);
    return SA_OK;
}

extern "C" int sa_vec_field_row(const sa_vec_t* sum32, const sa_vec_t* cnt32, uint32_t need, float boost, int has_boost, int first,
                                sa_vec_t* fsum32, sa_vec_t* fmax32) {
    SA_VEC_SAME(sum32, cnt32); SA_VEC_SAME(sum32, fsum32); SA_VEC_SAME(sum32, fmax32);
    SA_ARG(!sum32->f64 && !cnt32->f64 && !fsum32->f64 && !fmax32->f64, "expected 32-bit vectors");
    SA_VEC_RUN(sum32, sa_k_vec_field_row, (const float*)sum32->d, (const u32*)cnt32->d, need, boost, has_boost, first,
               (float*)fsum32->d, (float*)fmax32->d, sum32->n);
    return SA_OK;
}

extern "C" int sa_vec_field_finish(const sa_vec_t* fsum32, const sa_vec_t* fmax32, float tie, sa_vec_t* out32) {
    SA_VEC_SAME(fsum32, fmax32); SA_VEC_SAME(fsum32, out32);
    SA_ARG(!fsum32->f64 && !fmax32->f64 && !out32->f64, "expected 32-bit vectors");
    SA_VEC_RUN(fsum32, sa_k_vec_field_finish, (const float*)fsum32->d, (const float*)fmax32->d, tie, (float*)out32->d, fsum32->n);
    return SA_OK;
}

extern "C" int sa_vec_add32(sa_vec_t* dst32, const sa_vec_t* src32, int first) {
    SA_VEC_SAME(dst32, src32);
    SA_ARG(!dst32->f64 && !src32->f64, "expected 32-bit vectors");
    SA_VEC_RUN(dst32, sa_k_vec_add32, (float*)dst32->d, (const float*)src32->d, first, dst32->n);
    return SA_OK;
}

extern "C" int sa_vec_add_where(sa_vec_t* dst, const sa_vec_t* extra32, const sa_vec_t* mask) {
    SA_VEC_SAME(dst, extra32); SA_VEC_SAME(dst, mask);
    SA_ARG(!extra32->f64 && dst->f64 == mask->f64, "expected (T, f32, T)");
    if (dst->f64) SA_VEC_RUN(dst, (sa_k_vec_add_where<double, double>), (double*)dst->d, (const float*)extra32->d, (const double*)mask->d, dst->n);
    else SA_VEC_RUN(dst, (sa_k_vec_add_where<float, float>), (float*)dst->d, (const float*)extra32->d, (const float*)mask->d, dst->n);
    return SA_OK;
}

// number of docs with tf32[i] > 0 and mask[i] > 0 (the subset-local docfreq of a sliced array)
extern "C" int sa_vec_count_where(const sa_vec_t* tf32, const sa_vec_t* mask, uint64_t* out) {
    SA_VEC_SAME(tf32, mask);
    SA_ARG(out && !tf32->f64, "expected (f32, mask)");
    SA_HIP(hipSetDevice(tf32->device));
    hipStream_t st = sa_vec_stream(tf32->device);
    unsigned long long* d_c = nullptr;
    SA_HIP(hipMalloc(&d_c, sizeof(unsigned long long)));
    SA_HIP(hipMemsetAsync(d_c, 0, sizeof(unsigned long long), st));
    if (tf32->n) {
        if (mask->f64) hipLaunchKernelGGL((sa_k_vec_count_where<double>), dim3(sa_vgrid(tf32->n)), dim3(256), 0, st, (const float*)tf32->d, (const double*)mask->d, d_c, tf32->n);
        else hipLaunchKernelGGL((sa_k_vec_count_where<float>), dim3(sa_vgrid(tf32->n)), dim3(256), 0, st, (const float*)tf32->d, (const float*)mask->d, d_c, tf32->n);
    }
    unsigned long long h = 0;
    SA_HIP(hipMemcpyAsync(&h, d_c, sizeof(h), hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    hipFree(d_c);
    *out = h;
    return SA_OK;
}

// The next dense call on `ix` made by this thread (sa_index_bm25_dense, sa_index_termfreqs_dense*,
// sa_index_phrase_freqs_dense*, sa_index_bm25_phrase_dense*) writes its float32[n_docs] result, times
// boost, into `out32` on the device instead of copying it to the host (pass any non-null `out`).
static thread_local struct { const sa_index* ix; sa_vec* v; float boost; int has_boost; } tl_vec = {nullptr, nullptr, 1.f, 0};

extern "C" int sa_index_select_vec(sa_index_t* ix, sa_vec_t* out32, float boost, int has_boost) {
    SA_ARG(ix, "null index");
    if (!out32) { tl_vec.ix = nullptr; tl_vec.v = nullptr; return SA_OK; }
    SA_ARG(!out32->f64 && out32->n == ix->n_docs && out32->device == ix->device, "vector must be float32[n_docs] on the index's device");
    tl_vec.ix = ix; tl_vec.v = out32; tl_vec.boost = boost; tl_vec.has_boost = has_boost;
    return SA_OK;
}

// a diversion is pending for this thread's next dense call on `ix`; clear = drop it
bool sa_vec_target_pending(const sa_index* ix, bool clear) {
    const bool pending = tl_vec.ix == ix;
    if (pending && clear) { tl_vec.ix = nullptr; tl_vec.v = nullptr; }
    return pending;
}

// the diversion pending for this thread's next dense call on `ix`, handed to a kernel that writes the vector itself
// (sa_index_bm25_dense's one-launch route): destination, boost; the diversion is consumed
bool sa_vec_target_take(const sa_index* ix, float** dst, float* boost, int* has_boost) {
    if (tl_vec.ix != ix || !tl_vec.v || !tl_vec.v->n) return false;
    *dst = (float*)tl_vec.v->d; *boost = tl_vec.boost; *has_boost = tl_vec.has_boost;
    tl_vec.ix = nullptr; tl_vec.v = nullptr;
    return true;
}

// called by sa_emit_dense / sa_emit_zeros (sa_index.hip): true if the result was diverted into a vector
bool sa_emit_to_vec(sa_index* ix, const float* d_vec) {
    if (tl_vec.ix != ix) return false;
    sa_vec* v = tl_vec.v;
    const float boost = tl_vec.boost;
    const int has_boost = tl_vec.has_boost;
    tl_vec.ix = nullptr; tl_vec.v = nullptr;
    if (!v->n) return true;
    if (d_vec) hipLaunchKernelGGL(sa_k_vec_scale_copy, dim3(sa_vgrid(v->n)), dim3(256), 0, ix->stream, d_vec, boost, has_boost, (float*)v->d, v->n);
    else hipMemsetAsync(v->d, 0, (size_t)v->n * 4, ix->stream);
    return true;
}


// ---- the dense calls with an explicit destination (header: sa_dense_dest_t): the subset / device-vector routes without a
//      selection that outlives a call.  The selection mechanism underneath is per thread and is armed and consumed inside one
//      call here; a failed call disarms it.
static int sa_dest_arm(sa_index_t* ix, const sa_dense_dest_t* d) {
    if (!d) return SA_OK;
    SA_ARG(!(d->rows && d->vec), "sa_dense_dest: rows or vec, not both");
    if (d->vec) return sa_index_select_vec(ix, d->vec, d->boost, d->has_boost);
    if (d->rows) return sa_index_select_rows(ix, d->rows, d->n_rows);
    return SA_OK;
}
static int sa_dest_done(sa_index_t* ix, const sa_dense_dest_t* d, int rc) {
    if (rc != SA_OK && d) {
        const std::string keep = sa_last_error();
        if (d->vec) sa_index_select_vec(ix, nullptr, 1.f, 0);
        if (d->rows) sa_index_select_rows(ix, nullptr, 0);
        sa_set_error("%s", keep.c_str());
    }
    return rc;
}
#define SA_TO(call)                                   \
    SA_ARG(ix, "null index");                         \
    float dummy_ = 0.f;                               \
    void* out_ = (dest && dest->vec) ? (void*)&dummy_ : (void*)out; \
    (void)out_;                                       \
    SA_TRY(sa_dest_arm(ix, dest));                    \
    return sa_dest_done(ix, dest, call)

extern "C" int sa_index_termfreqs_dense_to(sa_index_t* ix, uint32_t term, const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_termfreqs_dense(ix, term, (float*)out_));
}
extern "C" int sa_index_termfreqs_dense_posn_to(sa_index_t* ix, uint32_t term, int64_t min_posn, int64_t max_posn,
                                                const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_termfreqs_dense_posn(ix, term, min_posn, max_posn, (float*)out_));
}
extern "C" int sa_index_bm25_dense_to(sa_index_t* ix, const uint32_t* terms, const float* idf, int n_query_terms, float k1, float b,
                                      const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_bm25_dense(ix, terms, idf, n_query_terms, k1, b, (float*)out_));
}
extern "C" int sa_index_phrase_freqs_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                              const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_phrase_freqs_dense(ix, terms, n_terms, slop, (float*)out_));
}
extern "C" int sa_index_phrase_freqs_dense_posn_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                                   int64_t max_posn, const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_phrase_freqs_dense_posn(ix, terms, n_terms, slop, min_posn, max_posn, (float*)out_));
}
extern "C" int sa_index_bm25_phrase_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, float idf, float k1,
                                             float b, const sa_dense_dest_t* dest, float* out) {
    SA_TO(sa_index_bm25_phrase_dense(ix, terms, n_terms, slop, idf, k1, b, (float*)out_));
}
extern "C" int sa_index_bm25_phrase_dense_posn_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                                  int64_t max_posn, float idf, float k1, float b, const sa_dense_dest_t* dest,
                                                  float* out) {
    SA_TO(sa_index_bm25_phrase_dense_posn(ix, terms, n_terms, slop, min_posn, max_posn, idf, k1, b, (float*)out_));
}
extern "C" int sa_index_similarity_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                            int64_t max_posn, int kind, double idf, double k1, double b,
                                            const sa_dense_dest_t* dest, void* out) {
    SA_TO(sa_index_similarity_dense(ix, terms, n_terms, slop, min_posn, max_posn, kind, idf, k1, b, out_));
}

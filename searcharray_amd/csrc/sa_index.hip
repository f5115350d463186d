// sa_index.hip -- index upload and on-device derivation of TF postings / document frequencies.
//
// Replaces, on the device, what the reference computes lazily per term on the host:
//   df  = len(unique(words >> 36))                  reference unique.pyx:87-104, middle_out.py:521-528
//   tf  = popcount64_reduce(words, 36, 0x3FFFF)     reference popcount.pyx:212-237, middle_out.py:501-512
// for ALL terms at once (= PosnBitArray.warm(), middle_out.py:337-342, without the >255 cut-off):
// one segmented popcount-reduce over the whole term-major word array, segment = (term, doc).
#include "sa_index.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"
#include <algorithm>

#include <math.h>
#include <stdlib.h>
#include <new>

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static thread_local std::string g_sa_error;

void sa_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_sa_error = buf;
}

extern "C" const char* sa_last_error(void) { return g_sa_error.c_str(); }
extern "C" int sa_abi_version(void) { return SA_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// Options (include/searcharray_hip.h, Part 0; sa_options.hpp)
// ---------------------------------------------------------------------------------------------
namespace {
struct SaOptName { const char* name; size_t off; };
static const SaOptName g_sa_opt_names[] = {
#define SA_X(f) {#f, offsetof(sa_options_fields, f)},
    SA_OPTION_LIST(SA_X)
#undef SA_X
};
constexpr int SA_N_OPTS = (int)(sizeof(g_sa_opt_names) / sizeof(g_sa_opt_names[0]));
static int64_t* sa_opt_slot(sa_options_t* o, const char* name) {
    for (int i = 0; i < SA_N_OPTS; i++)
        if (!strcmp(g_sa_opt_names[i].name, name)) return (int64_t*)((char*)o + g_sa_opt_names[i].off);
    return nullptr;
}
thread_local bool t_sa_opts_set = false;
thread_local sa_options_t t_sa_opts;
}  // namespace

extern "C" void sa_options_init(sa_options_t* o) {
    if (!o) return;
    o->struct_size = sizeof(sa_options_t);
    for (int i = 0; i < SA_N_OPTS; i++) *(int64_t*)((char*)o + g_sa_opt_names[i].off) = SA_OPT_UNSET;
}
extern "C" int sa_option_count(void) { return SA_N_OPTS; }
extern "C" const char* sa_option_name(int i) { return i >= 0 && i < SA_N_OPTS ? g_sa_opt_names[i].name : nullptr; }
extern "C" int sa_options_set(sa_options_t* o, const char* name, int64_t value) {
    SA_ARG(o && name, "null argument");
    int64_t* slot = sa_opt_slot(o, name);
    if (!slot) { sa_set_error("unknown option '%s'", name); return SA_ERR_ARG; }
    *slot = value;
    return SA_OK;
}
extern "C" int sa_options_get(const sa_options_t* o, const char* name, int64_t* value_out) {
    SA_ARG(o && name && value_out, "null argument");
    const int64_t* slot = sa_opt_slot(const_cast<sa_options_t*>(o), name);
    if (!slot) { sa_set_error("unknown option '%s'", name); return SA_ERR_ARG; }
    *value_out = *slot;
    return SA_OK;
}
// The process defaults: everything unset, plus the ONE environment variable the library reads -- SA_OPTS="name=value,name=value"
// (a debug override for binaries whose caller cannot pass options), parsed once.
const sa_options_t& sa_options_process_defaults() {
    static const sa_options_t defaults = [] {
        sa_options_t o;
        sa_options_init(&o);
        if (const char* env = getenv("SA_OPTS")) {
            std::string spec(env);
            size_t pos = 0;
            while (pos < spec.size()) {
                size_t end = spec.find(',', pos);
                if (end == std::string::npos) end = spec.size();
                const std::string item = spec.substr(pos, end - pos);
                const size_t eq = item.find('=');
                if (eq != std::string::npos) {
                    const std::string name = item.substr(0, eq);
                    if (int64_t* slot = sa_opt_slot(&o, name.c_str())) *slot = (int64_t)strtoll(item.c_str() + eq + 1, nullptr, 10);
                    else fprintf(stderr, "libsearcharray_hip: SA_OPTS: unknown option '%s' ignored\n", name.c_str());
                }
                pos = end + 1;
            }
        }
        return o;
    }();
    return defaults;
}
extern "C" int sa_options_process_defaults_get(sa_options_t* out) {
    SA_ARG(out, "null argument");
    *out = sa_options_process_defaults();
    return SA_OK;
}
static int sa_options_check(const sa_options_t* o) {
    if (o && o->struct_size != sizeof(sa_options_t)) {
        sa_set_error("sa_options_t: struct_size %llu, this library expects %zu (fill it with sa_options_init)", (unsigned long long)o->struct_size, sizeof(sa_options_t));
        return SA_ERR_ARG;
    }
    return SA_OK;
}
extern "C" int sa_options_set_thread_defaults(const sa_options_t* o) {
    SA_TRY(sa_options_check(o));
    t_sa_opts_set = o != nullptr;
    if (o) t_sa_opts = *o;
    return SA_OK;
}
// what a handle created by this thread starts from, per switch: the thread's default, else `fallback` (a batch: its index's
// option), else the process default
// (field by field: a thread default that sets ONE switch does not erase the index's other options or the SA_OPTS override)
sa_options_t sa_options_for_new_handle(const sa_options_t* fallback) {
    sa_options_t o = sa_options_process_defaults();
#define SA_X(f) if (fallback && sa_opt_is_set(fallback->f)) o.f = fallback->f;
    SA_OPTION_LIST(SA_X)
#undef SA_X
    if (t_sa_opts_set) {
#define SA_X(f) if (sa_opt_is_set(t_sa_opts.f)) o.f = t_sa_opts.f;
        SA_OPTION_LIST(SA_X)
#undef SA_X
    }
    return o;
}
bool sa_options_thread_defaults(sa_options_t* out) {
    if (t_sa_opts_set && out) *out = t_sa_opts;
    return t_sa_opts_set;
}
extern "C" int sa_index_set_options(sa_index_t* ix, const sa_options_t* o) {
    SA_ARG(ix && o, "null argument");
    SA_TRY(sa_options_check(o));
    std::lock_guard<std::mutex> g(ix->mu);
    ix->opts = *o;
    return SA_OK;
}
extern "C" int sa_index_get_options(sa_index_t* ix, sa_options_t* out) {
    SA_ARG(ix && out, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    *out = ix->opts;
    return SA_OK;
}


extern "C" int sa_device_count(int* out_count) {
    SA_ARG(out_count, "out_count is null");
    int n = 0;
    SA_HIP(hipGetDeviceCount(&n));
    *out_count = n;
    return SA_OK;
}

extern "C" int sa_device_name(int device, char* buf, int buf_len) {
    SA_ARG(buf && buf_len > 0, "buf");
    hipDeviceProp_t p;
    SA_HIP(hipGetDeviceProperties(&p, device));
    snprintf(buf, (size_t)buf_len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return SA_OK;
}

__global__ void __launch_bounds__(1024)
sa_k_scan_chunks(u32* __restrict__ counts, u32 nchunks, u32* __restrict__ total_out) {
    __shared__ u32 red[16];
    u32 carry = 0;
    for (u32 base = 0; base < nchunks; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nchunks ? counts[i] : 0u;
        u32 tot;
        const u32 ex = sa_block_excl_scan<16>(v, red, &tot);
        if (i < nchunks) counts[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

int sa_index_scratch(sa_index* ix, size_t bytes, void** out) {
    if (bytes > ix->scratch_bytes) {
        if (ix->d_scratch) SA_HIP(hipFree(ix->d_scratch));
        ix->d_scratch = nullptr;
        ix->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        SA_HIP(hipMalloc(&ix->d_scratch, want));
        ix->scratch_bytes = want;
    }
    *out = ix->d_scratch;
    return SA_OK;
}

// A dense call on one of the index's lanes (sa_index.hpp, DenseLane).  Constructed with the index lock held: takes a free lane
// (waits for one otherwise) and installs its stream and scratch as the index's, so the call's body -- written against
// ix->stream and sa_index_scratch -- enqueues on the lane.  finish(): records the lane's event, puts the index's own state back,
// RELEASES the lock and waits for the event outside it; other threads' calls enqueue meanwhile.  A call that fails before
// finish() is wound up by the destructor (under the lock, synchronising the lane).
SaDenseLaneScope::SaDenseLaneScope(sa_index* ix_, std::unique_lock<std::mutex>& lk_) : ix(ix_), lk(lk_) {
    for (;;) {
        for (auto& ln : ix->dense_lane)
            if (!ln.busy) { lane = &ln; break; }
        if (lane) break;
        ix->dense_lane_cv.wait(lk);
    }
    lane->busy = true;
    if (!lane->stream && hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) { rc = SA_ERR_HIP; sa_set_error("dense lane: hipStreamCreate failed"); }
    if (rc == SA_OK && !lane->done && hipEventCreateWithFlags(&lane->done, hipEventDisableTiming) != hipSuccess) { rc = SA_ERR_HIP; sa_set_error("dense lane: hipEventCreate failed"); }
    swap();
}
void SaDenseLaneScope::swap() {
    std::swap(ix->stream, lane->stream);
    std::swap(ix->d_scratch, lane->scratch);
    std::swap(ix->scratch_bytes, lane->scratch_bytes);
    std::swap(ix->d_rows_scratch, lane->rows);
    std::swap(ix->rows_scratch_bytes, lane->rows_bytes);
    std::swap(ix->d_sim_scratch, lane->sim);
    std::swap(ix->sim_scratch_bytes, lane->sim_bytes);
    swapped = !swapped;
}
int SaDenseLaneScope::finish() {
    hipEvent_t done = lane->done;
    const hipError_t e = hipEventRecord(done, ix->stream);      // (ix->stream IS the lane's stream here)
    swap();
    if (e != hipSuccess) { sa_set_error("dense lane: hipEventRecord failed"); return SA_ERR_HIP; }
    lk.unlock();
    const hipError_t w = hipEventSynchronize(done);
    lk.lock();
    if (w != hipSuccess) { sa_set_error("dense call failed on the device: %s", hipGetErrorString(w)); return SA_ERR_HIP; }
    return SA_OK;
}
SaDenseLaneScope::~SaDenseLaneScope() {
    if (swapped) {                                              // (left early: nothing of this call may still be in flight on the lane)
        hipStreamSynchronize(ix->stream);
        swap();
    }
    lane->busy = false;
    ix->dense_lane_cv.notify_one();
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------

// bit i of term_start is set iff a (non-empty) term's words begin at i
__global__ void sa_k_mark_term_starts(const u64* __restrict__ term_off, u32 n_terms, u32* __restrict__ bits) {
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < n_terms; t += gridDim.x * blockDim.x) {
        const u64 a = term_off[t], b = term_off[t + 1];
        if (a < b) atomicOr(&bits[a >> 5], 1u << (a & 31));
    }
}

// A word is a posting head when it opens a new (term, doc) group.  The compaction runs over one
// SEGMENT of the index at a time -- a run of whole terms with fewer than 2^31 words -- so its 32-bit
// element indices are segment-relative; i_base / p_base place the segment in the index.
struct PostingHeads {
    const u64* words;          // first word of the segment
    const u32* term_start_bits;// bit (i_base + i) set iff a term starts at word i_base + i
    const u64* term_off;
    u32 n_terms;
    u32 n_words;               // words in the segment
    u64 i_base;                // index of the segment's first word
    u64 p_base;                // index of the segment's first posting
    const float* doc_lens;
    u64* tfp;          // out: fat postings (whole index)
    u64* tf_off;       // out: tf_off[t] for non-empty terms (others fixed up on the host)
    u32* err;          // out: set to 1 when a word names a doc id >= n_docs
    u64 n_docs;
    int dl_packed;

    __device__ __forceinline__ bool is_start(u32 i) const { const u64 g = i_base + i; return (term_start_bits[g >> 5] >> (g & 31)) & 1u; }
    __device__ __forceinline__ bool flag(u32 i) const {
        if (i == 0 || is_start(i)) return true;          // a segment starts with a term
        return (words[i] >> SA_KEY_SHIFT) != (words[i - 1] >> SA_KEY_SHIFT);
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 w = words[i];
        const u64 doc = w >> SA_KEY_SHIFT;
        u32 tf = (u32)__popcll(w & SA_LSB_MASK);
        for (u32 j = i + 1; j < n_words && !flag(j); j++) tf += (u32)__popcll(words[j] & SA_LSB_MASK);
        u64 dl = 0;
        if (doc >= n_docs) { *err = 1u; }
        else if (dl_packed) dl = (u64)(u32)doc_lens[doc];
        tfp[p_base + pos] = (doc << SA_KEY_SHIFT) | (dl << SA_LSB_BITS) | (u64)tf;
        if (is_start(i)) {
            // last term t with term_off[t] <= i (empty terms share the offset and sort before it)
            const u64 g = i_base + i;
            u32 lo = 0, hi = n_terms;            // invariant: term_off[lo] <= g < term_off[hi]
            while (hi - lo > 1) {
                const u32 mid = lo + ((hi - lo) >> 1);
                if (term_off[mid] <= g) lo = mid; else hi = mid;
            }
            tf_off[lo] = p_base + pos;
        }
    }
};

// tile directory: row s = term dir_terms[s]; entry j = first posting with doc >= j * tile_docs
// doc directory row of one term: the first word of every doc the term occurs in
__global__ void __launch_bounds__(256)
sa_k_build_docdir(const u64* __restrict__ words, u32 n, u32* __restrict__ row, u32* __restrict__ top_block) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u64 doc = words[i] >> SA_KEY_SHIFT;
        if (i == 0 || (words[i - 1] >> SA_KEY_SHIFT) != doc) row[doc] = i;
        // a word in the LAST 18-position block (positions >= 18 * (2^18 - 1)): only then can a header be the
        // "h - 1" / "h + 1" neighbour of a header of ANOTHER doc (sa_spans.hip probes through this directory)
        if (((words[i] >> SA_LSB_BITS) & SA_LSB_MASK) == SA_LSB_MASK) *top_block = 1u;
    }
}

// per term: bit 0 = its first word has header 0 (doc 0, block 0), bit 1 = its last word has the largest header.
// sa_spans.hip needs "header 0 in L" per slop query; with these it is host arithmetic instead of a launch.
__global__ void __launch_bounds__(256)
sa_k_term_edges(const u64* __restrict__ words, const u64* __restrict__ term_off, u32 n_terms, unsigned char* __restrict__ out) {
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < n_terms; t += gridDim.x * blockDim.x) {
        const u64 a = term_off[t], b = term_off[t + 1];
        unsigned char e = 0;
        if (b > a) {
            if ((words[a] & SA_HEADER_MASK) == 0) e |= 1;
            if ((words[b - 1] & SA_HEADER_MASK) == SA_HEADER_MASK) e |= 2;
        }
        out[t] = e;
    }
}

// does any word sit in a document's last 18-position block?  (*flag |= 1)
__global__ void __launch_bounds__(256)
sa_k_any_top_block(const u64* __restrict__ words, u64 n, u32* __restrict__ flag) {
    bool any = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        any = any || ((words[i] >> SA_LSB_BITS) & SA_LSB_MASK) == SA_LSB_MASK;
    if (any) *flag = 1u;
}

// dense tf row of one term from its fat postings
__global__ void __launch_bounds__(256)
sa_k_build_tf8(const u64* __restrict__ tfp, u32 n, unsigned char* __restrict__ row, u32* __restrict__ bits) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u64 x = tfp[i];
        const u64 doc = x >> SA_KEY_SHIFT;
        const u32 tf = (u32)(x & SA_LSB_MASK);
        row[doc] = (unsigned char)(tf < 255u ? tf : 255u);
        atomicOr(&bits[doc >> 5], 1u << (doc & 31u));
    }
}

__global__ void sa_k_build_tile_dir(const u64* __restrict__ tfp, const u64* __restrict__ tf_off,
                                    const u32* __restrict__ dir_terms, u32 n_dir_terms, u32 n_tiles,
                                    u32 tile_docs, u32* __restrict__ tile_dir) {
    const u64 total = (u64)n_dir_terms * (n_tiles + 1);
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 s = (u32)(e / (n_tiles + 1)), j = (u32)(e % (n_tiles + 1));
        const u32 t = dir_terms[s];
        const u64 base = tf_off[t];
        const u32 cnt = (u32)(tf_off[t + 1] - base);
        const u64 key = ((u64)j * tile_docs) << SA_KEY_SHIFT;
        tile_dir[e] = sa_lower_bound(tfp + base, 0, cnt, key, SA_KEY_MASK);
    }
}

// dense tf: zero-filled by a memset, then scatter (as_dense, reference roaringish_ops.pyx:84-98)
__global__ void sa_k_scatter_tf(const u64* __restrict__ tfp, u64 lo, u64 hi, float* __restrict__ out) {
    for (u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = tfp[i];
        out[p >> SA_KEY_SHIFT] = (float)(u32)(p & SA_LSB_MASK);
    }
}

__global__ void sa_k_split_postings(const u64* __restrict__ tfp, u64 lo, u64 hi, u64 doc_base,
                                    u64* __restrict__ ids, float* __restrict__ tfs) {
    for (u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = tfp[i];
        ids[i - lo] = (p >> SA_KEY_SHIFT) + doc_base;
        tfs[i - lo] = (float)(u32)(p & SA_LSB_MASK);
    }
}

// ---------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------
void sa_index_free(sa_index* ix) {
    if (!ix) return;
    hipSetDevice(ix->device);
    if (ix->d_words) hipFree(ix->d_words);
    if (ix->d_term_off) hipFree(ix->d_term_off);
    if (ix->d_tfp) hipFree(ix->d_tfp);
    if (ix->d_tf_off) hipFree(ix->d_tf_off);
    if (ix->d_doc_lens) hipFree(ix->d_doc_lens);
    if (ix->d_tile_dir) hipFree(ix->d_tile_dir);
    if (ix->d_dir_slot) hipFree(ix->d_dir_slot);
    if (ix->d_docdir) hipFree(ix->d_docdir);
    if (ix->d_dd_slot) hipFree(ix->d_dd_slot);
    if (ix->d_tf8) hipFree(ix->d_tf8);
    if (ix->d_tfbits) hipFree(ix->d_tfbits);
    if (ix->d_tf8_slot) hipFree(ix->d_tf8_slot);
    ix->impacts.reset();
    if (ix->d_scratch) hipFree(ix->d_scratch);
    for (auto& ln : ix->dense_lane) {
        if (ln.stream) { hipStreamSynchronize(ln.stream); hipStreamDestroy(ln.stream); }
        if (ln.done) hipEventDestroy(ln.done);
        if (ln.scratch) hipFree(ln.scratch);
        if (ln.rows) hipFree(ln.rows);
        if (ln.sim) hipFree(ln.sim);
    }
    for (int i = 0; i < 3; i++) {
        if (ix->lane_scratch[i]) hipFree(ix->lane_scratch[i]);
        if (ix->lane_stream[i]) hipStreamDestroy(ix->lane_stream[i]);
    }
    if (ix->d_span_batch) hipFree(ix->d_span_batch);
    if (ix->d_span_counts) hipFree(ix->d_span_counts);
    if (ix->h_span_jobs) hipHostFree(ix->h_span_jobs);
    if (ix->h_flags) hipHostFree(ix->h_flags);
    if (ix->d_flags) hipFree(ix->d_flags);
    if (ix->ev_span_jobs) hipEventDestroy(ix->ev_span_jobs);
    if (ix->d_rows_scratch) hipFree(ix->d_rows_scratch);
    if (ix->d_sim_scratch) hipFree(ix->d_sim_scratch);
    if (ix->ev0) hipEventDestroy(ix->ev0);
    if (ix->ev1) hipEventDestroy(ix->ev1);
    if (ix->sstream) hipStreamDestroy(ix->sstream);
    if (ix->xstream) hipStreamDestroy(ix->xstream);
    if (ix->stream) hipStreamDestroy(ix->stream);
    delete ix;
}

// device, stream and the doc lengths (shared by both ways of creating an index)
int sa_index_setup(sa_index* ix, const float* doc_lens) {
    SA_HIP(hipSetDevice(ix->device));
    {
        hipDeviceProp_t prop;
        SA_HIP(hipGetDeviceProperties(&prop, ix->device));
        ix->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    }
    SA_HIP(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    // doc lengths ride in the postings when they are integers that fit 18 bits
    // (always true for indexes built by SearchArray.index: reference indexing.py:141-142)
    ix->dl_packed = true;
    ix->max_doc_len = 0;
    for (u64 d = 0; d < ix->n_docs; d++) {
        const float v = doc_lens[d];
        if (!(v >= 0.f) || v > 262143.f || v != floorf(v)) { ix->dl_packed = false; break; }
        if ((u32)v > ix->max_doc_len) ix->max_doc_len = (u32)v;
    }
    SA_HIP(hipMalloc(&ix->d_doc_lens, (ix->n_docs ? ix->n_docs : 1) * sizeof(float)));
    SA_HIP(hipMemcpyAsync(ix->d_doc_lens, doc_lens, ix->n_docs * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    return SA_OK;
}

static int sa_index_build(sa_index* ix, const u64* words, const u64* term_off, const float* doc_lens) {
    const u32 V = ix->n_terms;
    const u64 W = ix->n_words;
    SA_TRY(sa_index_setup(ix, doc_lens));
    hipStream_t st = ix->stream;
    SA_HIP(hipMalloc(&ix->d_words, (W + SA_WORDS_PAD) * sizeof(u64)));
    SA_HIP(hipMalloc(&ix->d_term_off, ((size_t)V + 1) * sizeof(u64)));
    SA_HIP(hipMemcpyAsync(ix->d_words, words, W * sizeof(u64), hipMemcpyHostToDevice, st));
    SA_HIP(hipMemcpyAsync(ix->d_term_off, term_off, ((size_t)V + 1) * sizeof(u64), hipMemcpyHostToDevice, st));
    return sa_index_derive(ix);
}

// Everything derived from the resident words: fat TF postings, tile directory, doc directory.
// Needs d_words, d_term_off, d_doc_lens on the device and h_term_off on the host.
int sa_index_derive(sa_index* ix) {
    const u32 V = ix->n_terms;
    const u64 W = ix->n_words;
    hipStream_t st = ix->stream;
    SA_HIP(hipMalloc(&ix->d_tf_off, ((size_t)V + 1) * sizeof(u64)));
    SA_HIP(hipMalloc(&ix->d_dir_slot, ((size_t)V + 1) * sizeof(u32)));

    // ---- derive postings ----
    // Segments: runs of whole terms with < 2^31 words each (a posting group never straddles a term
    // boundary), so the 32-bit stream compaction can serve shards far beyond 2^32 words.
    struct Seg { u64 w0, w1, p0; u32 cnt; };
    std::vector<Seg> segs;
    {
        u64 SEG_MAX = 1ull << 31;
        if (sa_opt_is_set(ix->opts.seg_words)) {                // test hook: force small segments
            const u64 v = (u64)ix->opts.seg_words;
            if (v > 0 && v < SEG_MAX) SEG_MAX = v;
        }
        u32 t = 0;
        while (t < V) {
            const u64 w0 = ix->h_term_off[t];
            u32 e = t;
            while (e < V && ix->h_term_off[e + 1] - w0 <= SEG_MAX) e++;
            if (e == t) {                                        // one term alone exceeds the segment size
                if (ix->h_term_off[t + 1] - w0 > (1ull << 31)) {
                    sa_set_error("term %u has more than 2^31 roaringish words", t);
                    return SA_ERR_UNSUPPORTED;
                }
                e = t + 1;
            }
            if (ix->h_term_off[e] > w0) segs.push_back({w0, ix->h_term_off[e], 0, 0});
            t = e;
        }
    }
    u64 max_seg = 0;
    for (const Seg& sg : segs) max_seg = sg.w1 - sg.w0 > max_seg ? sg.w1 - sg.w0 : max_seg;
    const u32 max_chunks = sa_compact_chunks((u32)max_seg);
    const size_t bits_words = (size_t)(W / 32 + 2);
    u32 *d_bits = nullptr, *d_chunks = nullptr, *d_total = nullptr;
    SA_HIP(hipMalloc(&d_bits, bits_words * sizeof(u32)));
    SA_HIP(hipMalloc(&d_chunks, ((size_t)max_chunks * (segs.size() ? segs.size() : 1) + 1) * sizeof(u32)));
    SA_HIP(hipMalloc(&d_total, 2 * sizeof(u32)));
    SA_HIP(hipMemsetAsync(d_total, 0, 2 * sizeof(u32), st));
    SA_HIP(hipMemsetAsync(d_bits, 0, bits_words * sizeof(u32), st));
    SA_HIP(hipMemsetAsync(ix->d_tf_off, 0xFF, ((size_t)V + 1) * sizeof(u64), st));
    if (V) hipLaunchKernelGGL(sa_k_mark_term_starts, dim3(sa_div_up(V, 256) < 1024 ? sa_div_up(V, 256) : 1024),
                              dim3(256), 0, st, ix->d_term_off, V, d_bits);

    PostingHeads ph;
    ph.term_start_bits = d_bits; ph.term_off = ix->d_term_off;
    ph.n_terms = V; ph.doc_lens = ix->d_doc_lens;
    ph.tfp = nullptr; ph.tf_off = ix->d_tf_off; ph.dl_packed = ix->dl_packed ? 1 : 0;
    ph.err = d_total + 1; ph.n_docs = ix->n_docs;
    // pass 1: count the postings of every segment (chunk counts are kept for the emit pass)
    u64 P = 0;
    for (size_t k = 0; k < segs.size(); k++) {
        Seg& sg = segs[k];
        const u32 n = (u32)(sg.w1 - sg.w0);
        u32* chunks = d_chunks + k * max_chunks;
        ph.words = ix->d_words + sg.w0; ph.n_words = n; ph.i_base = sg.w0; ph.p_base = 0;
        hipLaunchKernelGGL((sa_k_compact_count<PostingHeads>), dim3(sa_compact_grid(n)), dim3(SA_CT), 0, st, ph,
                           (const u32*)nullptr, n, chunks);
        hipLaunchKernelGGL(sa_k_scan_chunks, dim3(1), dim3(1024), 0, st, chunks, sa_compact_chunks(n), d_total);
        SA_HIP(hipMemcpyAsync(&sg.cnt, d_total, sizeof(u32), hipMemcpyDeviceToHost, st));
        SA_HIP(hipStreamSynchronize(st));
        sg.p0 = P;
        P += sg.cnt;
    }
    // worst case one posting per word; allocated exactly after counting
    SA_HIP(hipMalloc(&ix->d_tfp, ((size_t)P + 1) * sizeof(u64)));
    ph.tfp = ix->d_tfp;
    // pass 2: emit
    for (size_t k = 0; k < segs.size(); k++) {
        const Seg& sg = segs[k];
        const u32 n = (u32)(sg.w1 - sg.w0);
        ph.words = ix->d_words + sg.w0; ph.n_words = n; ph.i_base = sg.w0; ph.p_base = sg.p0;
        hipLaunchKernelGGL((sa_k_compact_emit<PostingHeads>), dim3(sa_compact_grid(n)), dim3(SA_CT), 0, st, ph,
                           (const u32*)nullptr, n, (const u32*)(d_chunks + k * max_chunks));
    }
    ix->n_postings = P;

    // tf_off: non-empty terms were written by the kernel; empty terms take the next offset
    ix->h_tf_off.resize((size_t)V + 1);
    SA_HIP(hipMemcpyAsync(ix->h_tf_off.data(), ix->d_tf_off, ((size_t)V + 1) * sizeof(u64), hipMemcpyDeviceToHost, st));
    u32 bad_doc = 0;
    SA_HIP(hipMemcpyAsync(&bad_doc, d_total + 1, sizeof(u32), hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    SA_HIP(hipGetLastError());
    if (bad_doc) {
        sa_set_error("words reference a doc id >= n_docs (%llu)", (unsigned long long)ix->n_docs);
        return SA_ERR_ARG;
    }
    ix->h_tf_off[V] = P;
    for (i64 t = (i64)V - 1; t >= 0; t--)
        if (ix->h_term_off[t] == ix->h_term_off[t + 1]) ix->h_tf_off[t] = ix->h_tf_off[t + 1];
    for (u32 t = 0; t < V; t++) {
        if (ix->h_tf_off[t] > ix->h_tf_off[t + 1]) {
            sa_set_error("internal: tf offsets not monotone at term %u (words not sorted by doc within a term?)", t);
            return SA_ERR_STATE;
        }
    }
    SA_HIP(hipMemcpyAsync(ix->d_tf_off, ix->h_tf_off.data(), ((size_t)V + 1) * sizeof(u64), hipMemcpyHostToDevice, st));
    SA_HIP(hipFree(d_bits));
    SA_HIP(hipFree(d_chunks));
    SA_HIP(hipFree(d_total));

    // ---- tile directory for frequent terms ----
    ix->n_tiles = ix->n_docs ? sa_div_up(ix->n_docs, ix->tile_docs) : 0;
    {
        // terms with at least ~1/8 posting per tile get a directory row; rarer terms are short
        // enough that a workgroup's lower-bound search stays inside a few cache lines
        const int div = (int)sa_opt(ix->opts.dir_div, 8);
        ix->dir_min_df = div > 0 ? ix->n_tiles / (u32)div : 4 * ix->n_tiles;
        if (div < 0) ix->dir_min_df = (u32)(-div) * ix->n_tiles;
        if (ix->dir_min_df < 64) ix->dir_min_df = 64;
    }
    std::vector<u32> slot((size_t)V + 1, 0xFFFFFFFFu), dir_terms;
    for (u32 t = 0; t < V; t++) {
        const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
        if (df >= ix->dir_min_df) {
            slot[t] = (u32)dir_terms.size();
            dir_terms.push_back(t);
        }
    }
    ix->n_dir_terms = (u32)dir_terms.size();
    SA_HIP(hipMemcpyAsync(ix->d_dir_slot, slot.data(), ((size_t)V + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
    const u64 entries = (u64)ix->n_dir_terms * (ix->n_tiles + 1);
    SA_HIP(hipMalloc(&ix->d_tile_dir, (entries ? entries : 1) * sizeof(u32)));
    if (entries) {
        u32* d_dir_terms = nullptr;
        SA_HIP(hipMalloc(&d_dir_terms, dir_terms.size() * sizeof(u32)));
        SA_HIP(hipMemcpyAsync(d_dir_terms, dir_terms.data(), dir_terms.size() * sizeof(u32), hipMemcpyHostToDevice, st));
        const u32 grid = entries / 256 + 1 < 65536 ? (u32)(entries / 256 + 1) : 65536;
        hipLaunchKernelGGL(sa_k_build_tile_dir, dim3(grid), dim3(256), 0, st, ix->d_tfp, ix->d_tf_off,
                           d_dir_terms, ix->n_dir_terms, ix->n_tiles, ix->tile_docs, ix->d_tile_dir);
        SA_HIP(hipStreamSynchronize(st));
        SA_HIP(hipFree(d_dir_terms));
    }
    // ---- dense tf rows of the frequent terms (dynamic pruning looks candidates up in them) ----
    {
        const int div = (int)sa_opt(ix->opts.tf8_div, 128);
        std::vector<std::pair<u64, u32>> cand;
        if (div > 0 && ix->n_docs > 0)
            for (u32 t = 0; t < V; t++) {
                const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
                if (df >= 64 && df * (u64)div >= ix->n_docs) cand.push_back({df, t});
            }
        std::sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        u64 budget_rows = ix->n_docs ? (ix->n_postings * 8 + (64ull << 20)) / ix->n_docs : 0;
        if (sa_opt_is_set(ix->opts.tf8_maxrows)) budget_rows = (u64)ix->opts.tf8_maxrows;
        if (cand.size() > budget_rows) cand.resize((size_t)budget_rows);
        if (cand.size() > 4096) cand.resize(4096);
        ix->n_tf8_terms = (u32)cand.size();
        std::vector<u32>& slot8 = ix->h_tf8_slot;
        slot8.assign((size_t)V + 1, SA_DD_NONE);
        for (u32 r = 0; r < ix->n_tf8_terms; r++) slot8[cand[r].second] = r;
        SA_HIP(hipMalloc(&ix->d_tf8_slot, ((size_t)V + 1) * sizeof(u32)));
        SA_HIP(hipMemcpyAsync(ix->d_tf8_slot, slot8.data(), ((size_t)V + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
        const size_t bytes = (size_t)ix->n_tf8_terms * ix->n_docs;
        SA_HIP(hipMalloc(&ix->d_tf8, bytes ? bytes : 4));
        if (bytes) SA_HIP(hipMemsetAsync(ix->d_tf8, 0, bytes, st));
        ix->tfbits_words = (ix->n_docs + 31) / 32;
        const size_t bbytes = (size_t)ix->n_tf8_terms * ix->tfbits_words * sizeof(u32);
        SA_HIP(hipMalloc(&ix->d_tfbits, bbytes ? bbytes : 4));
        if (bbytes) SA_HIP(hipMemsetAsync(ix->d_tfbits, 0, bbytes, st));
        for (u32 r = 0; r < ix->n_tf8_terms; r++) {
            const u32 t = cand[r].second;
            const u32 n = (u32)cand[r].first;
            const u32 grid = n / 256 + 1 < 16384 ? n / 256 + 1 : 16384;
            hipLaunchKernelGGL(sa_k_build_tf8, dim3(grid), dim3(256), 0, st, ix->d_tfp + ix->h_tf_off[t], n,
                               ix->d_tf8 + (size_t)r * ix->n_docs, ix->d_tfbits + (size_t)r * ix->tfbits_words);
        }
    }
    // ---- doc directory for the terms phrase probes hit hardest ----
    {
        // terms with at least one word per SA_DOCDIR_DIV docs (default 32), most frequent first,
        // within a memory budget of the size of the word array itself
        const int div = (int)sa_opt(ix->opts.docdir_div, 32);
        std::vector<std::pair<u64, u32>> cand;
        if (div > 0 && ix->n_docs > 0)
            for (u32 t = 0; t < V; t++) {
                const u64 w = ix->h_term_off[t + 1] - ix->h_term_off[t];
                if (w >= 64 && w * (u64)div >= ix->n_docs && w < 0xFFFFFFFFull) cand.push_back({w, t});
            }
        std::sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        const u64 budget_rows = ix->n_docs ? (ix->n_words * 8 + (64ull << 20)) / (ix->n_docs * 4) : 0;
        if (cand.size() > budget_rows) cand.resize((size_t)budget_rows);
        if (cand.size() > 4096) cand.resize(4096);
        ix->n_dd_terms = (u32)cand.size();
        std::vector<u32>& dd_slot = ix->h_dd_slot;
        dd_slot.assign((size_t)V + 1, SA_DD_NONE);
        for (u32 r = 0; r < ix->n_dd_terms; r++) dd_slot[cand[r].second] = r;
        SA_HIP(hipMalloc(&ix->d_dd_slot, ((size_t)V + 1) * sizeof(u32)));
        SA_HIP(hipMemcpyAsync(ix->d_dd_slot, dd_slot.data(), ((size_t)V + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
        const size_t dd_bytes = (size_t)ix->n_dd_terms * ix->n_docs * sizeof(u32);
        SA_HIP(hipMalloc(&ix->d_docdir, dd_bytes ? dd_bytes : 4));
        if (dd_bytes) SA_HIP(hipMemsetAsync(ix->d_docdir, 0xFF, dd_bytes, st));
        u32* d_top = nullptr;
        SA_HIP(hipMalloc(&d_top, ((size_t)ix->n_dd_terms + 1) * sizeof(u32)));
        if (hipMemsetAsync(d_top, 0, ((size_t)ix->n_dd_terms + 1) * sizeof(u32), st) != hipSuccess) {
            hipFree(d_top);
            sa_set_error("doc directory build failed (memset)");
            return SA_ERR_HIP;
        }
        for (u32 r = 0; r < ix->n_dd_terms; r++) {
            const u32 t = cand[r].second;
            const u32 n = (u32)cand[r].first;
            const u32 grid = n / 256 + 1 < 16384 ? n / 256 + 1 : 16384;
            hipLaunchKernelGGL(sa_k_build_docdir, dim3(grid), dim3(256), 0, st, ix->d_words + ix->h_term_off[t], n,
                               ix->d_docdir + (size_t)r * ix->n_docs, d_top + r);
        }
        ix->h_dd_top.assign((size_t)ix->n_dd_terms + 1, 0);
        hipError_t e1 = hipMemcpyAsync(ix->h_dd_top.data(), d_top, ((size_t)ix->n_dd_terms + 1) * sizeof(u32), hipMemcpyDeviceToHost, st);
        hipError_t e2 = hipStreamSynchronize(st);
        hipFree(d_top);
        if (e1 != hipSuccess || e2 != hipSuccess) { sa_set_error("doc directory build failed"); return SA_ERR_HIP; }
    }
    // ---- first / last header of every term (slop queries: sa_spans.hip) ----
    {
        ix->h_term_edge.assign((size_t)V + 1, 0);
        if (V > 0) {
            unsigned char* d_edge = nullptr;
            SA_HIP(hipMalloc(&d_edge, (size_t)V + 8));                  // (+ a word for sa_k_any_top_block)
            u32* d_any = (u32*)(d_edge + (((size_t)V + 3) & ~(size_t)3));
            u32 h_any = 0;
            SA_HIP(hipMemsetAsync(d_any, 0, sizeof(u32), st));
            if (ix->n_words > 0)
                hipLaunchKernelGGL(sa_k_any_top_block, dim3((u32)std::min<u64>(ix->n_words / 1024 + 1, 4096)), dim3(256), 0, st,
                                   (const u64*)ix->d_words, ix->n_words, d_any);
            hipError_t e0 = hipMemcpyAsync(&h_any, d_any, sizeof(u32), hipMemcpyDeviceToHost, st);
            hipLaunchKernelGGL(sa_k_term_edges, dim3(V / 256 + 1 < 4096 ? V / 256 + 1 : 4096), dim3(256), 0, st,
                               (const u64*)ix->d_words, (const u64*)ix->d_term_off, V, d_edge);
            hipError_t e1 = hipMemcpyAsync(ix->h_term_edge.data(), d_edge, (size_t)V, hipMemcpyDeviceToHost, st);
            hipError_t e2 = hipStreamSynchronize(st);
            hipFree(d_edge);
            if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) { sa_set_error("term edge flags failed"); return SA_ERR_HIP; }
            ix->any_top_block = h_any != 0;
        }
    }
    SA_HIP(hipStreamSynchronize(st));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_index_create(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                               const uint64_t* words, const uint64_t* term_off, const float* doc_lens,
                               float avg_doc_len, uint64_t corpus_size, uint32_t tile_docs,
                               sa_index_t** out) {
    SA_ARG(out, "out is null");
    SA_ARG(term_off, "term_off is null");
    SA_ARG(n_docs == 0 || doc_lens, "doc_lens is null");
    SA_ARG(n_docs <= (1ull << 28), "a shard holds at most 2^28 docs (28-bit roaringish key)");
    if (tile_docs == 0) tile_docs = SA_DEFAULT_TILE_DOCS;
    SA_ARG(tile_docs == 1024 || tile_docs == 2048 || tile_docs == 4096 || tile_docs == 8192 ||
               tile_docs == 16384 || tile_docs == 32768,
           "tile_docs must be 1024, 2048, 4096, 8192, 16384 or 32768");
    const u64 W = term_off[n_terms];
    SA_ARG(term_off[0] == 0, "term_off[0] must be 0");
    SA_ARG(W == 0 || words, "words is null");
    for (u32 t = 0; t < n_terms; t++) SA_ARG(term_off[t] <= term_off[t + 1], "term_off must be non-decreasing");

    sa_index* ix = new (std::nothrow) sa_index();
    if (!ix) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    ix->opts = sa_options_for_new_handle(nullptr);
    ix->device = device;
    ix->n_docs = n_docs; ix->doc_base = doc_base; ix->corpus_size = corpus_size;
    ix->n_terms = n_terms; ix->avg_doc_len = avg_doc_len; ix->n_words = W;
    ix->tile_docs = tile_docs;
    ix->h_term_off.assign(term_off, term_off + n_terms + 1);
    int rc = sa_index_build(ix, words, term_off, doc_lens);
    if (rc != SA_OK) { sa_index_free(ix); return rc; }
    *out = ix;
    return SA_OK;
}

extern "C" int sa_index_words(sa_index_t* ix, uint64_t* words_out, uint64_t* term_off_out) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_HIP(hipStreamSynchronize(ix->stream));
    if (words_out && ix->n_words) SA_HIP(hipMemcpy(words_out, ix->d_words, ix->n_words * sizeof(u64), hipMemcpyDeviceToHost));
    if (term_off_out) memcpy(term_off_out, ix->h_term_off.data(), ((size_t)ix->n_terms + 1) * sizeof(u64));
    return SA_OK;
}

extern "C" int sa_index_destroy(sa_index_t* ix) {
    if (!ix) return SA_OK;
    sa_index_free(ix);
    return SA_OK;
}

extern "C" int sa_index_docfreq(sa_index_t* ix, uint32_t term, uint64_t* out) {
    SA_ARG(ix && out, "null argument");
    *out = term < ix->n_terms ? ix->h_tf_off[term + 1] - ix->h_tf_off[term] : 0;
    return SA_OK;
}

extern "C" int sa_index_docfreqs(sa_index_t* ix, uint64_t* out) {
    SA_ARG(ix && out, "null argument");
    for (u32 t = 0; t < ix->n_terms; t++) out[t] = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
    return SA_OK;
}

// ---- subset output: the rerank / sliced-array use case (reference arr[mask].score(...), postings.py:
// 619-627, 702-703) wants a few rows of a dense vector; gathering them on the device keeps the PCIe
// copy proportional to the slice instead of to the index.
struct RowSelection { const sa_index* ix = nullptr; const uint64_t* rows = nullptr; uint64_t n = 0; };
static thread_local RowSelection tl_rows;

extern "C" int sa_index_select_rows(sa_index_t* ix, const uint64_t* rows, uint64_t n_rows) {
    SA_ARG(ix, "null index");
    SA_ARG(n_rows == 0 || rows, "rows is null");
    tl_rows.ix = rows ? ix : nullptr;
    tl_rows.rows = rows;
    tl_rows.n = n_rows;
    return SA_OK;
}

template <class T>
__global__ void __launch_bounds__(256)
sa_k_gather_rows(const T* __restrict__ vec, const u64* __restrict__ rows, u64 n, u64 n_docs, T* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        out[i] = rows[i] < n_docs ? vec[rows[i]] : (T)0;
}

void sa_emit_zeros(sa_index* ix, float* out) {
    if (sa_emit_to_vec(ix, nullptr)) { hipStreamSynchronize(ix->stream); return; }
    u64 n = ix->n_docs;
    if (tl_rows.ix == ix) { n = tl_rows.n; tl_rows = RowSelection(); }
    if (n) memset(out, 0, n * sizeof(float));
}

// ---- the other stock similarities (reference similarity.py:41-89) applied to a tf vector on the
// device.  Each follows numpy's evaluation of the reference's expression operation by operation
// (float32 arrays, Python-float constants rounded to float32 when they meet an array, np.float64 idf
// promoting the last product to float64), so results are bit-identical to the reference's.
struct SimSelection { const sa_index* ix = nullptr; int kind = 0; double idf = 0, k1 = 0, b = 0; };
static thread_local SimSelection tl_sim;

struct SimParams { float k1, kb, one_minus_b, k1_plus_1, avg; double idf; };

template <int KIND>
__global__ void __launch_bounds__(256)
sa_k_similarity(const float* __restrict__ tf, const float* __restrict__ doc_lens, SimParams p, u64 n,
                float* __restrict__ out32, double* __restrict__ out64) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const float f = tf[i], dl = doc_lens[i];
        if (KIND == SA_SIM_CLASSIC) {
            // idf * np.sqrt(term_freqs) * (1.0 / np.sqrt(doc_lens))   similarity.py:83-88
            // (sqrtf / '/' are correctly rounded under hipcc's defaults; __fsqrt_rn is the NATIVE approximation)
            const float length_norm = __fdiv_rn(1.0f, sqrtf(dl));
            out64[i] = __dmul_rn(__dmul_rn(p.idf, (double)sqrtf(f)), (double)length_norm);
        } else {
            // k1 * (1 - b + b * doc_lens / avg_doc_lens)               similarity.py:50,63-64
            const float norm = __fmul_rn(p.k1, __fadd_rn(p.one_minus_b, __fdiv_rn(__fmul_rn(p.kb, dl), p.avg)));
            const float den = __fadd_rn(f, norm);
            if (KIND == SA_SIM_BM25_IMPACT) out32[i] = __fdiv_rn(f, den);
            else out64[i] = __dmul_rn(p.idf, (double)__fdiv_rn(__fmul_rn(f, p.k1_plus_1), den));
        }
    }
}

template <class T>
static int sa_emit_typed(sa_index* ix, const T* d_vec, T* out) {
    hipStream_t st = ix->stream;
    if (tl_rows.ix != ix) {
        SA_HIP(hipMemcpyAsync(out, d_vec, ix->n_docs * sizeof(T), hipMemcpyDeviceToHost, st));
        return SA_OK;
    }
    const RowSelection sel = tl_rows;
    tl_rows = RowSelection();                        // consumed by this call
    if (sel.n == 0) return SA_OK;
    const size_t need = sel.n * (sizeof(u64) + sizeof(T)) + 256;
    if (ix->rows_scratch_bytes < need) {
        if (ix->d_rows_scratch) SA_HIP(hipFree(ix->d_rows_scratch));
        ix->d_rows_scratch = nullptr; ix->rows_scratch_bytes = 0;
        SA_HIP(hipMalloc(&ix->d_rows_scratch, need));
        ix->rows_scratch_bytes = need;
    }
    u64* d_rows = (u64*)ix->d_rows_scratch;
    T* d_sel = (T*)(d_rows + sel.n);
    SA_HIP(hipMemcpyAsync(d_rows, sel.rows, sel.n * sizeof(u64), hipMemcpyHostToDevice, st));
    const u32 grid = sel.n / 256 + 1 < 4096 ? (u32)(sel.n / 256 + 1) : 4096u;
    hipLaunchKernelGGL((sa_k_gather_rows<T>), dim3(grid), dim3(256), 0, st, d_vec, (const u64*)d_rows, sel.n, ix->n_docs, d_sel);
    SA_HIP(hipMemcpyAsync(out, d_sel, sel.n * sizeof(T), hipMemcpyDeviceToHost, st));
    return SA_OK;
}

// the tf vector of a dense call turned into the selected similarity's scores, then emitted
static int sa_emit_similarity(sa_index* ix, const float* d_tf, void* out) {
    const SimSelection sim = tl_sim;
    tl_sim = SimSelection();                         // consumed by this call
    const u64 N = ix->n_docs;
    hipStream_t st = ix->stream;
    SimParams p;
    p.k1 = (float)sim.k1; p.kb = (float)sim.b; p.one_minus_b = (float)(1.0 - sim.b);
    p.k1_plus_1 = (float)(sim.k1 + 1.0); p.avg = ix->avg_doc_len; p.idf = sim.idf;
    const u32 grid = sa_div_up(N ? N : 1, 256) < 8192 ? sa_div_up(N ? N : 1, 256) : 8192;
    if (sim.kind == SA_SIM_BM25_IMPACT) {
        float* d = const_cast<float*>(d_tf);         // in place: the tf vector is this call's scratch
        hipLaunchKernelGGL((sa_k_similarity<SA_SIM_BM25_IMPACT>), dim3(grid), dim3(256), 0, st, d_tf, ix->d_doc_lens, p, N, d, (double*)nullptr);
        return sa_emit_typed<float>(ix, d, (float*)out);
    }
    const size_t need = (N + 1) * sizeof(double);
    if (ix->sim_scratch_bytes < need) {
        if (ix->d_sim_scratch) SA_HIP(hipFree(ix->d_sim_scratch));
        ix->d_sim_scratch = nullptr; ix->sim_scratch_bytes = 0;
        SA_HIP(hipMalloc(&ix->d_sim_scratch, need));
        ix->sim_scratch_bytes = need;
    }
    double* d64 = (double*)ix->d_sim_scratch;
    if (sim.kind == SA_SIM_BM25_LEGACY)
        hipLaunchKernelGGL((sa_k_similarity<SA_SIM_BM25_LEGACY>), dim3(grid), dim3(256), 0, st, d_tf, ix->d_doc_lens, p, N, (float*)nullptr, d64);
    else
        hipLaunchKernelGGL((sa_k_similarity<SA_SIM_CLASSIC>), dim3(grid), dim3(256), 0, st, d_tf, ix->d_doc_lens, p, N, (float*)nullptr, d64);
    return sa_emit_typed<double>(ix, d64, (double*)out);
}

int sa_emit_dense(sa_index* ix, const float* d_vec, float* out) {
    if (tl_sim.ix == ix) return sa_emit_similarity(ix, d_vec, out);
    if (sa_emit_to_vec(ix, d_vec)) return SA_OK;
    return sa_emit_typed<float>(ix, d_vec, out);
}

extern "C" int sa_index_similarity_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                         int64_t min_posn, int64_t max_posn, int kind, double idf, double k1,
                                         double b, void* out) {
    SA_ARG(ix && out && terms, "null argument");
    SA_ARG(kind == SA_SIM_BM25_IMPACT || kind == SA_SIM_BM25_LEGACY || kind == SA_SIM_CLASSIC,
           "kind must be SA_SIM_BM25_IMPACT, SA_SIM_BM25_LEGACY or SA_SIM_CLASSIC");
    SA_ARG(n_terms >= 1, "no terms");
    if (sa_vec_target_pending(ix, true)) {
        sa_set_error("sa_index_select_vec cannot be combined with sa_index_similarity_dense (float32 BM25 / tf results only)");
        return SA_ERR_STATE;
    }
    tl_sim.ix = ix; tl_sim.kind = kind; tl_sim.idf = idf; tl_sim.k1 = k1; tl_sim.b = b;
    const int rc = n_terms == 1
        ? sa_index_termfreqs_dense_posn(ix, terms[0], min_posn, max_posn, (float*)out)
        : sa_index_phrase_freqs_dense_posn(ix, terms, n_terms, slop, min_posn, max_posn, (float*)out);
    tl_sim = SimSelection();                         // an argument error leaves nothing pending
    return rc;
}

extern "C" int sa_index_termfreqs_dense(sa_index_t* ix, uint32_t term, float* out) {
    SA_ARG(ix && out, "null argument");
    std::unique_lock<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SaDenseLaneScope lane(ix, g);
    SA_TRY(lane.rc);
    void* scratch;
    SA_TRY(sa_index_scratch(ix, (ix->n_docs + 1) * sizeof(float), &scratch));
    float* d_out = (float*)scratch;
    SA_HIP(hipMemsetAsync(d_out, 0, ix->n_docs * sizeof(float), ix->stream));
    if (term < ix->n_terms) {
        const u64 lo = ix->h_tf_off[term], hi = ix->h_tf_off[term + 1];
        if (hi > lo) {
            const u32 grid = sa_div_up(hi - lo, 256) < 4096 ? sa_div_up(hi - lo, 256) : 4096;
            hipLaunchKernelGGL(sa_k_scatter_tf, dim3(grid), dim3(256), 0, ix->stream, ix->d_tfp, lo, hi, d_out);
        }
    }
    SA_TRY(sa_emit_dense(ix, d_out, out));
    SA_TRY(lane.finish());
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_index_termfreqs_sparse(sa_index_t* ix, uint32_t term, uint64_t* doc_ids_out,
                                         float* tfs_out, int64_t* n_out) {
    SA_ARG(ix && n_out, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    *n_out = 0;
    if (term >= ix->n_terms) return SA_OK;
    const u64 lo = ix->h_tf_off[term], hi = ix->h_tf_off[term + 1];
    const u64 n = hi - lo;
    if (n == 0) return SA_OK;
    SA_ARG(doc_ids_out && tfs_out, "null output");
    SA_HIP(hipSetDevice(ix->device));
    void* scratch;
    SA_TRY(sa_index_scratch(ix, n * (sizeof(u64) + sizeof(float)) + 64, &scratch));
    u64* d_ids = (u64*)scratch;
    float* d_tfs = (float*)(d_ids + n);
    const u32 grid = sa_div_up(n, 256) < 4096 ? sa_div_up(n, 256) : 4096;
    hipLaunchKernelGGL(sa_k_split_postings, dim3(grid), dim3(256), 0, ix->stream, ix->d_tfp, lo, hi,
                       ix->doc_base, d_ids, d_tfs);
    SA_HIP(hipMemcpyAsync(doc_ids_out, d_ids, n * sizeof(u64), hipMemcpyDeviceToHost, ix->stream));
    SA_HIP(hipMemcpyAsync(tfs_out, d_tfs, n * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    SA_HIP(hipStreamSynchronize(ix->stream));
    SA_HIP(hipGetLastError());
    *n_out = (int64_t)n;
    return SA_OK;
}

extern "C" int sa_index_synchronize(sa_index_t* ix) {
    SA_ARG(ix, "null index");
    SA_HIP(hipSetDevice(ix->device));
    // (everything enqueued for this device: the index's streams and the streams of its batches)
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_index_info(sa_index_t* ix, sa_index_info_t* out) {
    SA_ARG(ix && out, "null argument");
    out->n_docs = ix->n_docs; out->doc_base = ix->doc_base; out->corpus_size = ix->corpus_size;
    out->n_words = ix->n_words; out->n_postings = ix->n_postings;
    out->n_terms = ix->n_terms; out->tile_docs = ix->tile_docs; out->n_tiles = ix->n_tiles;
    out->n_dir_terms = ix->n_dir_terms;
    out->device = ix->device;
    out->dl_packed = ix->dl_packed ? 1 : 0;
    out->n_docdir_terms = ix->n_dd_terms;
    out->n_tf8_terms = ix->n_tf8_terms;
    out->hbm_bytes = ix->n_words * 8 + ix->n_postings * 8 + ((u64)ix->n_terms + 1) * 20 + ix->n_docs * 4 +
                     (u64)ix->n_dir_terms * (ix->n_tiles + 1) * 4 + (u64)ix->n_dd_terms * ix->n_docs * 4 + (u64)ix->n_tf8_terms * (ix->n_docs + ix->tfbits_words * 4) + ix->scratch_bytes +
                     (ix->impacts ? ix->impacts->n * 8 : 0);
    return SA_OK;
}

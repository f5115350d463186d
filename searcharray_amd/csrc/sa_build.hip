// sa_build.hip -- index build on the device: token stream -> roaringish words.
//
// Replaces, for SearchArray.index, the host step of the reference indexer: the stable sort of the
// (term, doc, position) triples by term (indexing.py:102-115) and RoaringishEncoder.encode
// (roaringish.py:93-142): group positions by (term, doc, position // 18) and OR the bits
// 1 << (position % 18) into one 64-bit word per group.  The tokenizer is arbitrary Python and stays
// on the host; what it produces -- one term id per token, docs back to back in position order -- is
// all this needs.
//
//   1. pairs   key = term id, value = doc << 24 | position      (position < 18 * 2^18 < 2^23)
//   2. stable LSD radix sort of the pairs by key (rocPRIM, ceil(log2 V) bits): tokens arrive in
//      (doc, position) order, so the result is ordered by (term, doc, position)
//   3. one stream compaction: a token opens a word when its (term, doc, position // 18) differs from
//      its predecessor's; the head ORs the <= 18 bits of its group          -> words, word terms
//   4. term_off from the word terms (empty terms share their successor's offset)
// then the usual derivation (sa_index_derive).  The words are byte-identical to the host encoder's.
#include "sa_index.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"
#include <algorithm>
#include <new>
#include <numeric>
#include <vector>

#define SA_POS_BITS 24

// token i -> (term, doc << 24 | position); doc found by a search of the doc offsets
__global__ void __launch_bounds__(256)
sa_k_token_pairs(const u32* __restrict__ tokens, const u64* __restrict__ doc_ptr, u64 n_docs, u64 n_tok, u32 n_terms,
                 u32* __restrict__ keys, u64* __restrict__ vals, u32* __restrict__ err) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_tok; i += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n_docs;                   // last doc with doc_ptr[doc] <= i
        while (hi - lo > 1) {
            const u64 mid = lo + ((hi - lo) >> 1);
            if (doc_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        const u64 pos = i - doc_ptr[lo];
        const u32 term = tokens[i];
        if (term >= n_terms) *err = 1u;
        if (pos >= (u64)SA_LSB_BITS << SA_LSB_BITS) *err = 2u;     // position >= 18 * 2^18 (MAX_POSN)
        keys[i] = term;
        vals[i] = (lo << SA_POS_BITS) | pos;
    }
}

// a sorted token opens a roaringish word when (term, doc, position // 18) changes
struct TokenWordHeads {
    const u32* keys;
    const u64* vals;
    u32 n;
    u64* words;        // out
    u32* wterm;        // out: term of each word
    __device__ __forceinline__ u64 group(u32 i) const {
        const u64 v = vals[i];
        return ((v >> SA_POS_BITS) << SA_POS_BITS) | ((v & ((1ull << SA_POS_BITS) - 1)) / SA_LSB_BITS);
    }
    __device__ __forceinline__ bool flag(u32 i) const {
        return i == 0 || keys[i] != keys[i - 1] || group(i) != group(i - 1);
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 v = vals[i];
        const u64 doc = v >> SA_POS_BITS;
        const u64 p0 = v & ((1ull << SA_POS_BITS) - 1);
        const u64 blk = p0 / SA_LSB_BITS;
        u64 bits = 1ull << (p0 % SA_LSB_BITS);
        for (u32 j = i + 1; j < n && !flag(j); j++) bits |= 1ull << ((vals[j] & ((1ull << SA_POS_BITS) - 1)) % SA_LSB_BITS);
        words[pos] = (doc << SA_KEY_SHIFT) | (blk << SA_LSB_BITS) | bits;
        wterm[pos] = keys[i];
    }
};

// term_off[t] = first word whose term is >= t
__global__ void __launch_bounds__(256)
sa_k_term_offsets(const u32* __restrict__ wterm, u32 n_words, u32 n_terms, u64* __restrict__ term_off) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_words; i += gridDim.x * blockDim.x) {
        const u32 prev = i == 0 ? 0u : wterm[i - 1] + 1u;          // first term not yet closed
        const u32 cur = i == n_words ? n_terms : wterm[i];
        for (u32 t = prev; t <= cur && t <= n_terms; t++) term_off[t] = i;
    }
}

// stable sort of (key, value) pairs by the low `bits` bits of the key (sa_sort.hip)
int sa_sort_pairs_by_key(u32* keys_in, u32* keys_out, u64* vals_in, u64* vals_out, u32 n, int bits, hipStream_t st);

extern "C" int sa_index_create_from_tokens(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                                           const uint32_t* tokens, const uint64_t* doc_ptr, const float* doc_lens,
                                           float avg_doc_len, uint64_t corpus_size, uint32_t tile_docs,
                                           sa_index_t** out) {
    SA_ARG(out && doc_ptr, "null argument");
    SA_ARG(n_docs == 0 || doc_lens, "doc_lens is null");
    SA_ARG(n_docs <= (1ull << 28), "a shard holds at most 2^28 docs (28-bit roaringish key)");
    if (tile_docs == 0) tile_docs = SA_DEFAULT_TILE_DOCS;
    SA_ARG(tile_docs == 1024 || tile_docs == 2048 || tile_docs == 4096 || tile_docs == 8192 ||
               tile_docs == 16384 || tile_docs == 32768,
           "tile_docs must be 1024, 2048, 4096, 8192, 16384 or 32768");
    const u64 n_tok = doc_ptr[n_docs];
    SA_ARG(doc_ptr[0] == 0, "doc_ptr[0] must be 0");
    SA_ARG(n_tok == 0 || tokens, "tokens is null");
    SA_ARG(n_tok < 0xFFFFF000ull, "more than 2^32 tokens per shard is not supported yet");
    for (u64 d = 0; d < n_docs; d++) SA_ARG(doc_ptr[d] <= doc_ptr[d + 1], "doc_ptr must be non-decreasing");

    sa_index* ix = new (std::nothrow) sa_index();
    if (!ix) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    ix->opts = sa_options_for_new_handle(nullptr);
    ix->device = device;
    ix->n_docs = n_docs; ix->doc_base = doc_base; ix->corpus_size = corpus_size;
    ix->n_terms = n_terms; ix->avg_doc_len = avg_doc_len; ix->n_words = 0;
    ix->tile_docs = tile_docs;
    const u32 V = n_terms;
    u32 *d_tokens = nullptr, *d_keys = nullptr, *d_keys2 = nullptr, *d_wterm = nullptr, *d_chunks = nullptr, *d_cnt = nullptr;
    u64 *d_doc_ptr = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
    auto cleanup = [&]() {
        hipFree(d_tokens); hipFree(d_keys); hipFree(d_keys2); hipFree(d_wterm); hipFree(d_chunks); hipFree(d_cnt);
        hipFree(d_doc_ptr); hipFree(d_vals); hipFree(d_vals2);
    };
    auto fail = [&](int code) { cleanup(); sa_index_free(ix); return code; };
#define SA_HIP_F(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { sa_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return fail(SA_ERR_HIP); } } while (0)
    if (sa_index_setup(ix, doc_lens) != SA_OK) return fail(SA_ERR_HIP);
    hipStream_t st = ix->stream;
    const u32 n = (u32)n_tok;
    SA_HIP_F(hipMalloc(&d_tokens, ((size_t)n + 1) * sizeof(u32)));
    SA_HIP_F(hipMalloc(&d_doc_ptr, ((size_t)n_docs + 1) * sizeof(u64)));
    SA_HIP_F(hipMalloc(&d_keys, ((size_t)n + 1) * sizeof(u32)));
    SA_HIP_F(hipMalloc(&d_keys2, ((size_t)n + 1) * sizeof(u32)));
    SA_HIP_F(hipMalloc(&d_vals, ((size_t)n + 1) * sizeof(u64)));
    SA_HIP_F(hipMalloc(&d_vals2, ((size_t)n + 1) * sizeof(u64)));
    SA_HIP_F(hipMalloc(&d_cnt, 4 * sizeof(u32)));
    SA_HIP_F(hipMemsetAsync(d_cnt, 0, 4 * sizeof(u32), st));
    SA_HIP_F(hipMemcpyAsync(d_tokens, tokens, (size_t)n * sizeof(u32), hipMemcpyHostToDevice, st));
    SA_HIP_F(hipMemcpyAsync(d_doc_ptr, doc_ptr, ((size_t)n_docs + 1) * sizeof(u64), hipMemcpyHostToDevice, st));
    u32 W = 0;
    if (n) {
        const u32 grid = n / 256 + 1 < 65536 ? n / 256 + 1 : 65536;
        hipLaunchKernelGGL(sa_k_token_pairs, dim3(grid), dim3(256), 0, st, (const u32*)d_tokens, (const u64*)d_doc_ptr, n_docs,
                           n_tok, V, d_keys, d_vals, d_cnt + 1);
        u32 err = 0;
        SA_HIP_F(hipMemcpyAsync(&err, d_cnt + 1, sizeof(u32), hipMemcpyDeviceToHost, st));
        SA_HIP_F(hipStreamSynchronize(st));
        if (err == 1) { sa_set_error("a token names a term id >= n_terms (%u)", V); return fail(SA_ERR_ARG); }
        if (err == 2) { sa_set_error("Document length exceeds maximum of %u", (unsigned)(SA_LSB_BITS << SA_LSB_BITS)); return fail(SA_ERR_ARG); }
        int bits = 1;
        while (bits < 32 && (1ull << bits) < (u64)V) bits++;
        if (sa_sort_pairs_by_key(d_keys, d_keys2, d_vals, d_vals2, n, bits, st) != SA_OK) return fail(SA_ERR_HIP);
        hipFree(d_keys); d_keys = nullptr;
        hipFree(d_vals); d_vals = nullptr;
        hipFree(d_tokens); d_tokens = nullptr;
        // words: count, allocate exactly, emit
        const u32 nchunks = sa_compact_chunks(n);
        SA_HIP_F(hipMalloc(&d_chunks, ((size_t)nchunks + 8) * sizeof(u32)));
        TokenWordHeads h;
        h.keys = d_keys2; h.vals = d_vals2; h.n = n; h.words = nullptr; h.wterm = nullptr;
        const u32 cgrid = sa_compact_grid(n);
        hipLaunchKernelGGL((sa_k_compact_count<TokenWordHeads>), dim3(cgrid), dim3(SA_CT), 0, st, h, (const u32*)nullptr, n, d_chunks);
        hipLaunchKernelGGL(sa_k_scan_chunks, dim3(1), dim3(1024), 0, st, d_chunks, nchunks, d_cnt);
        SA_HIP_F(hipMemcpyAsync(&W, d_cnt, sizeof(u32), hipMemcpyDeviceToHost, st));
        SA_HIP_F(hipStreamSynchronize(st));
        SA_HIP_F(hipMalloc(&ix->d_words, ((size_t)W + SA_WORDS_PAD) * sizeof(u64)));
        SA_HIP_F(hipMalloc(&d_wterm, ((size_t)W + 1) * sizeof(u32)));
        h.words = ix->d_words; h.wterm = d_wterm;
        hipLaunchKernelGGL((sa_k_compact_emit<TokenWordHeads>), dim3(cgrid), dim3(SA_CT), 0, st, h, (const u32*)nullptr, n, d_chunks);
    } else {
        SA_HIP_F(hipMalloc(&ix->d_words, SA_WORDS_PAD * sizeof(u64)));
        SA_HIP_F(hipMalloc(&d_wterm, sizeof(u32)));
    }
    ix->n_words = W;
    SA_HIP_F(hipMalloc(&ix->d_term_off, ((size_t)V + 1) * sizeof(u64)));
    {
        const u32 grid = W / 256 + 1 < 16384 ? W / 256 + 1 : 16384;
        hipLaunchKernelGGL(sa_k_term_offsets, dim3(grid), dim3(256), 0, st, (const u32*)d_wterm, W, V, ix->d_term_off);
    }
    ix->h_term_off.resize((size_t)V + 1);
    SA_HIP_F(hipMemcpyAsync(ix->h_term_off.data(), ix->d_term_off, ((size_t)V + 1) * sizeof(u64), hipMemcpyDeviceToHost, st));
    SA_HIP_F(hipStreamSynchronize(st));
    SA_HIP_F(hipGetLastError());
    cleanup();
    d_tokens = d_keys = d_keys2 = d_wterm = d_chunks = d_cnt = nullptr;
    d_doc_ptr = d_vals = d_vals2 = nullptr;
#undef SA_HIP_F
    int rc = sa_index_derive(ix);
    if (rc != SA_OK) { sa_index_free(ix); return rc; }
    *out = ix;
    return SA_OK;
}

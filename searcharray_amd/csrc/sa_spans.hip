// sa_spans.hip -- slop > 0 phrase matching on the device (BASELINE config 5).
//
// The reference treats slop > 0 as a two-stage search (phrase/spans.py:71-187):
//   1. _intersect_all: keep, per term, the words whose 46-bit header lies in a candidate set built
//      from pairwise header intersections / adjacencies of (term 0, term i), widened by +-1;
//   2. _span_freqs (roaringish/spans.pyx:189-319): per document a serial state machine over at
//      most 512 "active spans", counting complete non-overlapping spans.
//
// Stage 1 only depends on header SETS (every merge / intersect in it is followed by a masked
// membership test), so it is restated as a per-word predicate evaluated with lower-bound probes:
//     in_t(h)        term t has a word with header h
//     Lset_i(h) = in_0(h)&in_i(h) | in_i(h)&in_0(h-1) | in_0(h)&in_i(h-1)      (spans.py:79-90)
//     Rset_i(h) = in_0(h)&in_i(h) | in_0(h)&in_i(h+1) | in_i(h)&in_0(h+1)
//     L = AND_i Lset_i,  R = AND_i Rset_i                                       (spans.py:92-100)
//     keep word with header h  <=>  L(h) | R(h) | R(h-1) | (L(h+1) unless header 0 is in L)
//                                                                                (spans.py:106-118)
// followed by one stable compaction per term.  The "unless": the reference forms `L - 1` with
// unsigned arithmetic (spans.py:108), so when L contains header 0 (doc 0, block 0) that element
// wraps to 2^64 - 2^18, the array handed to the sorted merges is no longer sorted, everything after
// the wrapped element -- the whole `L - 1` part -- ends up behind a value larger than any real
// header, and the final sorted slice never reaches it.  The goldens (outputs of the reference
// itself) pin this: e.g. query [8, 1, 3] on tests/golden/zipf_small.
//
// Stage 2 runs one thread per document group (documents are independent).  The k-th thread takes
// the k-th document group of EVERY term -- the reference walks the terms' cursors in lock step
// (spans.pyx:223-304) and does not re-align them by key -- and replays the state machine with the
// span table in a per-thread slab of global memory.  Two quirks of the reference are part of the
// observable behaviour and are reproduced: position bits are `1 << (p % 64)` evaluated as a 32-bit
// shift (count mod 32, sign-extended), and a rejected extension leaves its position bit set.
// A document that fills the 512-entry table is undefined behaviour in the reference (it indexes
// past the arrays); here, as in oracle/spans.c, it takes the reference's "full" rule
// (min over terms of the summed popcounts).
#include "sa_index.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"

#define SA_SPAN_MAX_TERMS 16
#define SA_NSPANS 512
#define SA_SPAN_CHUNK_DOCS 8192          // documents per state-machine launch (20 KiB of span table each)

struct SpanTerms {
    const u64* words[SA_SPAN_MAX_TERMS];
    u32 len[SA_SPAN_MAX_TERMS];
    int T;
};

__device__ __forceinline__ bool sa_has_header(const u64* __restrict__ a, u32 n, u64 h) {
    const u32 j = sa_lower_bound(a, 0, n, h, SA_HEADER_MASK);
    return j < n && (a[j] & SA_HEADER_MASK) == h;
}

__device__ bool sa_span_Lset(const SpanTerms& st, u64 h) {
    const u64 unit = 1ull << SA_LSB_BITS;
    const bool a0 = sa_has_header(st.words[0], st.len[0], h);
    const bool a0m = sa_has_header(st.words[0], st.len[0], h - unit);
    for (int i = 1; i < st.T; i++) {
        const bool bi = sa_has_header(st.words[i], st.len[i], h);
        const bool bim = sa_has_header(st.words[i], st.len[i], h - unit);
        if (!((a0 && bi) || (bi && a0m) || (a0 && bim))) return false;
    }
    return true;
}

__device__ bool sa_span_Rset(const SpanTerms& st, u64 h) {
    const u64 unit = 1ull << SA_LSB_BITS;
    const bool a0 = sa_has_header(st.words[0], st.len[0], h);
    const bool a0p = sa_has_header(st.words[0], st.len[0], h + unit);
    for (int i = 1; i < st.T; i++) {
        const bool bi = sa_has_header(st.words[i], st.len[i], h);
        const bool bip = sa_has_header(st.words[i], st.len[i], h + unit);
        if (!((a0 && bi) || (a0 && bip) || (bi && a0p))) return false;
    }
    return true;
}

__global__ void sa_k_span_wrap_flag(const SpanTerms st, u32* __restrict__ wrap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *wrap = sa_span_Lset(st, 0ull) ? 1u : 0u;
}

// stage 1: candidate words of term t, compacted into out (stable)
struct SpanCandidates {
    SpanTerms st;
    int t;
    u64* out;
    const u32* wrap;                // 1: header 0 is in L, the `L - 1` widening is lost (see top)
    __device__ __forceinline__ bool flag(u32 i) const {
        const u64 unit = 1ull << SA_LSB_BITS;
        const u64 h = st.words[t][i] & SA_HEADER_MASK;
        if (sa_span_Lset(st, h) || sa_span_Rset(st, h) || sa_span_Rset(st, h - unit)) return true;
        return !*wrap && sa_span_Lset(st, h + unit);
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = st.words[t][i]; }
};

// document-group heads of a compacted candidate array
struct DocHeads {
    const u64* words; u32* out;
    __device__ __forceinline__ bool flag(u32 i) const { return i == 0 || (words[i] >> SA_KEY_SHIFT) != (words[i - 1] >> SA_KEY_SHIFT); }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = i; }
};

struct SpanMachineParams {
    const u64* cand[SA_SPAN_MAX_TERMS];       // candidate words of each term
    const u32* n_cand[SA_SPAN_MAX_TERMS];     // device lengths
    const u32* heads[SA_SPAN_MAX_TERMS];      // start index of each document group
    const u32* n_heads[SA_SPAN_MAX_TERMS];
    int T;
    u32 slop;
    u32 doc_begin;                            // first group handled by this launch
    u64* slab;                                // [SA_SPAN_CHUNK_DOCS][5][SA_NSPANS]: terms, posns, beg, end, collected
    u32* counts;                              // dense per-doc counts (atomically accumulated)
    u64 n_docs;
};

__device__ __forceinline__ u64 sa_posn_mask(i64 p) {
    // reference spans.pyx:108-109 as compiled: 32-bit shift, count mod 32, sign-extended
    const int m = (int)(1u << ((u32)(p % 64) & 31u));
    return (u64)(i64)m;
}

__device__ __forceinline__ i64 sa_iabs(i64 v) { return v < 0 ? -v : v; }

__global__ void __launch_bounds__(64) sa_k_span_machine(const SpanMachineParams p) {
    const u32 local = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = p.doc_begin + local;
    if (local >= SA_SPAN_CHUNK_DOCS || k >= *p.n_heads[0]) return;
    u64* s_terms = p.slab + (u64)local * 5 * SA_NSPANS;
    u64* s_posns = s_terms + SA_NSPANS;
    i64* s_beg = (i64*)(s_posns + SA_NSPANS);
    i64* s_end = s_beg + SA_NSPANS;
    const u64 num_terms = (u64)p.T;
    const u64 max_span_width = num_terms + p.slop;
    u32 cursor = 0;
    bool full = false;
    u64 last_key = 0;
    u64 sum_pop[SA_SPAN_MAX_TERMS];

    for (int t = 0; t < p.T; t++) {
        sum_pop[t] = 0;
        const u32 ng = *p.n_heads[t];
        if (k >= ng) continue;                                   // this term has no k-th document group
        const u32 lo = p.heads[t][k];
        const u32 hi = (k + 1 < ng) ? p.heads[t][k + 1] : *p.n_cand[t];
        const u64 curr_term_mask = 1ull << t;
        bool gave_up = false;
        for (u32 wi = lo; wi < hi && !gave_up; wi++) {
            const u64 w = p.cand[t][wi];
            last_key = w >> SA_KEY_SHIFT;
            const u64 payload_base = ((w >> SA_LSB_BITS) & SA_LSB_MASK) * SA_LSB_BITS;
            u64 bits = w & SA_LSB_MASK;
            sum_pop[t] += (u64)__popcll(bits);
            while (bits != 0) {
                const i64 curr_posn = (i64)(payload_base + (u64)(__ffsll((long long)bits) - 1));
                bits &= bits - 1;
                const u64 posn_mask = sa_posn_mask(curr_posn);
                if (cursor >= SA_NSPANS) { full = true; break; }
                s_terms[cursor] = curr_term_mask; s_posns[cursor] = posn_mask;
                s_beg[cursor] = curr_posn; s_end[cursor] = curr_posn;
                const u32 end = cursor;
                cursor++;
                for (u32 si = 0; si < end; si++) {
                    const u64 st = s_terms[si], sp = s_posns[si];
                    const u64 nt = (u64)__popcll(st), np = (u64)__popcll(sp);
                    if (nt < num_terms && np == num_terms) continue;
                    if (st & curr_term_mask) continue;           // term already in the span: nothing changes
                    const u64 sp2 = sp | posn_mask;
                    s_posns[si] = sp2;                           // the position bit stays even if rejected
                    const u64 new_unique = (u64)__popcll(sp2);
                    const u64 proposed = (u64)sa_iabs(curr_posn - s_beg[si]);
                    if (np == new_unique || proposed > max_span_width) continue;
                    s_terms[si] = st | curr_term_mask;
                    if (cursor < SA_NSPANS) {
                        s_terms[cursor] = st | curr_term_mask; s_posns[cursor] = sp2 & ~posn_mask;
                        s_beg[cursor] = s_beg[si]; s_end[cursor] = s_end[si];
                        cursor++;
                        full = false;
                    } else {
                        full = true;
                    }
                    s_end[si] = curr_posn;
                }
                if (cursor >= SA_NSPANS) break;
            }
            // reference compaction (spans.pyx:140-154) never removes a span (widths are bounded by
            // construction), so a full table stays full: skip the rest of this term's words
            if (cursor >= SA_NSPANS) gave_up = true;
        }
    }
    u32 incr;
    if (full) {
        u64 mn = 0;
        for (int t = 0; t < p.T; t++) if (mn == 0 || sum_pop[t] < mn) mn = sum_pop[t];
        incr = (u32)mn;
    } else {
        // _collect_spans, spans.pyx:157-186: walk the spans in order; a complete span narrower than
        // max_width either replaces the first collected span it overlaps AND is shorter than, or is
        // appended.  Collected (beg, end) pairs are packed into the fifth 512-entry lane of the slab.
        u64* col = (u64*)(s_end + SA_NSPANS);
        u32 ncol = 0;
        for (u32 si = 0; si < cursor; si++) {
            const bool complete = ((u64)__popcll(s_terms[si]) == num_terms) || ((u64)__popcll(s_posns[si]) == num_terms);
            const i64 b = s_beg[si], e = s_end[si];
            const i64 width = sa_iabs(e - b);
            if (!complete || (u64)width >= max_span_width) continue;
            bool replaced = false;
            for (u32 c = 0; c < ncol; c++) {
                const i64 cb = (i64)(col[c] >> 32), ce = (i64)(col[c] & 0xFFFFFFFFull);
                if (b <= ce && e >= cb && width < sa_iabs(ce - cb)) {
                    col[c] = ((u64)b << 32) | (u64)e;
                    replaced = true;
                    break;
                }
            }
            if (!replaced) col[ncol++] = ((u64)b << 32) | (u64)e;
        }
        incr = ncol;
    }
    if (incr && last_key < p.n_docs) atomicAdd(&p.counts[last_key], incr);
}

__global__ void __launch_bounds__(256)
sa_k_counts_to_float(const u32* __restrict__ counts, float* __restrict__ out, u64 n) {
    for (u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (u64)gridDim.x * blockDim.x) out[d] = (float)counts[d];
}

// dense slop > 0 phrase counts of terms[0..T) -> *d_out (float[n_docs], inside the index scratch)
int sa_span_counts_device(sa_index* ix, const u32* terms, int T, int slop, const PosnFilter& filt, float** d_out) {
    if (T > SA_SPAN_MAX_TERMS) { sa_set_error("slop phrases support at most %d terms", SA_SPAN_MAX_TERMS); return SA_ERR_UNSUPPORTED; }
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    SpanTerms terms_dev;
    memset(&terms_dev, 0, sizeof(terms_dev));
    terms_dev.T = T;
    bool known = true;
    size_t total_len = 0, max_len = 0;
    for (int t = 0; t < T; t++) {
        if (terms[t] >= ix->n_terms) { known = false; continue; }
        const u64 off = ix->h_term_off[terms[t]];
        terms_dev.words[t] = ix->d_words + off;
        terms_dev.len[t] = (u32)(ix->h_term_off[terms[t] + 1] - off);
        total_len += terms_dev.len[t];
        if (terms_dev.len[t] > max_len) max_len = terms_dev.len[t];
    }
    const size_t slab_words = (size_t)SA_SPAN_CHUNK_DOCS * 5 * SA_NSPANS;
    const size_t chunk_words = sa_compact_chunks((u32)(max_len + 1)) + 8;
    const size_t filt_bytes = filt.active ? (total_len + 64 * (size_t)T) * 8 : 0;
    const size_t need = (N + 64) * 8 + (total_len + 64 * T) * 12 + slab_words * 8 + chunk_words * 4 + filt_bytes + 64 * 1024;
    void* scratch;
    SA_TRY(sa_index_scratch(ix, need, &scratch));
    char* base = (char*)scratch;
    size_t used = 0;
    auto take = [&](size_t bytes) { char* p = base + used; used += (bytes + 255) & ~(size_t)255; return p; };
    float* running = (float*)take((N + 1) * 4);
    u32* counts = (u32*)take((N + 1) * 4);
    u32* cnt = (u32*)take(4 * SA_SPAN_MAX_TERMS * 4);          // [t] n_cand, [16 + t] n_heads, [32] wrap flag
    u32* chunks = (u32*)take(chunk_words * 4);
    u64* slab = (u64*)take(slab_words * 8);
    *d_out = running;
    SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));
    SA_HIP(hipMemsetAsync(counts, 0, N * sizeof(u32), st));
    SA_HIP(hipMemsetAsync(cnt, 0, 4 * SA_SPAN_MAX_TERMS * 4, st));
    if (!known || N == 0 || total_len == 0) return SA_OK;
    if (filt.active) {
        const u64* ptrs[SA_SPAN_MAX_TERMS];
        u64* bufs[SA_SPAN_MAX_TERMS];
        u32 lens[SA_SPAN_MAX_TERMS];
        for (int t = 0; t < T; t++) {
            ptrs[t] = terms_dev.words[t]; lens[t] = terms_dev.len[t];
            bufs[t] = (u64*)take(((size_t)lens[t] + 1) * 8);
        }
        if (used > need) { sa_set_error("internal: span scratch exhausted"); return SA_ERR_STATE; }
        SA_TRY(sa_posn_filter_terms(ix, filt, T, ptrs, lens, bufs, cnt + 3 * SA_SPAN_MAX_TERMS, chunks));
        SA_HIP(hipMemsetAsync(cnt, 0, 3 * SA_SPAN_MAX_TERMS * 4, st));
        for (int t = 0; t < T; t++) { terms_dev.words[t] = ptrs[t]; terms_dev.len[t] = lens[t]; }
    }

    hipLaunchKernelGGL(sa_k_span_wrap_flag, dim3(1), dim3(64), 0, st, terms_dev, cnt + 2 * SA_SPAN_MAX_TERMS);
    SpanMachineParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.T = T; mp.slop = (u32)slop; mp.slab = slab; mp.counts = counts; mp.n_docs = N;
    for (int t = 0; t < T; t++) {
        u64* cand = (u64*)take(((size_t)terms_dev.len[t] + 1) * 8);
        u32* heads = (u32*)take(((size_t)terms_dev.len[t] + 1) * 4);
        if (used > need) { sa_set_error("internal: span scratch exhausted"); return SA_ERR_STATE; }
        mp.cand[t] = cand; mp.n_cand[t] = cnt + t; mp.heads[t] = heads; mp.n_heads[t] = cnt + SA_SPAN_MAX_TERMS + t;
        if (terms_dev.len[t] == 0) continue;
        SpanCandidates sc;
        sc.st = terms_dev; sc.t = t; sc.out = cand; sc.wrap = cnt + 2 * SA_SPAN_MAX_TERMS;
        sa_compact(sc, (const u32*)nullptr, terms_dev.len[t], chunks, cnt + t, st);
        DocHeads dh;
        dh.words = cand; dh.out = heads;
        sa_compact(dh, cnt + t, terms_dev.len[t], chunks, cnt + SA_SPAN_MAX_TERMS + t, st);
    }
    u32 n_groups = 0;
    SA_HIP(hipMemcpyAsync(&n_groups, cnt + SA_SPAN_MAX_TERMS, sizeof(u32), hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    for (u32 begin = 0; begin < n_groups; begin += SA_SPAN_CHUNK_DOCS) {
        mp.doc_begin = begin;
        const u32 n = n_groups - begin < SA_SPAN_CHUNK_DOCS ? n_groups - begin : SA_SPAN_CHUNK_DOCS;
        hipLaunchKernelGGL(sa_k_span_machine, dim3((n + 63) / 64), dim3(64), 0, st, mp);
    }
    const u64 g = (N + 255) / 256;
    hipLaunchKernelGGL(sa_k_counts_to_float, dim3((u32)(g < 8192 ? (g ? g : 1) : 8192)), dim3(256), 0, st, counts, running, N);
    return SA_OK;
}

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
// Stage 2 takes the k-th document group of EVERY term together -- the reference walks the terms' cursors in lock
// step (spans.pyx:223-304) and does not re-align them by key -- and replays the state machine per group: one
// thread per group with its span table in LDS, groups in work order (sa_k_span_machine_flat); the groups whose table
// outgrows the LDS column again with a whole wave each (sa_k_span_machine_wave).  One pipeline per query, on one
// stream, nothing read back: flags -> one compaction for all terms (candidate words + document heads) -> work order
// -> the two machines, accumulating into the dense float result.  Two quirks of the reference are part of the
// observable behaviour and are reproduced: position bits are `1 << (p % 64)` evaluated as a 32-bit
// shift (count mod 32, sign-extended), and a rejected extension leaves its position bit set.
// A document that fills the 512-entry table is undefined behaviour in the reference (it indexes
// past the arrays); here it behaves exactly as oracle/spans.c's guarded restatement does -- the reference's "full"
// rule (min over terms of the summed popcounts), incl. which words still count (sa_span_doc) -- and the tests
// compare such documents bit for bit like all others.
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_scan.hpp"
#include "../../include/searcharray_hip.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#define SA_SPAN_MAX_TERMS 32
#define SA_NSPANS 512
#define SA_SPAN_THREADS 65536            // resident state-machine threads (12 KiB of span table each)

struct SpanTerms {
    const u64* words[SA_SPAN_MAX_TERMS];
    const u32* dd[SA_SPAN_MAX_TERMS];    // doc directory row of the term (index of each doc's first word), or null
    u32 len[SA_SPAN_MAX_TERMS];
    u32 off[SA_SPAN_MAX_TERMS + 1];      // prefix sums of len: position of each term in the flag array
    u64 n_docs;
    int T;
};

// Membership of the three headers h-1, h, h+1 (in units of one 18-position block) in a sorted word
// list: bit 0 / 1 / 2.  One lower-bound search; the three headers are neighbours in sorted order.
// The reference's unsigned arithmetic wraps at both ends (h-1 of header 0, h+1 of the largest
// header); the wrapped values are looked up like any other header, as numpy does.
__device__ __forceinline__ u32 sa_header_triple(const u64* __restrict__ a, u32 n, u64 h) {
    const u64 unit = 1ull << SA_LSB_BITS;
    if (n == 0) return 0;
    const u64 hm = h - unit, hp = h + unit;
    u32 bits = 0;
    if (h == 0) {                                    // h-1 wrapped to the largest header
        if ((a[n - 1] & SA_HEADER_MASK) == hm) bits |= 1u;
    }
    if (hp == 0) {                                   // h+1 wrapped to header 0
        if ((a[0] & SA_HEADER_MASK) == 0) bits |= 4u;
    }
    u32 j = sa_lower_bound(a, 0, n, h == 0 ? h : hm, SA_HEADER_MASK);
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if (j >= n) break;
        const u64 x = a[j] & SA_HEADER_MASK;
        if (h != 0 && x == hm) bits |= 1u;
        else if (x == h) bits |= 2u;
        else if (hp != 0 && x == hp) bits |= 4u;
        else break;
        j++;
    }
    return bits;
}

// The same through the term's doc directory: one 4-byte load finds the doc's first word (or rejects the doc),
// and the doc's few words are scanned for the three headers -- instead of a binary search of the whole list
// (20 dependent loads for a term with a million words).  Valid for a term without a word in the last
// 18-position block (sa_k_build_docdir checks): then h - 1 and h + 1 can only match words of h's own doc,
// because a neighbour in ANOTHER doc would have to be that doc's block 2^18 - 1 (and the wrapped neighbours
// of header 0 / of the largest header do not exist either).
__device__ __forceinline__ u32 sa_header_triple_dd(const u64* __restrict__ a, u32 n, const u32* __restrict__ dd, u64 n_docs, u64 h) {
    const u64 unit = 1ull << SA_LSB_BITS;
    const u64 doc = h >> SA_KEY_SHIFT;
    if (((h >> SA_LSB_BITS) & SA_LSB_MASK) == SA_LSB_MASK) return sa_header_triple(a, n, h);   // the probe itself sits in a last block
    if (doc >= n_docs) return 0;
    u32 j = dd[doc];
    if (j == SA_DD_ABSENT) return 0;
    const u64 hm = h - unit, hp = h + unit;
    const bool has_prev = (h & ~SA_KEY_MASK) != 0;                      // block > 0: h - 1 is in the same doc
    u32 bits = 0;
    for (; j < n; j++) {
        const u64 x = a[j] & SA_HEADER_MASK;
        if ((x >> SA_KEY_SHIFT) != doc || x > hp) break;
        if (x == h) bits |= 2u;
        else if (has_prev && x == hm) bits |= 1u;
        else if (x == hp) bits |= 4u;
    }
    return bits;
}

// candidate predicate of one header (see the top of the file); m[i] = sa_header_triple of term i
template <int TT>
__device__ __forceinline__ bool sa_span_keep(const u32* m, int T, bool wrap) {
    const bool a0m = m[0] & 1u, a0 = m[0] & 2u, a0p = m[0] & 4u;
    bool L = true, R = true, Rm = true, Lp = true;   // L(h), R(h), R(h-1), L(h+1)
#pragma unroll
    for (int i = 1; i < (TT ? TT : SA_SPAN_MAX_TERMS); i++) {
        if (i >= T) break;
        const bool bim = m[i] & 1u, bi = m[i] & 2u, bip = m[i] & 4u;
        L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
        R &= (a0 && bi) || (a0 && bip) || (bi && a0p);
        Rm &= (a0m && bim) || (a0m && bi) || (bim && a0);
        Lp &= (a0p && bip) || (bip && a0) || (a0p && bi);
    }
    return L || R || Rm || (!wrap && Lp);
}

// header 0 in L?  (the `L - 1` widening is lost then, see the top of the file).  Also clears the query's device
// counters (cnt layout: sa_span_counts_device), so no separate fill is enqueued for them.
#define SA_SPAN_SORT_MIN 131072         // words of term 0 (an upper bound of the document groups) from which the groups are put in work order
#define SA_SPAN_INLINE_SCAN 4096       // up to this many chunks (8 M words) the emit pass scans the chunk counts itself
#define SA_SPAN_NBINS 32                 // work bins of the fast pass (positions of a document group, saturated)
#define SA_SPAN_CNT_BINS (5 * SA_SPAN_MAX_TERMS)                     // [.. + NBINS) bin sizes, [.. + 2 NBINS) bin cursors
#define SA_SPAN_CNT_WORDS (5 * SA_SPAN_MAX_TERMS + 2 * SA_SPAN_NBINS)
#define SA_SPAN_CNT_WRAP (2 * SA_SPAN_MAX_TERMS)
__global__ void __launch_bounds__(256) sa_k_span_wrap_flag(const SpanTerms st, u32* __restrict__ cnt) {
    if (blockIdx.x != 0) return;
    if (cnt && threadIdx.x < SA_SPAN_CNT_WORDS && threadIdx.x != SA_SPAN_CNT_WRAP) cnt[threadIdx.x] = 0u;
    if (threadIdx.x != 0) return;
    u32* wrap = cnt + SA_SPAN_CNT_WRAP;
    bool L = true;
    const u32 m0 = sa_header_triple(st.words[0], st.len[0], 0ull);
    for (int i = 1; i < st.T; i++) {
        const u32 mi = sa_header_triple(st.words[i], st.len[i], 0ull);
        const bool a0m = m0 & 1u, a0 = m0 & 2u, bim = mi & 1u, bi = mi & 2u;
        L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
    }
    *wrap = L ? 1u : 0u;
}

// stage 1a: keep-flag of every word of every term (one launch; the predicate costs one search
// per phrase term and is evaluated once, the compaction below only reads the flags).  The same launch clears the
// dense per-doc counts the state machines accumulate into.  TT: the number of terms when it is 2, 3 or 4 (the loops
// over the terms unroll and the per-term results stay in registers -- the generic code spends most of its
// instructions on indexing them), 0: any number.
// (bx / gx: the block's index and the number of blocks of THIS phrase's launch -- blockIdx.x / gridDim.x for a launch of
//  its own, or the phrase's share of a batched launch whose blockIdx.y picks the phrase: sa_k_span_*_multi below)
template <int TT>
__device__ __forceinline__ void
sa_span_flags_body(const SpanTerms& st, const u32* __restrict__ wrap, const int wrap_host, u32* __restrict__ cnt_clear,
                   unsigned char* __restrict__ flags, float* __restrict__ counts, const u32 bx, const u32 gx) {
    const int T = TT ? TT : st.T;
    const u32 total = st.off[T];
    // "header 0 in L": from sa_k_span_wrap_flag (filtered lists), or worked out on the host from the index's per-term
    // edge flags -- then this launch also clears the query's counters and no launch precedes it
    const bool wr = wrap ? *wrap != 0 : wrap_host != 0;
    if (cnt_clear && bx == 0 && threadIdx.x < SA_SPAN_CNT_WORDS) cnt_clear[threadIdx.x] = 0u;
    // (counts == null: the batched route's count vectors are kept zero by the launch that ranks them)
    if (counts)
        for (u64 d = (u64)bx * blockDim.x + threadIdx.x; d < st.n_docs; d += (u64)gx * blockDim.x) counts[d] = 0.f;
    for (u32 g = bx * blockDim.x + threadIdx.x; g < total; g += gx * blockDim.x) {
        int t = 0;
#pragma unroll
        for (int i = 1; i < (TT ? TT : SA_SPAN_MAX_TERMS); i++) t += (i < T && g >= st.off[i]) ? 1 : 0;
        const u32 gi = g - st.off[t], gn = st.len[t];
        const u64* const own = st.words[t];
        const u64 h = own[gi] & SA_HEADER_MASK;
        // the word's OWN term needs no probe: headers are unique and sorted within a term, so h - 1 / h + 1 are in the
        // list iff they are the neighbouring words' headers (the wrapped neighbours of header 0 / of the largest
        // header: the list's last / first word) -- two loads from the lines the wave is reading anyway
        const u64 own_prev = own[gi > 0u ? gi - 1u : gn - 1u] & SA_HEADER_MASK;
        const u64 own_next = own[gi + 1u < gn ? gi + 1u : 0u] & SA_HEADER_MASK;
        u32 m[TT ? TT : SA_SPAN_MAX_TERMS];
        bool possible = true;
#pragma unroll
        for (int i = 0; i < (TT ? TT : SA_SPAN_MAX_TERMS); i++) m[i] = 0;
        // The probes are chains of dependent loads (directory entry -> the doc's first words), and the kernel is bound
        // by their latency: those of the first four terms are issued together -- all directory entries, then the
        // first two words of the doc in every term -- instead of term after term.
        constexpr int TU = 4;
        const u64 unit = 1ull << SA_LSB_BITS;
        const u64 doc = h >> SA_KEY_SHIFT, hm = h - unit, hp = h + unit;
        const bool has_prev = (h & ~SA_KEY_MASK) != 0;
        const bool plain = ((h >> SA_LSB_BITS) & SA_LSB_MASK) != SA_LSB_MASK && doc < st.n_docs;   // (else: sa_header_triple_dd's own handling)
        u32 jj[TU];
        u64 x0[TU], x1[TU];
        bool via[TU];
#pragma unroll
        for (int i = 0; i < TU; i++) {
            via[i] = i < T && i != t && st.dd[i] != nullptr && plain;
            jj[i] = via[i] ? st.dd[i][doc] : SA_DD_ABSENT;
        }
#pragma unroll
        for (int i = 0; i < TU; i++) {
            const bool have = via[i] && jj[i] != SA_DD_ABSENT;
            x0[i] = (have && jj[i] < st.len[i]) ? st.words[i][jj[i]] : ~0ull;
            x1[i] = (have && jj[i] + 1u < st.len[i]) ? st.words[i][jj[i] + 1u] : ~0ull;
        }
#pragma unroll
        for (int i = 0; i < (TT && TT < TU ? TT : TU); i++) {
            if (i < T && possible) {
                if (via[i]) {
                    u32 bits = 0;
                    auto step = [&](const u64 xw) -> bool {
                        const u64 x = xw & SA_HEADER_MASK;
                        if ((x >> SA_KEY_SHIFT) != doc || x > hp) return false;
                        if (x == h) bits |= 2u;
                        else if (has_prev && x == hm) bits |= 1u;
                        else if (x == hp) bits |= 4u;
                        return true;
                    };
                    if (step(x0[i]) && step(x1[i]))
                        for (u32 j = jj[i] + 2u; j < st.len[i]; j++)
                            if (!step(st.words[i][j])) break;
                    m[i] = bits;
                } else if (i == t) {
                    m[i] = 2u | (own_prev == hm ? 1u : 0u) | (own_next == hp ? 4u : 0u);
                } else {
                    m[i] = st.dd[i] ? sa_header_triple_dd(st.words[i], st.len[i], st.dd[i], st.n_docs, h)
                                    : sa_header_triple(st.words[i], st.len[i], h);
                }
                // every set needs term 0 and term i around h: nothing there -> no candidate
                if (m[i] == 0) possible = false;
            }
        }
        for (int i = TU; i < T && possible; i++) {
            if (i == t) m[i] = 2u | (own_prev == hm ? 1u : 0u) | (own_next == hp ? 4u : 0u);
            else m[i] = st.dd[i] ? sa_header_triple_dd(st.words[i], st.len[i], st.dd[i], st.n_docs, h)
                                 : sa_header_triple(st.words[i], st.len[i], h);
            if (m[i] == 0) possible = false;
        }
        flags[g] = (possible && sa_span_keep<TT>(m, T, wr)) ? 1 : 0;
    }
}

template <int TT>
__global__ void __launch_bounds__(256)
sa_k_span_flags(const SpanTerms st, const u32* __restrict__ wrap, const int wrap_host, u32* __restrict__ cnt_clear,
                unsigned char* __restrict__ flags, float* __restrict__ counts) {
    sa_span_flags_body<TT>(st, wrap, wrap_host, cnt_clear, flags, counts, blockIdx.x, gridDim.x);
}

// stage 1b: ONE stable compaction for all terms and both outputs -- the candidate words of each term, and the index
// (in the compacted array) of the first candidate word of each document: count per chunk, a scan per (term, output),
// emit.  A candidate word opens a document group iff no earlier word of the same doc (they are neighbours in the
// term's list, a handful at most) is a candidate.
struct SpanChunkTab { u32 coff[SA_SPAN_MAX_TERMS + 1]; };      // chunks of term t: [coff[t], coff[t + 1])
struct SpanCompactOut { u64* cand[SA_SPAN_MAX_TERMS]; u32* heads[SA_SPAN_MAX_TERMS]; unsigned char* gpos[SA_SPAN_MAX_TERMS]; };   // gpos: positions per document group (saturated)

// (the two preceding words and their flags are requested together with the word itself -- a group rarely has more
//  than three words, and one after the other the loads would be a chain of dependent round trips)
__device__ __forceinline__ bool sa_span_opens_doc(const u64* __restrict__ w, const unsigned char* __restrict__ f, u32 i) {
    const u32 i1 = i >= 1u ? i - 1u : i, i2 = i >= 2u ? i - 2u : i;
    const u64 w0 = w[i], w1 = w[i1], w2 = w[i2];
    const bool f1 = f[i1] != 0, f2 = f[i2] != 0;
    const u64 doc = w0 >> SA_KEY_SHIFT;
    if (i < 1u || (w1 >> SA_KEY_SHIFT) != doc) return true;
    if (f1) return false;
    if (i < 2u || (w2 >> SA_KEY_SHIFT) != doc) return true;
    if (f2) return false;
    for (u32 j = i2; j > 0;) {
        j--;
        if ((w[j] >> SA_KEY_SHIFT) != doc) break;
        if (f[j]) return false;
    }
    return true;
}

__device__ __forceinline__ void
sa_span_compact_count_body(const SpanTerms& st, const SpanChunkTab& ck, const unsigned char* __restrict__ flags,
                           u32* __restrict__ chunk_counts, u32 n_chunks, const u32 bx, const u32 gx) {
    __shared__ u32 red[SA_CW + 1];
    for (u32 c = bx; c < n_chunks; c += gx) {
        int t = 0;
        while (c >= ck.coff[t + 1]) t++;
        const u32 base = (c - ck.coff[t]) * SA_CHUNK, n = st.len[t];
        const u64* w = st.words[t];
        const unsigned char* f = flags + st.off[t];
        u32 nc = 0, nh = 0;
#pragma unroll
        for (int j = 0; j < SA_CI; j++) {
            const u32 i = base + j * SA_CT + threadIdx.x;
            if (i < n && f[i]) {
                nc++;
                if (sa_span_opens_doc(w, f, i)) nh++;
            }
        }
        const u32 tot = sa_block_sum<SA_CW>(nc | (nh << 16), red);          // (a chunk has 2048 elements)
        if (threadIdx.x == 0) { chunk_counts[c] = tot & 0xFFFFu; chunk_counts[n_chunks + c] = tot >> 16; }
    }
}

__global__ void __launch_bounds__(SA_CT)
sa_k_span_compact_count(const SpanTerms st, const SpanChunkTab ck, const unsigned char* __restrict__ flags,
                        u32* __restrict__ chunk_counts, u32 n_chunks) {
    sa_span_compact_count_body(st, ck, flags, chunk_counts, n_chunks, blockIdx.x, gridDim.x);
}

// block b: exclusive scan of the chunk counts of term b % T, output b / T; totals -> cnt[t] / cnt[16 + t]
__global__ void __launch_bounds__(1024)
sa_k_span_compact_scan(const SpanChunkTab ck, int T, u32* __restrict__ chunk_counts, u32 n_chunks, u32* __restrict__ cnt) {
    __shared__ u32 red[16];
    const int t = (int)blockIdx.x % T, kind = (int)blockIdx.x / T;
    u32* counts = chunk_counts + (u32)kind * n_chunks + ck.coff[t];
    const u32 nck = ck.coff[t + 1] - ck.coff[t];
    u32 carry = 0;
    for (u32 base = 0; base < nck; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nck ? counts[i] : 0u;
        u32 tot;
        const u32 ex = sa_block_excl_scan<16>(v, red, &tot);
        if (i < nck) counts[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) cnt[kind * SA_SPAN_MAX_TERMS + t] = carry;
}

__device__ __forceinline__ void
sa_span_compact_emit_body(const SpanTerms& st, const SpanChunkTab& ck, const unsigned char* __restrict__ flags,
                          const u32* __restrict__ chunk_off, u32 n_chunks, const SpanCompactOut& out, u32* __restrict__ totals,
                          const u32 bx, const u32 gx) {
    __shared__ u32 wc[SA_CI][SA_CW];
    __shared__ u32 red[2][SA_CW];
    const int lane = sa_lane(), wave = sa_wave_id();
    const u64 lt = (1ull << lane) - 1ull;
    for (u32 c = bx; c < n_chunks; c += gx) {
        int t = 0;
        while (c >= ck.coff[t + 1]) t++;
        const u32 base = (c - ck.coff[t]) * SA_CHUNK, n = st.len[t];
        const u64* w = st.words[t];
        const unsigned char* f = flags + st.off[t];
        u32 rank[SA_CI];                                           // candidate rank | head rank << 16, within the wave's round
        u32 mine = 0;                                              // bit j: candidate; bit 8 + j: opens a document
#pragma unroll
        for (int j = 0; j < SA_CI; j++) {
            const u32 i = base + j * SA_CT + threadIdx.x;
            const bool fl = i < n && f[i];
            const bool hd = fl && sa_span_opens_doc(w, f, i);
            const u64 bc = __ballot(fl), bh = __ballot(hd);
            rank[j] = (u32)__popcll(bc & lt) | ((u32)__popcll(bh & lt) << 16);
            if (lane == 0) wc[j][wave] = (u32)__popcll(bc) | ((u32)__popcll(bh) << 16);
            mine |= ((fl ? 1u : 0u) << j) | ((hd ? 1u : 0u) << (8 + j));
        }
        __syncthreads();
        u32 off_c, off_h;
        if (totals) {
            // no scan launch: chunk_off still holds the COUNTS, and the block adds up those of the term's chunks in
            // front of its own (a few hundred values at most -- the host takes this route for <= SA_SPAN_INLINE_SCAN
            // chunks); the term's last chunk publishes the totals
            u32 sc = 0, sh = 0;
            for (u32 x = ck.coff[t] + threadIdx.x; x < c; x += SA_CT) { sc += chunk_off[x]; sh += chunk_off[n_chunks + x]; }
            sc = sa_wave_sum(sc); sh = sa_wave_sum(sh);
            if (lane == 0) { red[0][wave] = sc; red[1][wave] = sh; }
            __syncthreads();
            off_c = 0; off_h = 0;
#pragma unroll
            for (int wv = 0; wv < SA_CW; wv++) { off_c += red[0][wv]; off_h += red[1][wv]; }
            if (c + 1u == ck.coff[t + 1] && threadIdx.x == 0) {
                totals[t] = off_c + chunk_off[c];
                totals[SA_SPAN_MAX_TERMS + t] = off_h + chunk_off[n_chunks + c];
            }
        } else {
            off_c = chunk_off[c]; off_h = chunk_off[n_chunks + c];
        }
        u64* cand = out.cand[t];
        u32* heads = out.heads[t];
#pragma unroll
        for (int j = 0; j < SA_CI; j++) {
#pragma unroll
            for (int wv = 0; wv < SA_CW; wv++) {
                if (wv == wave && ((mine >> j) & 1u)) {
                    const u32 pos = off_c + (rank[j] & 0xFFFFu);
                    cand[pos] = w[base + j * SA_CT + threadIdx.x];
                    if ((mine >> (8 + j)) & 1u) {
                        // opens a document group: its index in the compacted array, and how many positions the group holds
                        const u32 i = base + j * SA_CT + threadIdx.x;
                        const u64 doc = w[i] >> SA_KEY_SHIFT;
                        // (a group rarely has more than three words: the next two are requested together)
                        const u32 i1 = i + 1u < n ? i + 1u : i, i2 = i + 2u < n ? i + 2u : i;
                        const u64 wa = w[i], wb = w[i1], wc = w[i2];
                        const bool fb = f[i1] != 0, fc = f[i2] != 0;
                        u32 np = (u32)__popc((u32)(wa & SA_LSB_MASK));
                        const bool sb = i1 != i && (wb >> SA_KEY_SHIFT) == doc, sc = sb && i2 != i1 && (wc >> SA_KEY_SHIFT) == doc;
                        if (sb && fb) np += (u32)__popc((u32)(wb & SA_LSB_MASK));
                        if (sc && fc) np += (u32)__popc((u32)(wc & SA_LSB_MASK));
                        if (sc)
                            for (u32 x = i + 3u; x < n && (w[x] >> SA_KEY_SHIFT) == doc; x++)
                                if (f[x]) np += (u32)__popc((u32)(w[x] & SA_LSB_MASK));
                        heads[off_h + (rank[j] >> 16)] = pos;
                        out.gpos[t][off_h + (rank[j] >> 16)] = (unsigned char)(np < 255u ? np : 255u);
                    }
                }
                off_c += wc[j][wv] & 0xFFFFu;
                off_h += wc[j][wv] >> 16;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(SA_CT)
sa_k_span_compact_emit(const SpanTerms st, const SpanChunkTab ck, const unsigned char* __restrict__ flags,
                       const u32* __restrict__ chunk_off, u32 n_chunks, const SpanCompactOut out, u32* __restrict__ totals) {
    sa_span_compact_emit_body(st, ck, flags, chunk_off, n_chunks, out, totals, blockIdx.x, gridDim.x);
}

// document-group heads of a compacted candidate array
struct DocHeads {
    const u64* words; u32* out;
    __device__ __forceinline__ bool flag(u32 i) const { return i == 0 || (words[i] >> SA_KEY_SHIFT) != (words[i - 1] >> SA_KEY_SHIFT); }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = i; }
};

// One active span: term set, position-bit set, first and last position.  The reference keeps four
// uint64 / int64 arrays; 16 bytes hold the same information (<= 16 terms; the position mask is a
// sign-extended int32, see sa_posn_mask32; positions are < 2^23), so a span is one 16-byte access.
struct alignas(16) SpanEnt { u32 terms; int posns; int beg; int end; };

struct SpanMachineParams {
    const u64* cand[SA_SPAN_MAX_TERMS];       // candidate words of each term
    const u32* n_cand[SA_SPAN_MAX_TERMS];     // device lengths
    const u32* heads[SA_SPAN_MAX_TERMS];      // start index of each document group
    const u32* n_heads[SA_SPAN_MAX_TERMS];
    int T;
    u32 slop;
    u32 n_threads;                            // G: resident threads, thread g owns column g of the slabs
    SpanEnt* ents;                            // [SA_NSPANS][G] span tables, interleaved for coalescing
    u64* col;                                 // [SA_NSPANS][G] collected (beg, end) pairs
    u32* counts;                              // dense per-doc counts (atomically accumulated) -- the kernel-level mirror; or
    float* fcounts;                           //   the same as floats (the dense result itself: small integers, exact in any order)
    u64 n_docs;
    u32* over_list;                           // LDS passes: document groups whose table outgrew the column; full-table
    u32* over_cnt;                            //   pass: the groups to take (null: all of them)
    const u32* in_list;                       // heavy pass: the groups the fast pass abandoned
    const u32* in_cnt;
    const u32* order;                         // fast pass: the document groups sorted by work (sa_k_span_bin_*), or null
    unsigned char* touched;                   // batched route: [key >> touch_shift] = 1 wherever a count is added (the ranking
    u32 touch_shift;                          //   launch reads the touched tiles only), or null
};

__device__ __forceinline__ void sa_span_add(const SpanMachineParams& p, u64 key, u32 incr) {
    if (incr == 0 || key >= p.n_docs) return;
    if (p.fcounts) unsafeAtomicAdd(&p.fcounts[key], (float)incr);
    else atomicAdd(&p.counts[key], incr);
    if (p.touched) p.touched[key >> p.touch_shift] = 1;
}

// reference spans.pyx:108-109 as compiled: `1 << (p % 64)` is a 32-bit shift (count mod 32) whose
// int result is sign-extended to 64 bits.  Kept as the int32; OR / AND-NOT commute with the sign
// extension, and the 64-bit popcount is recovered by sa_popc_sext.
__device__ __forceinline__ int sa_posn_mask32(int p) { return (int)(1u << ((u32)(p % 64) & 31u)); }
__device__ __forceinline__ u32 sa_popc_sext(int v) { return (u32)__popc((u32)v) + (v < 0 ? 32u : 0u); }
__device__ __forceinline__ int sa_iabs32(int v) { return v < 0 ? -v : v; }

// Stage 2: one thread per document group (documents are independent).  Thread k takes the k-th document group of
// EVERY term -- the reference walks the terms' cursors in lock step (spans.pyx:223-304) and does not re-align them
// by key -- and replays the state machine.
//
// Two passes over two table placements:
//   fast   one thread per document group, the whole span table in LDS -- a column per lane, 16-byte entries, lane
//          L's entry i at (i * 64 + L) * 16: whatever rows the lanes of a wave are at, they hit different banks.
//          SA_SPAN_LDS spans per lane (the collected spans reuse the slots of the spans already visited) keep a
//          wave at 12 KiB of LDS, i.e. 13 waves per CU to hide the latency of the per-lane word loads.  A typical document needs a handful of spans; a
//          document whose table would outgrow the column is ABANDONED here (nothing counted) and put on a list;
//   heavy  the listed documents again, one WAVE each (sa_k_span_machine_wave below).
// (sa_k_span_machine -- one thread per document with the full 512-span table in the thread's column of a global
//  slab -- remains as the kernel-level mirror's machine and as the SA_SPAN_FAST=0 route the tests compare with.)
// Returns false when the table capacity CAP_E / CAP_C is exceeded before the reference's own limit (fast pass only).
#define SA_SPAN_LDS 12
#define SA_SPAN_PMAX 16

// _collect_spans, spans.pyx:157-186: walk the spans in order; a complete span narrower than max_width either
// replaces the first collected span it overlaps AND is shorter than, or is appended.  Returns false when more than
// CAP_C spans would be collected (fast pass only).
template <int CAP_C, class Ents, class Col>
__device__ __forceinline__ bool sa_span_collect(const Ents& ents, const Col& col, const u32 cursor, const u32 num_terms,
                                                const int max_span_width, u32* n_out) {
    u32 ncol = 0;
    for (u32 si = 0; si < cursor; si++) {
        const SpanEnt e = ents[si];
        const bool complete = ((u32)__popc(e.terms) == num_terms) || (sa_popc_sext(e.posns) == num_terms);
        const int b = e.beg, en = e.end;
        const int width = sa_iabs32(en - b);
        if (!complete || width >= max_span_width) continue;
        bool replaced = false;
        for (u32 c = 0; c < ncol; c++) {
            const u64 cc = col[c];
            const int cb = (int)(cc >> 32), ce = (int)(cc & 0xFFFFFFFFull);
            if (b <= ce && en >= cb && width < sa_iabs32(ce - cb)) {
                col[c] = ((u64)(u32)b << 32) | (u64)(u32)en;
                replaced = true;
                break;
            }
        }
        if (!replaced) {
            if (CAP_C < SA_NSPANS && ncol >= (u32)CAP_C) return false;
            col[ncol] = ((u64)(u32)b << 32) | (u64)(u32)en;
            ncol++;
        }
    }
    *n_out = ncol;
    return true;
}

template <int CAP_E, int CAP_C, class Ents, class Col>
__device__ __forceinline__ bool sa_span_doc(const SpanMachineParams& p, const u32 k, const Ents& ents, const Col& col,
                                            u32* incr_out, u64* key_out) {
    const u32 num_terms = (u32)p.T;
    const int max_span_width = (int)(num_terms + p.slop);
    constexpr bool CAN_FILL = CAP_E >= SA_NSPANS;                // only the full-table pass can reach the reference's limit
    u32 cursor = 0;
    bool full = false;
    u64 last_key = 0;
    u32 sum_pop[CAN_FILL ? SA_SPAN_MAX_TERMS : 1];               // (indexed by a run-time term: private memory)
    for (int t = 0; t < p.T; t++) {
        if (CAN_FILL) sum_pop[t] = 0;
        const u32 ng = *p.n_heads[t];
        if (k >= ng) continue;                                   // this term has no k-th document group
        const u32 lo = p.heads[t][k];
        const u32 hi = (k + 1 < ng) ? p.heads[t][k + 1] : *p.n_cand[t];
        const u32 curr_term_mask = 1u << t;
        // Spans appended while term t's positions are taken -- its fresh spans and its forks -- all contain term t, and
        // the reference's visit of such a span changes nothing (spans.pyx: "term already in the span"); the spans
        // from before the term never contain it.  So a position only VISITS the spans that existed when its term began.
        const u32 tstart = cursor;
        bool gave_up = false;
        for (u32 wi = lo; wi < hi && !gave_up; wi++) {
            const u64 w = p.cand[t][wi];
            last_key = w >> SA_KEY_SHIFT;
            const int payload_base = (int)(((w >> SA_LSB_BITS) & SA_LSB_MASK) * SA_LSB_BITS);
            u32 bits = (u32)(w & SA_LSB_MASK);
            if (CAN_FILL) sum_pop[t] += (u32)__popc(bits);
            while (bits != 0) {
                const int curr_posn = payload_base + (__ffs((int)bits) - 1);
                bits &= bits - 1;
                const int posn_mask = sa_posn_mask32(curr_posn);
                if (cursor >= SA_NSPANS) { full = true; break; }
                if (CAP_E < SA_NSPANS && cursor >= (u32)CAP_E) return false;
                SpanEnt fresh;
                fresh.terms = curr_term_mask; fresh.posns = posn_mask; fresh.beg = curr_posn; fresh.end = curr_posn;
                ents[cursor] = fresh;
                const u32 end = tstart;
                cursor++;
                for (u32 si = 0; si < end; si++) {
                    SpanEnt e = ents[si];
                    const u32 nt = (u32)__popc(e.terms), np = sa_popc_sext(e.posns);
                    if (nt < num_terms && np == num_terms) continue;
                    if (e.terms & curr_term_mask) continue;          // term already in the span: nothing changes
                    const int sp2 = e.posns | posn_mask;
                    const u32 new_unique = sa_popc_sext(sp2);
                    const int proposed = sa_iabs32(curr_posn - e.beg);
                    if (np == new_unique || proposed > max_span_width) {
                        if (sp2 != e.posns) { e.posns = sp2; ents[si] = e; }   // the position bit stays even if rejected
                        continue;
                    }
                    if (cursor < SA_NSPANS) {
                        if (CAP_E < SA_NSPANS && cursor >= (u32)CAP_E) return false;
                        SpanEnt fork;
                        fork.terms = e.terms | curr_term_mask; fork.posns = sp2 & ~posn_mask;
                        fork.beg = e.beg; fork.end = e.end;
                        ents[cursor] = fork;
                        cursor++;
                        full = false;
                    } else {
                        full = true;
                    }
                    e.terms |= curr_term_mask; e.posns = sp2; e.end = curr_posn;
                    ents[si] = e;
                }
                if (cursor >= SA_NSPANS) break;
            }
            // reference compaction (spans.pyx:140-154) never removes a span (widths are bounded by
            // construction), so a full table stays full.  The reference then looks for the term's next DOCUMENT and
            // continues there (spans.pyx:283-291): the rest of this document's words of the term is skipped -- unless
            // this is the term's last document group: its search finds no other key, nothing is skipped and every
            // remaining word still adds to the term's summed popcounts (the "full" rule's input).  Kept, like
            // oracle/spans.c: tests/test_phrase.py::test_slop_span_table_overflow_matches_the_oracle.
            if (cursor >= SA_NSPANS && k + 1 < ng) gave_up = true;
        }
    }
    u32 incr;
    if (CAN_FILL && full) {
        u32 mn = 0;
        for (int t = 0; t < p.T; t++) if (mn == 0 || sum_pop[t] < mn) mn = sum_pop[t];
        incr = mn;
    } else {
        if (!sa_span_collect<CAP_C>(ents, col, cursor, num_terms, max_span_width, &incr)) return false;
    }
    *incr_out = incr;
    *key_out = last_key;
    return true;
}

// Work bins for the fast pass.  With one document per lane a wave runs as long as its busiest document, and the
// machine's work grows with the square of a document's positions: taken in index order, a wave's busiest document
// has ~4 x the average work.  So the document groups are sorted by their number of positions (a counting sort over
// SA_SPAN_NBINS bins: sizes, then a scatter through per-bin cursors) and a wave takes 64 neighbours of that order.
// Busiest bins FIRST: the long waves start while the grid is full and the short ones fill the gaps (measured: lightest
// first costs +17 us on the heaviest 2-term query, and so does sorting inside blocks of 8192 groups in one launch --
// the same wave homogeneity, but every block's long waves start late).
struct SpanBinParams {
    const unsigned char* gpos[SA_SPAN_MAX_TERMS];
    const u32* n_heads[SA_SPAN_MAX_TERMS];
    int T;
    unsigned char* bin;                       // [n_heads[0]] bin of each document group
    u32* sizes;                               // [NBINS] (zeroed by sa_k_span_wrap_flag)
    u32* cursors;                             // [NBINS]
    u32* order;                               // [n_heads[0]] out: document groups, busiest bin first
};

__global__ void __launch_bounds__(1024) sa_k_span_bin_count(const SpanBinParams bp) {
    __shared__ u32 h[SA_SPAN_NBINS];
    if (threadIdx.x < SA_SPAN_NBINS) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 n = *bp.n_heads[0];
    const u32 per = (n + gridDim.x - 1) / gridDim.x;
    const u32 lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (u32 k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        u32 np = 0;
        for (int t = 0; t < bp.T; t++) if (k < *bp.n_heads[t]) np += bp.gpos[t][k];
        const u32 b = np < (u32)SA_SPAN_NBINS - 1u ? np : (u32)SA_SPAN_NBINS - 1u;
        bp.bin[k] = (unsigned char)b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SA_SPAN_NBINS && h[threadIdx.x]) atomicAdd(&bp.sizes[threadIdx.x], h[threadIdx.x]);
}

__global__ void __launch_bounds__(1024) sa_k_span_bin_scatter(const SpanBinParams bp) {
    __shared__ u32 h[SA_SPAN_NBINS], at[SA_SPAN_NBINS];
    if (threadIdx.x < SA_SPAN_NBINS) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 n = *bp.n_heads[0];
    const u32 per = (n + gridDim.x - 1) / gridDim.x;
    const u32 lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (u32 k = lo + threadIdx.x; k < hi; k += blockDim.x) atomicAdd(&h[bp.bin[k]], 1u);
    __syncthreads();
    if (threadIdx.x < SA_SPAN_NBINS) {
        u32 start = 0;                                           // bins in descending order
        for (u32 b = threadIdx.x + 1u; b < (u32)SA_SPAN_NBINS; b++) start += bp.sizes[b];
        at[threadIdx.x] = h[threadIdx.x] ? start + atomicAdd(&bp.cursors[threadIdx.x], h[threadIdx.x]) : 0u;
    }
    __syncthreads();
    for (u32 k = lo + threadIdx.x; k < hi; k += blockDim.x) bp.order[atomicAdd(&at[bp.bin[k]], 1u)] = k;
}

// Fast pass: thread k = document group k, span table of CE entries per lane in LDS.
//
// The reference's machine is four nested loops (terms, words, position bits, spans); with one document per lane a
// wave would pay, at every level, for the lane with the most iterations at THAT level -- the product of the maxima,
// ~10 x what its busiest document needs.  So the document is first unrolled into its position list (term, position;
// in the machine's order: term by term, word by word, bit by bit; PMAX entries per lane in LDS), and the machine then
// runs as ONE flat loop per lane whose iteration is "take the next position if the current one has visited all its
// spans, then visit one span": a wave runs as long as its busiest document, and the loop holds no global load.
// Documents with more than PMAX positions or more than CE spans are ABANDONED (nothing counted) to p.over_list for
// the heavy pass -- one counter update per wave, the abandoned lanes take consecutive slots.
// TT: the number of terms when it is 2 or 3 (the loops over the terms resolve at compile time), 0: any number.
// lane L's span table in LDS: entry i at (i * 64 + L); the collected spans reuse the table's own storage -- collecting
// walks the spans in order and has at most as many collected spans as spans already visited, so collected span c lives
// in the (dead) slot of span c
// (S: lanes that share the table -- 64 documents per wave, or 32 / 16 with tables of twice / four times the rows in the
//  same LDS: the doc-parallel route's documents with many positions)
template <int S = 64>
struct SpanEntColS {
    SpanEnt* base;
    struct Ref {
        SpanEnt* q;
        __device__ __forceinline__ operator SpanEnt() const { return *q; }
        __device__ __forceinline__ void operator=(const SpanEnt& e) const { *q = e; }
    };
    __device__ __forceinline__ Ref operator[](u32 i) const { return Ref{base + i * (u32)S}; }
};
template <int S = 64>
struct SpanColColS {
    u64* base;
    __device__ __forceinline__ u64& operator[](u32 i) const { return base[i * (u32)(2 * S)]; }
};
typedef SpanEntColS<64> SpanEntCol;

// The machine (sa_span_doc's loops, flattened) over the lane's position list s_pos[q * 64 + lane] = term << 24 | position,
// q < npos, in the machine's order (term by term).  The iteration is branch-free -- the kernel is bound by instruction
// issue, and on divergent branches most of what is issued is exec-mask bookkeeping: every lane computes both outcomes
// and writes through selected addresses, row CE of the table being a scratch row for the writes that do not happen.
// A position visits the spans that existed when its TERM began (see sa_span_doc).  Returns false when the table
// outgrew CE entries (nothing counted); else *incr_out = the document's count.
template <int CE, int PMAX, int S = 64>
__device__ __forceinline__ bool sa_span_flat_loop(const SpanEntColS<S>& ents, const u32* s_pos, const u32 lane, const u32 npos,
                                                  const u32 num_terms, const int max_span_width, u32* incr_out) {
    // The spans a term's positions still have to visit are a set that only shrinks -- a skipped span (blocked, or
    // already holding the term) is skipped by every later position of the term, skipped spans do not change -- kept as a
    // bit mask: all spans from before the term when it begins, walked by find-first-set, a span leaves it when a visit
    // finds or makes it dead.  The visits that remain happen in the same order, so forks are appended exactly as before.
    static_assert(CE < 32, "the active set is a 32-bit mask");
    u32 cursor = 0, pi = 0, curr_term_mask = 0, active = 0, todo = 0;
    int curr_posn = 0, posn_mask = 0;
    bool abandoned = false;
    bool alive = true;
    while (alive) {
        // the current position has visited every span it has to: take the next one
        const bool need = todo == 0;
        const bool done = need && pi >= npos;
        const bool fresh_it = need && !done;
        const u32 pv = s_pos[(pi < (u32)PMAX ? pi : (u32)PMAX - 1u) * (u32)S + lane];
        const u32 new_mask = 1u << (pv >> 24);
        active = (fresh_it && new_mask != curr_term_mask) ? (1u << cursor) - 1u : active;      // (cursor <= CE < 32)
        curr_posn = fresh_it ? (int)(pv & 0xFFFFFFu) : curr_posn;
        curr_term_mask = fresh_it ? new_mask : curr_term_mask;
        posn_mask = sa_posn_mask32(curr_posn);
        const bool over_f = fresh_it && cursor >= (u32)CE;
        {
            SpanEnt fresh;
            fresh.terms = curr_term_mask; fresh.posns = posn_mask; fresh.beg = curr_posn; fresh.end = curr_posn;
            ents[(fresh_it && !over_f) ? cursor : (u32)CE] = fresh;
        }
        todo = fresh_it ? active : todo;
        cursor += fresh_it ? 1u : 0u;
        pi += fresh_it ? 1u : 0u;
        // visit the next span of the set
        const bool vis = !done && !over_f && todo != 0;
        const u32 slot = vis ? (u32)__builtin_ctz(todo) : (u32)CE;
        todo = vis ? todo & (todo - 1u) : todo;
        const SpanEnt e = ents[slot];
        const u32 nt = (u32)__popc(e.terms), np = sa_popc_sext(e.posns);
        const bool act = vis && !((nt < num_terms && np == num_terms) || (e.terms & curr_term_mask));
        const int sp2 = e.posns | posn_mask;
        const u32 new_unique = sa_popc_sext(sp2);
        const int proposed = sa_iabs32(curr_posn - e.beg);
        const bool rej = np == new_unique || proposed > max_span_width;    // (the position bit stays even if rejected)
        const bool fork_it = act && !rej;
        const bool over_k = fork_it && cursor >= (u32)CE;
        SpanEnt upd, fork;
        upd.terms = fork_it ? (e.terms | curr_term_mask) : e.terms;
        upd.posns = act ? sp2 : e.posns;
        upd.beg = e.beg;
        upd.end = fork_it ? curr_posn : e.end;
        ents[slot] = upd;
        fork.terms = e.terms | curr_term_mask; fork.posns = sp2 & ~posn_mask; fork.beg = e.beg; fork.end = e.end;
        ents[(fork_it && !over_k) ? cursor : (u32)CE] = fork;
        cursor += (fork_it && !over_k) ? 1u : 0u;
        // dead for this term from here on: skipped now, extended now, or blocked by the bit this visit left
        const bool dead = !act || fork_it || (nt < num_terms && new_unique == num_terms);
        active = (vis && dead) ? active & ~(1u << slot) : active;
        abandoned = over_f || over_k;
        alive = !done && !abandoned;
    }
    if (abandoned) return false;
    u32 incr = 0;
    sa_span_collect<SA_NSPANS>(ents, SpanColColS<S>{(u64*)ents.base}, cursor, num_terms, max_span_width, &incr);
    *incr_out = incr;
    return true;
}

template <int CE, int PMAX, int TT>
__device__ __forceinline__ void sa_span_machine_flat_body(const SpanMachineParams& p, const u32 bx) {
    __shared__ alignas(16) SpanEnt s_ents[(CE + 1) * 64];        // lane L's entry i at (i * 64 + L): conflict-free whatever i each lane is at; row CE: scratch
    __shared__ u32 s_pos[PMAX * 64];                             // term << 24 | position
    const u32 lane = threadIdx.x;
    const u32 n_items = *p.n_heads[0];
    const u32 item = bx * 64u + lane;
    const u32 k = (item < n_items && p.order) ? p.order[item] : item;
    const int T = TT ? TT : p.T;
    const u32 num_terms = (u32)T;
    const int max_span_width = (int)(num_terms + p.slop);
    const SpanEntCol ents{s_ents + lane};
    bool abandoned = false;
    if (item < n_items) {
        // ---- the document's positions, in the machine's order.  The loads are what this kernel waits for (group
        //      bounds -> words, per term), so those of the first four terms are issued together: all the bounds, then
        //      the first two words of every term; further words and further terms (rare) take the plain loop.
        u32 npos = 0;
        u64 last_key = 0;
        auto push_word = [&](const int t, const u64 w) {
            last_key = w >> SA_KEY_SHIFT;
            const u32 payload_base = (u32)((w >> SA_LSB_BITS) & SA_LSB_MASK) * (u32)SA_LSB_BITS;
            u32 bits = (u32)(w & SA_LSB_MASK);
            const u32 nb = (u32)__popc(bits);
            if (npos + nb <= (u32)PMAX) {
                u32 q = npos;
                while (bits != 0) {
                    s_pos[q * 64u + lane] = ((u32)t << 24) | (payload_base + (u32)(__ffs((int)bits) - 1));
                    bits &= bits - 1;
                    q++;
                }
            }
            npos += nb;
        };
        constexpr int TU = 4;
        u32 lo[TU], hi[TU];
        u64 w0[TU], w1[TU];
#pragma unroll
        for (int t = 0; t < TU; t++) {
            lo[t] = 0; hi[t] = 0;
            if (t < T) {
                const u32 ng = *p.n_heads[t];
                if (k < ng) {                                    // (else: this term has no k-th document group)
                    lo[t] = p.heads[t][k];
                    hi[t] = (k + 1 < ng) ? p.heads[t][k + 1] : *p.n_cand[t];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TU; t++) {
            w0[t] = 0; w1[t] = 0;
            if (hi[t] > lo[t]) w0[t] = p.cand[t][lo[t]];
            if (hi[t] > lo[t] + 1u) w1[t] = p.cand[t][lo[t] + 1u];
        }
#pragma unroll
        for (int t = 0; t < TU; t++) {
            if (hi[t] > lo[t]) push_word(t, w0[t]);
            if (hi[t] > lo[t] + 1u) push_word(t, w1[t]);
            for (u32 wi = lo[t] + 2u; wi < hi[t]; wi++) push_word(t, p.cand[t][wi]);
        }
        for (int t = TU; t < T; t++) {
            const u32 ng = *p.n_heads[t];
            if (k >= ng) continue;
            const u32 l = p.heads[t][k];
            const u32 h = (k + 1 < ng) ? p.heads[t][k + 1] : *p.n_cand[t];
            for (u32 wi = l; wi < h; wi++) push_word(t, p.cand[t][wi]);
        }
        abandoned = npos > (u32)PMAX;
        if (!abandoned) {
            u32 incr = 0;
            if (sa_span_flat_loop<CE, PMAX>(ents, s_pos, lane, npos, num_terms, max_span_width, &incr)) sa_span_add(p, last_key, incr);
            else abandoned = true;
        }
    }
    const u64 ab = __ballot(abandoned);
    if (ab) {
        u32 slot = 0;
        if (lane == (u32)__builtin_ctzll(ab)) slot = atomicAdd(p.over_cnt, (u32)__popcll(ab));
        slot = (u32)__shfl((int)slot, __builtin_ctzll(ab), SA_WAVE);
        if (abandoned) p.over_list[slot + (u32)__popcll(ab & ((1ull << lane) - 1ull))] = k;
    }
}

template <int CE, int PMAX, int TT>
__global__ void __launch_bounds__(64) sa_k_span_machine_flat(const SpanMachineParams p) {
    sa_span_machine_flat_body<CE, PMAX, TT>(p, blockIdx.x);
}

// Heavy documents: ONE WAVE per document group.  The span table (the reference's full 512 entries) is in LDS; the
// positions are still taken one after the other, but what the reference does for one position -- visit every
// span that existed before it, extend / fork -- is done for 64 spans at a time, one per lane: the visits are
// independent of each other (a visit reads and writes its own span; forks are appended behind the spans that
// existed, in span order -- a ballot prefix gives each fork its slot -- and are not visited for the same
// position).  The cost of a document drops from positions x spans dependent steps of one lane to
// positions x ceil(spans / 64) steps of a wave, and heavy documents no longer form the kernel's tail.
// one word of term `curr_term_mask`: its positions, one after the other, against the spans that existed when the term
// began (tstart; see sa_span_doc), 64 spans at a time.  Stops when the table is full.
__device__ __forceinline__ void sa_span_wave_word(SpanEnt* s_ents, const u64 w, const u32 curr_term_mask, const u32 tstart,
                                                  const u32 num_terms, const int max_span_width, const u32 lane,
                                                  u32& cursor, bool& full) {
    const u64 lt = (1ull << lane) - 1ull;
    const int payload_base = (int)(((w >> SA_LSB_BITS) & SA_LSB_MASK) * SA_LSB_BITS);
    u32 bits = (u32)(w & SA_LSB_MASK);
    while (bits != 0) {
        const int curr_posn = payload_base + (__ffs((int)bits) - 1);
        bits &= bits - 1;
        const int posn_mask = sa_posn_mask32(curr_posn);
        if (cursor >= SA_NSPANS) { full = true; break; }
        if (lane == 0) {
            SpanEnt fresh;
            fresh.terms = curr_term_mask; fresh.posns = posn_mask; fresh.beg = curr_posn; fresh.end = curr_posn;
            s_ents[cursor] = fresh;
        }
        const u32 end = tstart;
        cursor++;
        for (u32 base = 0; base < end; base += 64u) {
            const u32 si = base + lane;
            bool fork_it = false;
            SpanEnt e;
            e.terms = 0; e.posns = 0; e.beg = 0; e.end = 0;
            if (si < end) {
                e = s_ents[si];
                const u32 nt = (u32)__popc(e.terms), np = sa_popc_sext(e.posns);
                const bool skip = (nt < num_terms && np == num_terms) || (e.terms & curr_term_mask);
                if (!skip) {
                    const int sp2 = e.posns | posn_mask;
                    const u32 new_unique = sa_popc_sext(sp2);
                    const int proposed = sa_iabs32(curr_posn - e.beg);
                    if (np == new_unique || proposed > max_span_width) {
                        if (sp2 != e.posns) { e.posns = sp2; s_ents[si] = e; }   // the position bit stays even if rejected
                    } else {
                        fork_it = true;
                    }
                }
            }
            // forks of this chunk, in span order, behind what is already there
            const u64 fb = __ballot(fork_it);
            if (fb) {
                const u32 rank = (u32)__popcll(fb & lt), total = (u32)__popcll(fb);
                const u32 room = SA_NSPANS - cursor;                   // (cursor <= 512 here)
                if (fork_it) {
                    const int sp2 = e.posns | posn_mask;
                    if (rank < room) {
                        SpanEnt fork;
                        fork.terms = e.terms | curr_term_mask; fork.posns = sp2 & ~posn_mask;
                        fork.beg = e.beg; fork.end = e.end;
                        s_ents[cursor + rank] = fork;
                    }
                    e.terms |= curr_term_mask; e.posns = sp2; e.end = curr_posn;
                    s_ents[si] = e;
                }
                // the reference takes the forks one by one: successes while there is room, then failures
                full = total > room;
                cursor += total < room ? total : room;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (cursor >= SA_NSPANS) break;
    }
}

// the document's count once its words are through: the reference's "full" rule (min over the terms of the summed
// popcounts; lane t holds term t's sum) or _collect_spans (spans.pyx:157-186), spans in order; the search for the
// first collected span a new one overlaps and undercuts runs over 64 collected spans at a time
__device__ __forceinline__ u32 sa_span_wave_finish(SpanEnt* s_ents, const u32 cursor, const bool full, const u32 my_sum,
                                                   const int T, const int max_span_width, const u32 lane) {
    const u32 num_terms = (u32)T;
    u64* const s_col2 = (u64*)s_ents;                            // collected span c in the (dead) slot of span c: s_col2[2 c]
    if (full) {
        u32 mn = 0;
        for (int t = 0; t < T; t++) {
            const u32 sp = (u32)__builtin_amdgcn_readlane((int)my_sum, t);
            if (mn == 0 || sp < mn) mn = sp;
        }
        return mn;
    }
    u32 ncol = 0;
    for (u32 si = 0; si < cursor; si++) {
        const SpanEnt e = s_ents[si];
        __builtin_amdgcn_wave_barrier();                 // every lane has the span before its slot is reused below
        const bool complete = ((u32)__popc(e.terms) == num_terms) || (sa_popc_sext(e.posns) == num_terms);
        const int b = e.beg, en = e.end;
        const int width = sa_iabs32(en - b);
        if (!complete || width >= max_span_width) continue;
        bool replaced = false;
        for (u32 cb0 = 0; cb0 < ncol && !replaced; cb0 += 64u) {
            const u32 c = cb0 + lane;
            bool hit = false;
            if (c < ncol) {
                const u64 cc = s_col2[2u * c];
                const int cb = (int)(cc >> 32), ce = (int)(cc & 0xFFFFFFFFull);
                hit = b <= ce && en >= cb && width < sa_iabs32(ce - cb);
            }
            const u64 hb = __ballot(hit);
            if (hb) {
                if (lane == (u32)__builtin_ctzll(hb)) s_col2[2u * c] = ((u64)(u32)b << 32) | (u64)(u32)en;
                replaced = true;
            }
        }
        if (!replaced) {
            if (lane == 0) s_col2[2u * ncol] = ((u64)(u32)b << 32) | (u64)(u32)en;
            ncol++;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return ncol;
}

__device__ __forceinline__ void sa_span_machine_wave_body(const SpanMachineParams& p, const u32 bx, const u32 gx) {
    __shared__ alignas(16) SpanEnt s_ents[SA_NSPANS];
    const u32 lane = threadIdx.x;
    const u32 n_items = *p.in_cnt;
    const u32 num_terms = (u32)p.T;
    const int max_span_width = (int)(num_terms + p.slop);
    for (u32 item = bx; item < n_items; item += gx) {
        const u32 k = p.in_list[item];
        u32 cursor = 0;
        bool full = false;
        u64 last_key = 0;
        u32 my_sum = 0;                                          // lane t: summed popcounts of term t
        __builtin_amdgcn_wave_barrier();
        for (int t = 0; t < p.T; t++) {
            const u32 ng = *p.n_heads[t];
            if (k >= ng) continue;
            const u32 lo = p.heads[t][k];
            const u32 hi = (k + 1 < ng) ? p.heads[t][k + 1] : *p.n_cand[t];
            const u32 tstart = cursor;
            bool gave_up = false;
            for (u32 wi = lo; wi < hi && !gave_up; wi++) {
                const u64 w = p.cand[t][wi];
                last_key = w >> SA_KEY_SHIFT;
                if (lane == (u32)t) my_sum += (u32)__popc((u32)(w & SA_LSB_MASK));
                sa_span_wave_word(s_ents, w, 1u << t, tstart, num_terms, max_span_width, lane, cursor, full);
                if (cursor >= SA_NSPANS && k + 1 < ng) gave_up = true;       // (not in the term's last document group: see sa_span_doc)
            }
        }
        const u32 incr = sa_span_wave_finish(s_ents, cursor, full, my_sum, p.T, max_span_width, lane);
        if (lane == 0) sa_span_add(p, last_key, incr);
    }
}

__global__ void __launch_bounds__(64) sa_k_span_machine_wave(const SpanMachineParams p) {
    sa_span_machine_wave_body(p, blockIdx.x, gridDim.x);
}

// ---- the doc-parallel route: whole lists, two to four terms -----------------------------------------------------------
// When no word of the phrase's lists sits in a document's last 18-position block (no such word in the whole index:
// sa_index::any_top_block, from the build; or every term with a doc directory row, which records it per term:
// sa_header_triple_dd), and header 0 is not in L (no wrapped `L - 1`, see the top of the
// file; with two terms that widening is redundant and the condition is not needed), the candidate predicate of a word
// only looks at words of the word's OWN document, in every term.  Then
// (a) whether a word is a candidate can be worked out by the thread that holds the document's words, no flag array;
// (b) a document has candidate words in ALL terms or in none: L(h) needs every term at h or h - 1 and term 0 at h or
//     h - 1, and each of those words is itself kept (by L(h) or by L((h - 1) + 1)); R(h), R(h - 1), L(h + 1) alike.
//     So the k-th document group of every term is the SAME document, and "the k-th group of every term" (what the
//     reference's lock-step walk takes, and what the general route reproduces through stable compactions) is "the
//     document" -- no ranks, no compaction, any order.
// ONE launch, nothing but the lists read and the dense result written (earlier forms of this route -- count / emit /
// machine over record buffers in HBM, then sort blocks + work lists -- moved 2-3 x the lists and were bound by what
// bound the general route: passes that wait for memory, then a pass that waits for instruction issue, one after the
// other).  A block of 256 threads takes 512 neighbouring documents -- of the RAREST term's list (512 neighbouring words
// of it: the ones that open a document stand for it; the dense result is cleared beforehand) or, when that list is
// about as long as the collection, of the collection itself (SpanDocParams::anchor):
//   gather  one document per thread and round (the documents' words are read in doc order; a term's words of the
//           document are found through its directory row or by a search on the key: sa_span_first): candidate words, their positions -> the document's bin (0 none, npos up to 32, else heavy) and,
//           for up to 8 positions, the position list -- term << 10 | position - 18 x the document's first block, 16
//           bits, in the machine's order -- in LDS;
//   order   the block's documents by bin, most positions first (counting sort in LDS);
//   machine each wave takes the next chunk of that order when it is done with its last (an LDS counter: the first
//           chunks are the long ones), one document per lane, the lane machine (sa_span_flat_loop8a) on the lists
//           where they lie: what the general route's global counting sort + scattered gathers produce (waves of equal
//           work, busiest first) without leaving the CU.  Chunks of 8 documents while they have more than 16 positions,
//           of 16 above 8, else 64 (two terms; 32 for three and four): a lane of an 8- / 16- / 32-lane chunk gets 80 /
//           40 / 20 table rows instead of 12 in the wave's 6 KiB, and its positions (unpacked, or read again from the
//           lists) a place behind them -- so that a document with many positions, whose table would outgrow a 64-lane
//           column, still runs as a lane;
//   heavy   documents beyond that, and lanes whose table outgrew its column even so: a wave each, once the block's lane
//           work is done and its tables are free (sa_span_wave_doc: the words through the directory, candidate test
//           per lane, the 512-span table of 16-byte entries in 8 KiB of them).
// Blocks in different phases share a CU (four fit: 40 KB of LDS each), so the gather's memory latency hides behind
// other blocks' machines.  Span entries are 8 bytes here (position bits 32, first position 23, last - first 5, terms
// 4): T + slop <= 15.
#define SA_SPAN_DW 4                     // words of one term per document the gather holds in registers
#define SA_SPAN_DB 36                    // bins: [npos] for 1 <= npos <= 32, [33] heavy (the block is 8 bytes under 21 LDS granules of 1280 bytes with it: six blocks per CU)
#define SA_SPAN_FT 256                   // threads of a block
#define SA_SPAN_FD 512                   // documents of a block
#define SA_SPAN_PC 8                     // positions of a document the gather keeps in LDS (= positions of a 64-lane chunk's lane)
#define SA_SPAN_PMAXF 32                 // positions of a document a lane takes at all
#ifndef SA_SPAN_NTAB
#define SA_SPAN_NTAB 4                   // waves of a block that own span tables and run the lane machines (the block's other waves only gather and rank)
#endif
#ifndef SA_SPAN_NTAB_BATCH
#define SA_SPAN_NTAB_BATCH 2             // the same for a batch launch that fills the device: 12 KiB of tables less per block, six resident blocks per CU instead of four
#endif
#ifndef SA_SPAN_FROWS
#define SA_SPAN_FROWS 12                 // table rows of a lane of a 64-lane chunk, incl. the scratch row: 12 x 64 x 8 = 6 KiB per wave
#endif

struct SpanDocParams {
    SpanTerms st;                        // (dd[t] == null: a term without a directory row, its documents are found by search)
    u32 slop;
    int anchor;                          // the rarest term: a block takes 512 words of ITS list and looks at the documents they open
                                         //   (the dense result is cleared beforehand); -1: 512 documents (and clears their results)
    float* counts;                       // the dense result
    // batched route (sa_k_span_doc_fused_multi): the result vector comes from a pool that is all zeros between runs, so nothing is
    // cleared; [doc >> touch_shift] of `touched` = 1 wherever a count is written (the ranking launch reads those tiles only)
    unsigned char* touched;
    u32 touch_shift;
    u32 block0, n_blocks;                // the phrase's blocks (block0: where they would start in a phrase-after-phrase launch; the shared launch goes by SpanSlot)
    // ... or no result vector at all (rank.cand != null): the block turns its documents' counts into BM25 scores and ranks them
    // itself -- the batch's pruned top-k selection over the block's 512 documents -- so only candidates above the phrase's
    // bound leave the kernel
    SpanRankCtx rank;
    float idf;
    u32 row;                             // the phrase's row in the batch
};

__device__ __forceinline__ void sa_span_doc_put(const SpanDocParams& p, u64 doc, u32 incr) {
    p.counts[doc] = (float)incr;
    if (p.touched) p.touched[doc >> p.touch_shift] = 1;
}

// index of the document's first word in term t's list -- through the term's doc directory row, or, for a term without
// one, by a search on the 28-bit key -- or SA_DD_ABSENT
__device__ __forceinline__ u32 sa_span_first(const SpanTerms& st, const int t, const u64 doc) {
    if (st.dd[t]) return sa_glob(st.dd[t])[doc];
    const u32 j = sa_lower_bound(st.words[t], 0, st.len[t], doc << SA_KEY_SHIFT, SA_KEY_MASK);
    return (j < st.len[t] && (sa_glob(st.words[t])[j] >> SA_KEY_SHIFT) == doc) ? j : SA_DD_ABSENT;
}

// sa_header_triple_dd with the document's first word found as above (no word of these lists sits in a last block)
__device__ __forceinline__ u32 sa_header_triple_doc(const SpanTerms& st, const int t, const u64 h) {
    const u64 unit = 1ull << SA_LSB_BITS;
    const u64 doc = h >> SA_KEY_SHIFT;
    if (doc >= st.n_docs) return 0;
    u32 j = sa_span_first(st, t, doc);
    if (j == SA_DD_ABSENT) return 0;
    const u64* const a = st.words[t];
    const u32 n = st.len[t];
    const u64 hm = h - unit, hp = h + unit;
    const bool has_prev = (h & ~SA_KEY_MASK) != 0;                      // block > 0: h - 1 is in the same doc
    u32 bits = 0;
    for (; j < n; j++) {
        const u64 x = a[j] & SA_HEADER_MASK;
        if ((x >> SA_KEY_SHIFT) != doc || x > hp) break;
        if (x == h) bits |= 2u;
        else if (has_prev && x == hm) bits |= 1u;
        else if (x == hp) bits |= 4u;
    }
    return bits;
}

// candidate predicate of the word with header h, probing every term within the word's document
template <int TT>
__device__ __forceinline__ bool sa_span_keep_word(const SpanTerms& st, const u64 h) {
    u32 m[TT];
#pragma unroll
    for (int i = 0; i < TT; i++) m[i] = 0;
    bool possible = true;
#pragma unroll
    for (int i = 0; i < TT; i++) {
        if (possible) m[i] = sa_header_triple_doc(st, i, h);
        if (m[i] == 0) possible = false;
    }
    return possible && sa_span_keep<TT>(m, TT, false);
}

// positions of the document's candidate words, any number of words (the count pass, documents with more than
// SA_SPAN_DW words of a term)
template <int TT>
__device__ __forceinline__ u32 sa_span_doc_npos_slow(const SpanTerms& st, const u64 doc) {
    u32 npos = 0;
#pragma unroll
    for (int t = 0; t < TT; t++) {
        const u32 n = st.len[t];
        for (u32 j = sa_span_first(st, t, doc); j < n; j++) {
            const u64 w = st.words[t][j];
            if ((w >> SA_KEY_SHIFT) != doc) break;
            if (sa_span_keep_word<TT>(st, w & SA_HEADER_MASK)) npos += (u32)__popc((u32)(w & SA_LSB_MASK));
        }
    }
    return npos;
}

// The document's words of every term (at most SA_SPAN_DW each within 30 blocks, else *many) and which of them are
// candidates (bit q of keep[t]), from the document's own words in registers.
// the first SA_SPAN_DW words of every term's words of the document (j0: where each term's begin) and the one behind them
template <int TT>
__device__ __forceinline__ void sa_span_doc_words_load(const SpanTerms& st, const u32 (&j0)[TT], u64 (&W)[TT][SA_SPAN_DW], u64 (&X)[TT]) {
#pragma unroll
    for (int t = 0; t < TT; t++) {
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++) {
            const u32 idx = j0[t] + (u32)q;
            W[t][q] = idx < st.len[t] ? sa_glob(st.words[t])[idx] : ~0ull;
        }
        const u32 idx = j0[t] + (u32)SA_SPAN_DW;
        X[t] = idx < st.len[t] ? sa_glob(st.words[t])[idx] : ~0ull;
    }
}
template <int TT>
__device__ __forceinline__ bool sa_span_doc_words_eval(const u64 doc, u64 (&W)[TT][SA_SPAN_DW], const u64 (&X)[TT], u32 (&c)[TT], u32 (&keep)[TT],
                                                       bool* many, u32* first_blk);
template <int TT>
__device__ __forceinline__ bool sa_span_doc_words(const SpanTerms& st, const u64 doc, u64 (&W)[TT][SA_SPAN_DW], u32 (&c)[TT],
                                                  u32 (&keep)[TT], bool* many, u32* first_blk = nullptr) {
    u32 j0[TT];
    bool all = true;
#pragma unroll
    for (int t = 0; t < TT; t++) { j0[t] = all ? sa_span_first(st, t, doc) : SA_DD_ABSENT; all = all && j0[t] != SA_DD_ABSENT; }
#pragma unroll
    for (int t = 0; t < TT; t++) { c[t] = 0; keep[t] = 0; }
    *many = false;
    if (!all) return false;
    u64 X[TT];                                                   // the word behind the ones held
    sa_span_doc_words_load<TT>(st, j0, W, X);
    return sa_span_doc_words_eval<TT>(doc, W, X, c, keep, many, first_blk);
}
// (c, keep zero and *many false on entry)
template <int TT>
__device__ __forceinline__ bool sa_span_doc_words_eval(const u64 doc, u64 (&W)[TT][SA_SPAN_DW], const u64 (&X)[TT], u32 (&c)[TT], u32 (&keep)[TT],
                                                       bool* many, u32* first_blk) {
#pragma unroll
    for (int t = 0; t < TT; t++) {
        bool run = true;
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++) {
            run = run && (W[t][q] >> SA_KEY_SHIFT) == doc;
            c[t] += run ? 1u : 0u;
        }
        if (run && (X[t] >> SA_KEY_SHIFT) == doc) *many = true;
    }
    if (*many) return true;
    // Presence of each term per 18-position block, as bits relative to the document's first block (bit 1 = that
    // block): the sets of the top of the file become shifts and ANDs over all the document's words at once --
    // in_i(h - 1) at bit h is P[i] << 1, in_i(h + 1) is P[i] >> 1 -- instead of three compares per pair of words.
    u32 blk[TT][SA_SPAN_DW];
    u32 lo_blk = 0xFFFFFFFFu, hi_blk = 0;
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++) {
            blk[t][q] = (u32)(W[t][q] >> SA_LSB_BITS) & (u32)SA_LSB_MASK;
            if ((u32)q < c[t]) {
                lo_blk = blk[t][q] < lo_blk ? blk[t][q] : lo_blk;
                hi_blk = blk[t][q] > hi_blk ? blk[t][q] : hi_blk;
            }
        }
    if (hi_blk - lo_blk > 29u) { *many = true; return true; }   // (a document longer than 500 positions: the slow path)
    if (first_blk) *first_blk = lo_blk;
    u32 P[TT];
#pragma unroll
    for (int t = 0; t < TT; t++) {
        P[t] = 0;
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++)
            if ((u32)q < c[t]) P[t] |= 1u << (blk[t][q] - lo_blk + 1u);
    }
    u32 L = ~0u, R = ~0u;
#pragma unroll
    for (int i = 1; i < TT; i++) {
        L &= (P[0] & P[i]) | (P[i] & (P[0] << 1)) | (P[0] & (P[i] << 1));        // spans.py:79-90
        R &= (P[0] & P[i]) | (P[0] & (P[i] >> 1)) | (P[i] & (P[0] >> 1));
    }
    const u32 K = L | R | (R << 1) | (L >> 1);                   // L(h) | R(h) | R(h - 1) | L(h + 1)   (spans.py:106-118)
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++)
            if ((u32)q < c[t] && ((K >> (blk[t][q] - lo_blk + 1u)) & 1u)) keep[t] |= 1u << q;
    return true;
}

// the positions of the document's candidate words, in the machine's order (term by term, word by word, bit by bit), at
// most `cap` of them: dst[n * stride] = term << 24 | position
template <int TT>
__device__ __forceinline__ void sa_span_doc_positions(const u64 (&W)[TT][SA_SPAN_DW], const u32 (&keep)[TT], u32* dst, const u32 stride, const u32 cap) {
    u32 n = 0;
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++)
            if ((keep[t] >> q) & 1u) {
                const u64 w = W[t][q];
                const u32 payload_base = (u32)((w >> SA_LSB_BITS) & SA_LSB_MASK) * (u32)SA_LSB_BITS;
                u32 bits = (u32)(w & SA_LSB_MASK);
                while (bits != 0 && n < cap) {
                    dst[n * stride] = ((u32)t << 24) | (payload_base + (u32)(__ffs((int)bits) - 1));
                    bits &= bits - 1;
                    n++;
                }
            }
}

// the same as 16-bit entries relative to the document's first block (the gather's lists in LDS): term << 10 | position -
// 18 * first_blk (< 540: sa_span_doc_words keeps documents within 30 blocks)
template <int TT>
__device__ __forceinline__ void sa_span_doc_positions16(const u64 (&W)[TT][SA_SPAN_DW], const u32 (&keep)[TT], const u32 first_blk,
                                                        unsigned short* dst, const u32 stride, const u32 cap) {
    u32 n = 0;
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int q = 0; q < SA_SPAN_DW; q++)
            if ((keep[t] >> q) & 1u) {
                const u64 w = W[t][q];
                const u32 rel_base = ((u32)((w >> SA_LSB_BITS) & SA_LSB_MASK) - first_blk) * (u32)SA_LSB_BITS;
                u32 bits = (u32)(w & SA_LSB_MASK);
                while (bits != 0 && n < cap) {
                    dst[n * stride] = (unsigned short)(((u32)t << 10) | (rel_base + (u32)(__ffs((int)bits) - 1)));
                    bits &= bits - 1;
                    n++;
                }
            }
}

// ---- span entries in 8 bytes: position bits [0, 32), first position [32, 55), last - first + 16 [55, 60), terms [60, 64)
//      (|last - first| <= T + slop: a span's end only moves to a position within the window of its start)
__device__ __forceinline__ u64 sa_ent8_pack(const u32 terms, const int posns, const int beg, const int end) {
    return (u64)(u32)posns | ((u64)((u32)beg & 0x7FFFFFu) << 32) | ((u64)((u32)(end - beg + 16) & 31u) << 55) | ((u64)(terms & 15u) << 60);
}
__device__ __forceinline__ SpanEnt sa_ent8_unpack(const u64 v) {
    SpanEnt e;
    e.posns = (int)(u32)v;
    e.beg = (int)((u32)(v >> 32) & 0x7FFFFFu);
    e.end = e.beg + (int)((u32)(v >> 55) & 31u) - 16;
    e.terms = (u32)(v >> 60);
    return e;
}

// sa_span_flat_loop over 8-byte entries: the lane's entry i at ents[i * S] (row CE: scratch), its positions at
// pos[q * pstride], q < npos.  Returns false when the table outgrew CE entries.
// pos(q): the lane's q-th position as term << 24 | position
template <int CE, int PM, int S, class Pos>
__device__ __forceinline__ bool sa_span_flat_loop8(u64* ents, const Pos& pos, const u32 npos, const u32 num_terms,
                                                   const int max_span_width, u32* incr_out) {
    u32 cursor = 0, pi = 0, si = 0, end = 0, curr_term_mask = 0, tstart = 0;
    int curr_posn = 0, posn_mask = 0;
    bool abandoned = false;
    bool alive = true;
    // the first term's positions only open spans -- there is nothing for them to visit: a loop of its own, a fraction
    // of the general iteration
    if (npos != 0) {
        const u32 first_term = pos(0u) >> 24;
        curr_term_mask = 1u << first_term;
        while (pi < npos && cursor < (u32)CE) {
            const u32 pv = pos(pi);
            if ((pv >> 24) != first_term) break;
            const int q = (int)(pv & 0xFFFFFFu);
            ents[cursor * (u32)S] = sa_ent8_pack(curr_term_mask, sa_posn_mask32(q), q, q);
            cursor++; pi++;
        }
    }
    while (alive) {
        const bool need = si >= end;
        const bool done = need && pi >= npos;
        const bool fresh_it = need && !done;
        const u32 pv = pos(pi < (u32)PM ? pi : (u32)PM - 1u);
        const u32 new_mask = 1u << (pv >> 24);
        tstart = (fresh_it && new_mask != curr_term_mask) ? cursor : tstart;
        curr_posn = fresh_it ? (int)(pv & 0xFFFFFFu) : curr_posn;
        curr_term_mask = fresh_it ? new_mask : curr_term_mask;
        posn_mask = sa_posn_mask32(curr_posn);
        const bool over_f = fresh_it && cursor >= (u32)CE;
        ents[((fresh_it && !over_f) ? cursor : (u32)CE) * (u32)S] = sa_ent8_pack(curr_term_mask, posn_mask, curr_posn, curr_posn);
        end = fresh_it ? tstart : end;
        si = fresh_it ? 0u : si;
        cursor += fresh_it ? 1u : 0u;
        pi += fresh_it ? 1u : 0u;
        const bool vis = !done && !over_f && si < end;
        const u32 slot = vis ? si : (u32)CE;
        const SpanEnt e = sa_ent8_unpack(ents[slot * (u32)S]);
        const u32 nt = (u32)__popc(e.terms), np = sa_popc_sext(e.posns);
        const bool act = vis && !((nt < num_terms && np == num_terms) || (e.terms & curr_term_mask));
        const int sp2 = e.posns | posn_mask;
        const u32 new_unique = sa_popc_sext(sp2);
        const int proposed = sa_iabs32(curr_posn - e.beg);
        const bool rej = np == new_unique || proposed > max_span_width;    // (the position bit stays even if rejected)
        const bool fork_it = act && !rej;
        const bool over_k = fork_it && cursor >= (u32)CE;
        ents[slot * (u32)S] = sa_ent8_pack(fork_it ? (e.terms | curr_term_mask) : e.terms, act ? sp2 : e.posns, e.beg, fork_it ? curr_posn : e.end);
        ents[((fork_it && !over_k) ? cursor : (u32)CE) * (u32)S] = sa_ent8_pack(e.terms | curr_term_mask, sp2 & ~posn_mask, e.beg, e.end);
        cursor += (fork_it && !over_k) ? 1u : 0u;
        si += vis ? 1u : 0u;
        abandoned = over_f || over_k;
        alive = !done && !abandoned;
    }
    if (abandoned) return false;
    // _collect_spans (sa_span_collect): collected span c in the (dead) slot of span c, as first << 32 | last
    u32 ncol = 0;
    for (u32 i = 0; i < cursor; i++) {
        const SpanEnt e = sa_ent8_unpack(ents[i * (u32)S]);
        const bool complete = ((u32)__popc(e.terms) == num_terms) || (sa_popc_sext(e.posns) == num_terms);
        const int b = e.beg, en = e.end;
        const int width = sa_iabs32(en - b);
        if (!complete || width >= max_span_width) continue;
        bool replaced = false;
        for (u32 c = 0; c < ncol; c++) {
            const u64 cc = ents[c * (u32)S];
            const int cb = (int)(cc >> 32), ce = (int)(cc & 0xFFFFFFFFull);
            if (b <= ce && en >= cb && width < sa_iabs32(ce - cb)) {
                ents[c * (u32)S] = ((u64)(u32)b << 32) | (u64)(u32)en;
                replaced = true;
                break;
            }
        }
        if (!replaced) {
            ents[ncol * (u32)S] = ((u64)(u32)b << 32) | (u64)(u32)en;
            ncol++;
        }
    }
    *incr_out = ncol;
    return true;
}

// The last document with candidate words + 1 (the overflow rule's "last document group of the term": with aligned groups,
// of every term).  Looked up when a table fills, i.e. hardly ever: from the end of the collection, 64 documents at a time.
template <int TT>
__device__ __forceinline__ u32 sa_span_doc_last(const SpanDocParams& p, const u32 lane) {
    for (u64 top = p.st.n_docs; top > 0; top = top > 64 ? top - 64 : 0) {
        const u64 doc = top - 1 - lane;
        bool has = false;
        if (lane < top) {
            bool all = true;
#pragma unroll
            for (int t = 0; t < TT; t++) all = all && sa_span_first(p.st, t, doc) != SA_DD_ABSENT;
            has = all && sa_span_doc_npos_slow<TT>(p.st, doc) != 0;
        }
        const u64 m = __ballot(has);
        if (m) return (u32)(top - (u64)__builtin_ctzll(m));           // lane l holds doc top - 1 - l: the lowest lane is the last doc
    }
    return 0;
}

// one document through the wave machine: its words through the directory, 64 at a time -- candidate test per lane,
// then the candidates one after the other.
template <int TT>
__device__ __forceinline__ void sa_span_wave_doc(const SpanDocParams& p, const u64 doc, SpanEnt* s_ents, const u32 lane, u32* slot = nullptr) {
#pragma unroll
    for (int t = 0; t < TT; t++)
        if (sa_span_first(p.st, t, doc) == SA_DD_ABSENT) return;
    const int max_span_width = (int)((u32)TT + p.slop);
    u32 last_doc1 = 0xFFFFFFFFu;                                 // (not looked up yet)
    u32 cursor = 0, my_sum = 0;
    bool full = false;
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < TT; t++) {
        const u32 tstart = cursor;
        const u32 n = p.st.len[t];
        const u64* const w = p.st.words[t];
        bool gave_up = false;
        for (u32 j = sa_span_first(p.st, t, doc); j < n && !gave_up; j += 64u) {
            const u32 idx = j + lane;
            const u64 wv = idx < n ? w[idx] : ~0ull;
            const bool same = (wv >> SA_KEY_SHIFT) == doc;
            const bool kp = same && sa_span_keep_word<TT>(p.st, wv & SA_HEADER_MASK);
            const u64 sb = __ballot(same);
            u64 todo = __ballot(kp);
            // (the document's words are a prefix of the 64: candidates behind a word of another document do not exist)
            while (todo != 0 && !gave_up) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)wv, l);
                const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(wv >> 32), l);
                const u64 wl = ((u64)hi << 32) | lo;
                if (lane == (u32)t) my_sum += (u32)__popc((u32)(wl & SA_LSB_MASK));
                sa_span_wave_word(s_ents, wl, 1u << t, tstart, (u32)TT, max_span_width, lane, cursor, full);
                if (cursor >= SA_NSPANS) {                                 // (see sa_span_doc)
                    if (last_doc1 == 0xFFFFFFFFu) last_doc1 = sa_span_doc_last<TT>(p, lane);
                    if ((u32)doc + 1u != last_doc1) gave_up = true;
                }
            }
            if (sb != ~0ull) break;
        }
    }
    const u32 incr = sa_span_wave_finish(s_ents, cursor, full, my_sum, TT, max_span_width, lane);
    if (lane == 0 && slot) *slot = incr;                                     // (the block stores its documents' counts together)
    else if (lane == 0 && incr) sa_span_doc_put(p, doc, incr);
    __builtin_amdgcn_wave_barrier();
}

// The same with an ACTIVE SET per term (CE < 64).  A span a position skips -- its table is blocked (a partial span whose
// position bits already number T) or it already holds the position's term -- is skipped by every later position of that
// term as well: skipped spans do not change, a blocked span stays blocked, an extended one keeps the term.  So the
// spans a term's positions still have to visit are a set that only shrinks: a bit mask (all spans from before the term
// when it begins), walked by find-first-set; a span leaves it when a visit finds or makes it dead.  The visits that remain
// happen in the same order (position by position, spans ascending), so forks are appended exactly as before.  Two terms:
// the first position of term 1 settles nearly every span of term 0 (extended or blocked), the others only open their own
// span -- a + b + a iterations instead of a + a x b.
template <int CE, int PM, int S, class Pos>
__device__ __forceinline__ bool sa_span_flat_loop8a(u64* ents, const Pos& pos, const u32 npos, const u32 num_terms,
                                                    const int max_span_width, u32* incr_out) {
    static_assert(CE < 64, "the active set is a 64-bit mask");
    typedef typename std::conditional<(CE < 32), u32, u64>::type M;
    u32 cursor = 0, pi = 0, curr_term_mask = 0;
    M active = 0, todo = 0;
    int curr_posn = 0, posn_mask = 0;
    bool abandoned = false;
    bool alive = true;
    if (npos != 0) {                                             // (the first term's positions only open spans)
        const u32 first_term = pos(0u) >> 24;
        curr_term_mask = 1u << first_term;
        while (pi < npos && cursor < (u32)CE) {
            const u32 pv = pos(pi);
            if ((pv >> 24) != first_term) break;
            const int q = (int)(pv & 0xFFFFFFu);
            ents[cursor * (u32)S] = sa_ent8_pack(curr_term_mask, sa_posn_mask32(q), q, q);
            cursor++; pi++;
        }
    }
    while (alive) {
        const bool need = todo == 0;
        const bool done = need && pi >= npos;
        const bool fresh_it = need && !done;
        const u32 pv = pos(pi < (u32)PM ? pi : (u32)PM - 1u);
        const u32 new_mask = 1u << (pv >> 24);
        active = (fresh_it && new_mask != curr_term_mask) ? (M)(((M)1 << cursor) - (M)1) : active;     // (cursor <= CE)
        curr_posn = fresh_it ? (int)(pv & 0xFFFFFFu) : curr_posn;
        curr_term_mask = fresh_it ? new_mask : curr_term_mask;
        posn_mask = sa_posn_mask32(curr_posn);
        const bool over_f = fresh_it && cursor >= (u32)CE;
        ents[((fresh_it && !over_f) ? cursor : (u32)CE) * (u32)S] = sa_ent8_pack(curr_term_mask, posn_mask, curr_posn, curr_posn);
        todo = fresh_it ? active : todo;
        cursor += fresh_it ? 1u : 0u;
        pi += fresh_it ? 1u : 0u;
        // visit the next span of the set
        const bool vis = !done && !over_f && todo != 0;
        const u32 si = vis ? (u32)(sizeof(M) == 4 ? __builtin_ctz((u32)todo) : __builtin_ctzll((u64)todo)) : (u32)CE;
        todo = vis ? (M)(todo & (todo - (M)1)) : todo;
        const SpanEnt e = sa_ent8_unpack(ents[si * (u32)S]);
        const u32 nt = (u32)__popc(e.terms), np = sa_popc_sext(e.posns);
        const bool act = vis && !((nt < num_terms && np == num_terms) || (e.terms & curr_term_mask));
        const int sp2 = e.posns | posn_mask;
        const u32 new_unique = sa_popc_sext(sp2);
        const int proposed = sa_iabs32(curr_posn - e.beg);
        const bool rej = np == new_unique || proposed > max_span_width;    // (the position bit stays even if rejected)
        const bool fork_it = act && !rej;
        const bool over_k = fork_it && cursor >= (u32)CE;
        ents[si * (u32)S] = sa_ent8_pack(fork_it ? (e.terms | curr_term_mask) : e.terms, act ? sp2 : e.posns, e.beg, fork_it ? curr_posn : e.end);
        ents[((fork_it && !over_k) ? cursor : (u32)CE) * (u32)S] = sa_ent8_pack(e.terms | curr_term_mask, sp2 & ~posn_mask, e.beg, e.end);
        cursor += (fork_it && !over_k) ? 1u : 0u;
        // dead for this term from here on: skipped now, extended now, or blocked by the bit this visit left
        const bool dead = !act || fork_it || (nt < num_terms && new_unique == num_terms);
        active = (vis && dead) ? (M)(active & ~((M)1 << si)) : active;
        abandoned = over_f || over_k;
        alive = !done && !abandoned;
    }
    if (abandoned) return false;
    u32 ncol = 0;                                                // (_collect_spans: as in sa_span_flat_loop8)
    for (u32 i = 0; i < cursor; i++) {
        const SpanEnt e = sa_ent8_unpack(ents[i * (u32)S]);
        const bool complete = ((u32)__popc(e.terms) == num_terms) || (sa_popc_sext(e.posns) == num_terms);
        const int b = e.beg, en = e.end;
        const int width = sa_iabs32(en - b);
        if (!complete || width >= max_span_width) continue;
        bool replaced = false;
        for (u32 c = 0; c < ncol; c++) {
            const u64 cc = ents[c * (u32)S];
            const int cb = (int)(cc >> 32), ce = (int)(cc & 0xFFFFFFFFull);
            if (b <= ce && en >= cb && width < sa_iabs32(ce - cb)) {
                ents[c * (u32)S] = ((u64)(u32)b << 32) | (u64)(u32)en;
                replaced = true;
                break;
            }
        }
        if (!replaced) {
            ents[ncol * (u32)S] = ((u64)(u32)b << 32) | (u64)(u32)en;
            ncol++;
        }
    }
    *incr_out = ncol;
    return true;
}

template <int CE, int PM, int S, class Pos>
__device__ __forceinline__ bool sa_span_lane_machine(u64* ents, const Pos& pos, const u32 npos, const u32 num_terms,
                                                     const int max_span_width, u32* incr_out) {
    if constexpr (CE < 64) return sa_span_flat_loop8a<CE, PM, S>(ents, pos, npos, num_terms, max_span_width, incr_out);
    else return sa_span_flat_loop8<CE, PM, S>(ents, pos, npos, num_terms, max_span_width, incr_out);
}

// S lanes of a wave take S neighbours of the block's order, from `start`.  64 lanes: the positions where the gather
// left them; fewer: behind the tables (PM x S words), from the gather's list if it holds them, else from the words again.
template <int CE, int PM, int S, int TT>
__device__ __forceinline__ void sa_span_doc_chunk(const SpanDocParams& p, u64* tab, const unsigned short* s_plist, u32* s_pbase,
                                                  unsigned char* s_bin, const unsigned short* s_order, const u32* s_doc, const u32 lane, const u32 start, const u32 n,
                                                  const bool staged) {
    const bool have = lane < (u32)S && start + lane < n;
    const u32 local = have ? s_order[start + lane] : 0u;
    const u32 npos = have ? s_bin[local] : 0u;
    const u64 doc = s_doc[local];
    u32 incr = 0;
    bool ok;
    if (S == 64) {
        const unsigned short* const pl = s_plist + local;
        const u32 base = s_pbase[local];
        ok = sa_span_lane_machine<CE, PM, S>(tab + lane, [&](const u32 q) -> u32 {
            const u32 v = pl[q * SA_SPAN_FD];
            return ((v >> 10) << 24) | (base + (v & 1023u));
        }, npos, (u32)TT, (int)((u32)TT + p.slop), &incr);
    } else {
        u32* const s_pos = (u32*)(tab + (size_t)(CE + 1) * S);
        if (have) {
            if (npos <= (u32)SA_SPAN_PC) {
                const u32 base = s_pbase[local];
                for (u32 q = 0; q < npos; q++) {
                    const u32 v = s_plist[q * SA_SPAN_FD + local];
                    s_pos[q * (u32)S + lane] = ((v >> 10) << 24) | (base + (v & 1023u));
                }
            } else {
                u64 W[TT][SA_SPAN_DW];
                u32 c[TT], keep[TT];
                bool many = false;
                sa_span_doc_words<TT>(p.st, doc, W, c, keep, &many);
                sa_span_doc_positions<TT>(W, keep, s_pos + lane, (u32)S, npos);
            }
        }
        const u32* const pl = s_pos + (have ? lane : 0u);
        ok = sa_span_lane_machine<CE, PM, S>(tab + (have ? lane : 0u), [&](const u32 q) -> u32 { return pl[q * (u32)S]; },
                                           npos, (u32)TT, (int)((u32)TT + p.slop), &incr);
    }
    if (staged) __builtin_amdgcn_wave_barrier();                               // (every lane has read its s_pbase before any slot is reused)
    if (have) {
        // staged: the document's slot of s_pbase -- its position base is not needed any more -- takes the count
        if (staged) s_pbase[local] = ok ? incr : 0u;
        else if (ok && incr) sa_span_doc_put(p, doc, incr);
        if (!ok) s_bin[local] = (unsigned char)(SA_SPAN_PMAXF + 1);            // its table outgrew the column: marked heavy -- a wave of its own below
    }
}

// -DSA_PROBE (scripts/build_probe.sh; never in the product build): shader cycles a block's wave 0 spends per phase of the doc-parallel
// body, summed over the launch (s_memtime at the phase boundaries), read by sa_debug_span_probe_read
#ifdef SA_PROBE
__device__ unsigned long long g_sa_span_probe[8];
#define SA_SPP(i) do { if (threadIdx.x == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_sa_span_probe[i], (unsigned long long)(t_ - sp_last)); sp_last = t_; } } while (0)
extern "C" int sa_debug_span_probe_read(unsigned long long* out8, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return SA_ERR_HIP;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sa_span_probe), 8 * 8) != hipSuccess) return SA_ERR_HIP;
    if (clear) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sa_span_probe), z, 8 * 8) != hipSuccess) return SA_ERR_HIP; }
    return SA_OK;
}
#else
#define SA_SPP(i) do { } while (0)
#endif

template <int TT, int NTAB_>
__device__ __forceinline__ void sa_span_doc_fused_body(const SpanDocParams& p, const u32 block) {
#ifdef SA_PROBE
    u64 sp_last = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NW = SA_SPAN_FT / 64, ROUNDS = SA_SPAN_FD / SA_SPAN_FT;
    constexpr int TABW = SA_SPAN_FROWS * 64;                     // a wave's tables, in 8-byte words
    __shared__ unsigned short s_plist[SA_SPAN_PC * SA_SPAN_FD];  // position-major: position q of local document d at [q * FD + d]
    __shared__ u32 s_pbase[SA_SPAN_FD];                          // 18 x the document's first block
    __shared__ u32 s_doc[SA_SPAN_FD];                            // the slot's document
    constexpr int NTAB = NTAB_ < NW ? NTAB_ : NW;                // waves with tables: they run the lane machines
    __shared__ alignas(16) u64 s_tab[NTAB * TABW];
    __shared__ unsigned char s_bin[SA_SPAN_FD];
    __shared__ unsigned short s_order[SA_SPAN_FD];               // the machines' work order; afterwards the list of the heavy documents
    __shared__ u32 s_h[SA_SPAN_DB], s_first[SA_SPAN_DB], s_cur[SA_SPAN_DB];
    __shared__ u32 s_nheavy, s_next;
    constexpr int HW = (int)(NTAB * TABW * 8 / (SA_NSPANS * sizeof(SpanEnt)));      // waves that find room for a full 512-span table afterwards
    static_assert(HW >= 1, "the block's tables must hold one full table");
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < SA_SPAN_DB) { s_h[threadIdx.x] = 0; s_cur[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { s_nheavy = 0; s_next = 0; }
    __syncthreads();
    constexpr u32 PMAX = SA_SPAN_PMAXF, HEAVY = SA_SPAN_PMAXF + 1;
    const u64 lo = (u64)block * SA_SPAN_FD;
    // Batched route over ALL documents (a heavy phrase of a batch): the block's counts are STAGED in LDS -- a document's slot of
    // s_pbase, free once its machine has read it -- and stored together at the end, 2 KB of whole lines: the L2 writes through,
    // so a 4-byte store per matching document is a partial-line write to HBM each (measured: 5.3 GB of writes for 0.5 GB of
    // count vectors on the bench's 256-phrase batch).
    const bool ranked = p.rank.cand != nullptr;
    const bool staged = ranked || (p.touched != nullptr && p.anchor < 0);
    // ---- gather: bins and short position lists
    // (the slot's document: the (lo + local)-th document, or the one the (lo + local)-th word of the rarest term's list opens -- a word
    //  whose predecessor belongs to the same document opens none)
    auto slot_doc = [&](const u32 local, bool* valid_out) -> u64 {
        u64 doc = lo + local;
        bool valid = doc < p.st.n_docs;
        if (p.anchor >= 0) {
            const auto aw = sa_glob(p.st.words[p.anchor]);
            valid = doc < p.st.len[p.anchor];
            if (valid) {
                const u64 i = doc;
                doc = aw[i] >> SA_KEY_SHIFT;
                valid = (i == 0 || (aw[i - 1] >> SA_KEY_SHIFT) != doc) && doc < p.st.n_docs;
            }
        } else if (valid && !p.touched && !ranked) {
            p.counts[doc] = 0.f;
        }
        *valid_out = valid;
        return doc;
    };
    // (what the gather keeps of a document whose words are held: its bin, and the short position list of a light one)
    auto slot_bin = [&](const u32 local, const u64 doc, const bool have_words, u64 (&W)[TT][SA_SPAN_DW], const u32 (&keep)[TT], const bool many, const u32 first_blk) -> u32 {
        u32 bin = 0;
        if (have_words) {
            u32 npos = 0;
            if (many) {
                npos = sa_span_doc_npos_slow<TT>(p.st, doc);
            } else {
#pragma unroll
                for (int t = 0; t < TT; t++)
#pragma unroll
                    for (int q = 0; q < SA_SPAN_DW; q++)
                        if ((keep[t] >> q) & 1u) npos += (u32)__popc((u32)(W[t][q] & SA_LSB_MASK));
                if (npos != 0 && npos <= (u32)SA_SPAN_PC) {
                    sa_span_doc_positions16<TT>(W, keep, first_blk, s_plist + local, SA_SPAN_FD, npos);
                    s_pbase[local] = first_blk * (u32)SA_LSB_BITS;
                }
            }
            bin = npos == 0 ? 0u : ((many || npos > PMAX) ? HEAVY : npos);
        }
        return bin;
    };
    bool compacted = false;                                     // (uniform: the ranked fast path moves its documents to the front of the slots)
    if (TT == 2 && ROUNDS == 2 && p.st.dd[0] && p.st.dd[1] && p.st.len[0] && p.st.len[1]) {       // (uniform)
        // Two terms with doc-directory rows (round 6): BOTH documents of a thread are in flight together and no load sits inside a
        // branch (the compiler waits for every load in flight where a branch that contains one joins): the directory cells of both
        // terms for both documents, then all their words -- two dependent round trips per block where the loop below takes six.
        // A slot without a document reads cell 0 / a clamped word and discards it.  (The probe build's phase cycles put half of a
        // block's time into this phase; registers are not what limits the four blocks per CU -- LDS is.)
        u64 docs_[2]; bool valid_[2]; u32 j0_[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
            docs_[r] = slot_doc(local, &valid_[r]);
            s_doc[local] = (u32)docs_[r];
        }
        // (the anchor term's cell is not read: a document that the (lo + local)-th word of the anchor's list opens has that word as its
        //  first -- every lane reads cell 0 of that row instead, one line for the wave where the cells of 64 scattered documents are 64)
        u32 dcell[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int t = 0; t < 2; t++) dcell[r][t] = sa_glob(p.st.dd[t])[(valid_[r] && p.anchor != t) ? docs_[r] : 0ull];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const u32 own = (u32)(lo + (u32)r * SA_SPAN_FT + threadIdx.x);
#pragma unroll
            for (int t = 0; t < 2; t++) dcell[r][t] = p.anchor == t ? own : dcell[r][t];
            j0_[r][0] = valid_[r] ? dcell[r][0] : SA_DD_ABSENT;
            j0_[r][1] = j0_[r][0] != SA_DD_ABSENT ? dcell[r][1] : SA_DD_ABSENT;
        }
        compacted = ranked && p.rank.k <= 32u;                      // (k <= 32: a wave may raise several of the phrase's bound slots -- sa_tile_topk_pruned -- which a compacted block needs)
        if (compacted) {                                            // (uniform)
            // COMPACTION: only the documents that hold BOTH terms go on -- to the front of the block's slots, in slot order.  A slot
            // that opens no document (anchor words of a document's second, third ... word) or whose document lacks the other term
            // kept its lane idle through everything that follows (29 of 64 lanes active per VALU instruction, measured): after the
            // compaction the waves behind the last document skip the candidate test and the position lists altogether.  (The
            // slot <-> document mapping is free in the ranked batch: the block's results leave as (score, doc) candidates.  The
            // unranked routes store whole lines of counts by slot = document and keep their slots.)
            bool has[2];
#pragma unroll
            for (int r = 0; r < 2; r++) has[r] = j0_[r][1] != SA_DD_ABSENT;
            const u64 b0 = __ballot(has[0]), b1 = __ballot(has[1]);
            if (lane == 0) { s_first[wave] = (u32)__popcll(b0); s_first[(u32)NW + wave] = (u32)__popcll(b1); }
            __syncthreads();
            u32 base0 = 0, base1 = 0, n_all = 0;
#pragma unroll
            for (int i = 0; i < 2 * NW; i++) {
                const u32 c = s_first[i];
                base0 += (u32)i < wave ? c : 0u;
                base1 += (u32)i < (u32)NW + wave ? c : 0u;
                n_all += c;
            }
            const u64 ltm = (1ull << lane) - 1ull;
            const u32 pos[2] = {base0 + (u32)__popcll(b0 & ltm), base1 + (u32)__popcll(b1 & ltm)};
            u32* const tmp0 = (u32*)s_tab;                          // (the span tables are not in use before the machines)
            u32* const tmp1 = tmp0 + SA_SPAN_FD;
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (has[r]) { s_doc[pos[r]] = (u32)docs_[r]; tmp0[pos[r]] = j0_[r][0]; tmp1[pos[r]] = j0_[r][1]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
                const bool have = local < n_all;
                if (!have) s_doc[local] = 0u;
                docs_[r] = have ? (u64)s_doc[local] : 0ull;
                j0_[r][0] = have ? tmp0[local] : SA_DD_ABSENT;
                j0_[r][1] = have ? tmp1[local] : SA_DD_ABSENT;
            }
        }
        // The document's words, SA_SPAN_DW + 1 of them per term, as 16 + 16 + 8 bytes: a load instruction of 64 scattered lanes costs the
        // L1 a tag lookup per lane whatever its width, and this phase is what a block waits for.  The loads run up to four words past
        // the term's list -- into the next term's, or into the padding behind the index's words (SA_WORDS_PAD) -- and what lies past
        // the list is discarded.
        static_assert(SA_SPAN_DW == 4 && SA_WORDS_PAD >= SA_SPAN_DW, "five words per term and document: two 16-byte loads and one of 8");
        u64 WW[2][2][SA_SPAN_DW], XX[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const u32 last = p.st.len[t] - 1u;
                const bool there = j0_[r][t] != SA_DD_ABSENT;
                const auto wp = sa_glob(p.st.words[t]) + (there ? j0_[r][t] : 0u);
                sa_w2 a, b;
                __builtin_memcpy(&a, (const void*)wp, 16);
                __builtin_memcpy(&b, (const void*)(wp + 2), 16);
                const u64 c = wp[4];
                const u64 L[SA_SPAN_DW + 1] = {a.x, a.y, b.x, b.y, c};
#pragma unroll
                for (int q = 0; q <= SA_SPAN_DW; q++) {
                    const bool in = there && j0_[r][t] + (u32)q <= last;      // (j0 <= last: no wrap)
                    if (q < SA_SPAN_DW) WW[r][t][q < SA_SPAN_DW ? q : 0] = in ? L[q] : ~0ull; else XX[r][t] = in ? L[q] : ~0ull;
                }
            }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
            u32 c[2] = {0, 0}, keep[2] = {0, 0};
            bool many = false;
            u32 first_blk = 0;
            const bool all = j0_[r][1] != SA_DD_ABSENT;
            u64 (&Wr)[TT][SA_SPAN_DW] = reinterpret_cast<u64 (&)[TT][SA_SPAN_DW]>(WW[r]);
            const u64 (&Xr)[TT] = reinterpret_cast<const u64 (&)[TT]>(XX[r]);
            u32 (&cr)[TT] = reinterpret_cast<u32 (&)[TT]>(c);
            u32 (&kr)[TT] = reinterpret_cast<u32 (&)[TT]>(keep);
#ifdef SA_EXP_NOEVAL
            const bool have_words = all && (Wr[0][0] ^ Wr[1][0] ^ Xr[0] ^ Xr[1] ^ Wr[0][3] ^ Wr[1][3]) == 0x1234567ull;
#else
            const bool have_words = all && sa_span_doc_words_eval<TT>(docs_[r], Wr, Xr, cr, kr, &many, &first_blk);
#endif
            const u32 bin = slot_bin(local, docs_[r], have_words, Wr, kr, many, first_blk);
            if (bin) atomicAdd(&s_h[bin], 1u);
            if (staged && (bin == 0u || bin == HEAVY)) s_pbase[local] = 0u;      // (no machine in the chunks: nothing counted so far; a slot without a document never counts)
            s_bin[local] = (unsigned char)bin;
        }
    } else {
        u64 W[TT][SA_SPAN_DW];
        u32 c[TT], keep[TT];
#pragma unroll 1
        for (int r = 0; r < ROUNDS; r++) {
            const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
            bool valid = false;
            const u64 doc = slot_doc(local, &valid);
            s_doc[local] = (u32)doc;
            u32 bin = 0;
            if (valid) {
                bool many = false;
                u32 first_blk = 0;
                const bool have_words = sa_span_doc_words<TT>(p.st, doc, W, c, keep, &many, &first_blk);
                bin = slot_bin(local, doc, have_words, W, keep, many, first_blk);
                if (bin) atomicAdd(&s_h[bin], 1u);
            }
            if (staged && (bin == 0u || bin == HEAVY)) s_pbase[local] = 0u;      // (no machine in the chunks: nothing counted so far; a slot without a document never counts)
            s_bin[local] = (unsigned char)bin;
        }
    }
    SA_SPP(0);                                                  // gather (this wave's share)
    __syncthreads();
#ifdef SA_EXP_GATHER_ONLY                                        // (SA_EXP_*: timing experiments of scripts/gpu_r6_slop_exp.sh -- a phase removed, results wrong; never in the product build)
    return;
#endif
    SA_SPP(1);                                                  // ... and the wait for the block's other waves
    // ---- order: s_first[b] = documents with more than b positions (lane x of wave 0: bin PMAX - x)
    if (wave == 0) {
        const u32 b = lane <= PMAX ? PMAX - lane : 0u;
        const u32 mine = (lane < PMAX) ? s_h[b] : 0u;             // (bin 0: no document)
        u32 incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 a = __shfl_up(incl, o, SA_WAVE);
            if (lane >= (u32)o) incl += a;
        }
        if (lane <= PMAX) s_first[b] = incl - mine;              // (lane PMAX: bin 0 = all of them)
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < ROUNDS; r++) {
        const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
        const u32 bin = s_bin[local];
        if (bin && bin != HEAVY) s_order[s_first[bin] + atomicAdd(&s_cur[bin], 1u)] = (unsigned short)local;
    }
    __syncthreads();
    SA_SPP(2);                                                  // order
    // ---- machine: chunks of 8 documents while they have more than 16 positions, of 16 above 8, else 64; a wave takes the
    //      next chunk when it is done with its last (the first chunks are the long ones)
    {
        const u32 n = s_first[0], n_c = s_first[2 * SA_SPAN_PC], n_cb = s_first[SA_SPAN_PC];
        const u32 k_c = (n_c + 7u) / 8u;
        const u32 start_b = 8u * k_c < n ? 8u * k_c : n;
        const u32 k_b = n_cb > start_b ? (n_cb - start_b + 15u) / 16u : 0u;
        const u32 start_a = start_b + 16u * k_b < n ? start_b + 16u * k_b : n;
        // (three and four terms: 15 % of the documents need more than 12 spans, 3 % more than 16 -- 32 documents per
        //  wave with 20 rows each there, a wave of their own for the rest)
        constexpr u32 AL = TT == 2 ? 64u : 32u;
        const u32 k_a = (n - start_a + AL - 1u) / AL;
        u64* const tab = s_tab + (size_t)(wave < (u32)NTAB ? wave : 0u) * TABW;
        constexpr int R = SA_SPAN_FROWS;
        // 8 lanes: 32 positions x 8 x 4 B = 1 KiB behind (TABW - 128) / 8 rows; 16 lanes: 16 x 16 x 4 B = 1 KiB behind (TABW - 128) / 16 rows
#ifdef SA_EXP_NOMACH
        if (false)
#endif
        for (; wave < (u32)NTAB; ) {
            u32 ck = 0;
            if (lane == 0) ck = atomicAdd(&s_next, 1u);
            ck = (u32)__builtin_amdgcn_readfirstlane((int)ck);
            if (ck >= k_c + k_b + k_a) break;
            if (ck < k_c) sa_span_doc_chunk<(TABW - 128) / 8 - 1, 4 * SA_SPAN_PC, 8, TT>(p, tab, s_plist, s_pbase, s_bin, s_order, s_doc, lane, 8u * ck, n, staged);
            else if (ck < k_c + k_b) sa_span_doc_chunk<(TABW - 128) / 16 - 1, 2 * SA_SPAN_PC, 16, TT>(p, tab, s_plist, s_pbase, s_bin, s_order, s_doc, lane, start_b + 16u * (ck - k_c), n, staged);
            else if (TT == 2) sa_span_doc_chunk<R - 1, SA_SPAN_PC, 64, TT>(p, tab, s_plist, s_pbase, s_bin, s_order, s_doc, lane, start_a + 64u * (ck - k_c - k_b), n, staged);
            else sa_span_doc_chunk<(TABW - 128) / 32 - 1, SA_SPAN_PC, 32, TT>(p, tab, s_plist, s_pbase, s_bin, s_order, s_doc, lane, start_a + 32u * (ck - k_c - k_b), n, staged);
            __builtin_amdgcn_wave_barrier();
        }
    }
    SA_SPP(3);                                                  // machines (this wave's chunks)
    __syncthreads();
    SA_SPP(4);                                                  // ... and the wait for the block's other waves
    // ---- heavy documents and outgrown tables: a wave each, the block's tables now being free (HW full tables fit)
    // (their list: the documents the gather found heavy and the ones a lane machine marked -- into s_order, which the machines are done with)
#pragma unroll 1
    for (int r = 0; r < ROUNDS; r++) {
        const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
        if (s_bin[local] == HEAVY) s_order[atomicAdd(&s_nheavy, 1u)] = (unsigned short)local;
    }
    __syncthreads();
    const u32 nh = s_nheavy;
    if (wave < (u32)HW)
        for (u32 i = wave; i < nh; i += (u32)HW)
            sa_span_wave_doc<TT>(p, s_doc[s_order[i]], (SpanEnt*)s_tab + (size_t)wave * SA_NSPANS, lane, staged ? &s_pbase[s_order[i]] : nullptr);
    SA_SPP(5);                                                  // heavy documents
#ifdef SA_EXP_NORANK
    if (ranked) return;
#endif
    if (ranked) {
        // counts -> BM25 (the reference's operation order, similarity.py:24-38 / bm25.pyx:19-23, as sa_k_dense_topk_tiles forms it)
        // in place, then the pruned selection of the phrase batches over the block's documents (s_doc: their ids)
        __syncthreads();
        const float one_minus_b = 1.0f - p.rank.b;
        u32 slot_val = 0xFFFFFFFFu;
        if ((threadIdx.x & 63u) < 32u)
            slot_val = __hip_atomic_load(&sa_glob(p.rank.slots)[p.row * 32u + (threadIdx.x & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float* const acc = (float*)s_pbase;
        for (int r = 0; r < ROUNDS; r++) {
            const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
            const u32 c = s_pbase[local];
            float sc = 0.f;
            if (c != 0u) {
                const float t = (float)c;
                const float norm = __fmul_rn(p.rank.k1, __fadd_rn(one_minus_b, __fmul_rn(p.rank.b, __fdiv_rn(sa_glob(p.rank.doc_lens)[s_doc[local]], p.rank.avgdl))));
                sc = __fmul_rn(__fdiv_rn(t, __fadd_rn(t, norm)), p.idf);
            }
            acc[local] = sc;                                             // (each thread rewrites its own two slots)
        }
        __syncthreads();
        sa_tile_topk_pruned<SA_SPAN_FD, SA_SPAN_FT>(acc, slot_val, p.row, block, p.rank.doc_base, p.rank.k, p.rank.slots, p.rank.cand,
                                                    p.rank.cand_cap, p.rank.cand_cnt, s_doc, compacted ? 8u : 0u);
    } else if (staged) {
        __syncthreads();
        bool any = false;
        for (int r = 0; r < ROUNDS; r++) {
            const u32 local = (u32)r * SA_SPAN_FT + threadIdx.x;
            const u64 doc = lo + local;
            const u32 c = s_pbase[local];
            if (doc < p.st.n_docs) p.counts[doc] = (float)c;              // (zeros too: whole lines)
            any = any || (doc < p.st.n_docs && c != 0u);
        }
        if (__any(any) && (threadIdx.x & 63u) == 0u) {
            p.touched[lo >> p.touch_shift] = 1;
            p.touched[(lo + SA_SPAN_FD - 1 < p.st.n_docs ? lo + SA_SPAN_FD - 1 : p.st.n_docs - 1) >> p.touch_shift] = 1;
        }
    }
    SA_SPP(6);                                                  // scoring + ranking (or the staged counts' store)
#ifdef SA_PROBE
    if (threadIdx.x == 0) atomicAdd(&g_sa_span_probe[7], 1ull);    // blocks
#endif
}

template <int TT>
__global__ void __launch_bounds__(SA_SPAN_FT) sa_k_span_doc_fused(const SpanDocParams p) {
    sa_span_doc_fused_body<TT, SA_SPAN_NTAB>(p, blockIdx.x);
}

struct SpanSlot { u32 job, round; };   // eight blocks of a shared launch: job (index in the launch's job array), its round
// B phrases in ONE launch.  Block b of the launch: XCD x = b % 8 (the hardware deals consecutive blocks round-robin to the XCDs),
// slot s = b / 8; work[s] = {job, r} says whose block it is -- the r-th block of the x-th EIGHTH of that job's blocks (of its
// documents: a job's blocks are in doc order).  So XCD x walks the x-th eighth of every job of the launch: the doc directory rows
// and words of a frequent term that several phrases share stay in ONE L2 per doc range and are found there by the next phrase
// (round 4 dealt a job's blocks round-robin: every XCD saw every eighth block of every phrase, no line was ever found again --
// 3.08 x the algorithmic bytes at an L2 hit rate of 14 %).  The host writes the slots so that the jobs of a BUNDLE (neighbours
// in the launch's order: phrases over the same frequent term) take turns, round after round: only a few blocks of a phrase are
// resident at a time, and the ones that follow find the phrase's ranking bounds (SpanRankCtx::slots) already raised -- with a
// phrase's blocks back to back all of them ranked against empty bounds, every wave of a heavy phrase took the exact top-k path.
template <int TT, int NTAB>
__global__ void __launch_bounds__(SA_SPAN_FT) sa_k_span_doc_fused_multi(const SpanDocParams* __restrict__ jobs, const SpanSlot* __restrict__ work) {
    const SpanSlot w = work[blockIdx.x >> 3];
    // (a reference, not a copy: a private copy of the job lives in scratch -- its term arrays are indexed at run time -- and
    // every access of it is HBM traffic; through the pointer the fields are scalar loads at a wave-uniform address)
    const SpanDocParams& p = jobs[w.job];
    const u32 per = (p.n_blocks + 7u) >> 3;
    const u32 wb = (blockIdx.x & 7u) * per + w.round;
    if (wb >= p.n_blocks) return;
    sa_span_doc_fused_body<TT, NTAB>(p, wb);
}

static bool sa_opt_span_doc(const sa_index* ix) { return sa_opt(ix->opts.span_doc, 1) != 0; }

template <int TT>
static void sa_span_doc_launch(const SpanDocParams& p, dim3 fg, hipStream_t st) {
    hipLaunchKernelGGL((sa_k_span_doc_fused<TT>), fg, dim3(SA_SPAN_FT), 0, st, p);
}

static int sa_span_counts_doc_route(sa_index* ix, const SpanTerms& terms_dev, int T, int slop, float** d_out) {
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    void* scratch;
    SA_TRY(sa_index_scratch(ix, (N + 64) * 4, &scratch));
    SpanDocParams p;
    memset(&p, 0, sizeof(p));
    p.st = terms_dev;
    p.st.off[0] = 0;
    for (int t = 0; t < T; t++) p.st.off[t + 1] = p.st.off[t] + p.st.len[t];
    p.slop = (u32)slop;
    p.counts = (float*)scratch;
    *d_out = p.counts;
    // the documents to look at: those of the rarest term -- unless its list is about as long as the collection (then
    // every document, in doc order: no search for the openers, nothing to clear beforehand)
    int rarest = 0;
    for (int t = 1; t < T; t++) if (terms_dev.len[t] < terms_dev.len[rarest]) rarest = t;
    p.anchor = 2 * (u64)terms_dev.len[rarest] >= N ? -1 : rarest;
    if (p.anchor >= 0) SA_HIP(hipMemsetAsync(p.counts, 0, N * sizeof(float), st));
    if (sa_opt(ix->opts.trace, 0)) fprintf(stderr, "slop doc route: %s\n", p.anchor >= 0 ? "over the rarest term's documents" : "over all documents");
    const u64 slots = p.anchor >= 0 ? (u64)terms_dev.len[rarest] : N;
    const dim3 fg((u32)((slots + SA_SPAN_FD - 1) / SA_SPAN_FD));
    switch (T) {
    case 2: sa_span_doc_launch<2>(p, fg, st); break;
    case 3: sa_span_doc_launch<3>(p, fg, st); break;
    default: sa_span_doc_launch<4>(p, fg, st); break;
    }
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// ---- B slop phrases in SHARED launches (phrase batches, BASELINE config 5) ------------------------------------------
// A sampled slop phrase is five short launches, and a batch of them is bound by the host's launch rate (round 2: 25 K
// phrases/s over two streams).  Here every stage is ONE launch for all phrases of a class (2, 3, 4 terms, more):
// blockIdx.y picks the phrase, whose parameters -- exactly the structs the single-phrase kernels take by value -- sit
// in a device array, and blockIdx.x is the block's index in the phrase's own share of the grid.
struct SpanJob {
    SpanTerms st;
    SpanChunkTab ck;
    SpanCompactOut co;
    SpanMachineParams mp;
    int wrap_host;
    u32* cnt;
    unsigned char* flags;
    u32* chunks;
    u32 n_chunks;
    u32 g_flags, g_chunks, g_flat, g_wave;                       // blocks of this phrase in the shared launches
};

template <int TT>
__global__ void __launch_bounds__(256) sa_k_span_flags_multi(const SpanJob* __restrict__ jobs) {
    const SpanJob& J = jobs[blockIdx.y];
    if (blockIdx.x >= J.g_flags) return;
    sa_span_flags_body<TT>(J.st, (const u32*)nullptr, J.wrap_host, J.cnt, J.flags, (float*)nullptr, blockIdx.x, J.g_flags);
}

__global__ void __launch_bounds__(SA_CT) sa_k_span_compact_count_multi(const SpanJob* __restrict__ jobs) {
    const SpanJob& J = jobs[blockIdx.y];
    if (blockIdx.x >= J.g_chunks) return;
    sa_span_compact_count_body(J.st, J.ck, J.flags, J.chunks, J.n_chunks, blockIdx.x, J.g_chunks);
}

__global__ void __launch_bounds__(SA_CT) sa_k_span_compact_emit_multi(const SpanJob* __restrict__ jobs) {
    const SpanJob& J = jobs[blockIdx.y];
    if (blockIdx.x >= J.g_chunks) return;
    sa_span_compact_emit_body(J.st, J.ck, J.flags, J.chunks, J.n_chunks, J.co, J.cnt, blockIdx.x, J.g_chunks);   // (inline scan: the host checks)
}

template <int CE, int PMAX, int TT>
__global__ void __launch_bounds__(64) sa_k_span_machine_flat_multi(const SpanJob* __restrict__ jobs) {
    const SpanJob& J = jobs[blockIdx.y];
    if (blockIdx.x >= J.g_flat) return;
    sa_span_machine_flat_body<CE, PMAX, TT>(J.mp, blockIdx.x);
}

__global__ void __launch_bounds__(64) sa_k_span_machine_wave_multi(const SpanJob* __restrict__ jobs) {
    const SpanJob& J = jobs[blockIdx.y];
    if (blockIdx.x >= J.g_wave) return;
    sa_span_machine_wave_body(J.mp, blockIdx.x, J.g_wave);
}

// slow pass: a resident grid strides over the listed document groups (all of them when over_list is null:
// the kernel-level mirror sa_span_search), full tables in the thread's column of the global slab
__global__ void __launch_bounds__(64) sa_k_span_machine(const SpanMachineParams p) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 G = p.n_threads;
    if (g >= G) return;
    const u32 n_items = p.over_list ? *p.over_cnt : *p.n_heads[0];
    struct EntSlab {
        SpanEnt* base; u64 G;
        struct Ref {
            SpanEnt* q;
            __device__ __forceinline__ operator SpanEnt() const { return *q; }
            __device__ __forceinline__ void operator=(const SpanEnt& e) const { *q = e; }
        };
        __device__ __forceinline__ Ref operator[](u32 i) const { return Ref{base + (u64)i * G}; }
    };
    struct ColSlab {
        u64* base; u64 G;
        __device__ __forceinline__ u64& operator[](u32 i) const { return base[(u64)i * G]; }
    };
    const EntSlab ents{p.ents + g, (u64)G};
    const ColSlab col{p.col + g, (u64)G};
    for (u32 i = g; i < n_items; i += G) {
        const u32 k = p.over_list ? p.over_list[i] : i;
        u32 incr = 0;
        u64 key = 0;
        sa_span_doc<SA_NSPANS, SA_NSPANS>(p, k, ents, col, &incr, &key);
        sa_span_add(p, key, incr);
    }
}

// dense slop > 0 phrase counts of terms[0..T) -> *d_out (float[n_docs], inside the index scratch)
int sa_span_counts_device(sa_index* ix, const u32* terms, int T, int slop, const PosnFilter& filt, float** d_out) {
    if (T > SA_SPAN_MAX_TERMS) { sa_set_error("slop phrases support at most %d terms", SA_SPAN_MAX_TERMS); return SA_ERR_UNSUPPORTED; }
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    SpanTerms terms_dev;
    memset(&terms_dev, 0, sizeof(terms_dev));
    terms_dev.T = T;
    terms_dev.n_docs = ix->n_docs;
    bool known = true;
    size_t total_len = 0, max_len = 0;
    for (int t = 0; t < T; t++) {
        if (terms[t] >= ix->n_terms) { known = false; continue; }
        const u64 off = ix->h_term_off[terms[t]];
        terms_dev.words[t] = ix->d_words + off;
        terms_dev.len[t] = (u32)(ix->h_term_off[terms[t] + 1] - off);
        // probes through the doc directory (whole, unfiltered lists of frequent terms without a top-block word)
        if (!filt.active && ix->n_dd_terms > 0 && sa_opt(ix->opts.span_docdir, 1) != 0) {
            const u32 sl = ix->h_dd_slot[terms[t]];
            if (sl != SA_DD_NONE && sl < ix->h_dd_top.size() && ix->h_dd_top[sl] == 0)
                terms_dev.dd[t] = ix->d_docdir + (size_t)sl * ix->n_docs;
        }
        total_len += terms_dev.len[t];
        if (terms_dev.len[t] > max_len) max_len = terms_dev.len[t];
    }
    if (total_len > 0xFFFFFFF0ull) { sa_set_error("slop phrase: more than 2^32 words in the phrase's terms"); return SA_ERR_UNSUPPORTED; }
    // the doc-parallel route: 2..4 known terms, all with a directory row (whole lists, no word in a last block), header
    // 0 not in L (host arithmetic on the per-term edge flags, as below)
    if (known && !filt.active && T >= 2 && T <= 4 && T + slop <= 15 && N > 0 && N < 0xFFFFFFF0ull && sa_opt_span_doc(ix) &&
        ix->h_term_edge.size() >= (size_t)ix->n_terms) {
        // no word of these lists in a document's last 18-position block: no such word in the index at all (the usual case),
        // or every term with a directory row that says so
        bool all_dd = true, nonempty = true;
        for (int t = 0; t < T; t++) { all_dd = all_dd && terms_dev.dd[t] != nullptr; nonempty = nonempty && terms_dev.len[t] > 0; }
        const bool local = nonempty && (all_dd || !ix->any_top_block);
        const unsigned char e0 = ix->h_term_edge[terms[0]];
        bool L = true;
        for (int i = 1; i < T; i++) {
            const unsigned char ei = ix->h_term_edge[terms[i]];
            const bool a0 = e0 & 1, a0m = e0 & 2, bi = ei & 1, bim = ei & 2;
            L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
        }
        // (two terms: the `L - 1` widening adds nothing -- a word at h with L(h + 1) has R(h), whichever clause of
        //  Lset(h + 1) holds and whichever term the word is of -- so its loss changes nothing either)
        // (it wins on every phrase measured, short lists included -- zipf-1M, slop 2, [0 1] 0.078 vs 0.130 ms, [5 6] 0.030 vs
        //  0.059, [20 30] 0.022 vs 0.034, [5 8 9] 0.040 vs 0.070: one launch against seven)
        const bool take = local && (T == 2 || !L);
        if (sa_opt(ix->opts.trace, 0)) fprintf(stderr, "slop route: %s (T %d, directory rows %d, header 0 in L %d)\n", take ? "doc-parallel" : "general", T, (int)all_dd, (int)L);
        if (take) return sa_span_counts_doc_route(ix, terms_dev, T, slop, d_out);
    }
    // resident state-machine threads: no more than there can be document groups
    u32 G = (u32)((terms_dev.len[0] + 63u) & ~63u);
    u32 g_max = SA_SPAN_THREADS;
    if (sa_opt(ix->opts.span_threads, 0) >= 64) g_max = ((u32)ix->opts.span_threads + 63u) & ~63u;   // tests: force the stride loop
    if (G > g_max) G = g_max;
    if (G == 0) G = 64;
    const size_t slab_bytes = (size_t)G * SA_NSPANS * (sizeof(SpanEnt) + sizeof(u64));
    // chunk counters: two rows (candidates, document heads) over the chunks of all terms; the position filter's
    // compactions (one list at a time) use the same words
    const size_t chunk_words = std::max<size_t>(2 * (total_len / SA_CHUNK + (size_t)T + 1), sa_compact_chunks((u32)(max_len + 1))) + 8;
    const size_t filt_bytes = filt.active ? (total_len + 64 * (size_t)T) * 8 : 0;
    const size_t need = (N + 64) * 8 + (total_len + 64 * T) * 14 + slab_bytes + chunk_words * 4 + filt_bytes + 64 * 1024 +
                        ((size_t)terms_dev.len[0] + 64) * 13 + 256 * (size_t)T;
    void* scratch;
    SA_TRY(sa_index_scratch(ix, need, &scratch));
    char* base = (char*)scratch;
    size_t used = 0;
    auto take = [&](size_t bytes) { char* p = base + used; used += (bytes + 255) & ~(size_t)255; return p; };
    float* running = (float*)take((N + 1) * 4);
    u32* cnt = (u32*)take(SA_SPAN_CNT_WORDS * 4);          // [t] n_cand, [16 + t] n_heads, [32] wrap flag, [48 + t] filter scratch, [64] abandoned groups
    u32* chunks = (u32*)take(chunk_words * 4);
    SpanEnt* ents = (SpanEnt*)take((size_t)G * SA_NSPANS * sizeof(SpanEnt));
    u64* col = (u64*)take((size_t)G * SA_NSPANS * sizeof(u64));
    unsigned char* flags = (unsigned char*)take(total_len + 64);
    u32* over_list = (u32*)take(((size_t)terms_dev.len[0] + 64) * 4);
    u32* order = (u32*)take(((size_t)terms_dev.len[0] + 64) * 4);
    unsigned char* bins = (unsigned char*)take((size_t)terms_dev.len[0] + 64);
    *d_out = running;
    if (!known || N == 0 || total_len == 0) {
        SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));
        return SA_OK;
    }
    // (no fills: sa_k_span_wrap_flag clears the counters, sa_k_span_flags the dense result the machines add into)
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
        for (int t = 0; t < T; t++) { terms_dev.words[t] = ptrs[t]; terms_dev.len[t] = lens[t]; }
    }
    terms_dev.off[0] = 0;
    for (int t = 0; t < T; t++) terms_dev.off[t + 1] = terms_dev.off[t] + terms_dev.len[t];
    if (terms_dev.off[T] == 0 || terms_dev.len[0] == 0) {         // (the position filter left nothing; term 0 drives the walk)
        SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));
        return SA_OK;
    }
    // header 0 in L?  Whole lists: host arithmetic on the index's per-term edge flags (sa_k_term_edges); lists cut
    // by a position filter: sa_k_span_wrap_flag looks at the filtered words.
    const u32* wrap_dev = nullptr;
    int wrap_host = 0;
    u32* cnt_clear = cnt;
    if (filt.active || ix->h_term_edge.size() < (size_t)ix->n_terms) {
        hipLaunchKernelGGL(sa_k_span_wrap_flag, dim3(1), dim3(256), 0, st, terms_dev, cnt);
        wrap_dev = cnt + SA_SPAN_CNT_WRAP;
        cnt_clear = nullptr;
    } else {
        const unsigned char e0 = ix->h_term_edge[terms[0]];
        bool L = true;
        for (int i = 1; i < T; i++) {
            const unsigned char ei = ix->h_term_edge[terms[i]];
            const bool a0 = e0 & 1, a0m = e0 & 2, bi = ei & 1, bim = ei & 2;
            L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
        }
        wrap_host = L ? 1 : 0;
    }
    {
        const u32 total = terms_dev.off[T];
        // (at least a block per 4096 documents: the launch also clears the dense result -- with one block for two
        //  50-word lists that was 0.12 ms of a 0.15 ms query on 1 M documents)
        const u32 by_words = total / 256 + 1 < 16384 ? total / 256 + 1 : 16384;
        const u32 by_docs = (u32)std::min<u64>(N / 4096 + 1, 1024);
        const u32 grid = std::max(by_words, by_docs);
        switch (T) {
        case 2: hipLaunchKernelGGL(sa_k_span_flags<2>, dim3(grid), dim3(256), 0, st, terms_dev, wrap_dev, wrap_host, cnt_clear, flags, running); break;
        case 3: hipLaunchKernelGGL(sa_k_span_flags<3>, dim3(grid), dim3(256), 0, st, terms_dev, wrap_dev, wrap_host, cnt_clear, flags, running); break;
        case 4: hipLaunchKernelGGL(sa_k_span_flags<4>, dim3(grid), dim3(256), 0, st, terms_dev, wrap_dev, wrap_host, cnt_clear, flags, running); break;
        default: hipLaunchKernelGGL(sa_k_span_flags<0>, dim3(grid), dim3(256), 0, st, terms_dev, wrap_dev, wrap_host, cnt_clear, flags, running); break;
        }
    }
    SpanMachineParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.T = T; mp.slop = (u32)slop; mp.ents = ents; mp.col = col; mp.fcounts = running; mp.n_docs = N; mp.n_threads = G;
    SpanChunkTab ck;
    SpanCompactOut co;
    memset(&ck, 0, sizeof(ck));
    memset(&co, 0, sizeof(co));
    for (int t = 0; t < T; t++) {
        u64* cand = (u64*)take(((size_t)terms_dev.len[t] + 1) * 8);
        u32* heads = (u32*)take(((size_t)terms_dev.len[t] + 1) * 4);
        if (used > need) { sa_set_error("internal: span scratch exhausted"); return SA_ERR_STATE; }
        mp.cand[t] = cand; mp.n_cand[t] = cnt + t; mp.heads[t] = heads; mp.n_heads[t] = cnt + SA_SPAN_MAX_TERMS + t;
        co.cand[t] = cand; co.heads[t] = heads;
        co.gpos[t] = (unsigned char*)take((size_t)terms_dev.len[t] + 1);
        if (used > need) { sa_set_error("internal: span scratch exhausted"); return SA_ERR_STATE; }
        ck.coff[t + 1] = ck.coff[t] + sa_compact_chunks(terms_dev.len[t]);
    }
    for (int t = T; t < SA_SPAN_MAX_TERMS; t++) ck.coff[t + 1] = ck.coff[t];
    {
        const u32 n_chunks = ck.coff[T];                       // (> 0: total_len > 0)
        const u32 grid = n_chunks < 16384u ? n_chunks : 16384u;
        hipLaunchKernelGGL(sa_k_span_compact_count, dim3(grid), dim3(SA_CT), 0, st, terms_dev, ck, (const unsigned char*)flags, chunks, n_chunks);
        const bool inline_scan = n_chunks <= (u32)SA_SPAN_INLINE_SCAN && n_chunks <= 16384u;       // (one chunk per block)
        if (!inline_scan) hipLaunchKernelGGL(sa_k_span_compact_scan, dim3(2 * T), dim3(1024), 0, st, ck, T, chunks, n_chunks, cnt);
        hipLaunchKernelGGL(sa_k_span_compact_emit, dim3(grid), dim3(SA_CT), 0, st, terms_dev, ck, (const unsigned char*)flags,
                           (const u32*)chunks, n_chunks, co, inline_scan ? cnt : (u32*)nullptr);
    }
    // fast pass (tables in LDS, one thread per document group), then the groups it abandoned with full tables
    if (sa_opt(ix->opts.span_fast, 1) != 0 && terms_dev.len[0] > 0) {
        mp.over_list = over_list; mp.over_cnt = cnt + 4 * SA_SPAN_MAX_TERMS;
        // (work order only when the document groups outnumber the lanes the device keeps resident -- 256 CUs x 13 waves
        //  x 64: below that every wave starts at once, the order changes nothing, and a light phrase saves two launches)
        const int sort_env = (int)sa_opt(ix->opts.span_sort, -1);
        if (sort_env != 0 && (sort_env > 0 || terms_dev.len[0] > (u32)SA_SPAN_SORT_MIN)) {
            SpanBinParams bp;
            memset(&bp, 0, sizeof(bp));
            for (int t = 0; t < T; t++) { bp.gpos[t] = co.gpos[t]; bp.n_heads[t] = mp.n_heads[t]; }
            bp.T = T; bp.bin = bins; bp.sizes = cnt + SA_SPAN_CNT_BINS; bp.cursors = cnt + SA_SPAN_CNT_BINS + SA_SPAN_NBINS; bp.order = order;
            const u32 bg = std::max<u32>(1u, std::min<u32>(256u, (terms_dev.len[0] + 2047u) / 2048u));
            hipLaunchKernelGGL(sa_k_span_bin_count, dim3(bg), dim3(1024), 0, st, bp);
            hipLaunchKernelGGL(sa_k_span_bin_scatter, dim3(bg), dim3(1024), 0, st, bp);
            mp.order = order;
        }
        const dim3 fg((terms_dev.len[0] + 63u) / 64u);
        // table size: two-term documents rarely need more than 12 spans; with three terms and more 15 % of them do
        // (3 % more than 16, < 1 % more than 20), and the heavy pass (a wave per document) costs more than the lower
        // residency of a larger table
        if (T == 2) hipLaunchKernelGGL((sa_k_span_machine_flat<SA_SPAN_LDS, SA_SPAN_PMAX, 2>), fg, dim3(64), 0, st, mp);
        else if (T == 3) hipLaunchKernelGGL((sa_k_span_machine_flat<20, 20, 3>), fg, dim3(64), 0, st, mp);
        else hipLaunchKernelGGL((sa_k_span_machine_flat<20, 20, 0>), fg, dim3(64), 0, st, mp);
        mp.in_list = over_list; mp.in_cnt = cnt + 4 * SA_SPAN_MAX_TERMS;
        // (a wave per abandoned group; 1024 blocks striding over the list were measured slower on the heaviest 2-term
        //  query, 0.155 vs 0.140 ms: it abandons thousands of groups)
        const u32 g2 = std::min<u32>(8192u, terms_dev.len[0]);
        hipLaunchKernelGGL(sa_k_span_machine_wave, dim3(g2), dim3(64), 0, st, mp);
    } else {
        hipLaunchKernelGGL(sa_k_span_machine, dim3(G / 64), dim3(64), 0, st, mp);       // every group, full tables in the global slab
    }
    return SA_OK;
}

// The slop phrases of a phrase batch through the shared launches above (sa_k_span_*_multi).  Phrase i = terms[i][0 .. T[i]),
// slop[i]; on return handled[i] != 0 says that its dense counts are (being) accumulated in d_out[i] (float[n_docs] inside
// the index's batch scratch, valid until the next call); the other phrases -- an unknown term, lists long enough for
// the work-order sort or the separate chunk scan, forced test switches -- are left to sa_span_counts_device.
// Everything is enqueued on `st`; the caller holds the index lock.
int sa_span_counts_batch(sa_index* ix, hipStream_t st, int n, const u32* const* terms, const int* T, const int* slop,
                         const float* idf, const u32* rows, float** d_out, unsigned char* handled,
                         const sa_dense_rank_job** d_rank_jobs, int* n_rank_jobs, u32 rank_tile_shift, const SpanRankCtx* rank) {
    for (int i = 0; i < n; i++) { handled[i] = 0; d_out[i] = nullptr; }
    *d_rank_jobs = nullptr; *n_rank_jobs = 0;
    const u64 N = ix->n_docs;
    if (n <= 0 || N == 0) return SA_OK;
    {
        if (sa_opt(ix->opts.span_fast, 1) == 0 || sa_opt(ix->opts.span_sort, -1) > 0 || sa_opt(ix->opts.span_multi, 1) == 0) return SA_OK;
        if (ix->h_term_edge.size() < (size_t)ix->n_terms) return SA_OK;
    }
    const bool use_dd = ix->n_dd_terms > 0 && sa_opt(ix->opts.span_docdir, 1) != 0;
    std::vector<SpanJob> jobs;
    std::vector<SpanDocParams> djobs[3];               // phrases of 2 / 3 / 4 terms on the doc-parallel route
    std::vector<int> drow[3];
    const bool doc_route = sa_opt_span_doc(ix) && sa_opt(ix->opts.span_doc_multi, 1) != 0;
    // the doc-parallel phrases rank their documents inside the kernel (no count vector) when the caller hands its ranking state
    const bool fused_rank = rank && rank->cand && sa_opt(ix->opts.span_doc_rank, 1) != 0;
    std::vector<int> job_row, job_class;
    std::vector<size_t> job_off;                       // scratch offset of each job
    size_t used = 0;
    auto take = [&](size_t bytes) { const size_t o = used; used += (bytes + 255) & ~(size_t)255; return o; };
    for (int i = 0; i < n; i++) {
        const int Ti = T[i];
        if (Ti < 2 || Ti > SA_SPAN_MAX_TERMS || slop[i] <= 0) continue;
        SpanJob J;
        memset(&J, 0, sizeof(J));
        J.st.T = Ti; J.st.n_docs = N;
        bool ok = true;
        size_t total_len = 0;
        for (int t = 0; t < Ti && ok; t++) {
            const u32 term = terms[i][t];
            if (term >= ix->n_terms) { ok = false; break; }
            const u64 off = ix->h_term_off[term];
            J.st.words[t] = ix->d_words + off;
            J.st.len[t] = (u32)(ix->h_term_off[term + 1] - off);
            if (use_dd) {
                const u32 sl = ix->h_dd_slot[term];
                if (sl != SA_DD_NONE && sl < ix->h_dd_top.size() && ix->h_dd_top[sl] == 0) J.st.dd[t] = ix->d_docdir + (size_t)sl * N;
            }
            total_len += J.st.len[t];
        }
        if (!ok || total_len == 0 || J.st.len[0] == 0 || total_len > 0x7FFFFFF0ull) continue;
        // the doc-parallel route (sa_k_span_doc_fused: gather, work order and span machines per block of documents, nothing but
        // the lists read) where the phrase qualifies -- the conditions of sa_span_counts_device -- whatever the lists' lengths:
        // ALL such phrases of the batch share one launch per term count
        if (doc_route && Ti <= 4 && Ti + slop[i] <= 15 && N < 0xFFFFFFF0ull) {
            bool all_dd = true, nonempty = true;
            for (int t = 0; t < Ti; t++) { all_dd = all_dd && J.st.dd[t] != nullptr; nonempty = nonempty && J.st.len[t] > 0; }
            const unsigned char e0 = ix->h_term_edge[terms[i][0]];
            bool L = true;
            for (int t = 1; t < Ti; t++) {
                const unsigned char ei = ix->h_term_edge[terms[i][t]];
                const bool a0 = e0 & 1, a0m = e0 & 2, bi = ei & 1, bim = ei & 2;
                L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
            }
            if (nonempty && (all_dd || !ix->any_top_block) && (Ti == 2 || !L)) {
                SpanDocParams P;
                memset(&P, 0, sizeof(P));
                P.st = J.st;
                P.st.off[0] = 0;
                for (int t = 0; t < Ti; t++) P.st.off[t + 1] = P.st.off[t] + P.st.len[t];
                P.slop = (u32)slop[i];
                int rarest = 0;
                for (int t = 1; t < Ti; t++) if (P.st.len[t] < P.st.len[rarest]) rarest = t;
                P.anchor = 2 * (u64)P.st.len[rarest] >= N ? -1 : rarest;
                const u64 slots = P.anchor >= 0 ? (u64)P.st.len[rarest] : N;
                P.n_blocks = (u32)((slots + SA_SPAN_FD - 1) / SA_SPAN_FD);
                djobs[Ti - 2].push_back(P);
                drow[Ti - 2].push_back(i);
                continue;
            }
        }
        if (J.st.len[0] > (u32)SA_SPAN_SORT_MIN) continue;                         // (would be put in work order)
        for (int t = 0; t < Ti; t++) J.st.off[t + 1] = J.st.off[t] + J.st.len[t];
        for (int t = 0; t < Ti; t++) J.ck.coff[t + 1] = J.ck.coff[t] + sa_compact_chunks(J.st.len[t]);
        for (int t = Ti; t < SA_SPAN_MAX_TERMS; t++) J.ck.coff[t + 1] = J.ck.coff[t];
        J.n_chunks = J.ck.coff[Ti];
        if (J.n_chunks == 0 || J.n_chunks > (u32)SA_SPAN_INLINE_SCAN) continue;
        {
            // header 0 in L?  (host arithmetic on the index's per-term edge flags, as in sa_span_counts_device)
            const unsigned char e0 = ix->h_term_edge[terms[i][0]];
            bool L = true;
            for (int t = 1; t < Ti; t++) {
                const unsigned char ei = ix->h_term_edge[terms[i][t]];
                const bool a0 = e0 & 1, a0m = e0 & 2, bi = ei & 1, bim = ei & 2;
                L &= (a0 && bi) || (bi && a0m) || (a0 && bim);
            }
            J.wrap_host = L ? 1 : 0;
        }
        // scratch of this phrase (offsets now, addresses once the buffer is known)
        job_off.push_back(used);
        take(SA_SPAN_CNT_WORDS * 4);
        take(((size_t)2 * J.n_chunks + 8) * 4);
        take(total_len + 64);
        take(((size_t)J.st.len[0] + 64) * 4);                        // groups the fast pass abandons
        for (int t = 0; t < Ti; t++) { take(((size_t)J.st.len[t] + 1) * 8); take(((size_t)J.st.len[t] + 1) * 4); take((size_t)J.st.len[t] + 1); }
        J.mp.T = Ti; J.mp.slop = (u32)slop[i]; J.mp.n_docs = N;
        J.g_flags = std::min<u32>(16384u, J.st.off[Ti] / 256u + 1u);
        J.g_chunks = std::min<u32>(16384u, J.n_chunks);
        J.g_flat = (J.st.len[0] + 63u) / 64u;
        J.g_wave = std::min<u32>(128u, J.st.len[0]);                // (the fast pass abandons a few per cent of the groups at most; the blocks stride)
        jobs.push_back(J);
        job_row.push_back(i);
        job_class.push_back(Ti == 2 ? 0 : Ti == 3 ? 1 : Ti == 4 ? 2 : 3);
    }
    const int nj = (int)jobs.size();
    const int nd = (int)(djobs[0].size() + djobs[1].size() + djobs[2].size());
    const int nv = nj + (fused_rank ? 0 : nd);         // count vectors / ranking jobs
    if (nj + nd == 0) return SA_OK;
    // jobs of a class are neighbours in the device array
    std::vector<int> order((size_t)nj);
    for (int j = 0; j < nj; j++) order[(size_t)j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return job_class[(size_t)a] < job_class[(size_t)b]; });
    // (the ranking jobs -- counts, idf, batch row of every phrase taken -- ride behind the span jobs in the same upload)
    const size_t span_jobs_bytes = ((size_t)nj * sizeof(SpanJob) + 255) & ~(size_t)255;
    const size_t rank_jobs_bytes = ((size_t)nv * sizeof(sa_dense_rank_job) + 255) & ~(size_t)255;
    // (... and behind the doc-parallel phrases' parameter blocks the slots of their launches: {job, round} per eight blocks)
    const size_t doc_jobs_bytes = ((size_t)nd * sizeof(SpanDocParams) + 255) & ~(size_t)255;
    size_t n_slots = 0;
    for (int c = 0; c < 3; c++) for (const SpanDocParams& P : djobs[c]) n_slots += (P.n_blocks + 7u) >> 3;
    const size_t jobs_bytes = span_jobs_bytes + rank_jobs_bytes + doc_jobs_bytes + ((n_slots * sizeof(SpanSlot) + 255) & ~(size_t)255);
    const size_t need = jobs_bytes + used + 4096;
    if (ix->span_batch_bytes < need) {
        SA_HIP(hipStreamSynchronize(st));
        if (ix->d_span_batch) SA_HIP(hipFree(ix->d_span_batch));
        ix->d_span_batch = nullptr; ix->span_batch_bytes = 0;
        SA_HIP(hipMalloc(&ix->d_span_batch, need + need / 4));
        ix->span_batch_bytes = need + need / 4;
    }
    if (ix->span_jobs_host_bytes < jobs_bytes) {
        if (ix->ev_span_jobs) SA_HIP(hipEventSynchronize(ix->ev_span_jobs));
        if (ix->h_span_jobs) SA_HIP(hipHostFree(ix->h_span_jobs));
        ix->h_span_jobs = nullptr; ix->span_jobs_host_bytes = 0;
        SA_HIP(hipHostMalloc(&ix->h_span_jobs, jobs_bytes * 2, 0));
        ix->span_jobs_host_bytes = jobs_bytes * 2;
    }
    // The dense count vectors come from a pool of their own that is all zeros between runs: the launch that ranks a
    // vector puts back a zero wherever it read a count (sa_k_dense_topk_tiles_multi), so no run clears B x n_docs
    // floats.  A run that did not get as far as its ranking launch leaves the pool marked dirty: cleared here.
    // (behind each vector: a byte per ranking tile, "has a count" -- same life cycle)
    const size_t cvec = (size_t)((N + 64) & ~(u64)63);
    const size_t cstride = cvec + ((((size_t)(N >> rank_tile_shift) + 1 + 3) / 4 + 63) & ~(size_t)63);
    if (ix->span_counts_cap < (size_t)nv * cstride) {
        SA_HIP(hipStreamSynchronize(st));
        if (ix->d_span_counts) SA_HIP(hipFree(ix->d_span_counts));
        ix->d_span_counts = nullptr; ix->span_counts_cap = 0;
        SA_HIP(hipMalloc(&ix->d_span_counts, (size_t)nv * cstride * sizeof(float)));
        ix->span_counts_cap = (size_t)nv * cstride;
        ix->span_counts_dirty = true;
    }
    if (ix->span_counts_dirty && ix->span_counts_cap) SA_HIP(hipMemsetAsync(ix->d_span_counts, 0, ix->span_counts_cap * sizeof(float), st));
    ix->span_counts_dirty = nv > 0;                                 // (until the caller has enqueued the ranking launch)
    if (!ix->ev_span_jobs) SA_HIP(hipEventCreateWithFlags(&ix->ev_span_jobs, hipEventDisableTiming));
    else SA_HIP(hipEventSynchronize(ix->ev_span_jobs));             // (the previous upload has left the host image)
    char* base = (char*)ix->d_span_batch + jobs_bytes;
    SpanJob* hj = (SpanJob*)ix->h_span_jobs;
    sa_dense_rank_job* hr = (sa_dense_rank_job*)((char*)ix->h_span_jobs + span_jobs_bytes);
    int class_first[5] = {0, 0, 0, 0, 0};
    for (int q = 0; q < nj; q++) {
        const int j = order[(size_t)q];
        SpanJob J = jobs[(size_t)j];
        size_t o = job_off[(size_t)j];
        auto next = [&](size_t bytes) { char* p = base + o; o += (bytes + 255) & ~(size_t)255; return p; };
        const int Ti = J.st.T;
        float* running = ix->d_span_counts + (size_t)q * cstride;
        J.cnt = (u32*)next(SA_SPAN_CNT_WORDS * 4);
        J.chunks = (u32*)next(((size_t)2 * J.n_chunks + 8) * 4);
        J.flags = (unsigned char*)next((size_t)J.st.off[Ti] + 64);
        u32* over_list = (u32*)next(((size_t)J.st.len[0] + 64) * 4);
        for (int t = 0; t < Ti; t++) {
            u64* cand = (u64*)next(((size_t)J.st.len[t] + 1) * 8);
            u32* heads = (u32*)next(((size_t)J.st.len[t] + 1) * 4);
            J.co.cand[t] = cand; J.co.heads[t] = heads;
            J.co.gpos[t] = (unsigned char*)next((size_t)J.st.len[t] + 1);
            J.mp.cand[t] = cand; J.mp.n_cand[t] = J.cnt + t; J.mp.heads[t] = heads; J.mp.n_heads[t] = J.cnt + SA_SPAN_MAX_TERMS + t;
        }
        J.mp.fcounts = running;
        J.mp.touched = (unsigned char*)(running + cvec); J.mp.touch_shift = rank_tile_shift;
        J.mp.over_list = over_list; J.mp.over_cnt = J.cnt + 4 * SA_SPAN_MAX_TERMS;
        J.mp.in_list = over_list; J.mp.in_cnt = J.cnt + 4 * SA_SPAN_MAX_TERMS;
        hj[q] = J;
        hr[q].counts = running; hr[q].touched = (unsigned char*)(running + cvec);
        hr[q].idf = idf[job_row[(size_t)j]]; hr[q].row = rows[job_row[(size_t)j]];
        d_out[job_row[(size_t)j]] = running;
        handled[job_row[(size_t)j]] = 1;
        class_first[job_class[(size_t)j] + 1] = q + 1;
    }
    for (int c = 1; c <= 4; c++) if (class_first[c] < class_first[c - 1]) class_first[c] = class_first[c - 1];
    // the doc-parallel phrases: vectors nj .. nv - 1 of the pool, their parameter blocks behind the ranking jobs
    SpanDocParams* hd = (SpanDocParams*)((char*)ix->h_span_jobs + span_jobs_bytes + rank_jobs_bytes);
    const SpanDocParams* dd_dev = (const SpanDocParams*)((char*)ix->d_span_batch + span_jobs_bytes + rank_jobs_bytes);
    u32 dfirst[4] = {0, 0, 0, 0}, dblocks[3] = {0, 0, 0};
    SpanSlot* const hw = (SpanSlot*)((char*)ix->h_span_jobs + span_jobs_bytes + rank_jobs_bytes + doc_jobs_bytes);
    const SpanSlot* const dw_dev = (const SpanSlot*)((char*)ix->d_span_batch + span_jobs_bytes + rank_jobs_bytes + doc_jobs_bytes);
    u32 wfirst[3] = {0, 0, 0};
    // jobs of a bundle take turns in the launch's slots (option span_bundle; 1: a phrase's blocks back to back)
    const u32 bundle = (u32)std::min<long long>(4096, std::max<long long>(1, sa_opt(ix->opts.span_bundle, 32)));
    {
        int q = nj;
        u32 k = 0, wk = 0;
        for (int c = 0; c < 3; c++) {
            dfirst[c] = k;
            u32 b0 = 0;
            // phrases that share their LONGEST list follow each other (sa_k_span_doc_fused_multi: what one phrase brought into an
            // XCD's L2 the next one finds there); every job starts at a multiple of 8 blocks
            std::vector<size_t> ord(djobs[c].size());
            for (size_t j = 0; j < ord.size(); j++) ord[j] = j;
            auto longest = [&](const SpanDocParams& P) {
                int best = 0;
                for (int t = 1; t < (int)P.st.T; t++) if (P.st.len[t] > P.st.len[best]) best = t;
                return std::make_pair(P.st.len[best], (const void*)P.st.words[best]);
            };
            std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
                const auto ka = longest(djobs[c][a]), kb = longest(djobs[c][b]);
                return ka.first != kb.first ? ka.first > kb.first : ka.second < kb.second;
            });
            wfirst[c] = wk;
            for (size_t j0 = 0; j0 < ord.size(); j0 += bundle) {
                const size_t j1 = std::min(ord.size(), j0 + (size_t)bundle);
                u32 rounds = 0;
                for (size_t jj = j0; jj < j1; jj++) rounds = std::max(rounds, (djobs[c][ord[jj]].n_blocks + 7u) >> 3);
                for (u32 r = 0; r < rounds; r++)
                    for (size_t jj = j0; jj < j1; jj++)
                        if (r < ((djobs[c][ord[jj]].n_blocks + 7u) >> 3)) { hw[wk].job = (u32)(jj); hw[wk].round = r; wk++; }
            }
            for (size_t jj = 0; jj < ord.size(); jj++, q++, k++) {
                const size_t j = ord[jj];
                SpanDocParams P = djobs[c][j];
                const int row = drow[c][j];
                P.block0 = b0;
                b0 += ((P.n_blocks + 7u) >> 3) << 3;
                if (fused_rank) {
                    P.rank = *rank; P.idf = idf[row]; P.row = rows[row];
                    q--;                                           // (no vector of the pool)
                } else {
                    float* running = ix->d_span_counts + (size_t)q * cstride;
                    P.counts = running;
                    P.touched = (unsigned char*)(running + cvec); P.touch_shift = rank_tile_shift;
                    hr[q].counts = running; hr[q].touched = P.touched;
                    hr[q].idf = idf[row]; hr[q].row = rows[row];
                    d_out[row] = running;
                }
                hd[k] = P;
                handled[row] = 1;
            }
            dblocks[c] = b0;
        }
        dfirst[3] = k;
    }
    SA_HIP(hipMemcpyAsync(ix->d_span_batch, hj, jobs_bytes, hipMemcpyHostToDevice, st));
    *d_rank_jobs = (const sa_dense_rank_job*)((char*)ix->d_span_batch + span_jobs_bytes);
    *n_rank_jobs = nv;
    SA_HIP(hipEventRecord(ix->ev_span_jobs, st));
    const SpanJob* dj = (const SpanJob*)ix->d_span_batch;
    for (int c = 0; c < 4; c++) {
        const int a = class_first[c], b = class_first[c + 1];
        if (b <= a) continue;
        u32 gf = 1, gc = 1, gl = 1, gw = 1;
        for (int q = a; q < b; q++) {
            gf = std::max(gf, hj[q].g_flags); gc = std::max(gc, hj[q].g_chunks);
            gl = std::max(gl, hj[q].g_flat); gw = std::max(gw, hj[q].g_wave);
        }
        const u32 nb = (u32)(b - a);
        const SpanJob* jc = dj + a;
        switch (c) {
        case 0: hipLaunchKernelGGL(sa_k_span_flags_multi<2>, dim3(gf, nb), dim3(256), 0, st, jc); break;
        case 1: hipLaunchKernelGGL(sa_k_span_flags_multi<3>, dim3(gf, nb), dim3(256), 0, st, jc); break;
        case 2: hipLaunchKernelGGL(sa_k_span_flags_multi<4>, dim3(gf, nb), dim3(256), 0, st, jc); break;
        default: hipLaunchKernelGGL(sa_k_span_flags_multi<0>, dim3(gf, nb), dim3(256), 0, st, jc); break;
        }
        hipLaunchKernelGGL(sa_k_span_compact_count_multi, dim3(gc, nb), dim3(SA_CT), 0, st, jc);
        hipLaunchKernelGGL(sa_k_span_compact_emit_multi, dim3(gc, nb), dim3(SA_CT), 0, st, jc);
        if (c == 0) hipLaunchKernelGGL((sa_k_span_machine_flat_multi<SA_SPAN_LDS, SA_SPAN_PMAX, 2>), dim3(gl, nb), dim3(64), 0, st, jc);
        else if (c == 1) hipLaunchKernelGGL((sa_k_span_machine_flat_multi<20, 20, 3>), dim3(gl, nb), dim3(64), 0, st, jc);
        else hipLaunchKernelGGL((sa_k_span_machine_flat_multi<20, 20, 0>), dim3(gl, nb), dim3(64), 0, st, jc);
        hipLaunchKernelGGL(sa_k_span_machine_wave_multi, dim3(gw, nb), dim3(64), 0, st, jc);
    }
    for (int c = 0; c < 3; c++) {
        const u32 cnt = dfirst[c + 1] - dfirst[c];
        if (cnt == 0 || dblocks[c] == 0) continue;
        const SpanDocParams* jc = dd_dev + dfirst[c];
        const SpanSlot* const wc = dw_dev + wfirst[c];
        // (span_lds_pad: MEASUREMENT HOOK -- unused dynamic LDS per block lowers the resident blocks per CU: the occupancy experiment of DESIGN 3.4)
        const u32 pad = (u32)std::min<long long>(120000, std::max<long long>(0, sa_opt(ix->opts.span_lds_pad, 0)));
        // Waves of a block with span tables.  A launch that fills the device is bound by how many blocks a CU holds -- a block's gather
        // waits for memory, its machines are short -- and the tables are most of a block's LDS: with two of the four waves running the
        // machines a block takes 26 KB instead of 39 and a CU holds six instead of four (the bench's 256-phrase batch: 0.588 -> 0.48 ms).
        // A launch of few blocks is bound by a block's own latency: all four waves run machines there (33 sampled phrases: -7 % with two).
        const long long tw = sa_opt(ix->opts.span_tab_waves, dblocks[c] > 4u * (u32)ix->n_cus ? SA_SPAN_NTAB_BATCH : SA_SPAN_NTAB);
        const bool few = tw < SA_SPAN_NTAB;
        if (sa_opt(ix->opts.trace, 0)) fprintf(stderr, "sa_span_counts_batch: doc-parallel launch of %u phrases, %u blocks, %d table waves per block\n", cnt, dblocks[c], few ? SA_SPAN_NTAB_BATCH : SA_SPAN_NTAB);
        if (c == 0) {
            if (few) hipLaunchKernelGGL((sa_k_span_doc_fused_multi<2, SA_SPAN_NTAB_BATCH>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
            else hipLaunchKernelGGL((sa_k_span_doc_fused_multi<2, SA_SPAN_NTAB>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
        } else if (c == 1) {
            if (few) hipLaunchKernelGGL((sa_k_span_doc_fused_multi<3, SA_SPAN_NTAB_BATCH>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
            else hipLaunchKernelGGL((sa_k_span_doc_fused_multi<3, SA_SPAN_NTAB>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
        } else {
            if (few) hipLaunchKernelGGL((sa_k_span_doc_fused_multi<4, SA_SPAN_NTAB_BATCH>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
            else hipLaunchKernelGGL((sa_k_span_doc_fused_multi<4, SA_SPAN_NTAB>), dim3(dblocks[c]), dim3(SA_SPAN_FT), pad, st, jc, wc);
        }
    }
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// ---- kernel-level mirror of span_search(posns, lengths, phrase_freqs, slop, ...) -------------------
// (reference roaringish/spans.pyx:322-330): the terms' candidate words back to back in `posns`,
// term t = posns[lengths[t] : lengths[t + 1]], as phrase/spans.py:171-187 hands them over after
// _intersect_all.  Runs stage 2 (the span machine above) on them and returns the documents whose
// count it raised, ascending, with the increment -- what the reference adds into its Counter.
struct NonZeroCounts {
    const u32* counts; u64* docs; u64* incr;
    __device__ __forceinline__ bool flag(u32 i) const { return counts[i] != 0; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { docs[pos] = i; incr[pos] = counts[i]; }
};

extern "C" int sa_span_search(const uint64_t* posns, const uint64_t* lengths, int n_terms, uint64_t slop,
                              uint64_t* docs_out, uint64_t* counts_out, int64_t* n_out) {
    SA_ARG(lengths && n_out, "null argument");
    *n_out = 0;
    SA_ARG(n_terms >= 1, "no terms");
    if (n_terms > SA_SPAN_MAX_TERMS) { sa_set_error("slop phrases support at most %d terms", SA_SPAN_MAX_TERMS); return SA_ERR_UNSUPPORTED; }
    const int T = n_terms;
    for (int t = 0; t < T; t++) SA_ARG(lengths[t] <= lengths[t + 1], "lengths must be non-decreasing");
    const u64 total = lengths[T] - lengths[0];
    SA_ARG(total < 0xFFFFF000ull, "array too long");
    if (lengths[1] == lengths[0]) return SA_OK;                       // term 0 drives the walk (spans.pyx:224)
    SA_ARG(posns && docs_out && counts_out, "null argument");
    // doc ids are bounded by the largest key of any term (each term's words are sorted)
    u64 n_docs = 0;
    for (int t = 0; t < T; t++)
        if (lengths[t + 1] > lengths[t]) {
            const u64 key = posns[lengths[t + 1] - 1] >> SA_KEY_SHIFT;
            if (key + 1 > n_docs) n_docs = key + 1;
        }
    struct Bufs {
        std::vector<void*> p;
        ~Bufs() { for (void* x : p) hipFree(x); }
        int take(void** out, size_t bytes) {
            SA_HIP(hipMalloc(out, bytes ? bytes : 8));
            p.push_back(*out);
            return SA_OK;
        }
    } bufs;
    hipStream_t st = 0;
    u64* d_words; u32 *d_counts, *d_cnt, *d_chunks, *d_heads;
    SA_TRY(bufs.take((void**)&d_words, (size_t)total * 8));
    SA_TRY(bufs.take((void**)&d_counts, ((size_t)n_docs + 1) * 4));
    SA_TRY(bufs.take((void**)&d_cnt, 4 * SA_SPAN_MAX_TERMS * 4));
    SA_TRY(bufs.take((void**)&d_heads, ((size_t)total + T) * 4));
    const size_t max_n = (size_t)(total > n_docs ? total : n_docs);
    SA_TRY(bufs.take((void**)&d_chunks, ((size_t)sa_compact_chunks((u32)max_n + 1) + 8) * 4));
    SA_HIP(hipMemcpyAsync(d_words, posns + lengths[0], (size_t)total * 8, hipMemcpyHostToDevice, st));
    SA_HIP(hipMemsetAsync(d_counts, 0, ((size_t)n_docs + 1) * 4, st));
    u32 h_cnt[4 * SA_SPAN_MAX_TERMS];
    memset(h_cnt, 0, sizeof(h_cnt));
    for (int t = 0; t < T; t++) h_cnt[t] = (u32)(lengths[t + 1] - lengths[t]);
    SA_HIP(hipMemcpyAsync(d_cnt, h_cnt, sizeof(h_cnt), hipMemcpyHostToDevice, st));
    u32 G = (h_cnt[0] + 63u) & ~63u;
    u32 g_max = SA_SPAN_THREADS;
    {
        const sa_options_t o = sa_options_for_new_handle(nullptr);      // (a Part-1 mirror: no index handle; the calling thread's options)
        if (sa_opt(o.span_threads, 0) >= 64) g_max = ((u32)o.span_threads + 63u) & ~63u;
    }
    if (G > g_max) G = g_max;
    SpanEnt* ents; u64* col;
    SA_TRY(bufs.take((void**)&ents, (size_t)G * SA_NSPANS * sizeof(SpanEnt)));
    SA_TRY(bufs.take((void**)&col, (size_t)G * SA_NSPANS * sizeof(u64)));
    SpanMachineParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.T = T; mp.slop = (u32)(slop > 0xFFFFFFFFull ? 0xFFFFFFFFull : slop); mp.ents = ents; mp.col = col;
    mp.counts = d_counts; mp.n_docs = n_docs; mp.n_threads = G;
    for (int t = 0; t < T; t++) {
        const u64 off = lengths[t] - lengths[0];
        mp.cand[t] = d_words + off; mp.n_cand[t] = d_cnt + t;
        mp.heads[t] = d_heads + off + t; mp.n_heads[t] = d_cnt + SA_SPAN_MAX_TERMS + t;
        if (h_cnt[t] == 0) continue;
        DocHeads dh;
        dh.words = d_words + off; dh.out = d_heads + off + t;
        sa_compact(dh, (const u32*)nullptr, h_cnt[t], d_chunks, d_cnt + SA_SPAN_MAX_TERMS + t, st);
    }
    hipLaunchKernelGGL(sa_k_span_machine, dim3(G / 64), dim3(64), 0, st, mp);
    u64 *d_docs, *d_incr;
    const size_t n_hit_max = (size_t)(n_docs < total ? n_docs : total) + 1;   // one hit per document group at most
    SA_TRY(bufs.take((void**)&d_docs, n_hit_max * 8));
    SA_TRY(bufs.take((void**)&d_incr, n_hit_max * 8));
    NonZeroCounts nz; nz.counts = d_counts; nz.docs = d_docs; nz.incr = d_incr;
    sa_compact(nz, (const u32*)nullptr, (u32)n_docs, d_chunks, d_cnt + 3 * SA_SPAN_MAX_TERMS, st);
    SA_HIP(hipGetLastError());
    u32 n = 0;
    SA_HIP(hipMemcpyAsync(&n, d_cnt + 3 * SA_SPAN_MAX_TERMS, 4, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    if (n) {
        SA_HIP(hipMemcpy(docs_out, d_docs, (size_t)n * 8, hipMemcpyDeviceToHost));
        SA_HIP(hipMemcpy(counts_out, d_incr, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    *n_out = n;
    return SA_OK;
}

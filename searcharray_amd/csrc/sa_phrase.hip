// sa_phrase.hip -- exact-phrase (slop = 0) match counts over the roaringish positional words.
//
// Two device paths, both producing the dense float32 phrase-frequency vector the reference's
// PosnBitArray.phrase_freqs returns (reference searcharray/phrase/middle_out.py:418-441):
//
//  * the GENERAL chain: a data-parallel restatement of the reference's bigram chain
//      compute_phrase_freqs / l2r / r2l / middle-out      middle_out.py:96-168
//      bigram_freqs and its helpers                       phrase/bigram_freqs.py:48-307
//      intersect_with_adjacents / intersect / merge       roaringish/intersect.pyx:213-275, merge.pyx:54-92
//    with identical observable semantics, including the same-term rule (bigram_freqs.py:48-101),
//    zero-payload continuation words and the T>=5 middle-out quirk.
//    The serial galloping two-pointer of the reference becomes: per lhs word, a lower-bound probe
//    into rhs on the 46-bit header (headers are unique within a term, so the match set is the
//    same), then stable stream compaction with device-resident lengths -- the whole chain is
//    enqueued on one stream with no host round trip.
//    Because only phrase_freqs[ids] = counts survives (ids-intersection + np.minimum over bigram
//    steps, middle_out.py:73-93), per-step counts are scatter-added into a dense u32 vector and the
//    running result is an elementwise min -- equal to the reference's sorted (ids, counts) algebra
//    (absent doc == count 0), without sort_merge_counts / popcount_reduce_at round trips.
//
//  * the FUSED kernel for phrases of pairwise-distinct terms: anchored on the rarest term, each
//    anchor word probes the other terms' words at headers h-1, h, h+1, aligns their 18-bit bitmaps
//    by the phrase offsets and counts popcount(AND) -- one pass, every word of the anchor read once.
//    For distinct terms the reference's chain equals the exact count of positions p with
//    t0@p, t1@p+1, ... (and, for the middle-out plan, the min of the two sub-phrase counts).
#include "sa_index.hpp"
#include <vector>
#include "sa_scan.hpp"
#include "sa_phrase_dev.hpp"
#include "../../include/searcharray_hip.h"

#include <stdlib.h>

#define SA_NONE 0xFFFFFFFFu
// (no limit on the terms of a phrase: the general chain takes them one bigram at a time, like the reference's
//  compute_phrase_freqs, middle_out.py:73-168; the fused kernel takes sub-phrases of up to SA_MAX_FUSED)
#define SA_MAX_FUSED 18

enum { CONT_LHS = 0, CONT_RHS = 1 };

// ---------------------------------------------------------------------------------------
// general chain kernels
// ---------------------------------------------------------------------------------------

// per lhs word: index of the rhs word with the same header / with header + 1, or SA_NONE
// (intersect_with_adjacents, intersect.pyx:213-275); clears *same when an inner pair differs
// (the same-term test `np.all(lhs_int == rhs_int)`, bigram_freqs.py:140).
__global__ void __launch_bounds__(256)
sa_k_probe(const u64* __restrict__ lhs, const u32* __restrict__ nl_dev, const u64* __restrict__ rhs,
           const u32* __restrict__ nr_dev, u32* __restrict__ jin, u32* __restrict__ jadj, u32* __restrict__ same) {
    const u32 nl = *nl_dev, nr = *nr_dev;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
        const u64 l = lhs[i];
        const u64 h = l & SA_HEADER_MASK;
        const u32 j = sa_lower_bound(rhs, 0, nr, h, SA_HEADER_MASK);
        u32 ji = SA_NONE, ja = SA_NONE;
        if (j < nr && (rhs[j] & SA_HEADER_MASK) == h) {
            ji = j;
            if (rhs[j] != l) *same = 0u;
        }
        const u64 h2 = h + (1ull << SA_LSB_BITS);
        const u32 j2 = sa_lower_bound(rhs, j, nr, h2, SA_HEADER_MASK);
        if (j2 < nr && (rhs[j2] & SA_HEADER_MASK) == h2) ja = j2;
        jin[i] = ji;
        jadj[i] = ja;
    }
}

// inner matches: bigram_freqs.py:104-155 (and the same-term branch :65-101)
struct InnerBigram {
    const u64* lhs; const u64* rhs; const u32* jin; const u32* same;
    u64* next_inner; u32* step; int cont;
    __device__ __forceinline__ bool flag(u32 i) const { return jin[i] != SA_NONE; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 l = lhs[i], r = rhs[jin[i]];
        u32 count;
        u64 nxt;
        if (*same) {
            const u64 ov = l & (r << 1);
            const int adj = __popcll(ov & SA_LSB_MASK);
            const int cons = __popcll((ov & (ov << 1)) & SA_LSB_MASK);
            count = (u32)(adj - ((cons + 1) >> 1));
            const u64 msbs = l & ~SA_LSB_MASK;
            nxt = (cont == CONT_RHS) ? ((((r << 1) & r) & SA_LSB_MASK) | msbs)
                                     : (msbs | ((l & (l >> 1)) & SA_LSB_MASK));
        } else {
            const u64 ov = (l & SA_LSB_MASK) & ((r & SA_LSB_MASK) >> 1);
            count = (u32)__popcll(ov);
            nxt = (cont == CONT_RHS) ? (((ov << 1) & SA_LSB_MASK) | (r & SA_HEADER_MASK))
                                     : (ov | (l & SA_HEADER_MASK));
        }
        next_inner[pos] = nxt;
        if (count) atomicAdd(&step[l >> SA_KEY_SHIFT], count);
    }
};

// cross-word matches: lhs bit 17 & rhs bit 0 at header + 1 (bigram_freqs.py:158-188)
struct AdjBigram {
    const u64* lhs; const u64* rhs; const u32* jadj;
    u64* next_adj; u32* step; int cont;
    __device__ __forceinline__ bool flag(u32 i) const {
        const u32 j = jadj[i];
        return j != SA_NONE && (lhs[i] & SA_UPPER_BIT) != 0 && (rhs[j] & 1ull) != 0;
    }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const {
        const u64 l = lhs[i], r = rhs[jadj[i]];
        next_adj[pos] = (cont == CONT_RHS) ? ((r & SA_HEADER_MASK) | 1ull) : ((l & SA_HEADER_MASK) | SA_UPPER_BIT);
        atomicAdd(&step[l >> SA_KEY_SHIFT], 1u);
    }
};

// _set_adjbit_at_header (bigram_freqs.py:191-210), part 1: OR the adjacency bit into the inner
// continuation word with the same header and mark that adjacent word as absorbed.
__global__ void __launch_bounds__(256)
sa_k_absorb_adj(u64* __restrict__ next_inner, const u32* __restrict__ ni_dev, const u64* __restrict__ next_adj,
                const u32* __restrict__ na_dev, u32* __restrict__ absorbed, int cont) {
    const u32 ni = *ni_dev, na = *na_dev;
    const u64 bit = (cont == CONT_RHS) ? 1ull : SA_UPPER_BIT;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < na; j += gridDim.x * blockDim.x) {
        const u64 h = next_adj[j] & SA_HEADER_MASK;
        const u32 idx = sa_lower_bound(next_inner, 0, ni, h, SA_HEADER_MASK);
        if (idx < ni && (next_inner[idx] & SA_HEADER_MASK) == h) {
            atomicOr((unsigned long long*)&next_inner[idx], (unsigned long long)bit);
            absorbed[j] = 1u;
        } else {
            absorbed[j] = 0u;
        }
    }
}

struct KeepUnabsorbed {
    const u64* next_adj; const u32* absorbed; u64* out;
    __device__ __forceinline__ bool flag(u32 j) const { return absorbed[j] == 0u; }
    __device__ __forceinline__ void emit(u32 j, u32 pos) const { out[pos] = next_adj[j]; }
};

// part 2: merge(next_inner, unabsorbed adj) (merge.pyx:54-92; the two sets have disjoint headers,
// so each element's output slot is its own rank plus its rank in the other array).
__global__ void __launch_bounds__(256)
sa_k_merge_by_rank(const u64* __restrict__ a, const u32* __restrict__ na_dev, const u64* __restrict__ b,
                   const u32* __restrict__ nb_dev, u64* __restrict__ out, u32* __restrict__ nout_dev) {
    const u32 na = *na_dev, nb = *nb_dev;
    const u32 total = na + nb;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < na) {
            const u64 x = a[i];
            out[i + sa_lower_bound(b, 0, nb, x, ~0ull)] = x;
        } else {
            const u32 j = i - na;
            const u64 x = b[j];
            out[j + sa_lower_bound(a, 0, na, x, ~0ull)] = x;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *nout_dev = total;
}

// running = min(running, step) and clear step (the ids-intersect + np.minimum of
// _intersect_bigram_matches, middle_out.py:73-93, in dense form)
__global__ void __launch_bounds__(256)
sa_k_min_step(float* __restrict__ running, u32* __restrict__ step, u64 n, int first) {
    for (u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (u64)gridDim.x * blockDim.x) {
        const float c = (float)step[d];
        step[d] = 0u;
        running[d] = first ? c : (c < running[d] ? c : running[d]);
    }
}

__global__ void __launch_bounds__(256)
sa_k_min2(float* __restrict__ a, const float* __restrict__ b, u64 n) {
    for (u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (u64)gridDim.x * blockDim.x)
        a[d] = b[d] < a[d] ? b[d] : a[d];
}

// ---------------------------------------------------------------------------------------
// fused kernel (pairwise-distinct terms)
// ---------------------------------------------------------------------------------------
struct FusedPhraseParams {
    const u64* ptr[SA_MAX_FUSED];    // each term's (possibly position-filtered) words
    u32 len[SA_MAX_FUSED];
    const u32* dd[SA_MAX_FUSED];     // doc directory row of the term (whole, unfiltered list), or null
    int T, anchor;
    u32* step;                   // dense per-doc match counts (u32, atomically accumulated) -- or
    float* fcounts;              //   the dense float result itself (one sub-phrase: no minimum to take; small integers,
                                 //   exact in any order)
};

__global__ void __launch_bounds__(256) sa_k_phrase_fused(const FusedPhraseParams p) {
    const u64* anc = p.ptr[p.anchor];
    const u32 na = p.len[p.anchor];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += gridDim.x * blockDim.x) {
        const u64 w = anc[i];
        const u64 m = sa_phrase_anchor_mask_win(w, p.T, p.anchor, [&](int t, u64 h, bool want_prev, bool want_next) -> u64 {
            const u64* a = p.ptr[t];
            const u32 n = p.len[t];
            if (p.dd[t]) return sa_window_docdir(a, n, p.dd[t], h, want_prev, want_next);   // uniform per term
            const u64 delta = 1ull << SA_LSB_BITS;
            u32 hint = 0;
            u64 win = 0;
            if (want_prev) win |= sa_payload_at(a, n, h - delta, hint);
            win |= sa_payload_at(a, n, h, hint) << 18;
            if (want_next) win |= sa_payload_at(a, n, h + delta, hint) << 36;
            return win;
        });
        if (m) {
            if (p.fcounts) unsafeAtomicAdd(&p.fcounts[w >> SA_KEY_SHIFT], (float)__popcll(m));
            else atomicAdd(&p.step[w >> SA_KEY_SHIFT], (u32)__popcll(m));
        }
    }
}

// ---------------------------------------------------------------------------------------
// the chain per DOCUMENT (phrases with repeated terms, frequent terms)
// ---------------------------------------------------------------------------------------
// Every operation of a bigram step -- intersect on the header, the header + 1 adjacency, the continuation words, the
// adjacency bit absorbed into the inner word of the same header, the merge -- relates words of ONE document, unless a
// word sits in a document's last 18-position block (header + 1 is then the next document's block 0).  So the whole chain
// can run per document: one thread per document of the rarest term (or per document, when that list is about as long as
// the collection) finds the document's words of every term -- through the term's doc directory row, or by a search on
// the key -- and runs the general chain's steps on them in LDS: the same formulas as InnerBigram / AdjBigram /
// sa_k_absorb_adj / sa_k_merge_by_rank above, the per-step counts summed over the document, the running minimum over
// the steps (_intersect_bigram_matches: absent = 0) -- one launch instead of ~20 per bigram.
// Two things are checked, and the general chain takes the phrase if a check fails:
//   * the reference's same-term test, `np.all(lhs_int == rhs_int)` over ALL matched pairs of a step (bigram_freqs.py:140),
//     is not local.  It is predicted -- true exactly for a chain's first step over twice the same term -- and every step
//     the prediction calls "different" records whether it had a matched pair and whether one differed: "pairs, none
//     different" (degenerate data) is a failed prediction;
//   * a document with more words of a term, or of a continuation, than a thread holds (four, then eight), or with a word
//     in a last block.
#define SA_PD_MAXT 32                    // terms of a phrase this route takes
#define SA_PD_THREADS 128

struct PhraseDocParams {
    const u64* words[SA_PD_MAXT];        // by phrase position
    const u32* dd[SA_PD_MAXT];
    u32 len[SA_PD_MAXT];
    int T;
    int plan;                            // 0: left to right, 1: right to left, 2: middle out at `shortest`
    int shortest;
    u32 same_mask;                       // bit s: step s (in execution order) is predicted "same term"
    int anchor;                          // phrase position of the rarest term: its documents are the ones looked at; -1: every
                                         //   document (a thread per doc: cheaper when the rarest list is about as long as that)
    u64 n_docs;
    float* counts;
    u32* flags;                          // [0] a document did not fit; [1 + 2 s] step s had a matched pair, [2 + 2 s] a pair that differed
};

// thread-private arrays in LDS: element i of this thread at a[i * SA_PD_THREADS]
struct PdArr {
    u64* a;
    __device__ __forceinline__ u64& operator[](u32 i) const { return a[i * SA_PD_THREADS]; }
};

// index of the document's first word in the list at phrase position t (through the directory row if the term has one,
// else a search on the 28-bit key), or SA_DD_ABSENT
__device__ __forceinline__ u32 sa_pd_first(const PhraseDocParams& p, const int t, const u64 doc) {
    if (p.dd[t]) return p.dd[t][doc];
    const u32 j = sa_lower_bound(p.words[t], 0, p.len[t], doc << SA_KEY_SHIFT, SA_KEY_MASK);
    return (j < p.len[t] && (p.words[t][j] >> SA_KEY_SHIFT) == doc) ? j : SA_DD_ABSENT;
}

// the document's words of the term at phrase position t -> dst; false: more than CAP words, or a word in the document's
// last 18-position block (its header + 1 is the next document's block 0: the chain is not local to the document then).  (All requested at once: one
// after the other, each waiting for the last to see whether the document goes on, they are a chain of round trips.)
template <int CAP>
__device__ __forceinline__ bool sa_pd_load(const PhraseDocParams& p, const int t, const u64 doc, const PdArr& dst, u32* n_out) {
    u32 n = 0;
    const u32 j0 = sa_pd_first(p, t, doc);
    if (j0 != SA_DD_ABSENT) {
        const u64* const w = p.words[t];
        const u32 len = p.len[t];
        u64 x[CAP + 1];
#pragma unroll
        for (int q = 0; q <= CAP; q++) x[q] = j0 + (u32)q < len ? w[j0 + (u32)q] : ~0ull;
        bool run = true;
#pragma unroll
        for (int q = 0; q <= CAP; q++) {
            run = run && (x[q] >> SA_KEY_SHIFT) == doc;
            if (run && q < CAP) dst[(u32)q] = x[q];
            if (run && ((x[q] >> SA_LSB_BITS) & SA_LSB_MASK) == SA_LSB_MASK) return false;
            n += run ? 1u : 0u;
        }
        if (n > CAP) return false;
    }
    *n_out = n;
    return true;
}

// One bigram step on the document's words: lhs (nl) x rhs (nr) -> the continuation in `out` (*nout), the step's count.
// ni / na: scratch.  Returns false when the continuation does not fit.
template <int CAP>
__device__ __forceinline__ bool sa_pd_step(const PdArr& lhs, const u32 nl, const PdArr& rhs, const u32 nr, const int cont, const bool same,
                                           const PdArr& ni_arr, const PdArr& na_arr, const PdArr& out, u32* nout, u32* count_out,
                                           bool* any_pair, bool* any_diff) {
    const u64 unit = 1ull << SA_LSB_BITS;
    u32 ni = 0, na = 0, count = 0;
    for (u32 i = 0; i < nl; i++) {
        const u64 l = lhs[i];
        const u64 h = l & SA_HEADER_MASK;
        u32 jin = SA_NONE, jadj = SA_NONE;
        for (u32 j = 0; j < nr; j++) {                          // (sa_k_probe: headers are unique within a list)
            const u64 rh = rhs[j] & SA_HEADER_MASK;
            if (rh == h) jin = j;
            if (rh == h + unit) jadj = j;
        }
        if (jin != SA_NONE) {                                   // InnerBigram
            const u64 r = rhs[jin];
            *any_pair = true;
            if (r != l) *any_diff = true;
            u64 nxt;
            if (same) {
                const u64 ov = l & (r << 1);
                const int adj = __popcll(ov & SA_LSB_MASK);
                const int cons = __popcll((ov & (ov << 1)) & SA_LSB_MASK);
                count += (u32)(adj - ((cons + 1) >> 1));
                const u64 msbs = l & ~SA_LSB_MASK;
                nxt = (cont == CONT_RHS) ? ((((r << 1) & r) & SA_LSB_MASK) | msbs) : (msbs | ((l & (l >> 1)) & SA_LSB_MASK));
            } else {
                const u64 ov = (l & SA_LSB_MASK) & ((r & SA_LSB_MASK) >> 1);
                count += (u32)__popcll(ov);
                nxt = (cont == CONT_RHS) ? (((ov << 1) & SA_LSB_MASK) | (r & SA_HEADER_MASK)) : (ov | (l & SA_HEADER_MASK));
            }
            ni_arr[ni++] = nxt;                                 // (ni <= nl <= CAP)
        }
        if (jadj != SA_NONE && (l & SA_UPPER_BIT) != 0 && (rhs[jadj] & 1ull) != 0) {      // AdjBigram
            na_arr[na++] = (cont == CONT_RHS) ? ((rhs[jadj] & SA_HEADER_MASK) | 1ull) : ((l & SA_HEADER_MASK) | SA_UPPER_BIT);
            count += 1u;
        }
    }
    // sa_k_absorb_adj: the adjacency bit into the inner continuation word of the same header; the others stay words of their own
    const u64 bit = (cont == CONT_RHS) ? 1ull : SA_UPPER_BIT;
    u32 nu = 0;
    for (u32 a = 0; a < na; a++) {
        const u64 x = na_arr[a];
        const u64 ha = x & SA_HEADER_MASK;
        bool found = false;
        for (u32 i = 0; i < ni; i++)
            if ((ni_arr[i] & SA_HEADER_MASK) == ha) { ni_arr[i] = ni_arr[i] | bit; found = true; }
        if (!found) na_arr[nu++] = x;                           // (nu <= a: in place)
    }
    // sa_k_merge_by_rank: both ascending, headers disjoint
    if (ni + nu > CAP) return false;
    u32 i = 0, a = 0, n = 0;
    while (i < ni || a < nu) {
        const bool take_inner = a >= nu || (i < ni && ni_arr[i] < na_arr[a]);
        out[n++] = take_inner ? ni_arr[i] : na_arr[a];
        i += take_inner ? 1u : 0u;
        a += take_inner ? 0u : 1u;
    }
    *nout = n;
    *count_out = count;
    return true;
}

template <int CAP>
__global__ void __launch_bounds__(SA_PD_THREADS) sa_k_phrase_docs(const PhraseDocParams p) {
    __shared__ u64 s_arr[4 * CAP * SA_PD_THREADS];
    __shared__ u32 s_flags[2 + 2 * SA_PD_MAXT];
    if (threadIdx.x < 2 + 2 * SA_PD_MAXT) s_flags[threadIdx.x] = 0;
    __syncthreads();
    const PdArr A{s_arr + threadIdx.x}, B{s_arr + CAP * SA_PD_THREADS + threadIdx.x},
                NI{s_arr + 2 * CAP * SA_PD_THREADS + threadIdx.x}, NA{s_arr + 3 * CAP * SA_PD_THREADS + threadIdx.x};
    // one thread per document of the rarest term -- per word of its list that opens a document -- or per document
    const u64 i = (u64)blockIdx.x * SA_PD_THREADS + threadIdx.x;
    bool opener = false;
    u64 doc = 0;
    if (p.anchor < 0) {
        doc = i;
        opener = doc < p.n_docs;
    } else if (i < p.len[p.anchor]) {
        const u64* const aw = p.words[p.anchor];
        doc = aw[i] >> SA_KEY_SHIFT;
        opener = (i == 0 || (aw[i - 1] >> SA_KEY_SHIFT) != doc) && doc < p.n_docs;
    }
    if (opener) {
        bool present = true;
        for (int t = 0; t < p.T && present; t++) present = t == p.anchor || sa_pd_first(p, t, doc) != SA_DD_ABSENT;
        // (a term repeated in the phrase is looked up once per position: the second lookup hits the same lines)
        float result = 0.f;                                     // (a term without a word here: that step counts 0, and so does the minimum)
        bool fits = true;
        if (present) {
            u32 best = 0xFFFFFFFFu;
            int step = 0;
            // chain over positions [a, b): left to right (the continuation replaces the left operand) or right to left
            auto chain = [&](const int a, const int b, const bool l2r) {
                u32 nl = 0, nr = 0;
                if (l2r) fits = fits && sa_pd_load<CAP>(p, a, doc, A, &nl);
                else fits = fits && sa_pd_load<CAP>(p, b - 1, doc, B, &nr);
                for (int k = 1; k < b - a && fits; k++) {
                    if (l2r) fits = sa_pd_load<CAP>(p, a + k, doc, B, &nr);
                    else fits = sa_pd_load<CAP>(p, b - 1 - k, doc, A, &nl);
                    if (!fits) break;
                    const bool same = (p.same_mask >> step) & 1u;
                    bool any_pair = false, any_diff = false;
                    u32 nout = 0, cnt = 0;
                    // the continuation goes where the operand it replaces was (sa_pd_step reads both operands before it writes)
                    fits = sa_pd_step<CAP>(A, nl, B, nr, l2r ? CONT_RHS : CONT_LHS, same, NI, NA, l2r ? A : B, &nout, &cnt, &any_pair, &any_diff);
                    if (!fits) break;
                    if (l2r) nl = nout; else nr = nout;
                    best = cnt < best ? cnt : best;
                    if (any_pair) s_flags[1 + 2 * step] = 1u;
                    if (any_diff) s_flags[2 + 2 * step] = 1u;
                    step++;
                }
            };
            if (p.plan == 0) chain(0, p.T, true);
            else if (p.plan == 1) chain(0, p.T, false);
            else { chain(0, p.shortest, true); chain(p.shortest, p.T, false); }
            result = best == 0xFFFFFFFFu ? 0.f : (float)best;
        }
        if (!fits) s_flags[0] = 1u;
        if (result != 0.f || p.anchor < 0) p.counts[doc] = result;       // (anchored: the vector is cleared before the launch)
    }
    __syncthreads();
    if (threadIdx.x < 2 + 2 * SA_PD_MAXT && s_flags[threadIdx.x]) p.flags[threadIdx.x] = 1u;
}

// the checks' flags to a page-locked host array (a store from the device instead of a copy engine round trip); the device
// copy is left cleared for the next use
__global__ void __launch_bounds__(128) sa_k_flags_out(u32* __restrict__ d_flags, u32* __restrict__ h_flags, u32 n) {
    if (threadIdx.x < n) { h_flags[threadIdx.x] = d_flags[threadIdx.x]; d_flags[threadIdx.x] = 0u; }
}

// tf (phrase counts) -> BM25 in place; reference bm25.pyx:11-25 over the dense phrase_freqs
__global__ void __launch_bounds__(256)
sa_k_bm25_from_tf(float* __restrict__ tf, const float* __restrict__ dl, float avgdl, float idf, float k1, float b, u64 n) {
    const float one_minus_b = 1.0f - b;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const float t = tf[i];
        const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl[i], avgdl))));
        tf[i] = __fmul_rn(__fdiv_rn(t, __fadd_rn(t, norm)), idf);
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    template <class T> T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > cap) return nullptr;
        T* p = (T*)(base + used);
        used += bytes;
        return p;
    }
};

struct DArr {              // device array of roaringish words with a device-resident length
    const u64* data;
    const u32* n_dev;
    u32 bound;             // host-known upper bound of the length
};

static inline u32 sa_grid_for(u64 n) {
    const u64 g = (n + 255) / 256;
    return (u32)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

struct StepScratch {
    u32 *jin, *jadj, *absorbed, *chunks, *cnt;   // cnt: [0]=ni [1]=na [2]=nu [3]=same
    u64 *next_inner, *next_adj, *ua;
    u32 cap;
};

// One bigram step on the stream: updates `step` (dense counts) and writes the continuation
// array into out (length -> out_n_dev).
static int sa_bigram_step(hipStream_t st, const DArr& lhs, const DArr& rhs, int cont, StepScratch& s,
                          u32* step, u64* out, u32* out_n_dev) {
    if (lhs.bound > s.cap) { sa_set_error("internal: phrase scratch too small"); return SA_ERR_STATE; }
    SA_HIP(hipMemsetAsync(s.cnt, 0, 3 * sizeof(u32), st));
    const u32 one = 1u;
    SA_HIP(hipMemcpyAsync(s.cnt + 3, &one, sizeof(u32), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sa_k_probe, dim3(sa_grid_for(lhs.bound)), dim3(256), 0, st, lhs.data, lhs.n_dev, rhs.data,
                       rhs.n_dev, s.jin, s.jadj, s.cnt + 3);
    InnerBigram ib;
    ib.lhs = lhs.data; ib.rhs = rhs.data; ib.jin = s.jin; ib.same = s.cnt + 3;
    ib.next_inner = s.next_inner; ib.step = step; ib.cont = cont;
    sa_compact(ib, lhs.n_dev, lhs.bound, s.chunks, s.cnt + 0, st);
    AdjBigram ab;
    ab.lhs = lhs.data; ab.rhs = rhs.data; ab.jadj = s.jadj; ab.next_adj = s.next_adj; ab.step = step; ab.cont = cont;
    sa_compact(ab, lhs.n_dev, lhs.bound, s.chunks, s.cnt + 1, st);
    hipLaunchKernelGGL(sa_k_absorb_adj, dim3(sa_grid_for(lhs.bound)), dim3(256), 0, st, s.next_inner, s.cnt + 0,
                       s.next_adj, s.cnt + 1, s.absorbed, cont);
    KeepUnabsorbed ku;
    ku.next_adj = s.next_adj; ku.absorbed = s.absorbed; ku.out = s.ua;
    sa_compact(ku, s.cnt + 1, lhs.bound, s.chunks, s.cnt + 2, st);
    hipLaunchKernelGGL(sa_k_merge_by_rank, dim3(sa_grid_for(2ull * lhs.bound)), dim3(256), 0, st, s.next_inner,
                       s.cnt + 0, s.ua, s.cnt + 2, out, out_n_dev);
    return SA_OK;
}

// min_posn / max_posn filter: the reference slices every term's words with payload_slice before
// matching (middle_out.py:434-437, roaringish.py:266-282), comparing the UNSHIFTED
// (word & msb_mask) with min_posn // 18 and max_posn // 18 -- kept as is.
struct PosnSlice {
    const u64* arr; u64 lo, hi; u64* out;
    __device__ __forceinline__ bool flag(u32 i) const { const u64 v = arr[i] & 0x0000000FFFFC0000ull; return v >= lo && v <= hi; }
    __device__ __forceinline__ void emit(u32 i, u32 pos) const { out[pos] = arr[i]; }
};

int sa_posn_filter_bounds(int64_t min_posn, int64_t max_posn, PosnFilter* f) {
    f->active = (min_posn >= 0 || max_posn >= 0);
    if (min_posn >= 0 && min_posn % SA_LSB_BITS != 0) {
        sa_set_error("min_payload must be a multiple of %d", SA_LSB_BITS);                 // roaringish.py:270-271
        return SA_ERR_ARG;
    }
    if (max_posn >= 0 && max_posn % SA_LSB_BITS != SA_LSB_BITS - 1) {
        sa_set_error("max_payload must be a multiple of %d - 1", SA_LSB_BITS);             // roaringish.py:272-273
        return SA_ERR_ARG;
    }
    f->lo = min_posn >= 0 ? (u64)min_posn / SA_LSB_BITS : 0;
    f->hi = max_posn >= 0 ? (u64)max_posn / SA_LSB_BITS : 0xFFFFFFFFFFFFFFFFull / SA_LSB_BITS;
    return SA_OK;
}

// Apply the filter to every term: compacts into `bufs[t]` and reads the new lengths back.
int sa_posn_filter_terms(sa_index* ix, const PosnFilter& f, int T, const u64** ptrs, u32* lens, u64** bufs,
                         u32* d_counts, u32* d_chunks) {
    hipStream_t st = ix->stream;
    for (int t = 0; t < T; t++) {
        if (lens[t] == 0) continue;
        PosnSlice ps; ps.arr = ptrs[t]; ps.lo = f.lo; ps.hi = f.hi; ps.out = bufs[t];
        sa_compact(ps, (const u32*)nullptr, lens[t], d_chunks, d_counts + t, st);
        ptrs[t] = bufs[t];
    }
    std::vector<u32> h((size_t)T);
    SA_HIP(hipMemcpyAsync(h.data(), d_counts, (size_t)T * sizeof(u32), hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < T; t++) if (lens[t]) lens[t] = h[t];
    return SA_OK;
}

// dense phrase counts of terms[0..T) into d_running (float[n_docs]); uses the index scratch.
// mode: 0 auto, 1 general chain, 2 fused.
static int sa_phrase_counts_device(sa_index* ix, const u32* terms, int T, int mode, const PosnFilter& filt,
                                   float** d_running_out) {
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    std::vector<u32> lens_v((size_t)T);
    std::vector<const u64*> ptrs_v((size_t)T);
    u32* const lens = lens_v.data();
    const u64** const ptrs = ptrs_v.data();
    u32 maxlen = 0;
    size_t total_len = 0;
    bool known = true, distinct = true;
    for (int t = 0; t < T; t++) {
        if (terms[t] >= ix->n_terms) { known = false; lens[t] = 0; ptrs[t] = ix->d_words; continue; }
        const u64 off = ix->h_term_off[terms[t]];
        ptrs[t] = ix->d_words + off;
        lens[t] = (u32)(ix->h_term_off[terms[t] + 1] - off);
        maxlen = lens[t] > maxlen ? lens[t] : maxlen;
        total_len += lens[t];
        for (int u = 0; u < t; u++) if (terms[u] == terms[t]) distinct = false;
    }
    const size_t M = (size_t)maxlen + 64;
    const size_t chunk_words = sa_compact_chunks((u32)(2 * M)) + 8;
    const size_t filt_bytes = filt.active ? (total_len + 64 * (size_t)T) * 8 : 0;
    const size_t need = (N + 64) * 12 + M * (3 * 4 + 3 * 8 + 2 * 8 * 2) + chunk_words * 4 + filt_bytes + (size_t)T * 4 + 16 * 1024;
    void* scratch;
    SA_TRY(sa_index_scratch(ix, need, &scratch));
    Arena ar;
    ar.base = (char*)scratch; ar.cap = need;
    float* running = ar.take<float>(N + 1);
    float* running2 = ar.take<float>(N + 1);
    u32* step = ar.take<u32>(N + 1);
    u32* lens_dev = ar.take<u32>((size_t)T + 8);              // [t] term lengths, then ping-pong counters
    StepScratch s;
    s.cap = (u32)M;
    s.jin = ar.take<u32>(M); s.jadj = ar.take<u32>(M); s.absorbed = ar.take<u32>(M);
    s.chunks = ar.take<u32>(chunk_words); s.cnt = ar.take<u32>(8);
    s.next_inner = ar.take<u64>(M); s.next_adj = ar.take<u64>(M); s.ua = ar.take<u64>(M);
    u64* pp[2] = {ar.take<u64>(2 * M), ar.take<u64>(2 * M)};
    if (!pp[1] || !s.ua || !running2) { sa_set_error("internal: phrase arena exhausted"); return SA_ERR_STATE; }
    *d_running_out = running;
    if (!known || N == 0) {                                     // TermMissingError -> zeros (postings.py:705-708)
        SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));
        return SA_OK;
    }
    if (filt.active) {
        std::vector<u64*> bufs_v((size_t)T);
        u64** const bufs = bufs_v.data();
        for (int t = 0; t < T; t++) {
            bufs[t] = ar.take<u64>((size_t)lens[t] + 1);
            if (!bufs[t]) { sa_set_error("internal: phrase arena exhausted"); return SA_ERR_STATE; }
        }
        SA_TRY(sa_posn_filter_terms(ix, filt, T, ptrs, lens, bufs, lens_dev, s.chunks));
    }
    SA_HIP(hipMemcpyAsync(lens_dev, lens, (size_t)T * sizeof(u32), hipMemcpyHostToDevice, st));
    u32* ppn = lens_dev + T;                                    // two ping-pong length counters

    // plan: reference compute_phrase_freqs, middle_out.py:154-168 (first shortest on ties)
    int shortest = 0;
    for (int t = 1; t < T; t++) if (lens[t] < lens[shortest]) shortest = t;
    const bool l2r_only = shortest <= 1, r2l_only = !l2r_only && shortest >= T - 2;

    // doc directory rows (valid for the unfiltered term lists only)
    std::vector<const u32*> dd_rows_v((size_t)T);
    const u32** const dd_rows = dd_rows_v.data();
    {
        const bool use_dd = !filt.active && ix->n_dd_terms > 0 && sa_opt(ix->opts.phrase_docdir, 1) != 0;
        std::vector<u32>& slots = ix->h_dd_slot;
        for (int t = 0; t < T; t++) {
            const u32 sl = (use_dd && terms[t] < ix->n_terms) ? slots[terms[t]] : SA_DD_NONE;
            dd_rows[t] = sl != SA_DD_NONE ? ix->d_docdir + (size_t)sl * ix->n_docs : nullptr;
        }
    }
    // phrases with repeated terms: the chain per document (sa_k_phrase_docs) -- unless its checks say that the prediction
    // of the same-term test failed or a document did not fit
    {
        // (pairwise-distinct terms: only where the fused kernel cannot take the phrase -- a sub-phrase of more than 18 terms)
        const int part = (l2r_only || r2l_only) ? T : (shortest > T - shortest ? shortest : T - shortest);
        const bool fused_can = distinct && part <= SA_MAX_FUSED && !ix->any_top_block;
        bool take = mode == 0 && !fused_can && sa_opt(ix->opts.phrase_docs, 1) != 0 && !filt.active && T >= 2 && T <= SA_PD_MAXT && N < 0xFFFFFFF0ull;
        for (int t = 0; t < T && take; t++) take = lens[t] > 0;
        if (take) {
            PhraseDocParams pd;
            memset(&pd, 0, sizeof(pd));
            for (int t = 0; t < T; t++) { pd.words[t] = ptrs[t]; pd.dd[t] = dd_rows[t]; pd.len[t] = lens[t]; }
            pd.T = T; pd.n_docs = N; pd.counts = running;
            pd.plan = l2r_only ? 0 : (r2l_only ? 1 : 2);
            pd.shortest = shortest;
            // (a rarest list about as long as the collection: a thread per document, nothing to clear)
            pd.anchor = 2 * (u64)lens[shortest] >= N ? -1 : shortest;
            // the chain's first step over twice the same term is the one whose matched pairs are all equal
            int n_steps = T - 1;
            if (pd.plan == 0) pd.same_mask = terms[0] == terms[1] ? 1u : 0u;
            else if (pd.plan == 1) pd.same_mask = terms[T - 2] == terms[T - 1] ? 1u : 0u;
            else {
                n_steps = T - 2;                                   // (shortest - 1) + (T - shortest - 1)
                pd.same_mask = (terms[0] == terms[1] ? 1u : 0u) | (terms[T - 2] == terms[T - 1] ? 1u << (shortest - 1) : 0u);
            }
            if (!ix->h_flags) SA_HIP(hipHostMalloc((void**)&ix->h_flags, 128 * sizeof(u32), 0));
            if (!ix->d_flags) {
                SA_HIP(hipMalloc(&ix->d_flags, 128 * sizeof(u32)));
                SA_HIP(hipMemsetAsync(ix->d_flags, 0, 128 * sizeof(u32), st));
            }
            u32* const flags = ix->d_flags;                        // (zeros: every use ends with sa_k_flags_out, and the call waits for it)
            pd.flags = flags;
            u32* const h_flags = ix->h_flags;
            const dim3 grid((u32)(((pd.anchor < 0 ? N : (u64)lens[shortest]) + SA_PD_THREADS - 1) / SA_PD_THREADS));
            bool ok = false;
            // four words per list first (twenty waves per CU); eight if a document does not fit
            for (int attempt = 0; attempt < 2 && !ok; attempt++) {
                if (pd.anchor >= 0) SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));
                if (attempt == 0) hipLaunchKernelGGL(sa_k_phrase_docs<4>, grid, dim3(SA_PD_THREADS), 0, st, pd);
                else hipLaunchKernelGGL(sa_k_phrase_docs<8>, grid, dim3(SA_PD_THREADS), 0, st, pd);
                hipLaunchKernelGGL(sa_k_flags_out, dim3(1), dim3(128), 0, st, flags, h_flags, (u32)(2 + 2 * SA_PD_MAXT));
                SA_HIP(hipStreamSynchronize(st));
                if (h_flags[0] != 0) continue;                    // a document did not fit (its steps' records are incomplete)
                ok = true;
                for (int k = 0; k < n_steps && ok; k++)
                    if (!((pd.same_mask >> k) & 1u) && h_flags[1 + 2 * k] != 0 && h_flags[2 + 2 * k] == 0) ok = false;
                break;                                            // (a failed same-term prediction: no capacity helps)
            }
            if (sa_opt(ix->opts.trace, 0)) fprintf(stderr, "phrase route: chain per document %s\n", ok ? "taken" : "abandoned (general chain)");
            if (ok) return SA_OK;
        }
    }
    SA_HIP(hipMemsetAsync(running, 0, N * sizeof(float), st));     // (the routes below accumulate)
    // (a sub-phrase the fused kernel would have to take whole must fit its 18-position window: longer ones go
    //  through the general chain, which has no such limit)
    const int longest_part = (l2r_only || r2l_only) ? T : (shortest > T - shortest ? shortest : T - shortest);
    // (and no word in a document's last 18-position block: a match across two documents -- bit 17 of one's last block, bit 0
    //  of the next one's first -- is the LEFT document's in the chain, the anchor's in the fused kernel; the reference's
    //  encoder cannot produce such words, hand-made ones take the chain)
    const bool use_fused = (mode == 2) || (mode == 0 && distinct && longest_part <= SA_MAX_FUSED && !ix->any_top_block);
    if (use_fused) {
        if (!distinct) { sa_set_error("fused phrase kernel needs pairwise-distinct terms"); return SA_ERR_ARG; }
        // sub-phrases the reference evaluates: the whole phrase (l2r / r2l plans) or the two halves
        int parts[2][2] = {{0, T}, {0, 0}};
        int nparts = 1;
        if (!l2r_only && !r2l_only) { parts[0][1] = shortest; parts[1][0] = shortest; parts[1][1] = T; nparts = 2; }
        // (one sub-phrase: its counts ARE the result -- straight into the float vector, no count vector to clear and to fold)
        if (nparts > 1) SA_HIP(hipMemsetAsync(step, 0, N * sizeof(u32), st));
        for (int pi = 0; pi < nparts; pi++) {
            const int a = parts[pi][0], b = parts[pi][1];
            FusedPhraseParams fp;
            memset(&fp, 0, sizeof(fp));
            fp.T = b - a; fp.step = step;
            fp.fcounts = nparts == 1 ? running : nullptr;
            int anchor = 0;
            for (int t = a; t < b; t++) {
                fp.ptr[t - a] = ptrs[t]; fp.len[t - a] = lens[t];
                fp.dd[t - a] = dd_rows[t];
                if (lens[t] < lens[a + anchor]) anchor = t - a;
            }
            fp.anchor = anchor;
            if (fp.T > SA_MAX_FUSED) { sa_set_error("fused phrase kernel: sub-phrase longer than 18 terms"); return SA_ERR_UNSUPPORTED; }
            if (fp.len[anchor] > 0)
                hipLaunchKernelGGL(sa_k_phrase_fused, dim3(sa_grid_for(fp.len[anchor])), dim3(256), 0, st, fp);
            if (nparts > 1) hipLaunchKernelGGL(sa_k_min_step, dim3(sa_grid_for(N)), dim3(256), 0, st, running, step, N, pi == 0 ? 1 : 0);
        }
        return SA_OK;
    }

    // general chain
    SA_HIP(hipMemsetAsync(step, 0, N * sizeof(u32), st));
    auto term_arr = [&](int t) {
        DArr a; a.data = ptrs[t]; a.n_dev = lens_dev + t; a.bound = lens[t]; return a;
    };
    auto chain = [&](int a, int b, bool left_to_right, float* dst) -> int {
        // enc = terms[a..b)
        int first = 1, cur = 0;
        if (left_to_right) {
            DArr lhs = term_arr(a);
            for (int t = a + 1; t < b; t++) {
                DArr rhs = term_arr(t);
                SA_TRY(sa_bigram_step(st, lhs, rhs, CONT_RHS, s, step, pp[cur], ppn + cur));
                hipLaunchKernelGGL(sa_k_min_step, dim3(sa_grid_for(N)), dim3(256), 0, st, dst, step, N, first);
                first = 0;
                const u32 nb = lhs.bound < rhs.bound ? 2 * lhs.bound : rhs.bound;     // cont headers come from rhs
                lhs.data = pp[cur]; lhs.n_dev = ppn + cur; lhs.bound = nb < rhs.bound ? nb : rhs.bound;
                cur ^= 1;
            }
        } else {
            DArr rhs = term_arr(b - 1);
            for (int t = b - 2; t >= a; t--) {
                DArr lhs = term_arr(t);
                SA_TRY(sa_bigram_step(st, lhs, rhs, CONT_LHS, s, step, pp[cur], ppn + cur));
                hipLaunchKernelGGL(sa_k_min_step, dim3(sa_grid_for(N)), dim3(256), 0, st, dst, step, N, first);
                first = 0;
                const u32 nb = rhs.bound < lhs.bound ? 2 * rhs.bound : lhs.bound;     // cont headers come from lhs
                rhs.data = pp[cur]; rhs.n_dev = ppn + cur; rhs.bound = nb < lhs.bound ? nb : lhs.bound;
                cur ^= 1;
            }
        }
        return SA_OK;
    };
    if (l2r_only) {
        SA_TRY(chain(0, T, true, running));
    } else if (r2l_only) {
        SA_TRY(chain(0, T, false, running));
    } else {
        SA_TRY(chain(0, shortest, true, running));
        SA_TRY(chain(shortest, T, false, running2));
        hipLaunchKernelGGL(sa_k_min2, dim3(sa_grid_for(N)), dim3(256), 0, st, running, running2, N);
    }
    return SA_OK;
}

// algorithmic bytes of a phrase: every word of every term read once
static u64 sa_phrase_alg_bytes(const sa_index* ix, const u32* terms, int T) {
    u64 b = 0;
    for (int t = 0; t < T; t++)
        if (terms[t] < ix->n_terms) b += 8 * (ix->h_term_off[terms[t] + 1] - ix->h_term_off[terms[t]]);
    return b;
}

static int sa_profile_begin(sa_index* ix) {
    if (!ix->ev0) { SA_HIP(hipEventCreate(&ix->ev0)); SA_HIP(hipEventCreate(&ix->ev1)); }
    SA_HIP(hipEventRecord(ix->ev0, ix->stream));
    return SA_OK;
}

static int sa_profile_end(sa_index* ix, u64 alg_bytes) {
    SA_HIP(hipEventRecord(ix->ev1, ix->stream));
    ix->last_alg_bytes = alg_bytes;
    ix->profile_pending = true;
    return SA_OK;
}

extern "C" int sa_index_last_profile(sa_index_t* ix, double* kernel_ms_out, uint64_t* alg_bytes_out) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->profile_pending) {
        SA_HIP(hipSetDevice(ix->device));
        SA_HIP(hipEventSynchronize(ix->ev1));
        float ms = 0.f;
        SA_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
        ix->last_kernel_ms = ms;
        ix->profile_pending = false;
    }
    if (kernel_ms_out) *kernel_ms_out = ix->last_kernel_ms;
    if (alg_bytes_out) *alg_bytes_out = ix->last_alg_bytes;
    return SA_OK;
}

// option phrase_mode: 0 the library chooses, 1 the general chain, 2 the fused kernel
static int sa_phrase_mode(const sa_index* ix) {
    const long long m = sa_opt(ix->opts.phrase_mode, 0);
    return m == 1 || m == 2 ? (int)m : 0;
}

static int sa_phrase_or_span(sa_index* ix, const u32* terms, int n_terms, int slop, const PosnFilter& filt, float** d_out) {
    if (slop == 0) return sa_phrase_counts_device(ix, terms, n_terms, sa_phrase_mode(ix), filt, d_out);
    return sa_span_counts_device(ix, terms, n_terms, slop, filt, d_out);
}

// for the phrase batches (sa_phrase_batch.hip): dense counts of any phrase / slop in the index scratch, and
// counts -> BM25 in place; the caller holds the index lock and enqueues on ix->stream
int sa_phrase_dense_counts_device(sa_index* ix, const u32* terms, int n_terms, int slop, float** d_out) {
    PosnFilter filt;
    return sa_phrase_or_span(ix, terms, n_terms, slop, filt, d_out);
}
void sa_launch_bm25_from_tf(sa_index* ix, float* d_tf, float idf, float k1, float b) {
    if (ix->n_docs) hipLaunchKernelGGL(sa_k_bm25_from_tf, dim3(sa_grid_for(ix->n_docs)), dim3(256), 0, ix->stream, d_tf,
                                       ix->d_doc_lens, ix->avg_doc_len, idf, k1, b, ix->n_docs);
}

extern "C" int sa_index_phrase_freqs_dense_posn(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                                int64_t min_posn, int64_t max_posn, float* out) {
    SA_ARG(ix && out && terms, "null argument");
    // reference middle_out.py:425-426
    if (n_terms < 2) { sa_set_error("Must have at least two terms"); return SA_ERR_ARG; }
    SA_ARG(slop >= 0, "slop < 0");
    PosnFilter filt;
    SA_TRY(sa_posn_filter_bounds(min_posn, max_posn, &filt));
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    float* d_running = nullptr;
    SA_TRY(sa_profile_begin(ix));
    SA_TRY(sa_phrase_or_span(ix, terms, n_terms, slop, filt, &d_running));
    SA_TRY(sa_profile_end(ix, sa_phrase_alg_bytes(ix, terms, n_terms)));
    SA_TRY(sa_emit_dense(ix, d_running, out));
    SA_HIP(hipStreamSynchronize(ix->stream));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_index_phrase_freqs_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, float* out) {
    return sa_index_phrase_freqs_dense_posn(ix, terms, n_terms, slop, -1, -1, out);
}

// SearchArray.score(phrase): phrase counts -> BM25 with the idf summed over the phrase's terms
// (reference postings.py:652-680, similarity.py:19-38); idf is computed by the host.
extern "C" int sa_index_bm25_phrase_dense_posn(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                               int64_t min_posn, int64_t max_posn, float idf, float k1, float b,
                                               float* out) {
    SA_ARG(ix && out && terms, "null argument");
    if (n_terms < 2) { sa_set_error("Must have at least two terms"); return SA_ERR_ARG; }
    SA_ARG(slop >= 0, "slop < 0");
    PosnFilter filt;
    SA_TRY(sa_posn_filter_bounds(min_posn, max_posn, &filt));
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    const u64 N = ix->n_docs;
    if (ix->avg_doc_len == 0.f) {                      // similarity.py:31-32
        sa_emit_zeros(ix, out);
        return SA_OK;
    }
    float* d_running = nullptr;
    SA_TRY(sa_phrase_or_span(ix, terms, n_terms, slop, filt, &d_running));
    if (N) hipLaunchKernelGGL(sa_k_bm25_from_tf, dim3(sa_grid_for(N)), dim3(256), 0, ix->stream, d_running,
                              ix->d_doc_lens, ix->avg_doc_len, idf, k1, b, N);
    SA_TRY(sa_emit_dense(ix, d_running, out));
    SA_HIP(hipStreamSynchronize(ix->stream));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_index_bm25_phrase_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                          float idf, float k1, float b, float* out) {
    return sa_index_bm25_phrase_dense_posn(ix, terms, n_terms, slop, -1, -1, idf, k1, b, out);
}

// single-term tf restricted to a position range: popcount-reduce over the filtered words
// (reference PosnBitArray.termfreqs with min/max posn, middle_out.py:489-497) scattered dense.
struct TfHeads {
    const u64* words; const u32* n_dev; float* out;
    __device__ __forceinline__ bool flag(u32 i) const { return i == 0 || (words[i] >> SA_KEY_SHIFT) != (words[i - 1] >> SA_KEY_SHIFT); }
    __device__ __forceinline__ void emit(u32 i, u32) const {
        const u32 n = *n_dev;
        const u64 doc = words[i] >> SA_KEY_SHIFT;
        u32 tf = 0;
        for (u32 j = i; j < n && (words[j] >> SA_KEY_SHIFT) == doc; j++) tf += (u32)__popcll(words[j] & SA_LSB_MASK);
        out[doc] = (float)tf;
    }
};

extern "C" int sa_index_termfreqs_dense_posn(sa_index_t* ix, uint32_t term, int64_t min_posn, int64_t max_posn, float* out) {
    SA_ARG(ix && out, "null argument");
    PosnFilter filt;
    SA_TRY(sa_posn_filter_bounds(min_posn, max_posn, &filt));
    if (!filt.active) return sa_index_termfreqs_dense(ix, term, out);
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    u32 len = 0;
    const u64* ptr = ix->d_words;
    if (term < ix->n_terms) {
        ptr = ix->d_words + ix->h_term_off[term];
        len = (u32)(ix->h_term_off[term + 1] - ix->h_term_off[term]);
    }
    const size_t chunk_words = sa_compact_chunks(len + 1) + 8;
    void* scratch;
    SA_TRY(sa_index_scratch(ix, (N + 64) * 4 + ((size_t)len + 64) * 8 + chunk_words * 4 + 4096, &scratch));
    Arena ar; ar.base = (char*)scratch; ar.cap = ix->scratch_bytes;
    float* d_out = ar.take<float>(N + 1);
    u64* buf = ar.take<u64>((size_t)len + 1);
    u32* chunks = ar.take<u32>(chunk_words);
    u32* cnt = ar.take<u32>(4);
    SA_HIP(hipMemsetAsync(d_out, 0, N * sizeof(float), st));
    SA_HIP(hipMemsetAsync(cnt, 0, 16, st));
    if (len) {
        PosnSlice ps; ps.arr = ptr; ps.lo = filt.lo; ps.hi = filt.hi; ps.out = buf;
        sa_compact(ps, (const u32*)nullptr, len, chunks, cnt, st);
        TfHeads th; th.words = buf; th.n_dev = cnt; th.out = d_out;
        sa_compact(th, cnt, len, chunks, cnt + 1, st);
    }
    SA_TRY(sa_emit_dense(ix, d_out, out));
    SA_HIP(hipStreamSynchronize(st));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// sa_phrase_dev.hpp -- device helpers shared by the phrase kernels (sa_phrase.hip: one phrase over
// the whole index; sa_phrase_batch.hip: many phrases, one workgroup per doc tile).
#pragma once
#include "sa_common.hpp"

// 18-bit payload of the word with header h in a[0, n), 0 if absent.  `hint` carries the previous
// probe's position so the three probes h-1, h, h+1 cost one search.
__device__ __forceinline__ u64 sa_payload_at(const u64* __restrict__ a, u32 n, u64 h, u32& hint) {
    const u32 j = sa_lower_bound(a, hint, n, h, SA_HEADER_MASK);
    hint = j;
    return (j < n && (a[j] & SA_HEADER_MASK) == h) ? (a[j] & SA_LSB_MASK) : 0ull;
}

// Exact phrase of T pairwise-distinct terms, anchored on term `anchor`: for the anchor's word w
// (doc, 18-position block, bitmap) return the bitmap of anchor positions that belong to a
// full match: bit p survives iff every other term t sits at position 18*blk + p + (t - anchor)
// of the same doc.  Term t's words around the block are lined up as a 54-bit window (payloads of
// headers h-1 | h | h+1, same doc only) and shifted by t - anchor; popcount of the result is the
// number of phrase occurrences whose anchor term lies in this word -- the quantity the reference
// accumulates bigram by bigram (bigram_freqs.py:48-307, middle_out.py:73-168).
// `window(t, h, want_prev, want_next)` returns term t's window around header h:
//   bits [0,18) payload of h-1 (only if want_prev), [18,36) payload of h, [36,54) of h+1 (only if
//   want_next); absent words contribute 0.
template <class WindowFn>
__device__ __forceinline__ u64 sa_phrase_anchor_mask_win(u64 w, int T, int anchor, WindowFn window) {
    const u64 delta = 1ull << SA_LSB_BITS;
    const u64 h = w & SA_HEADER_MASK;
    const u64 doc_key = w & SA_KEY_MASK;
    u64 m = w & SA_LSB_MASK;                     // bit p: anchor term at position 18*blk + p
    for (int t = 0; t < T && m; t++) {
        if (t == anchor) continue;
        const int d = t - anchor;                // term t must sit at anchor position + d, |d| < 18
        // the neighbouring blocks matter only on the side the offset points to, and only inside the doc
        const bool want_prev = d < 0 && (h & ~SA_KEY_MASK & SA_HEADER_MASK) != 0 && ((h - delta) & SA_KEY_MASK) == doc_key;
        const bool want_next = d > 0 && ((h + delta) & SA_KEY_MASK) == doc_key;
        const u64 win = window(t, h, want_prev, want_next);
        m &= (win >> (18 + d)) & SA_LSB_MASK;
    }
    return m;
}

// Window of a term around header h when the term has a doc directory row (sa_index.hpp): row[doc] is
// the index in a[0, n) of the doc's first word, or SA_DD_ABSENT.  One 4-byte load rejects docs that
// lack the term; otherwise the doc's first three words (ascending blocks) are fetched at once --
// independent loads -- and only docs with more blocks of this term walk on.
__device__ __forceinline__ u64 sa_window_docdir(const u64* __restrict__ a, u32 n, const u32* __restrict__ row,
                                                u64 h, bool want_prev, bool want_next) {
    const u64 delta = 1ull << SA_LSB_BITS;
    const u32 s = row[h >> SA_KEY_SHIFT];
    if (s == 0xFFFFFFFFu) return 0ull;
    const u64 hm = h - delta, hp = h + delta;
    u64 win = 0;
    u64 x[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x[j] = s + j < n ? a[s + j] : 0ull;
    auto take = [&](u64 xw, bool valid) -> bool {    // false: past the window (or the list)
        const u64 xh = xw & SA_HEADER_MASK;
        if (!valid || xh > hp) return false;
        const u64 pl = xw & SA_LSB_MASK;
        if (xh == h) win |= pl << 18;
        else if (xh == hm) { if (want_prev) win |= pl; }
        else if (xh == hp) { if (want_next) win |= pl << 36; }
        return true;
    };
    bool more = take(x[0], true) && take(x[1], s + 1 < n) && take(x[2], s + 2 < n);
    for (u32 j = s + 3; more && j < n; j++) more = take(a[j], true);
    return win;
}

// every term given as a sorted word array `slice(t, a, n)`: the window is up to three lower-bound
// probes (the second and third continue from the first one's position)
template <class SliceFn>
__device__ __forceinline__ u64 sa_phrase_anchor_mask(u64 w, int T, int anchor, SliceFn slice) {
    return sa_phrase_anchor_mask_win(w, T, anchor, [&](int t, u64 h, bool want_prev, bool want_next) -> u64 {
        const u64 delta = 1ull << SA_LSB_BITS;
        const u64* a;
        u32 n;
        slice(t, a, n);
        u32 hint = 0;
        u64 win = 0;
        if (want_prev) win |= sa_payload_at(a, n, h - delta, hint);
        win |= sa_payload_at(a, n, h, hint) << 18;
        if (want_next) win |= sa_payload_at(a, n, h + delta, hint) << 36;
        return win;
    });
}

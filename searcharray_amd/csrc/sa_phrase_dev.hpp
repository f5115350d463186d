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
// `slice(t, a, n)` yields term t's sorted words a[0, n) (the whole term, or its slice of a tile).
template <class SliceFn>
__device__ __forceinline__ u64 sa_phrase_anchor_mask(u64 w, int T, int anchor, SliceFn slice) {
    const u64 delta = 1ull << SA_LSB_BITS;
    const u64 h = w & SA_HEADER_MASK;
    const u64 doc_key = w & SA_KEY_MASK;
    u64 m = w & SA_LSB_MASK;                     // bit p: anchor term at position 18*blk + p
    for (int t = 0; t < T && m; t++) {
        if (t == anchor) continue;
        const int d = t - anchor;                // term t must sit at anchor position + d, |d| < 18
        const u64* a;
        u32 n;
        slice(t, a, n);
        u32 hint = 0;
        u64 win = 0;
        if (d < 0) {
            const u64 hm = h - delta;
            if ((h & ~SA_KEY_MASK & SA_HEADER_MASK) != 0 && (hm & SA_KEY_MASK) == doc_key)
                win |= sa_payload_at(a, n, hm, hint);
        }
        win |= sa_payload_at(a, n, h, hint) << 18;
        if (d > 0) {
            const u64 hp = h + delta;
            if ((hp & SA_KEY_MASK) == doc_key) win |= sa_payload_at(a, n, hp, hint) << 36;
        }
        m &= (win >> (18 + d)) & SA_LSB_MASK;
    }
    return m;
}

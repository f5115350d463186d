/*
 * oracle/spans.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Restatement of the reference's slop > 0 span state machine,
 * searcharray/roaringish/spans.pyx:70-319 (_span_freqs and its helpers).  The reference keeps at
 * most 512 ActiveSpans per document; once the table is full its own code indexes one past the
 * arrays (spans.pyx:238-246 with cursor == 512), which is undefined behaviour.  This restatement
 * guards those writes: a document that fills the table is reported through the reference's
 * "full" rule (min over terms of the summed popcounts, spans.pyx:306-311) and flagged in
 * *overflow_docs so tests can exclude it from bit-exact comparison.
 *
 * Output: a list of (key, increment) events in processing order; the caller accumulates them
 * like the reference's Counter (phrase/spans.py:175-187).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;

#define NSPANS 512

typedef struct {
    u64 terms[NSPANS];
    u64 posns[NSPANS];
    i64 beg[NSPANS];
    i64 end[NSPANS];
    u64 cursor;
} ActiveSpans;

static i64 iabs64(i64 v) { return v < 0 ? -v : v; }

/* spans.pyx:108-109 `return 1 << (curr_posn % 64)`: Cython emits the literal as a C `int`, so the
 * shift is a 32-bit one.  As built by gcc on x86-64 (the reference's only target) the count is
 * taken mod 32 and the int result is sign-extended into the 64-bit DTYPE_t: positions p and p+32
 * alias, and p % 32 == 31 sets bits 31..63.  This is the reference's observable behaviour and
 * the goldens (outputs of the reference itself) pin it. */
static u64 posn_mask_of(i64 curr_posn)
{
    const int32_t m = (int32_t)(1u << ((unsigned)(curr_posn % 64) & 31u));
    return (u64)(i64)m;
}
static i64 span_width(const ActiveSpans* s, u64 i) { return iabs64(s->end[i] - s->beg[i]); }   /* spans.pyx:94-95 */

/* spans.pyx:140-154 */
static void compact_spans(ActiveSpans* spans, u64 max_width)
{
    ActiveSpans* out = (ActiveSpans*)calloc(1, sizeof(ActiveSpans));
    for (u64 i = 0; i < spans->cursor && i < NSPANS; i++) {
        if ((u64)span_width(spans, i) > max_width) continue;
        if (__builtin_popcountll(spans->terms[i]) > 0) {
            out->terms[out->cursor] = spans->terms[i];
            out->posns[out->cursor] = spans->posns[i];
            out->beg[out->cursor] = spans->beg[i];
            out->end[out->cursor] = spans->end[i];
            out->cursor++;
        }
    }
    memcpy(spans, out, sizeof(ActiveSpans));
    free(out);
}

/* spans.pyx:157-186: count complete, non-overlapping spans (shorter replaces overlapping longer) */
static u64 collect_spans(const ActiveSpans* spans, u64 num_terms, u64 max_width)
{
    ActiveSpans* col = (ActiveSpans*)calloc(1, sizeof(ActiveSpans));
    u64 n = spans->cursor < NSPANS ? spans->cursor : NSPANS;
    for (u64 i = 0; i < n; i++) {
        const int complete = ((u64)__builtin_popcountll(spans->terms[i]) == num_terms) ||
                             ((u64)__builtin_popcountll(spans->posns[i]) == num_terms);        /* :125-128 */
        if (complete && (u64)span_width(spans, i) < max_width) {
            const i64 new_width = iabs64(spans->end[i] - spans->beg[i]);
            int overlaps = 0;
            for (u64 c = 0; c < col->cursor; c++) {
                if (spans->beg[i] <= col->end[c] && spans->end[i] >= col->beg[c]) {             /* :119-122 */
                    const i64 coll_width = iabs64(col->end[c] - col->beg[c]);
                    if (new_width < coll_width) {
                        col->terms[c] = spans->terms[i];
                        col->posns[c] = spans->posns[i];
                        col->beg[c] = spans->beg[i];
                        col->end[c] = spans->end[i];
                        overlaps = 1;
                        break;
                    }
                }
            }
            if (!overlaps) {
                col->terms[col->cursor] = spans->terms[i];
                col->posns[col->cursor] = spans->posns[i];
                col->beg[col->cursor] = spans->beg[i];
                col->end[col->cursor] = spans->end[i];
                col->cursor++;
            }
        }
    }
    u64 r = col->cursor;
    free(col);
    return r;
}

/* spans.pyx:189-319 */
long oracle_span_freqs(const u64* posns, long n_posns, const u64* lengths, long n_lengths, u64 slop,
                       u64* out_keys, float* out_incr, long out_cap, long* overflow_docs)
{
    const u64 key_mask = 0xFFFFFFF000000000ull, header_mask = 0xFFFFFFFFFFFC0000ull;
    const u64 key_bits = 28, lsb_bits = 18;
    const u64 payload_mask = ~header_mask;
    const u64 payload_msb_mask = header_mask & ~key_mask;
    const u64 num_terms = (u64)n_lengths - 1;
    const u64 max_span_width = num_terms + slop;
    u64 curr_idx[64], sum_popcount[64];
    ActiveSpans* spans = (ActiveSpans*)calloc(1, sizeof(ActiveSpans));
    long n_out = 0;
    int full = 0;
    u64 curr_key = 0, last_key = 0;
    *overflow_docs = 0;

    for (u64 i = 0; i < num_terms; i++) curr_idx[i] = lengths[i];

    while (curr_idx[0] < lengths[1]) {
        int overflowed = 0;
        for (u64 term_ord = 0; term_ord < num_terms; term_ord++) {
            if ((long)curr_idx[term_ord] < n_posns)                       /* guarded read (reference reads past the slice) */
                curr_key = (posns[curr_idx[term_ord]] & key_mask) >> (64 - key_bits);
            sum_popcount[term_ord] = 0;
            while (curr_idx[term_ord] < lengths[term_ord + 1]) {
                last_key = curr_key;
                u64 term = posns[curr_idx[term_ord]];
                const u64 payload_base = ((term & payload_msb_mask) >> lsb_bits) * lsb_bits;
                term &= payload_mask;
                const u64 curr_term_mask = 1ull << term_ord;
                sum_popcount[term_ord] += (u64)__builtin_popcountll(posns[curr_idx[term_ord]] & payload_mask);

                while (term != 0) {
                    const u64 set_idx = (u64)__builtin_ctzll(term);
                    term &= term - 1;
                    const i64 curr_posn = (i64)(set_idx + payload_base);
                    const u64 posn_mask = posn_mask_of(curr_posn);
                    if (spans->cursor >= NSPANS) { overflowed = 1; full = 1; break; }    /* reference: UB write */
                    spans->terms[spans->cursor] = curr_term_mask;
                    spans->posns[spans->cursor] = posn_mask;
                    spans->beg[spans->cursor] = curr_posn;
                    spans->end[spans->cursor] = curr_posn;
                    const u64 end = spans->cursor;
                    spans->cursor += 1;
                    for (u64 si = 0; si < end; si++) {
                        const u64 nt = (u64)__builtin_popcountll(spans->terms[si]);
                        const u64 np = (u64)__builtin_popcountll(spans->posns[si]);
                        if (nt < num_terms && np == num_terms) continue;
                        spans->terms[si] |= curr_term_mask;
                        const u64 nt_now = (u64)__builtin_popcountll(spans->terms[si]);
                        if (nt_now > nt) {
                            spans->posns[si] |= posn_mask;
                            const u64 new_unique = (u64)__builtin_popcountll(spans->posns[si]);
                            const u64 proposed_width = (u64)iabs64(curr_posn - spans->beg[si]);
                            if (np == new_unique || proposed_width > max_span_width) {
                                spans->terms[si] &= ~curr_term_mask;
                                continue;
                            }
                            if (spans->cursor < NSPANS) {
                                spans->terms[spans->cursor] = spans->terms[si];
                                spans->posns[spans->cursor] = spans->posns[si] & ~posn_mask;
                                spans->beg[spans->cursor] = spans->beg[si];
                                spans->end[spans->cursor] = spans->end[si];
                                spans->cursor += 1;
                                full = 0;
                            } else {
                                full = 1;
                            }
                            spans->end[si] = curr_posn;
                        }
                    }
                    if (spans->cursor >= NSPANS) break;
                }
                curr_idx[term_ord] += 1;
                if (curr_idx[term_ord] < lengths[term_ord + 1])
                    curr_key = (posns[curr_idx[term_ord]] & key_mask) >> (64 - key_bits);
                if (spans->cursor >= NSPANS) {
                    compact_spans(spans, max_span_width);
                    if (spans->cursor >= NSPANS) {
                        overflowed = 1;
                        for (u64 i = curr_idx[term_ord]; i < lengths[term_ord + 1]; i++) {
                            curr_key = (posns[i] & key_mask) >> (64 - key_bits);
                            if (curr_key != last_key) { curr_idx[term_ord] = i; break; }
                        }
                    }
                }
                if (curr_key != last_key) break;
            }
        }
        float incr;
        if (full) {
            u64 min_pop = 0;
            for (u64 t = 0; t < num_terms; t++)
                if (min_pop == 0 || sum_popcount[t] < min_pop) min_pop = sum_popcount[t];
            incr = (float)min_pop;
        } else {
            incr = (float)collect_spans(spans, num_terms, max_span_width);
        }
        if (n_out < out_cap) { out_keys[n_out] = last_key; out_incr[n_out] = incr; }
        n_out++;
        if (overflowed) (*overflow_docs)++;
        memset(spans, 0, sizeof(ActiveSpans));
        full = 0;
    }
    free(spans);
    return n_out;
}

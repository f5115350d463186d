// sa_phrase_batch.hip -- many exact phrases at once: phrase match counts -> BM25 -> top-k, all on
// the device (Part 2 of the C ABI, phrase flavour of sa_batch_*).
//
// Replaces the caller loop  `for phrase in phrases: scores = arr.score(phrase); top = argpartition`
// around reference SearchArray.score (postings.py:652-680) -> PosnBitArray.phrase_freqs
// (middle_out.py:418-446) -> compute_phrase_freqs (middle_out.py:73-168) -> bm25 (similarity.py:24-38,
// bm25.pyx:11-25).  sa_index_bm25_phrase_dense remains the one-phrase, dense-output drop-in; the batch
// keeps everything resident and returns only B x k (score, doc) pairs.
//
// Layout: documents are cut into tiles of SA_PTILE docs; one workgroup scores one (tile, phrase).  A
// doc's roaringish words are contiguous in every term's list, so a tile owns one contiguous slice of
// each phrase term (slice table built once per batch by lower-bound searches).  The workgroup walks
// the slice of the phrase's rarest term (the anchor), lines the other terms up against each anchor
// word (sa_phrase_anchor_mask_win: 54-bit windows, shifts, AND, popcount -- integer work, no MFMA;
// frequent terms are located through the index's doc directory, one 4-byte load per probe, rare
// ones by a lower-bound search of their short tile slice) and adds the match counts into per-doc
// LDS counters; counts -> fp32 BM25 with the reference's operation
// order; the pruned wave-level top-k (sa_tile_topk_pruned) appends the few docs that can still reach
// the query's top-k.  Tiles in which some phrase term has no word leave after two loads.
//
// The reference's evaluation plan is kept: with the shortest term list at index s, phrases with
// s <= 1 or s >= T-2 are counted whole; otherwise the count is min(count(terms[:s]),
// count(terms[s:])) per doc (middle_out.py:154-168, "middle out").
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_phrase_dev.hpp"
#include "../../include/searcharray_hip.h"
#include <algorithm>
#include <new>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#define SA_PTILE 2048
#define SA_PTHREADS 256
#define SA_PHRASE_BATCH_MAXT 18      // |t - anchor| must stay below the 18-position block width

struct PhraseTileParams {
    const u64* words;
    const float* doc_lens;
    u64 n_docs, doc_base;
    u32 n_tiles;
    const u32* plan;       // [B][4]: n_terms (0: unknown term -> no match), split, anchor0, anchor1
    const u32* bounds;     // [B][T][n_tiles+1]
    const u64* wbase;      // [B][T]
    const float* idf;      // [B]
    u32 B, T, k;
    float k1, b, avgdl;
    u32 cand_cap;
    u32* cand_cnt;
    u32* slots;
    u64* cand;
    const u32* terms;      // [B][T] term ids
    const u32* wlen;       // [B][T] words of each phrase term (whole list)
    const u32* docdir;     // index doc directory [n_dd_terms][n_docs]
    const u32* dd_slot;    // [n_terms]
    int use_docdir;        // 0: binary-search every probe (SA_PHRASE_DOCDIR=0, tests)
};

// first word of each phrase term at or after every tile boundary, relative to the term's base
__global__ void __launch_bounds__(256)
sa_k_make_word_bounds(const u64* __restrict__ words, const u64* __restrict__ term_off, u32 n_terms, u32 n_tiles,
                      u32 tile_docs, const u32* __restrict__ terms, u32 BT, u32* __restrict__ bounds,
                      u64* __restrict__ wbase, u32* __restrict__ wlen) {
    const u64 total = (u64)BT * (n_tiles + 1);
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 qt = (u32)(e / (n_tiles + 1)), tile = (u32)(e % (n_tiles + 1));
        const u32 term = terms[qt];
        u32 rel = 0, cnt = 0;
        u64 base = 0;
        if (term < n_terms) {
            base = term_off[term];
            cnt = (u32)(term_off[term + 1] - base);
            rel = sa_lower_bound(words + base, 0, cnt, ((u64)tile * tile_docs) << SA_KEY_SHIFT, SA_KEY_MASK);
        }
        bounds[e] = rel;
        if (tile == 0) { wbase[qt] = base; wlen[qt] = cnt; }
    }
}

template <int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS) sa_k_phrase_tiles(const PhraseTileParams p) {
    constexpr int E = TILE / THREADS;
    __shared__ u32 cnt_a[TILE];                      // match counts, later the fp32 scores
    __shared__ u32 cnt_b[TILE];                      // second half of a middle-out plan
    __shared__ u64 s_lo[SA_PHRASE_BATCH_MAXT], s_hi[SA_PHRASE_BATCH_MAXT];
    __shared__ u64 s_tbase[SA_PHRASE_BATCH_MAXT];    // first word of the term (whole list)
    __shared__ u32 s_tlen[SA_PHRASE_BATCH_MAXT];
    __shared__ u32 s_dd[SA_PHRASE_BATCH_MAXT];       // doc directory row of the term, or SA_DD_NONE
    const u32 tid = threadIdx.x;
    const u32 item = blockIdx.x;
    const u32 tile = item / p.B, q = item % p.B;     // tile-major like the BM25 tiles
    const u64 tile_base = (u64)tile * TILE;
    const u32* plan = p.plan + (u64)q * 4;
    const u32 Tq = plan[0], split = plan[1];
    if (Tq == 0) return;                             // a term is unknown: zeros (postings.py:705-708)
    u32 slot_val = 0xFFFFFFFFu;
    if ((tid & (SA_WAVE - 1)) < 32u)
        slot_val = __hip_atomic_load(&p.slots[q * 32u + (tid & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < Tq) {
        const u32 qt = q * p.T + tid;
        const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
        const u64 base = p.wbase[qt];
        s_lo[tid] = base + row[0];
        s_hi[tid] = base + row[1];
        s_tbase[tid] = base;
        s_tlen[tid] = p.wlen[qt];
        s_dd[tid] = p.use_docdir ? p.dd_slot[p.terms[qt]] : SA_DD_NONE;
    }
#pragma unroll
    for (int j = 0; j < E; j++) cnt_a[j * THREADS + tid] = 0;
    if (split) {
#pragma unroll
        for (int j = 0; j < E; j++) cnt_b[j * THREADS + tid] = 0;
    }
    __syncthreads();
    // every term is needed by some part: an empty slice means no doc of this tile matches
    bool empty = false;
    for (u32 t = 0; t < Tq; t++) empty |= (s_lo[t] == s_hi[t]);
    if (empty) return;                               // uniform

    // Probing term t for the anchor's doc: frequent terms have a doc directory (one 4-byte load
    // finds the doc's words or rejects the doc); the others are searched inside their tile slice,
    // which is short precisely because the term is rare.
    const int nparts = split ? 2 : 1;
    for (int part = 0; part < nparts; part++) {
        const int t0 = part == 0 ? 0 : (int)split;
        const int t1 = (split && part == 0) ? (int)split : (int)Tq;
        const int anchor = (int)plan[2 + part];      // index within [t0, t1)
        u32* cnt = part == 0 ? cnt_a : cnt_b;
        const u64 alo = s_lo[t0 + anchor];
        const u32 na = (u32)(s_hi[t0 + anchor] - alo);
        for (u32 i = tid; i < na; i += THREADS) {
            const u64 w = p.words[alo + i];
            const u64 m = sa_phrase_anchor_mask_win(w, t1 - t0, anchor, [&](int t, u64 h, bool want_prev, bool want_next) -> u64 {
                const u32 dd = s_dd[t0 + t];
                if (dd != SA_DD_NONE)                // uniform per term
                    return sa_window_docdir(p.words + s_tbase[t0 + t], s_tlen[t0 + t], p.docdir + (u64)dd * p.n_docs,
                                            h, want_prev, want_next);
                const u64 delta = 1ull << SA_LSB_BITS;
                const u64* a = p.words + s_lo[t0 + t];
                const u32 n = (u32)(s_hi[t0 + t] - s_lo[t0 + t]);
                u32 hint = 0;
                u64 win = 0;
                if (want_prev) win |= sa_payload_at(a, n, h - delta, hint);
                win |= sa_payload_at(a, n, h, hint) << 18;
                if (want_next) win |= sa_payload_at(a, n, h + delta, hint) << 36;
                return win;
            });
            if (m) atomicAdd(&cnt[(u32)((w >> SA_KEY_SHIFT) - tile_base)], (u32)__popcll(m));
        }
    }
    __syncthreads();

    // counts -> BM25 (bm25.pyx:19-23 operation order, every op rounded to fp32); a thread converts
    // exactly the elements it owns in the selection below, so no barrier is needed in between
    float* acc = (float*)cnt_a;
    const float one_minus_b = 1.0f - p.b;
    const float idf = p.idf[q];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const u32 e = j * THREADS + tid;
        u32 c = cnt_a[e];
        if (split) { const u32 c2 = cnt_b[e]; c = c2 < c ? c2 : c; }
        float s = 0.f;
        if (c) {
            const float tf = (float)c;
            const float dl = p.doc_lens[tile_base + e];
            const float norm = __fmul_rn(p.k1, __fadd_rn(one_minus_b, __fmul_rn(p.b, __fdiv_rn(dl, p.avgdl))));
            s = __fmul_rn(__fdiv_rn(tf, __fadd_rn(tf, norm)), idf);
        }
        acc[e] = s;
    }
    sa_tile_topk_pruned<TILE, THREADS>(acc, slot_val, q, tile, p.doc_base + tile_base, p.k, p.slots, p.cand,
                                       p.cand_cap, p.cand_cnt);
}

// Ranking of a dense count vector (one phrase of the dense route): a workgroup loads a tile of counts, turns them
// into BM25 scores (reference bm25.pyx:11-25, the op order of sa_k_bm25_from_tf -- fused here: one launch less per
// phrase, and a batch of short phrases is bound by the host's launch rate) and runs the same pruned selection as
// the phrase tiles
template <int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS)
sa_k_dense_topk_tiles(const float* __restrict__ counts, const float* __restrict__ dl, float avgdl, float idf, float k1, float b,
                      u64 n_docs, u64 doc_base, u32 q, u32 k, u32* __restrict__ slots,
                      u64* __restrict__ cand, u32 cand_cap, u32* __restrict__ cand_cnt) {
    constexpr int E = TILE / THREADS;
    __shared__ float acc[TILE];
    const u32 tid = threadIdx.x, tile = blockIdx.x;
    const u64 tile_base = (u64)tile * TILE;
    const float one_minus_b = 1.0f - b;
    u32 slot_val = 0xFFFFFFFFu;
    if ((tid & (SA_WAVE - 1)) < 32u)
        slot_val = __hip_atomic_load(&slots[q * 32u + (tid & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < E; j++) {                    // (a thread loads exactly the elements it owns in the selection)
        const u64 d = tile_base + (u64)(j * THREADS) + tid;
        float sc = 0.f;
        if (d < n_docs) {
            const float t = counts[d];
            if (t != 0.f) {                                      // (0 / (0 + norm) * idf = 0: docs without a match are not ranked)
                const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl[d], avgdl))));
                sc = __fmul_rn(__fdiv_rn(t, __fadd_rn(t, norm)), idf);
            }
        }
        acc[j * THREADS + tid] = sc;
    }
    sa_tile_topk_pruned<TILE, THREADS>(acc, slot_val, q, tile, doc_base + tile_base, k, slots, cand, cand_cap, cand_cnt);
}

// the same for several dense-route phrases in ONE launch: blockIdx.y picks the phrase (its counts, idf and batch row)
template <int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS)
sa_k_dense_topk_tiles_multi(const sa_dense_rank_job* __restrict__ jobs, const float* __restrict__ dl, float avgdl, float k1, float b,
                            u64 n_docs, u64 doc_base, u32 k, u32* __restrict__ slots, u64* __restrict__ cand, u32 cand_cap,
                            u32* __restrict__ cand_cnt) {
    constexpr int E = TILE / THREADS;
    __shared__ float acc[TILE];
    const sa_dense_rank_job J = jobs[blockIdx.y];
    const u32 tid = threadIdx.x, tile = blockIdx.x, q = J.row;
    if (J.touched[tile] == 0) return;                          // no count in this tile: nothing to rank, nothing to put back
    __syncthreads();
    if (tid == 0) J.touched[tile] = 0;
    const u64 tile_base = (u64)tile * TILE;
    const float one_minus_b = 1.0f - b;
    u32 slot_val = 0xFFFFFFFFu;
    if ((tid & (SA_WAVE - 1)) < 32u)
        slot_val = __hip_atomic_load(&slots[q * 32u + (tid & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < E; j++) {
        const u64 d = tile_base + (u64)(j * THREADS) + tid;
        float sc = 0.f;
        if (d < n_docs) {
            const float t = J.counts[d];
            if (t != 0.f) {
                ((float*)J.counts)[d] = 0.f;                     // the count vectors of this route are all zeros between runs (sa_span_counts_batch)
                const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl[d], avgdl))));
                sc = __fmul_rn(__fdiv_rn(t, __fadd_rn(t, norm)), J.idf);
            }
        }
        acc[j * THREADS + tid] = sc;
    }
    sa_tile_topk_pruned<TILE, THREADS>(acc, slot_val, q, tile, doc_base + tile_base, k, slots, cand, cand_cap, cand_cnt);
}

int sa_launch_phrase_tiles(sa_batch* bt, hipStream_t st) {
    sa_index* ix = bt->ix;
    if (bt->pn_tiles == 0 || bt->B == 0) return SA_OK;
    // the batch's options rule while it runs (the dense routes below read the index's; the caller holds the index lock)
    struct OptsScope { sa_index* ix; sa_options_t saved; ~OptsScope() { ix->opts = saved; } } opts_scope{ix, ix->opts};
    ix->opts = bt->opts;
    PhraseTileParams p;
    memset(&p, 0, sizeof(p));
    p.words = ix->d_words; p.doc_lens = ix->d_doc_lens;
    p.n_docs = ix->n_docs; p.doc_base = ix->doc_base; p.n_tiles = bt->pn_tiles;
    p.plan = bt->d_plan; p.bounds = bt->d_wbounds; p.wbase = bt->d_wbase; p.idf = bt->d_idf;
    p.B = bt->B; p.T = bt->T; p.k = bt->k; p.k1 = bt->k1; p.b = bt->b; p.avgdl = ix->avg_doc_len;
    p.cand_cap = bt->cand_cap; p.cand_cnt = bt->d_cand_cnt; p.slots = bt->d_slots; p.cand = bt->d_cand;
    p.terms = bt->d_terms; p.wlen = bt->d_wlen; p.docdir = ix->d_docdir; p.dd_slot = ix->d_dd_slot;
    p.use_docdir = ix->n_dd_terms > 0 ? 1 : 0;
    if (sa_opt(bt->opts.phrase_docdir, 1) == 0) p.use_docdir = 0;
    const u64 n_items = (u64)bt->B * bt->pn_tiles;
    if (bt->dense_rows.size() < bt->B) {             // (rows on the dense route have plan[0] == 0: their items leave at once)
        if (bt->ptile == 2048)
            hipLaunchKernelGGL((sa_k_phrase_tiles<2048, SA_PTHREADS>), dim3((u32)n_items), dim3(SA_PTHREADS), 0, st, p);
        else
            hipLaunchKernelGGL((sa_k_phrase_tiles<4096, SA_PTHREADS>), dim3((u32)n_items), dim3(SA_PTHREADS), 0, st, p);
    }
    // the dense route: counts of the whole shard by the single-phrase kernels (general bigram chain with the
    // same-term rule / span machine for slop > 0), BM25 in place, then the ranking kernel above
    // Lanes: a dense-route phrase is a chain of ~10 short launches (a sampled slop phrase: 0.06 ms of device time,
    // most of it launch and dependency latency), and the phrases of a batch share nothing but the index -- row i runs
    // on lane i mod SA_PHRASE_LANES (default 2, at most 4), every lane but the first with a stream and a scratch area of its
    // own.  The single-phrase kernels take their stream and scratch from the index, so a lane is SWAPPED IN around
    // the calls (the index lock is held for the whole run).  Measured, 33 slop-2 phrases at 1 M docs as one batch:
    // 14.3 K phrases/s on one lane, 22.2 K on two, 21.6 K on three, 20.5 K on four (the host's launch rate is the limit
    // then: ~10 launches per phrase).  Capturing the sequence into a HIP graph (run 2 captured, later runs one
    // hipGraphLaunch; fork / join of the lanes through events) was built and measured on ROCm 7.2: 5.9 K phrases/s --
    // ~17 us per node of the replayed 330-node graph -- so the launches stay direct.
    struct Lane {
        sa_index* ix; int j;                                   // j < 0: the index's own stream and scratch
        Lane(sa_index* i, int lane) : ix(i), j(lane) { flip(); }
        ~Lane() { flip(); }
        void flip() {
            if (j < 0) return;
            std::swap(ix->stream, ix->lane_stream[j]);
            std::swap(ix->d_scratch, ix->lane_scratch[j]);
            std::swap(ix->scratch_bytes, ix->lane_bytes[j]);
        }
    };
    int n_lanes = 2;
    n_lanes = (int)sa_opt(bt->opts.phrase_lanes, n_lanes);
    n_lanes = std::max(1, std::min(n_lanes, 4));
    if ((size_t)n_lanes > bt->dense_rows.size()) n_lanes = (int)bt->dense_rows.size();
    if (n_lanes > 1) {
        for (int i = 0; i < 2; i++)
            if (!bt->ev_side[i]) SA_HIP(hipEventCreateWithFlags(&bt->ev_side[i], hipEventDisableTiming));
        SA_HIP(hipEventRecord(bt->ev_side[0], st));           // (the run's state is reset)
        for (int j = 0; j + 1 < n_lanes; j++) {
            if (!ix->lane_stream[j]) SA_HIP(hipStreamCreateWithFlags(&ix->lane_stream[j], hipStreamNonBlocking));
            SA_HIP(hipStreamWaitEvent(ix->lane_stream[j], bt->ev_side[0], 0));
        }
    }
    // slop phrases first, ALL of them in shared launches (sa_span_counts_batch: five launches per class of phrases instead of
    // five per phrase) and ranked by one launch; what that route does not take (an unknown term, very long lists) and
    // the exact dense-route phrases (repeated terms, more than 18 terms) follow one by one
    std::vector<unsigned char> taken(bt->B, 0);
    {
        std::vector<const u32*> tp;
        std::vector<int> tn, ts;
        std::vector<u32> rows;
        for (u32 row : bt->dense_rows)
            if (bt->h_pslop[row] > 0) { rows.push_back(row); tp.push_back(&bt->h_pterms[(size_t)row * bt->T]); tn.push_back(bt->h_pn[row]); ts.push_back(bt->h_pslop[row]); }
        if (!rows.empty()) {
            std::vector<float*> outs(rows.size(), nullptr);
            std::vector<unsigned char> handled(rows.size(), 0);
            std::vector<float> idfs(rows.size());
            for (size_t i = 0; i < rows.size(); i++) idfs[i] = bt->h_pidf[rows[i]];
            const sa_dense_rank_job* d_rj = nullptr;
            int n_rj = 0;
            SpanRankCtx rc;
            rc.doc_lens = ix->d_doc_lens; rc.avgdl = ix->avg_doc_len; rc.k1 = bt->k1; rc.b = bt->b; rc.k = bt->k;
            rc.slots = bt->d_slots; rc.cand = bt->d_cand; rc.cand_cap = bt->cand_cap; rc.cand_cnt = bt->d_cand_cnt; rc.doc_base = ix->doc_base;
            SA_TRY(sa_span_counts_batch(ix, st, (int)rows.size(), tp.data(), tn.data(), ts.data(), idfs.data(), rows.data(), outs.data(),
                                        handled.data(), &d_rj, &n_rj, bt->ptile == 2048 ? 11u : 12u, &rc));
            for (size_t i = 0; i < rows.size(); i++) if (handled[i]) taken[rows[i]] = 1;
            if (n_rj > 0) {
                if (bt->ptile == 2048)
                    hipLaunchKernelGGL((sa_k_dense_topk_tiles_multi<2048, SA_PTHREADS>), dim3(bt->pn_tiles, (u32)n_rj), dim3(SA_PTHREADS), 0, st,
                                       d_rj, (const float*)ix->d_doc_lens, ix->avg_doc_len, bt->k1, bt->b, ix->n_docs,
                                       ix->doc_base, bt->k, bt->d_slots, bt->d_cand, bt->cand_cap, bt->d_cand_cnt);
                else
                    hipLaunchKernelGGL((sa_k_dense_topk_tiles_multi<4096, SA_PTHREADS>), dim3(bt->pn_tiles, (u32)n_rj), dim3(SA_PTHREADS), 0, st,
                                       d_rj, (const float*)ix->d_doc_lens, ix->avg_doc_len, bt->k1, bt->b, ix->n_docs,
                                       ix->doc_base, bt->k, bt->d_slots, bt->d_cand, bt->cand_cap, bt->d_cand_cnt);
                ix->span_counts_dirty = false;
            }
        }
    }
    u32 nth = 0;
    for (u32 row : bt->dense_rows) {
        if (taken[row]) continue;
        const int j = (int)(nth++ % (u32)n_lanes) - 1;
        const hipStream_t ls = j < 0 ? st : ix->lane_stream[j];
        Lane lane(ix, j);
        float* d_scores = nullptr;
        SA_TRY(sa_phrase_dense_counts_device(ix, &bt->h_pterms[(size_t)row * bt->T], bt->h_pn[row], bt->h_pslop[row], &d_scores));
        if (bt->ptile == 2048)
            hipLaunchKernelGGL((sa_k_dense_topk_tiles<2048, SA_PTHREADS>), dim3(bt->pn_tiles), dim3(SA_PTHREADS), 0, ls, (const float*)d_scores,
                               (const float*)ix->d_doc_lens, ix->avg_doc_len, bt->h_pidf[row], bt->k1, bt->b,
                               ix->n_docs, ix->doc_base, row, bt->k, bt->d_slots, bt->d_cand, bt->cand_cap, bt->d_cand_cnt);
        else
            hipLaunchKernelGGL((sa_k_dense_topk_tiles<4096, SA_PTHREADS>), dim3(bt->pn_tiles), dim3(SA_PTHREADS), 0, ls, (const float*)d_scores,
                               (const float*)ix->d_doc_lens, ix->avg_doc_len, bt->h_pidf[row], bt->k1, bt->b,
                               ix->n_docs, ix->doc_base, row, bt->k, bt->d_slots, bt->d_cand, bt->cand_cap, bt->d_cand_cnt);
    }
    for (int j = 0; j + 1 < n_lanes; j++) {                    // the merge waits for every lane
        SA_HIP(hipEventRecord(bt->ev_side[1], ix->lane_stream[j]));
        SA_HIP(hipStreamWaitEvent(st, bt->ev_side[1], 0));
    }
    return SA_OK;
}

extern "C" int sa_phrase_batch_create(sa_index_t* ix, const uint32_t* terms, const int32_t* n_terms, const float* idf,
                                      int n_phrases, int max_terms, int k, float k1, float b, sa_batch_t** out) {
    return sa_phrase_batch_create_ex(ix, terms, n_terms, nullptr, idf, n_phrases, max_terms, k, k1, b, out);
}

// everything of a phrase batch that depends on the phrases: validation, the plan per phrase, the upload image, the
// word-slice table -- enqueued on the index stream, nothing allocated, nothing waited for
static int sa_phrase_batch_fill(sa_batch* bt, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop, const float* idf) {
    sa_index* ix = bt->ix;
    const u32 B = bt->B, T = bt->T;
    std::vector<u32> dense_rows;
    for (u32 i = 0; i < B; i++) {
        // reference middle_out.py:425-426
        if (n_terms[i] < 2) { sa_set_error("Must have at least two terms"); return SA_ERR_ARG; }
        SA_ARG(n_terms[i] <= (int)T, "n_terms[i] > max_terms");
        SA_ARG(!slop || slop[i] >= 0, "slop < 0");
        if (slop && slop[i] > 0 && n_terms[i] > 32) { sa_set_error("slop phrases support at most 32 terms"); return SA_ERR_UNSUPPORTED; }
        // the tile kernel takes exact phrases of up to 18 pairwise-distinct terms; everything else the reference's
        // score() accepts -- repeated terms (same-term rule, bigram_freqs.py:48-101), longer phrases, slop > 0
        // (spans.py:71-187) -- is scored through the dense single-phrase path and ranked on the device
        bool dense = n_terms[i] > SA_PHRASE_BATCH_MAXT || (slop && slop[i] > 0);
        for (int t = 0; t < n_terms[i] && !dense; t++)
            for (int u = 0; u < t; u++)
                if (terms[(size_t)i * T + t] == terms[(size_t)i * T + u] && terms[(size_t)i * T + t] < ix->n_terms) dense = true;
        if (dense) dense_rows.push_back(i);
    }
    char* img = nullptr;
    SA_TRY(sa_batch_upload_begin(bt, &img));
    auto at = [&](const void* dptr) { return img + ((const char*)dptr - bt->d_up); };
    u32* h_terms = (u32*)at(bt->d_terms);
    float* h_idf = (float*)at(bt->d_idf);
    u32* h_perm = (u32*)at(bt->d_perm);
    u32* plan = (u32*)at(bt->d_plan);
    bt->perm.resize(B);
    for (u32 i = 0; i < B; i++) { bt->perm[i] = i; h_perm[i] = i; }
    // plan per phrase (host): reference compute_phrase_freqs, middle_out.py:154-168
    memset(plan, 0, (size_t)B * 4 * sizeof(u32));
    for (size_t i = 0; i < (size_t)B * T; i++) h_terms[i] = SA_NO_TERM;
    memcpy(h_idf, idf, (size_t)B * sizeof(float));
    bt->alg_bytes = 0; bt->postings_bytes = 0;
    bt->dense_rows = dense_rows;
    bt->h_pterms.assign(terms, terms + (size_t)B * T);
    bt->h_pn.assign(n_terms, n_terms + B);
    bt->h_pslop.assign(B, 0);
    if (slop) bt->h_pslop.assign(slop, slop + B);
    bt->h_pidf.assign(idf, idf + B);
    std::vector<char> is_dense(B, 0);
    for (u32 r : dense_rows) is_dense[r] = 1;
    for (u32 i = 0; i < B; i++) {
        const int Tq = n_terms[i];
        if (is_dense[i]) {                             // plan[0] stays 0: the tile kernel skips the row
            for (int t = 0; t < Tq; t++) {
                const u32 term = terms[(size_t)i * T + t];
                if (term < ix->n_terms) bt->postings_bytes += 8 * (ix->h_term_off[term + 1] - ix->h_term_off[term]);
            }
            continue;
        }
        u64 lens[SA_PHRASE_BATCH_MAXT];
        bool known = true;
        for (int t = 0; t < Tq; t++) {
            const u32 term = terms[(size_t)i * T + t];
            h_terms[(size_t)i * T + t] = term;
            if (term >= ix->n_terms) { known = false; lens[t] = 0; continue; }
            lens[t] = ix->h_term_off[term + 1] - ix->h_term_off[term];
            bt->postings_bytes += 8 * lens[t];
        }
        if (!known) continue;                          // plan[0] == 0: no match anywhere
        int shortest = 0;
        for (int t = 1; t < Tq; t++) if (lens[t] < lens[shortest]) shortest = t;   // first shortest on ties
        const bool whole = shortest <= 1 || shortest >= Tq - 2;
        const u32 split = whole ? 0u : (u32)shortest;
        auto anchor_of = [&](int a, int e) {
            int best = a;
            for (int t = a + 1; t < e; t++) if (lens[t] < lens[best]) best = t;
            return (u32)(best - a);
        };
        plan[(size_t)i * 4 + 0] = (u32)Tq;
        plan[(size_t)i * 4 + 1] = split;
        plan[(size_t)i * 4 + 2] = split ? anchor_of(0, (int)split) : anchor_of(0, Tq);
        plan[(size_t)i * 4 + 3] = split ? anchor_of((int)split, Tq) : 0u;
    }
    bt->alg_bytes = bt->postings_bytes;
    SA_TRY(sa_batch_upload_commit(bt));
    {
        const u64 total = (u64)B * T * (bt->pn_tiles + 1);
        const u32 grid = total / 256 + 1 < 65535 ? (u32)(total / 256 + 1) : 65535u;
        hipLaunchKernelGGL(sa_k_make_word_bounds, dim3(grid), dim3(256), 0, ix->stream, ix->d_words, ix->d_term_off,
                           ix->n_terms, bt->pn_tiles, bt->ptile, (const u32*)bt->d_terms, B * T, bt->d_wbounds,
                           bt->d_wbase, bt->d_wlen);
    }
    SA_HIP(hipGetLastError());
    return SA_OK;
}

extern "C" int sa_phrase_batch_create_ex(sa_index_t* ix, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                                         const float* idf, int n_phrases, int max_terms, int k, float k1, float b,
                                         sa_batch_t** out) {
    SA_ARG(ix && out && terms && n_terms && idf, "null argument");
    SA_ARG(n_phrases > 0 && max_terms >= 2, "empty batch");
    SA_ARG(k > 0 && k <= SA_KMAX, "k must be in [1, 1024]");
    SA_ARG(ix->doc_base + ix->n_docs <= 0xFFFFFFFFull, "global doc ids must fit 32 bits for top-k");
    const u32 B = (u32)n_phrases, T = (u32)max_terms;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    sa_batch* bt = new (std::nothrow) sa_batch();
    if (!bt) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    bt->opts = sa_options_for_new_handle(&ix->opts);
    bt->ix = ix; bt->B = B; bt->T = T; bt->k = (u32)k; bt->k1 = k1; bt->b = b;
    bt->kind = 1;
    bt->st = ix->stream;           // (the dense route swaps lanes in and out of the index: phrase batches stay on its stream)
    bt->ptile = SA_PTILE;
    if (bt->opts.ptile == 2048 || bt->opts.ptile == 4096) bt->ptile = (u32)bt->opts.ptile;
    bt->pn_tiles = (u32)((ix->n_docs + bt->ptile - 1) / bt->ptile);
    auto alloc = [&]() -> int {
        // upload block: terms [B][T], idf [B], perm [B], plan [B][4]
        const size_t o_terms = 0, o_idf = o_terms + (size_t)B * T * 4, o_perm = o_idf + (size_t)B * 4, o_plan = o_perm + (size_t)B * 4;
        SA_TRY(sa_batch_alloc_upload(bt, o_plan + (size_t)B * 16));
        bt->d_terms = (u32*)(bt->d_up + o_terms); bt->d_idf = (float*)(bt->d_up + o_idf);
        bt->d_perm = (u32*)(bt->d_up + o_perm); bt->d_plan = (u32*)(bt->d_up + o_plan);
        SA_HIP(hipMalloc(&bt->d_wbounds, ((size_t)B * T * (bt->pn_tiles + 1) + 1) * sizeof(u32)));
        SA_HIP(hipMalloc(&bt->d_wbase, (size_t)B * T * sizeof(u64)));
        SA_HIP(hipMalloc(&bt->d_wlen, (size_t)B * T * sizeof(u32)));
        // (every wave of a ranking unit appends at most k keys: 4 waves per tile of the tile route -- and per block of 512 documents
        //  of the slop phrases that rank inside the span kernel, sa_k_span_doc_fused_multi: 4 such blocks per 2048-doc tile)
        SA_TRY(sa_batch_alloc_topk(bt, bt->pn_tiles, (SA_PTHREADS / SA_WAVE) * (bt->ptile / 512u)));
        return SA_OK;
    };
    int rc = alloc();
    if (rc == SA_OK) rc = sa_phrase_batch_fill(bt, terms, n_terms, slop, idf);
    if (rc == SA_OK && hipStreamSynchronize(ix->stream) != hipSuccess) {
        sa_set_error("sa_phrase_batch_create: hipStreamSynchronize failed");
        rc = SA_ERR_HIP;
    }
    if (rc != SA_OK) { sa_batch_free(bt); return rc; }
    *out = bt;
    return SA_OK;
}

// A new set of phrases (same number, same max_terms, k, k1, b) in an existing phrase batch: see sa_batch_reset.
extern "C" int sa_phrase_batch_reset(sa_batch_t* bt, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                                     const float* idf) {
    SA_ARG(bt && bt->ix && terms && n_terms && idf, "null argument");
    SA_ARG(bt->kind == 1, "sa_phrase_batch_reset takes a phrase batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    return sa_phrase_batch_fill(bt, terms, n_terms, slop, idf);
}

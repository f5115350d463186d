// sa_bm25_params.hpp -- kernel parameter blocks shared by the BM25 scoring kernels (sa_bm25.hip: per-query and
// grouped tile kernels).
#pragma once
#include "sa_index.hpp"

struct alignas(16) sa_u64x2 { u64 x, y; };

// Impact stream layout (sa_impacts, sa_index.hpp): first cell of term `term` whose TF postings start at tf_base
// (4 cells of slack per term: a term's last posting is followed by at least one whole, 16-byte-aligned pair
//  of sentinels before the next term starts -- the pair every load past a slice's term is clamped to)
__host__ __device__ __forceinline__ u64 sa_imp_base(u64 tf_base, u32 term) { return (tf_base + 4ull * term + 1ull) & ~1ull; }
// first cell of that sentinel pair, for a term of df postings starting at `ibase`
__host__ __device__ __forceinline__ u64 sa_imp_sentinel(u64 ibase, u64 df) { return ibase + ((df + 1ull) & ~1ull); }

struct Bm25Params {
    // index
    const u64* tfp;
    const u64* tf_off;
    const u32* dir_slot;
    const u32* tile_dir;
    const float* doc_lens;
    u32 n_terms, n_tiles;
    u64 n_docs, doc_base;
    int dl_packed;
    // batch
    const u32* terms;      // [B][T]
    const float* idf;      // [B][T]
    const float* sattab;   // [SA_SAT_NTF][tab_w] saturation table (sa_k_make_sattab)
    u32 tab_w;             // 0: no table (doc lengths not packed in the postings); else 64/128
    const u32* bounds;     // [B][T][n_tiles+1] slice table (sa_k_make_bounds)
    const u64* qbase;      // [B][T] posting base of each query term
    const u64* imp;        // impact stream (sa_impacts, sa_index.hpp) or null: score the TF postings
    const u64* qbase_imp;  // [B][T][2] impact stream: first cell of each query term, first cell of the sentinel pair behind it
    u64 imp_tail;          // a cell of the impact stream that is a sentinel whatever happens (its last pair)
    u32 B, T, k;
    u32 tile0, tile_end;   // tiles [tile0, tile_end) of this launch (sa_k_bm25_tiles)
    float k1, b, avgdl;
    int pruned;            // 1: wave-level selection against a global bound (MODE 1); 0: block-level selection (MODE 0)
    int no_topk;           // timing experiments only: skip the per-tile selection
    u32 cand_per_tile;     // general mode: candidate slots per (query, tile) = k
    u32 cand_cap;          // pruned mode: capacity of each query's append list
    u32* cand_cnt;         // pruned mode: [B] append cursors
    u32* slots;            // pruned mode: [B][32] pruning slots (score bits)
    u32* hist;             // pruned mode, k > 32: [B][SA_HBINS] score histograms (null: use the slots)
    u32* gthr;             // pruned mode, k > 32: [B] cached bound (score bits)
    const u32* seed;       // [B] the bound every query STARTS with (score bits; sa_k_make_bounds from the terms' rank tables), or null
    // dynamic pruning (MaxScore): per query the terms in ascending idf order and the score a doc
    // can reach at most from the j smallest-idf terms alone
    const float* ub;       // [B][T+1] upper bounds (ub[0] = 0), or null: exhaustive scoring
    const u32* ub_order;   // [B][T] query-term index of the j-th smallest idf
    const unsigned char* tf8;   // index dense tf rows [n_tf8_terms][n_docs]
    const u32* tf8_slot;   // [n_terms]
    u32* stats;            // diagnostics (sa_batch_stats): [B] candidates scored by the sparse path, or null
    const u32* qlist;      // queries to scan (after the sparse path took the others), or null: all B
    u32 nq;                // number of queries to scan (= B without a list)
    const u32* nq_dev;     // the same on the device (sa_k_bm25_tiles_list)
    // outputs
    float* dense_out;      // [B][n_docs] or null
    u64* cand;             // [B][n_tiles][k] composite keys (global doc ids) or null
};

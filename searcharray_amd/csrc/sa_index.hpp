// sa_index.hpp -- the HBM-resident index object behind the C ABI (include/searcharray_hip.h).
//
// Resident arrays (one GPU = one doc-range shard [doc_base, doc_base + n_docs), doc ids in
// the arrays are shard-local):
//   words     u64[W]      roaringish positional words, term-major (reference ArrayDict layout,
//                         reference searcharray/phrase/memmap_arrays.py:15-56)
//   term_off  u64[V+1]    words of term t = words[term_off[t] : term_off[t+1]]
//   tfp       u64[P]      derived "fat" TF postings, term-major, doc-sorted:
//                         doc << 36 | doc_len << 18 | tf   (28 + 18 + 18 bits)
//                         = what the reference caches per term as (doc_ids, term_freqs)
//                         (reference middle_out.py:501-512) with the doc length riding along so
//                         scoring is one coalesced 64-bit stream with no doc_lens gather.
//   tf_off    u64[V+1]    postings of term t; df_t = tf_off[t+1] - tf_off[t]
//   tile_dir  u32[S][n_tiles+1]  for the S terms with df >= dir_min_df: first posting (relative to
//                         tf_off[t]) whose doc >= tile * tile_docs -- lets a workgroup find its
//                         slice of a posting list with one load instead of a search.
//   dir_slot  u32[V]      row of term t in tile_dir, or 0xFFFFFFFF
//   doc_lens  f32[n_docs]
#pragma once
#include "sa_options.hpp"
#include "sa_common.hpp"
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

struct sa_comm;   // RCCL communicator wrapper (sa_comm.hip)

// Docs per scoring tile when the caller does not choose: 8 KiB of fp32 accumulators per ONE-wave workgroup (no
// cross-wave barrier at all); measured best for the per-query BM25 tile kernel (2048 / 4096 / 8192 docs: 1.27 /
// 1.33 / 1.44 ms per 256-query launch at 10 M docs, k = 1000: 1.83 / 1.93 / 2.25 ms), on a par for dynamic pruning.
#define SA_DEFAULT_TILE_DOCS 2048u
#define SA_DD_ABSENT 0xFFFFFFFFu
// words allocated behind the index's roaringish words (never written, never part of a list): kernels that fetch a document's words
// with wide loads may read up to this many words past the end of a term's list -- of the last term's too
#define SA_WORDS_PAD 8
struct sa_w2 { u64 x, y; };                     // (two words fetched by one 16-byte load from an 8-byte-aligned address)
#define SA_DD_NONE 0xFFFFFFFFu

// Impact stream of one (k1, b, avgdl) instantiation of BM25 (sa_bm25.hip, sa_k_make_impacts): the TF
// postings with the per-posting factor  tf / (tf + k1 * ((1 - b) + b * dl / avgdl))  already evaluated
// (reference bm25.pyx:19-23, op for op), laid out for the exhaustive tile kernel:
//   imp[i] = (doc * 4) << 32 | float bits of the factor      (doc * 4: the byte offset of the doc's
//            fp32 accumulator once the tile base is subtracted)
// Term t starts at the EVEN index  (tf_off[t] + 4 t + 1) & ~1 ; every gap holds the sentinel
// 0xFFFF'FFFF'FFFF'FFFF (doc id no tile contains) and every term is followed by at least one whole
// 16-byte-aligned pair of sentinels, so a 16-byte pair load never sees another term's posting, loads
// can be clamped to that pair instead of bounds-tested per lane, and "doc inside this tile" is the
// only validity test a posting needs.
#define SA_TOPF_NR 22
static const unsigned sa_topf_ranks[SA_TOPF_NR] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64, 100, 128, 200, 256, 512, 1000, 1024};

// Stage directory (sa_stage.hip) of one impact stream, for the staged-tile route's own tiles of `docs` documents.  For every
// term with at least one posting per eight tiles (a ROW):
//   abs[row][j]   first posting (relative to the term's base) whose doc >= j * docs -- where a workgroup's cursor starts
//   cm[row][j]    postings of the term in tile j (low 16 bits) | the LARGEST factor among them, its fp32 pattern's upper 16 bits
//                 rounded up (high 16 bits): what the term can add to a score in that tile at most, per unit of weight
// Rarer terms are walked by the kernel's cursors and bounded by their largest factor in the shard.  Built on first use, cached
// per tile size with the stream.
struct sa_stagedir {
    int device = 0;
    u32 docs = 0, n_st = 0, n_rows = 0;
    u32* d_dir = nullptr;           // [n_rows][n_st + 1][2]: {abs, cm} of a (term, tile) side by side -- one 8-byte load gives the kernel the slice's
                                    // start, size and bound (the entry behind a row's last tile: abs = df, cm = 0)
    std::vector<u32> row;           // [n_terms] row of a term, or 0xFFFFFFFF
    ~sa_stagedir();
};

struct sa_impacts {
    int device = 0;
    float k1 = 0.f, b = 0.f, avgdl = 0.f;
    u64* d_imp = nullptr;
    u64 n = 0;                      // u64 cells incl. padding
    // Dense factor rows of the most frequent terms (df >= n_docs / 4): dense[slot][doc] = the term's factor in doc, 0 where
    // the doc lacks it.  The grouped kernel builds the BASE of a (tile, group) item from the row of the group's shared
    // first term -- lane = doc, 16-byte loads, a multiply and a 16-byte LDS store per four docs -- instead of unpacking and
    // scattering up to 1900 postings per tile; for such terms the row is no more bytes than the postings (4 B per doc
    // against 8 B per posting).
    float* d_dense = nullptr;
    u64 dense_stride = 0;           // floats per row (n_docs rounded up to whole tiles)
    std::vector<u32> dense_slot;    // [n_terms] row of a term, or 0xFFFFFFFF (host)
    u32 n_dense = 0;
    // Rank table of every term's factors (sa_k_make_topf): topf[t][i] = a lower bound (bin edge, within 0.2 %) of the
    // SA_TOPF_RANK(i)-th LARGEST factor among the term's postings in this shard, 0 where the term has fewer postings or
    // the factor is below 1/16.  w_t * topf[t][rank >= k] is a bound of the k-th best score of ANY query that holds t
    // with weight w_t >= 0 (all contributions are non-negative and fp32 sums of non-negatives never fall below a
    // summand): every query starts with the best such bound over its terms instead of 0 (sa_k_make_bounds).
    float* d_topf = nullptr;        // built by the first batch that can use it (sa_impacts_ensure_topf)
    bool topf_tried = false;
    // round 6 (staged-tile route, sa_stage.hip): the LARGEST factor of every term, exact (an upper bound of what a
    // posting of the term can contribute per unit of weight), built with the rank tables; host copies of both, so
    // that a query set's starting bounds and its terms' score bounds are formed on the host with the upload
    float* d_maxf = nullptr;        // [n_terms]
    std::vector<float> h_topf, h_maxf;
    // PROBE ROWS (staged-tile route): dense factor rows like d_dense, for every term with df >= n_docs / probe_div (default 128;
    // as many as a quarter of the free HBM takes, most frequent first).  A term that cannot be essential for a query
    // is not streamed at all: the few documents that survive the bound test on the query's other terms read its factor with
    // one 4-byte load from its row.  Built by the first batch that plans the route.
    float* d_probe = nullptr;
    u32* d_pbits = nullptr;         // presence bitmaps of the same terms: [n_probe][pbits_words] (bit doc & 31 of word doc >> 5) -- what a tile of the
                                    // staged-tile kernel stages of a probed term, so that only the documents that HOLD it wait for a probe
    u64 pbits_words = 0;            // words per bitmap row (n_docs / 32 rounded up to 32 words)
    u64 probe_stride = 0;           // floats per row (n_docs rounded up to 64)
    std::vector<u32> probe_slot;    // [n_terms] row of a term, or 0xFFFFFFFF (host)
    u32 n_probe = 0;
    bool probe_tried = false;
    std::vector<std::shared_ptr<sa_stagedir>> stagedirs;   // stage directories built so far (one per tile size in use)
    ~sa_impacts();
};

struct sa_index;
struct SaDenseLaneScope;
struct sa_index {
    sa_options_t opts;              // the handle's switches (sa_options.hpp): the creating thread's defaults, or sa_index_set_options
    int device = 0;
    int n_cus = 1;                  // compute units of the device (persistent grid sizing)
    hipStream_t stream = nullptr;
    u64 n_docs = 0, doc_base = 0, corpus_size = 0;
    u32 n_terms = 0;
    float avg_doc_len = 0.f;
    u64 n_words = 0, n_postings = 0;
    bool dl_packed = true;          // doc lengths are integers < 2^18 and ride in the postings
    u32 max_doc_len = 0;            // largest doc length (when dl_packed)

    u64* d_words = nullptr;
    u64* d_term_off = nullptr;
    u64* d_tfp = nullptr;
    u64* d_tf_off = nullptr;
    float* d_doc_lens = nullptr;

    u32 tile_docs = 0, n_tiles = 0, dir_min_df = 0, n_dir_terms = 0;
    u32* d_tile_dir = nullptr;
    u32* d_dir_slot = nullptr;

    // Doc directory of the frequent terms (phrase probes): docdir[slot][doc] = index, relative to the
    // term's first word, of the first roaringish word of `doc` in that term, or SA_DD_ABSENT.  Finding a
    // term's words of a given doc becomes one 4-byte load instead of a binary search of the term's list.
    u32 n_dd_terms = 0;
    u32* d_docdir = nullptr;         // [n_dd_terms][n_docs]
    u32* d_dd_slot = nullptr;        // [n_terms] directory row of a term, or SA_DD_NONE

    // Dense term-frequency rows of the frequent terms (dynamic pruning, sa_bm25.hip): tf8[slot][doc] =
    // min(tf, 255), 0 when the doc lacks the term.  Scoring a handful of candidate docs against a
    // frequent term is then one byte load per doc instead of streaming the term's postings.
    u32 n_tf8_terms = 0;
    unsigned char* d_tf8 = nullptr;  // [n_tf8_terms][n_docs]
    u32* d_tfbits = nullptr;         // [n_tf8_terms][tfbits_words] presence bitmap of the same terms (cache-resident probes)
    u64 tfbits_words = 0;            // u32 words per bitmap row = ceil(n_docs / 32)
    u32* d_tf8_slot = nullptr;       // [n_terms] row of a term, or SA_DD_NONE

    std::vector<u64> h_term_off, h_tf_off;
    std::vector<float> h_idf;        // idf of every term as the host formed it (sa_index_set_idf_table; sa_batch_step gathers from it)
    std::vector<u32> h_dd_slot;      // host copy of d_dd_slot
    std::vector<unsigned char> h_term_edge;   // [n_terms] bit 0: first word has header 0, bit 1: last word has the largest header
    bool any_top_block = true;       // some word of the index sits in a document's LAST 18-position block (positions >= 4.7 M:
                                     //   hardly ever) -- only then can `header + 1` of a word be another document's block 0
    std::vector<u32> h_dd_top;       // [n_dd_terms] 1: the term has a word in the last 18-position block (sa_spans.hip)
    std::vector<u32> h_tf8_slot;     // host copy of d_tf8_slot

    // reusable device scratch (grown on demand, guarded by mu)
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // further lanes of batched dense-route phrases (sa_phrase_batch.hip swaps a lane's stream and scratch in around
    // the single-phrase kernels): SA_PHRASE_LANES - 1 of them
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};
    void* lane_scratch[3] = {nullptr, nullptr, nullptr};
    size_t lane_bytes[3] = {0, 0, 0};

    // LANES of the dense single-call entry points (sa_index_bm25_dense / sa_index_termfreqs_dense and their `_to` twins): a call
    // holds the index lock only while it ENQUEUES -- on a lane's stream, into a lane's scratch -- and waits for its result
    // outside the lock, so the kernels and device-to-host copies of several threads' calls on ONE handle overlap, the way the
    // reference's GIL-releasing native kernels overlap under a ThreadPoolExecutor (reference test/test_tmdb.py:285-312,
    // roaringish/intersect.pyx:306,316,338,384).  struct SaDenseLaneScope (sa_index.hip) swaps a lane in and out.
    struct DenseLane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        void* scratch = nullptr;
        size_t scratch_bytes = 0;
        void* rows = nullptr;
        size_t rows_bytes = 0;
        void* sim = nullptr;        // float64 results of the f64 similarities: per lane, a call's D2H copy is still in flight when the lock is dropped
        size_t sim_bytes = 0;
        bool busy = false;
    };
    static constexpr int N_DENSE_LANES = 8;
    DenseLane dense_lane[N_DENSE_LANES];
    std::condition_variable dense_lane_cv;

    // slop phrases of a phrase batch in shared launches (sa_span_counts_batch): job array + per-phrase scratch, and the
    // page-locked image the jobs are uploaded from
    void* d_span_batch = nullptr;
    size_t span_batch_bytes = 0;
    void* h_span_jobs = nullptr;
    size_t span_jobs_host_bytes = 0;
    hipEvent_t ev_span_jobs = nullptr;
    float* d_span_counts = nullptr;  // pool of dense count vectors of that route, all zeros between runs (the ranking launch cleans up)
    size_t span_counts_cap = 0;      // floats
    bool span_counts_dirty = false;
    u32* h_flags = nullptr;          // 128 page-locked words: small device results the host decides on (sa_phrase.hip)
    u32* d_flags = nullptr;          // their device side: 128 words, all zeros between uses (sa_k_flags_out clears them)

    // row selection scratch (sa_index_select_rows): device copy of the selected doc ids + gathered values
    void* d_rows_scratch = nullptr;
    size_t rows_scratch_bytes = 0;
    void* d_sim_scratch = nullptr;              // float64[n_docs] results of the f64 similarities
    size_t sim_scratch_bytes = 0;

    // most recent impact stream (shared with the batches built for the same k1 / b)
    std::shared_ptr<sa_impacts> impacts;

    sa_comm* comm = nullptr;
    hipStream_t sstream = nullptr;   // side stream: the per-query kernel over a batch's ungrouped rows runs beside the grouped kernel
    hipStream_t xstream = nullptr;   // exchange stream: all-gather + cross-rank merge overlap the next batch's scoring

    // profile counters (sa_index_stats)
    double last_kernel_ms = 0.0;
    u64 last_alg_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profile_pending = false;

    std::mutex mu;                  // one in-flight call per index handle (C ABI is re-entrant
                                    // across handles and serialised per handle)
};

int sa_index_scratch(sa_index* ix, size_t bytes, void** out);
// Copy a dense float[n_docs] device vector to the host `out` -- or, if this thread selected rows for
// this index (sa_index_select_rows), only those rows: out[i] = d_vec[rows[i]].  Enqueued on the index
// stream; the caller synchronises.  Call with the index lock held.
int sa_emit_dense(sa_index* ix, const float* d_vec, float* out);
// the same for an all-zero result (no device work)
void sa_emit_zeros(sa_index* ix, float* out);
// sa_vec.hip: divert the result into a device vector if this thread asked for it (sa_index_select_vec)
bool sa_emit_to_vec(sa_index* ix, const float* d_vec);
bool sa_vec_target_pending(const sa_index* ix, bool clear);
bool sa_vec_target_take(const sa_index* ix, float** dst, float* boost, int* has_boost);
// index construction pieces shared by sa_index_create (sa_index.hip) and sa_index_create_from_tokens (sa_build.hip)
int sa_index_setup(sa_index* ix, const float* doc_lens);
int sa_index_derive(sa_index* ix);
void sa_index_free(sa_index* ix);
// min_posn / max_posn restriction of the positional words (reference roaringish.py:266-282)
struct PosnFilter {
    bool active = false;
    u64 lo = 0, hi = 0xFFFFFFFFFFFFFFFFull;       // compared with the UNSHIFTED (word & msb_mask), as the reference does
};
int sa_posn_filter_bounds(int64_t min_posn, int64_t max_posn, PosnFilter* f);
int sa_posn_filter_terms(sa_index* ix, const PosnFilter& f, int T, const u64** ptrs, u32* lens, u64** bufs,
                         u32* d_counts, u32* d_chunks);
// slop > 0 phrase counts (sa_spans.hip): dense float[n_docs] inside the index scratch
// dense phrase counts (any number of terms up to 128, repeated terms, slop) and counts -> BM25 (sa_phrase.hip)
int sa_phrase_dense_counts_device(sa_index* ix, const u32* terms, int n_terms, int slop, float** d_out);
void sa_launch_bm25_from_tf(sa_index* ix, float* d_tf, float idf, float k1, float b);
int sa_span_counts_device(sa_index* ix, const u32* terms, int T, int slop, const PosnFilter& filt, float** d_out);
// one phrase of the dense route to rank: dense counts (float[n_docs]), the phrase's idf, its row in the batch
// (touched[tile]: the vector has a count in that ranking tile -- set by the span machines, cleared by the ranking launch,
//  which leaves the other tiles' 8 KB unread)
struct sa_dense_rank_job { const float* counts; unsigned char* touched; float idf; u32 row; };
// the ranking state of a phrase batch, for kernels that rank their documents themselves (sa_k_span_doc_fused_multi): BM25
// parameters, the batch's bound slots and candidate lists (cand == null: nothing to rank into)
struct SpanRankCtx {
    const float* doc_lens; float avgdl, k1, b; u32 k;
    u32* slots; u64* cand; u32 cand_cap; u32* cand_cnt; u64 doc_base;
};
int sa_span_counts_batch(sa_index* ix, hipStream_t st, int n, const u32* const* terms, const int* T, const int* slop,
                         const float* idf, const u32* rows, float** d_out, unsigned char* handled,
                         const sa_dense_rank_job** d_rank_jobs, int* n_rank_jobs, u32 rank_tile_shift, const SpanRankCtx* rank);

// a dense call on one of the index's lanes (sa_index.hip)
struct SaDenseLaneScope {
    sa_index* ix;
    std::unique_lock<std::mutex>& lk;
    sa_index::DenseLane* lane = nullptr;
    bool swapped = false;
    int rc = SA_OK;
    SaDenseLaneScope(sa_index* ix_, std::unique_lock<std::mutex>& lk_);
    ~SaDenseLaneScope();
    int finish();
    void swap();
};

// sa_batch.hpp -- a resident batch of top-k queries (BM25 term disjunctions or exact phrases):
// per-batch tables, candidate lists, the per-shard merge and the cross-rank exchange buffers.
#pragma once
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include <vector>
#include <time.h>

#define SA_MAX_QTERMS 32
#define SA_KMAX 1024
#define SA_EVENT_RING 128
// postings per work item of the sparse candidate path (the lead phase of a small shard takes the small size)
#define SA_SP_CHUNK_LEAD 256
#define SA_SP_CHUNK 1024

// the plan of one slice of a query set (at most 256 device rows) on the staged-tile route: one launch (sa_stage.hip)
struct sa_stagedir;
struct sa_stage_slice {
    u32 q0 = 0, nq = 0;             // device rows [q0, q0 + nq)
    u32 U = 0, NS = 0;              // distinct terms of the slice; the first NS are staged, the others probed in their probe rows
    u32 docs = 0, tmax = 4;         // docs per stage tile; kernel instantiation (4 or 8 terms per query)
    u32 imp_bytes = 0;              // bytes of the stream from cell_base to the end of the slice's last staged term
    u64 cell_base = 0;              // smallest impact-stream cell of the slice's staged terms (the kernel's 32-bit offsets count from it)
    float cand_per_doc = 0.f;       // candidates per document the plan expects
    bool all_rowed = true;          // every staged term has a stage-directory row (the launch may let workgroups co-walk tile ranges)
    std::shared_ptr<sa_stagedir> dir;
};

// working arrays of sa_stage_plan (sa_stage.hip), kept in the batch between query sets
struct sa_stage_scratch {
    std::vector<u32> tmap;                     // [n_terms] term -> distinct index while a plan runs, all ones otherwise
    std::vector<unsigned short> di;            // [B][T] distinct index of a slot's term
    std::vector<float> ubs, seeds;             // [B][T] bound of the slot's term x weight; [B] starting bounds
    std::vector<unsigned char> probed, staged; // [B][T] the slot may be probed; [U] the term must be staged
    std::vector<u64> dist_df;
    std::vector<u32> dist_term, dist_probe, order, rank;
    std::vector<std::pair<u64, u32>> keys;
};

struct sa_batch {
    sa_options_t opts;              // the batch's switches: the creating thread's defaults, else its index's; sa_batch_set_options
    sa_index* ix = nullptr;
    hipStream_t st = nullptr;       // the stream this batch's work is enqueued on: its own (BM25 batches), or the index stream
    bool own_stream = false;
    bool state_clean = false;       // per-run state (bound slots, cursors, histograms, work-list cursor) is zero: the last merge left it so
    u32 B = 0, T = 0, k = 0;
    float k1 = 1.2f, b = 0.75f;
    std::vector<u32> perm;          // device row r holds caller query perm[r] (results: caller order)
    u32* d_seed = nullptr;          // [B] (upload block, zeroed by the host) the queries' starting bounds, raised by sa_k_make_bounds
    bool seed_on = false;           // the current query set has them (non-negative weights, k1 >= 0, 0 <= b <= 1, rank tables built)
    u64 host_ns[4] = {0, 0, 0, 0};  // sa_batch_host_times
    std::vector<float> step_idf;    // sa_batch_step: the query set's weights, gathered from the index's idf table
    // Everything a NEW set of queries changes on the device is one contiguous UPLOAD BLOCK (d_up) with a
    // page-locked host image: sa_batch_reset fills the image and enqueues ONE hipMemcpyAsync (+ the slice-table
    // kernel) -- no allocation, no blocking copy, no synchronisation.  The pointers below (d_terms ... d_bloom_off
    // of a BM25 batch; d_terms, d_idf, d_perm, d_plan of a phrase batch) point into it.  Two images alternate so
    // that a reset can be prepared while the copy of the previous one may still be in flight.
    char* d_up = nullptr;
    char* h_up[2] = {nullptr, nullptr};
    size_t up_bytes = 0;
    hipEvent_t ev_up[2] = {nullptr, nullptr};   // image i has been copied
    bool up_used[2] = {false, false};
    u32 up_n = 0;                   // resets so far (image = up_n & 1)
    // results to the host without a stream synchronisation: every run ends with an async copy of the B*k keys and
    // the overflow flag into a page-locked buffer on the exchange stream; sa_batch_fetch waits for ITS event only
    u64* h_res = nullptr;           // [B*k + 2]: keys, this shard's overflow flag, the ranks' overflow flag
    hipEvent_t ev_final = nullptr;  // d_final written (the stream of the last merge)
    hipEvent_t ev_res = nullptr;    // h_res written (exchange stream)
    bool res_pending = false;
    bool unfetched = false;         // a run's results have been queued for the host and not been fetched yet
    u32* d_xflag = nullptr;         // sharded: OR over the ranks of the overflow flags (travels with the all-gather); behind d_final
    u32 wl_cap = 0;                 // entries of d_wl
    size_t bloom_cap = 0;           // bytes of d_bloom (what the query sets so far needed x 1.5; grows in sa_batch_ensure_bloom)
    u32* d_terms = nullptr;
    u32* d_perm = nullptr;
    float* d_idf = nullptr;
    u64* d_cand = nullptr;          // [B][n_tiles][waves*k]: per-tile blocks, or per-query append lists
    float* d_sattab = nullptr;      // saturation table of this batch's (k1, b, avgdl)
    u32 tab_w = 0;
    u32* d_bounds = nullptr;        // [B][T][n_tiles+1] slice table
    u64* d_qbase = nullptr;         // [B][T]
    std::shared_ptr<sa_impacts> impacts;   // impact stream of this batch's (k1, b), or null (tile kernel reads the TF postings)
    u64* d_qbase_imp = nullptr;     // [B][T] base of each query term in the impact stream
    u32 cand_cap = 0;               // keys per query in d_cand
    bool cap_limited = false;       // cand_cap below the worst case: overflow must be checked
    u32* d_cand_cnt = nullptr;      // [B] append cursors (pruned selection)
    u32* d_slots = nullptr;         // [B][32] pruning slots
    u32* d_gthr = nullptr;          // [B] cached histogram bound (k > 32), inside the d_slots allocation
    u32* d_hist = nullptr;          // [B][SA_HBINS] score histograms (k > 32), inside the d_slots allocation
    float* d_ub = nullptr;          // [B][T+1] dynamic pruning: score bound of the j smallest-idf terms
    u32* d_ub_order = nullptr;      // [B][T] query-term index of the j-th smallest idf
    u32* d_stats = nullptr;         // diagnostics (sa_batch_stats), null unless enabled
    // sparse candidate path (sa_sparse.hip)
    u32* d_lead = nullptr;          // [B] query-term index of the lead term, or 0xFFFFFFFF
    u64* d_p1_off = nullptr;        // [B+1] prefix sums of the lead terms' df
    u32* d_route = nullptr;         // [B] 0: answered by the sparse path, 1: tiles
    u32* d_emask = nullptr;         // [B] essential query terms
    u64* d_p2_off = nullptr;        // [B+1] prefix sums of the phase-2 candidates
    u32* d_tile_q = nullptr;        // [B] queries left to the tile kernel, then [1] their number, [1] survivor count
    u32* d_qdf = nullptr;           // [B][T] postings of each query term in this shard
    u32* d_qrow8 = nullptr;         // [B][T] dense tf row of each query term
    u64* d_surv = nullptr;          // phase-2 survivors
    u32* d_bloom = nullptr;         // Bloom filters of the lead terms' docs (bytes), query q at d_bloom_off[q]
    u64* d_bloom_off = nullptr;     // [B]
    u32* d_bloom_shift = nullptr;   // [B] hash >> shift = cell
    size_t bloom_bytes = 0;
    u32 surv_cap = 0;
    u64 sparse_p1_total = 0, sparse_limit2 = 0, sparse_p2_max = 0;
    u32 sparse_chunk1 = SA_SP_CHUNK_LEAD;
    bool sparse_ok = false;         // tables built and the scoring formula admits the idf bound
    bool sparse_lazy = false;       // the pruning tables of the current query set have not been derived (sa_batch_fill_prune_tables on demand)
    u32* d_overflow = nullptr;      // set by the merge kernel when a candidate list ran over (checked at fetch); behind d_final
    u64* d_local = nullptr;         // [B][k] per-shard result
    u64* d_gather = nullptr;        // [2][nranks][B][k] (multi-GPU, double-buffered like d_xlocal)
    u64* d_xlocal = nullptr;        // [2][B][k] per-shard results handed to the exchange stream
    hipEvent_t ev_side[2] = {nullptr, nullptr};       // side stream: [0] the run's state is reset (index stream), [1] the ungrouped rows are scored (side stream)
    hipEvent_t ev_scored[2] = {nullptr, nullptr};     // d_xlocal[b] written (index stream)
    hipEvent_t ev_exchanged[2] = {nullptr, nullptr};  // d_xlocal[b] / d_gather[b] consumed (exchange stream)
    bool exchanged_valid[2] = {false, false};
    u32 xstep = 0;
    int gather_ranks = 0;
    u64* d_final = nullptr;         // [B][k]
    u64* d_xcand = nullptr;         // [B][nranks*k] regrouped gather
    int xcand_ranks = 0;
    std::vector<hipEvent_t> ev0, ev1;   // ring of (start, stop) events around the scoring kernel
    u32 ev_n = 0;                       // runs recorded since the last sa_batch_profile
    u64 alg_bytes = 0, postings_bytes = 0;
    bool ran = false;
    // grouped exhaustive scoring (sa_k_bm25_group_tiles): device rows [0, n_grouped_rows) belong to groups --
    // first the groups of queries that share their first term (rows [0, n_shared_rows)), then the loose groups;
    // the ungrouped rows follow
    u32 n_shared_rows = 0;
    u32* d_grp = nullptr;           // [n_groups][2] first row, rows (bit 31: a loose group)
    u64* d_wl = nullptr;            // (tile, row) items the grouped kernel leaves to the per-query kernel
    u32* d_wl_cnt = nullptr;
    u32* d_iota = nullptr;          // [B] 0 .. B-1 (query lists of the per-query kernel: rows [a, b) = d_iota + a)
    u32 n_groups = 0, n_grouped_rows = 0, grp_tt = 1, grp_tt_shift = 0, grp_cq = 16;
    bool last_route_sparse = false; // the last run took dynamic pruning (sa_batch_last_route)
    // staged-tile route (sa_stage.hip, round 6): the plan of the current query set lives in the upload block
    bool stage_ok = false;          // the current query set has a plan (sa_stage_plan)
    bool last_route_stage = false;  // the last run took the staged-tile route
    bool bounds_valid = false;      // d_bounds / d_qbase / d_qbase_imp hold the current query set's slice table (the staged route does not need it)
    u32 st_U = 0, st_NS = 0;        // distinct terms of the query set; the first st_NS are staged, the others probed in their probe rows
    u32 st_docs = 0;                // docs per stage tile
    float st_cand_per_doc = 0.f;    // candidates (postings of essential terms) per document the plan expects, all queries together
    u32 st_imp_bytes = 0;           // bytes of the stream from st_cell_base to the end of the set's last term
    u64 st_cell_base = 0;           // smallest impact-stream cell of the set's terms (the kernel's 32-bit offsets count from it)
    u32 st_tmax = 4;                // kernel instantiation: 4 or 8 terms per query
    std::shared_ptr<sa_stagedir> st_dir;
    std::vector<sa_stage_slice> st_slices;   // the current set's plan, slice by slice (empty: none); the st_* fields above repeat slice 0
    sa_stage_scratch st_work;       // the plan's working arrays (kept between plans: a plan allocates nothing)
    char* d_st = nullptr;           // the plan's region of the upload block (sa_stage_bind carves it)
    size_t st_bytes = 0;
    // phrase batches (sa_phrase_batch.hip): kind == 1
    int kind = 0;                   // 0: disjunctive BM25 over terms, 1: exact phrases
    u32 ptile = 0, pn_tiles = 0;    // docs per phrase tile and their number
    u32* d_plan = nullptr;          // [B][4]: n_terms, split, anchor of part 0, anchor of part 1
    u32* d_wbounds = nullptr;       // [B][T][pn_tiles+1] first word of the term in each tile (relative)
    u64* d_wbase = nullptr;         // [B][T] word base of each phrase term
    u32* d_wlen = nullptr;          // [B][T] words of each phrase term
    // phrases the tile kernel does not take (repeated terms, more than 18 terms, slop > 0): scored one after the
    // other through the dense single-phrase path, ranked on the device
    std::vector<u32> dense_rows;    // batch rows on the dense route
    std::vector<u32> h_pterms;      // [B][T] host copy of the terms
    std::vector<int> h_pn, h_pslop; // [B] terms / slop per phrase
    std::vector<float> h_pidf;      // [B]
};


int sa_comm_allgather_topk(sa_index* ix, const u64* d_local, u64* d_gather, size_t count, int* nranks_out,
                           hipStream_t st);

int sa_comm_allreduce_max_u32(sa_index* ix, u32* d_val, hipStream_t st);

// tile scoring + pruned selection of a phrase batch on stream st (sa_phrase_batch.hip)
int sa_launch_phrase_tiles(sa_batch* bt, hipStream_t st);
// shared by the two batch kinds (sa_bm25.hip)
int sa_batch_alloc_topk(sa_batch* bt, u32 n_tiles, u32 waves);
// the upload block (sa_batch.hpp: d_up / h_up): allocate, take the next host image, enqueue its copy
int sa_batch_alloc_upload(sa_batch* bt, size_t bytes);
// host time of the step's parts, cumulative nanoseconds (sa_batch_host_times): [0] sa_batch_fill up to the upload (grouping, pruning
// tables: CPU only), [1] its enqueues (upload copy + bounds launch), [2] sa_batch_run's enqueues, [3] number of fills
static inline u64 sa_now_ns() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (u64)ts.tv_sec * 1000000000ull + (u64)ts.tv_nsec;
}
int sa_batch_upload_begin(sa_batch* bt, char** image);
int sa_batch_upload_commit(sa_batch* bt);
void sa_batch_free(sa_batch* bt);
// staged-tile route (sa_stage.hip): bytes of a (B, T) plan in the upload block; plan a query set (device-row order) into the
// host image -- sets bt->stage_ok; launch the scoring kernel of a planned set
size_t sa_stage_upload_bytes(u32 B, u32 T);
int sa_stage_plan(sa_batch* bt, char* img, const u32* row_terms, const float* row_idf);
int sa_launch_stage(sa_batch* bt, const struct Bm25Params& p, hipStream_t st);
// dynamic pruning: lead-term candidates, routing, remaining essential candidates (sa_sparse.hip)
int sa_launch_sparse(sa_batch* bt, hipStream_t st);

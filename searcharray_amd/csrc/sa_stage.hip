// sa_stage.hip -- the STAGED-TILE route of BM25 top-k batches (round 6).
//
// Replaces, for a batch of B queries, the reference's caller loop
//     np.sum([arr.score(t) for t in query], axis=0)   (test/test_msmarco.py:353-354; score = postings.py:652-680 ->
//     as_dense roaringish_ops.pyx:84-98 -> _bm25_score bm25.pyx:11-25)   ->   np.argpartition (utils/sort.py:24)
// like sa_k_bm25_group_tiles does, with a different DECOMPOSITION:
//
//   * the unit of work is a TILE of `docs` documents, not a (tile, query) pair.  A persistent workgroup walks a contiguous
//     range of tiles.  Per tile it STAGES the slices of the batch's distinct terms -- the impact-stream postings
//     doc*4 << 32 | fp32 factor , doc-sorted -- from HBM into LDS, once, whatever the number of queries that share a term,
//     with coalesced 8-byte loads (the slice of the stream a tile needs is contiguous);
//   * the queries are then answered FROM LDS.  A query starts with a bound G that at least k documents are known to
//     reach (rank tables of its terms, then the histogram of the documents found so far: sa_topk.hpp).  What a term can add
//     to a score in THIS tile is bounded by weight x (largest factor among its postings of the tile) -- a block maximum the
//     stage directory holds per (term, tile).  The query's terms are kept in descending order of their bound in the shard;
//     the terms at the END of that order whose tile bounds TOGETHER stay below G are non-essential: a document that holds
//     only such terms cannot reach G (fp32 sums of non-negatives are monotone; the rounding of the sum is covered by a
//     margin).  Every other document of the tile with a chance appears in the slice of an ESSENTIAL term, so the
//     candidates of a (tile, query) pair are the postings of its essential terms in this tile -- a few documents instead
//     of every posting of every term;
//   * a term that cannot be essential for ANY query of the batch -- by its bound in the shard and the queries' starting
//     bounds -- and has a PROBE ROW (a dense fp32 factor row, sa_impacts::d_probe) is not streamed at all: it only
//     enters the bounds, and the few documents that pass the test on the staged terms read its factor with one load;
//   * candidates of all queries of the tile are flattened into one work list and taken one per LANE: the lane looks
//     the document up in the query's other staged terms (binary search of the staged slices) in descending-bound order and
//     gives up as soon as what it has found plus what the remaining terms could add stays below G; a document found in
//     an essential term of higher priority is that term's candidate (no document is evaluated twice);
//   * the few documents that pass are scored EXACTLY as the reference does -- factor * idf per term, each product
//     rounded to fp32, summed in QUERY-TERM order ((s0 + s1) + s2) + s3, bm25.pyx:19-23 + the caller's np.sum -- and
//     appended to the query's candidate list above its bound, exactly like the other scoring kernels do
//     (histogram bound, merge kernel and redo rules are shared: sa_bm25.hip).
//
// Nothing here is approximate: a document is dropped only when an upper bound of its exact score is below a lower
// bound of the k-th best score.  Results are bit-identical to the other routes and to the oracle.
//
// Roofline: HBM-bound streaming of the staged posting lists (8 bytes per posting, once per batch) -- integer /
// compare work and a scalar fp32 multiply-add per looked-up posting, no MFMA.
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_bm25_params.hpp"
#include "../../include/searcharray_hip.h"

#include <algorithm>
#include <math.h>
#include <new>

#define SA_ST_NT 512            // threads per workgroup
#define SA_ST_UMAX 1024         // distinct terms of a query set
#define SA_ST_BMAX 256          // queries of a query set
#define SA_ST_BW 32             // words of such a bitmap (tiles of at most 1024 docs)
#ifndef SA_ST_REFRESH_MASK
#define SA_ST_REFRESH_MASK 31    // a query's bound is re-derived from its histogram whenever its candidate list crosses a multiple of this + 1
#endif
#define SA_ST_REF 64            // queries whose bound is re-derived at the end of a tile pass
#define SA_ST_NONE 0xFFFFu      // "no term" in the queries' term tables
#define SA_ST_PROBE 0xFFFFu     // s_off: the term is not staged; its factors are probed in its probe row (high half: the row)
#define SA_ST_NOROW 0xFFFFFFFFu
#define SA_ST_MARGIN 1.0000153f // 1 + 2^-16: covers the fp32 roundings of a sum of up to 8 non-negative terms taken in another order

// 8-byte cells an LDS stage holds (TMAX = 4: the BASELINE shape; 8: wider query tables, smaller stage); two workgroups per CU
template <int TMAX> struct SaStCap { static constexpr int v = TMAX <= 4 ? 2304 : 1024; };
// probed terms whose presence bitmap a tile stages; finalists that wait for stage C
template <int TMAX> struct SaStNpb { static constexpr int v = TMAX <= 4 ? 128 : 64; };
template <int TMAX> struct SaStCcap { static constexpr int v = TMAX <= 4 ? 384 : 192; };

struct alignas(16) StTerm {
    u64 cell0;                  // first cell of the term in the impact stream
    u32 df;
    u32 row;                    // row of the stage directory, or SA_ST_NOROW: the kernel's cursor walks the term
    u32 probe;                  // probe row of a term that is not staged, else 0xFFFFFFFF
    u32 maxf;                   // fp32 pattern of the term's largest factor in the shard
    u32 pad0, pad1;
};

struct StageParams {
    const u64* imp;
    u64 cell_base;                        // smallest cell0 of the staged terms: the kernel addresses the stream with 32-bit BYTE offsets from it
    u32 imp_bytes, dir_bytes;             // sizes of the buffers the kernel reads through buffer resources (a read past the end returns 0)
    const u32* dir;                       // stage directory (sa_stagedir): [rows][n_st + 1] x {abs, cm}
    u32 cw;                               // workgroups of an XCD that CO-WALK a range of tiles (workgroup j of a group takes tiles j, j + cw, ...); 1: private ranges
    u32 docs, n_st;                       // docs per stage tile, tiles
    u64 n_docs, doc_base;
    const StTerm* terms; u32 U, NS;       // distinct terms; the first NS are staged, the others probed
    const float* probe; u64 probe_rows64; // probe rows (interleaved: sa_probe_cell), rows x 64
    const u32* pbits; u32 pbits_words, pbits_bytes;     // their presence bitmaps: words per row, bytes in all
    u32 NPB;                              // probed terms whose bitmap the tiles stage (the first NPB of them)
    u32 B, T, k;
    const unsigned short* pu;             // [B][T] distinct-term index of the query's term at POSITION i (staged terms by descending bound, then the probed ones), SA_ST_NONE: absent
    const float* pw;                      // [B][T] its weight
    const u32* inv;                       // [B] position of query term s: 4 bits each
    const u32* seed;                      // [B] starting bounds (score bits)
    u32* gthr; u32* hist;                 // [B] cached histogram bounds, [B][SA_HBINS] histograms
    u64* cand; u32 cand_cap; u32* cand_cnt;
    u32* flag;                            // set when a probed term turns out essential (cannot happen: the run is then redone on another route)
    u32 tpx, tpw;                         // tiles per XCD, per workgroup
};

sa_stagedir::~sa_stagedir() {
    if (d_dir) { (void)hipSetDevice(device); (void)hipFree(d_dir); }
}

__global__ void __launch_bounds__(256)
sa_k_build_stagedir(const u64* __restrict__ tfp, const u64* __restrict__ tf_off, const u32* __restrict__ row_terms, u32 n_rows,
                    u32 n_st, u32 docs, u32* __restrict__ dir) {
    const u64 total = (u64)n_rows * (n_st + 1);
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 r = (u32)(e / (n_st + 1)), j = (u32)(e % (n_st + 1));
        const u32 t = row_terms[r];
        const u64 base = tf_off[t];
        const u32 cnt = (u32)(tf_off[t + 1] - base);
        dir[2 * e] = sa_lower_bound(tfp + base, 0, cnt, ((u64)j * docs) << SA_KEY_SHIFT, SA_KEY_MASK);
        dir[2 * e + 1] = 0u;
    }
}

// cm[row][tile] = postings | largest factor's upper 16 bits, rounded up (a bound: the patterns of non-negative floats order like the values)
__global__ void __launch_bounds__(256)
sa_k_build_stagecm(const u64* __restrict__ imp, const u64* __restrict__ tf_off, const u32* __restrict__ row_terms, u32 n_rows,
                   u32 n_st, u32* __restrict__ dir) {
    const u64 total = (u64)n_rows * n_st;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 r = (u32)(e / n_st), j = (u32)(e % n_st);
        const u32 t = row_terms[r];
        const u64* cells = imp + sa_imp_base(tf_off[t], t);
        const u64 ei = (u64)r * (n_st + 1) + j;
        const u32 lo = dir[2 * ei], hi = dir[2 * (ei + 1)];
        u32 mx = 0;
        for (u32 i = lo; i < hi; i++) { const u32 f = (u32)cells[i]; mx = f > mx ? f : mx; }
        const u32 cnt = hi - lo < 0xFFFFu ? hi - lo : 0xFFFFu;
        dir[2 * ei + 1] = cnt | (((mx + 0xFFFFu) >> 16) << 16);
    }
}

// the stage directory of impact stream `im` for tiles of `docs` documents (built on first use; call with the index lock held)
static std::shared_ptr<sa_stagedir> sa_stagedir_get(sa_index* ix, sa_impacts* im, u32 docs) {
    for (auto& d : im->stagedirs) if (d && d->docs == docs) return d;
    std::shared_ptr<sa_stagedir> sd(new (std::nothrow) sa_stagedir());
    if (!sd) return nullptr;
    sd->device = ix->device; sd->docs = docs;
    sd->n_st = ix->n_docs ? sa_div_up(ix->n_docs, docs) : 0;
    sd->row.assign(ix->n_terms, SA_ST_NOROW);
    std::vector<u32> row_terms;
    const u64 min_df = std::max<u64>(32, sd->n_st / 8);           // at least one posting per eight tiles
    for (u32 t = 0; t < ix->n_terms; t++) {
        const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
        if (df >= min_df) { sd->row[t] = (u32)row_terms.size(); row_terms.push_back(t); }
    }
    sd->n_rows = (u32)row_terms.size();
    const u64 entries = (u64)sd->n_rows * (sd->n_st + 1), cms = (u64)sd->n_rows * sd->n_st;
    if (hipMalloc(&sd->d_dir, (entries ? entries : 1) * 2 * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); sd->d_dir = nullptr; return nullptr; }
    if (cms) {
        u32* d_rt = nullptr;
        if (hipMalloc(&d_rt, row_terms.size() * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        bool ok = hipMemcpyAsync(d_rt, row_terms.data(), row_terms.size() * sizeof(u32), hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        if (ok) {
            const u32 grid = entries / 256 + 1 < 65536 ? (u32)(entries / 256 + 1) : 65536u;
            hipLaunchKernelGGL(sa_k_build_stagedir, dim3(grid), dim3(256), 0, ix->stream, (const u64*)ix->d_tfp, (const u64*)ix->d_tf_off,
                               (const u32*)d_rt, sd->n_rows, sd->n_st, docs, sd->d_dir);
            hipLaunchKernelGGL(sa_k_build_stagecm, dim3(grid), dim3(256), 0, ix->stream, (const u64*)im->d_imp, (const u64*)ix->d_tf_off,
                               (const u32*)d_rt, sd->n_rows, sd->n_st, sd->d_dir);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ix->stream) == hipSuccess;
        }
        (void)hipFree(d_rt);
        if (!ok) { (void)hipGetLastError(); return nullptr; }
    }
    if (im->stagedirs.size() >= 4) im->stagedirs.erase(im->stagedirs.begin());     // (batches that still use an old one keep it alive)
    im->stagedirs.push_back(sd);
    return sd;
}

// ---- probe rows ------------------------------------------------------------------------------------------------
// Probe rows are interleaved in blocks of 64 documents: factor of (row, doc) at  (doc >> 6) * (rows * 64) + row * 64 + (doc & 63) --
// the rows of one stretch of documents sit together (a workgroup that probes ~100 rows for the documents of its tiles touches a
// few pages, not one per row: with row-major rows of 40 MB each the probes were page-table walks)
__device__ __host__ __forceinline__ u64 sa_probe_cell(u32 row, u64 doc, u64 rows64) { return (doc >> 6) * rows64 + ((u64)row << 6) + (doc & 63ull); }
// all rows in ONE launch: blockIdx.y = the row, its list's first cell and length in `jobs` (round 6's first version launched a kernel per
// row: 336 launches, 6.4 ms of the first batch at 10 M docs)
struct ProbeRowJob { u64 first, df; };
__global__ void __launch_bounds__(256)
sa_k_make_probe_rows(const u64* __restrict__ imp, const ProbeRowJob* __restrict__ jobs, u64 rows64, float* __restrict__ probe,
                     u32* __restrict__ bits, u64 bits_words) {
    const u32 row = blockIdx.y;
    const u64 first = jobs[row].first, df = jobs[row].df;
    u32* const rb = bits + (u64)row * bits_words;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < df; i += (u64)gridDim.x * blockDim.x) {
        const u64 c = imp[first + i];
        const u32 doc = (u32)(c >> 32) >> 2;
        probe[sa_probe_cell(row, (u64)doc, rows64)] = __uint_as_float((u32)c);
        atomicOr(&rb[doc >> 5], 1u << (doc & 31u));
    }
}

// the probe rows of impact stream `im` (built on first use; call with the index lock held)
static void sa_probe_rows_ensure(sa_index* ix, sa_impacts* im, const sa_options_t& o) {
    if (im->probe_tried) return;
    im->probe_tried = true;
    im->probe_slot.assign(ix->n_terms, 0xFFFFFFFFu);
    const long long div = sa_opt(o.probe_div, 128);
    if (div <= 0 || ix->n_docs == 0) return;
    std::vector<std::pair<u64, u32>> cand;
    for (u32 t = 0; t < ix->n_terms; t++) {
        const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
        if (df >= 64 && df * (u64)div >= ix->n_docs) cand.push_back({df, t});
    }
    if (cand.empty()) return;
    std::sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& c) { return a.first > c.first || (a.first == c.first && a.second < c.second); });
    im->probe_stride = (ix->n_docs + 63ull) & ~63ull;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    const u64 max_rows = std::min<u64>(1024, (u64)(free_b / 4) / (im->probe_stride * sizeof(float)));
    if (cand.size() > max_rows) cand.resize(max_rows);
    if (cand.empty()) return;
    const size_t bytes = cand.size() * im->probe_stride * sizeof(float);
    im->pbits_words = (((ix->n_docs + 31ull) >> 5) + 31ull) & ~31ull;
    const size_t bbytes = cand.size() * im->pbits_words * sizeof(u32);
    hipStream_t st = ix->stream;
    if (hipMalloc(&im->d_probe, bytes) != hipSuccess || hipMemsetAsync(im->d_probe, 0, bytes, st) != hipSuccess ||
        hipMalloc(&im->d_pbits, bbytes) != hipSuccess || hipMemsetAsync(im->d_pbits, 0, bbytes, st) != hipSuccess) {
        (void)hipGetLastError();
        if (im->d_probe) { (void)hipFree(im->d_probe); im->d_probe = nullptr; }
        if (im->d_pbits) { (void)hipFree(im->d_pbits); im->d_pbits = nullptr; }
        return;
    }
    std::vector<ProbeRowJob> jobs(cand.size());
    for (size_t r = 0; r < cand.size(); r++) {
        const u32 t = cand[r].second;
        jobs[r].first = sa_imp_base(ix->h_tf_off[t], t); jobs[r].df = cand[r].first;
        im->probe_slot[t] = (u32)r;
    }
    ProbeRowJob* d_jobs = nullptr;
    bool jobs_ok = hipMalloc(&d_jobs, jobs.size() * sizeof(ProbeRowJob)) == hipSuccess &&
                   hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(ProbeRowJob), hipMemcpyHostToDevice, st) == hipSuccess;
    if (jobs_ok) {
        const u64 longest = cand[0].first;                      // (most frequent first)
        const u32 gx = (u32)std::min<u64>(512, longest / 4096 + 1);     // (a row's blocks stride over its list: 16 postings per thread and more)
        hipLaunchKernelGGL(sa_k_make_probe_rows, dim3(gx, (u32)cand.size()), dim3(256), 0, st, (const u64*)im->d_imp, (const ProbeRowJob*)d_jobs,
                           (u64)cand.size() * 64ull, im->d_probe, im->d_pbits, (u64)im->pbits_words);
    }
    const bool rows_ok = jobs_ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (d_jobs) (void)hipFree(d_jobs);
    if (!rows_ok) {
        (void)hipGetLastError();
        (void)hipFree(im->d_probe); im->d_probe = nullptr;
        (void)hipFree(im->d_pbits); im->d_pbits = nullptr;
        im->probe_slot.assign(ix->n_terms, 0xFFFFFFFFu);
        return;
    }
    im->n_probe = (u32)cand.size();
}

// ---- the plan of a query set, in the batch's upload block ---------------------------------------------------
struct StLayout { size_t terms, pw, inv, pu, total; };
static StLayout sa_stage_layout(u32 B, u32 T) {
    StLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    L.terms = take((size_t)B * T * sizeof(StTerm));
    L.pw = take((size_t)B * T * 4);
    L.inv = take((size_t)B * 4); L.pu = take((size_t)B * T * 2);
    L.total = off;
    return L;
}
// (a query set of more than SA_ST_BMAX queries is planned and run in SLICES of SA_ST_BMAX device rows: one region of the layout each)
static u32 sa_stage_slices(u32 B) { return (B + (u32)SA_ST_BMAX - 1u) / (u32)SA_ST_BMAX; }
static size_t sa_stage_slice_bytes(u32 B, u32 T) { return sa_stage_layout(B < (u32)SA_ST_BMAX ? B : (u32)SA_ST_BMAX, T).total; }
size_t sa_stage_upload_bytes(u32 B, u32 T) { return (size_t)sa_stage_slices(B) * sa_stage_slice_bytes(B, T); }

// the bound the kernel forms from a cm word: the largest factor's fp32 pattern with its low 16 bits rounded UP
static float sa_st_round_up16(float f) { u32 b; memcpy(&b, &f, 4); b = ((b + 0xFFFFu) >> 16) << 16; float r; memcpy(&r, &b, 4); return r; }

// Plan the query set (row_terms / row_idf: [B][T] in device-row order) into the upload image: distinct terms (staged ones first, most
// frequent first, then the probed ones), per query the terms in the kernel's order with their weights, the starting bounds.
// Leaves bt->stage_ok false when the set is not for this route (the caller then takes another one).
// one slice: device rows [q0, q0 + B) of the set (row_terms / row_idf point at row q0); `base`: the slice's region of the image, h_seed:
// the starting bounds of its rows.  *ok = the slice has a plan (in `out`).
static int sa_stage_plan_slice(sa_batch* bt, char* base, u32* h_seed, const u32* row_terms, const float* row_idf, u32 q0, u32 B,
                               sa_stage_slice& out, bool* ok) {
    *ok = false;
    sa_index* ix = bt->ix;
    const u32 T = bt->T;
    sa_impacts* im = bt->impacts.get();
    const bool probing = sa_opt(bt->opts.stage_probe, 1) != 0;
    const StLayout L = sa_stage_layout(B, T);
    StTerm* h_terms = (StTerm*)(base + L.terms);
    float* h_pw = (float*)(base + L.pw);
    u32* h_inv = (u32*)(base + L.inv);
    unsigned short* h_pu = (unsigned short*)(base + L.pu);

    // distinct terms: di[slot] = index of the slot's term among the set's distinct terms (SA_ST_NONE: unknown term).  The map from
    // term ids is a table of the batch (n_terms cells, all ones between plans: the cells written are put back below) and every
    // working array lives in the batch -- a plan allocates nothing and hashes nothing (it runs once per fresh query set, on the host
    // thread that feeds the device)
    struct DT { u64 df; u32 term; u32 probe; bool staged; };
    sa_stage_scratch& W = bt->st_work;
    if (W.tmap.size() != ix->n_terms) W.tmap.assign(ix->n_terms, 0xFFFFFFFFu);
    const size_t BT = (size_t)B * T;
    W.di.resize(BT); W.ubs.resize(BT); W.probed.resize(BT); W.seeds.resize(B);
    W.dist_df.clear(); W.dist_term.clear(); W.dist_probe.clear();
    bool too_many = false;
    // (the per-term host tables are indexed by term id -- random cells of arrays far bigger than the caches: ask for all of them
    //  first, so that the misses overlap instead of being taken one after the other)
    u32 rank_idx = SA_TOPF_NR - 1;
    for (int i = SA_TOPF_NR - 1; i >= 0; i--) if (sa_topf_ranks[i] >= bt->k) rank_idx = (u32)i;
    for (size_t i = 0; i < BT; i++) {
        const u32 t = row_terms[i];
        if (t >= ix->n_terms) continue;
        __builtin_prefetch(&W.tmap[t]);
        __builtin_prefetch(&ix->h_tf_off[t]);
        __builtin_prefetch(&im->h_maxf[t]);
        __builtin_prefetch(&im->h_topf[(size_t)t * SA_TOPF_NR + rank_idx]);
        if (probing && t < im->probe_slot.size()) __builtin_prefetch(&im->probe_slot[t]);
    }
    for (size_t i = 0; i < BT; i++) {
        const u32 t = row_terms[i];
        if (t >= ix->n_terms) { W.di[i] = SA_ST_NONE; continue; }
        u32 u = W.tmap[t];
        if (u == 0xFFFFFFFFu) {
            u = (u32)W.dist_term.size();
            if (u >= (u32)SA_ST_UMAX) { too_many = true; W.di[i] = SA_ST_NONE; continue; }
            W.tmap[t] = u;
            W.dist_term.push_back(t);
            W.dist_df.push_back(ix->h_tf_off[t + 1] - ix->h_tf_off[t]);
            W.dist_probe.push_back(probing && im->d_probe && t < im->probe_slot.size() ? im->probe_slot[t] : 0xFFFFFFFFu);
        }
        W.di[i] = u;
    }
    const u32 U = (u32)W.dist_term.size();
    for (u32 u = 0; u < U; u++) W.tmap[W.dist_term[u]] = 0xFFFFFFFFu;          // (the table is all ones again)
    if (U == 0 || too_many) return SA_OK;
    W.staged.assign(U, 0);
    // insertion sorts of at most 8 slots (stable; std::stable_sort allocates)
    auto sort_slots = [](u32* a, u32 n, auto less) {
        for (u32 i = 1; i < n; i++) { const u32 x = a[i]; u32 j = i; while (j > 0 && less(x, a[j - 1])) { a[j] = a[j - 1]; j--; } a[j] = x; }
    };
    // per query: the starting bound; the terms it can PROBE -- those with a probe row whose bounds in the shard, taken
    // together from the smallest up, stay below the starting bound with the kernel's own arithmetic (so that the kernel,
    // whose tile bounds are never above these and whose bound G is never below the starting one, can never find them
    // essential); every other term must be STAGED, for every query
    const float seed_scale = (float)sa_opt(bt->opts.seed_scale_pct, 100) / 100.f;
    float* const ubs = W.ubs.data();
    float* const seeds = W.seeds.data();
    unsigned char* const probed = W.probed.data();
    const unsigned short* const di = W.di.data();
    double cand_df = 0.0;
    u32 dense_further = 0;                                      // queries with a dense term (df >= n_docs / 4) that is not their first
    for (u32 q = 0; q < B; q++) {
        const size_t qb = (size_t)q * T;
        float seed = 0.f;
        for (u32 s = 1; s < T; s++)
            if (di[qb + s] != SA_ST_NONE && W.dist_df[di[qb + s]] * 4ull >= ix->n_docs) { dense_further++; break; }
        for (u32 s = 0; s < T; s++) {
            probed[qb + s] = 0; ubs[qb + s] = 0.f;
            if (di[qb + s] == SA_ST_NONE) continue;
            const u32 t = row_terms[qb + s];
            const float w = row_idf[qb + s];
            ubs[qb + s] = sa_st_round_up16(im->h_maxf[t]) * w;          // (what the kernel forms from a cm word, at most)
            const float sd1 = (im->h_topf[(size_t)t * SA_TOPF_NR + rank_idx] * w) * seed_scale;   // (the arithmetic of sa_k_make_bounds)
            if (sd1 > seed) seed = sd1;
        }
        seeds[q] = seed;
        u32 os[8];
        for (u32 s = 0; s < T; s++) os[s] = s;
        sort_slots(os, T, [&](u32 a, u32 c) { return ubs[qb + a] < ubs[qb + c]; });
        {
            // the candidates this query has to expect: the postings of the terms that its starting bound leaves essential
            float sf = 0.f;
            for (u32 j = 0; j < T; j++) {
                sf += ubs[qb + os[j]];
                if (di[qb + os[j]] != SA_ST_NONE && !(seed > 0.f && sf * SA_ST_MARGIN < seed)) cand_df += (double)W.dist_df[di[qb + os[j]]];
            }
        }
        if (seed > 0.f) {
            // candidates for probing: smallest bound first (the slots with a probe row, in the order of `os`)
            float sfx = 0.f;
            for (u32 j = 0; j < T; j++) {
                const u32 s = os[j];
                if (di[qb + s] == SA_ST_NONE || W.dist_probe[di[qb + s]] == 0xFFFFFFFFu) continue;
                sfx = sfx + ubs[qb + s];
                if (!(sfx * SA_ST_MARGIN * 1.0001f < seed)) break;
                probed[qb + s] = 1;
            }
        }
        for (u32 s = 0; s < T; s++) if (di[qb + s] != SA_ST_NONE && !probed[qb + s]) W.staged[di[qb + s]] = 1;
    }
    // staged terms first, most frequent first; then the probed ones
    // (a counting sort by [probed][64 - bits of df]: heavy terms first so that the lanes of a wave own slices of similar length -- an
    //  ORDER OF MAGNITUDE is all that needs; a comparison sort of the terms was a third of the plan's host time)
    W.order.resize(U); W.rank.resize(U);
    {
        u32 cnt[2 * 65] = {0};
        auto bucket = [&](u32 u) -> u32 { const u64 d = W.dist_df[u]; return (W.staged[u] ? 0u : 65u) + (d ? (u32)__builtin_clzll(d) : 64u); };
        for (u32 u = 0; u < U; u++) cnt[bucket(u)]++;
        u32 run = 0;
        for (u32 i = 0; i < 2 * 65; i++) { const u32 c = cnt[i]; cnt[i] = run; run += c; }
        for (u32 u = 0; u < U; u++) W.order[cnt[bucket(u)]++] = u;
    }
    u32 NS = 0;
    u64 dfsum = 0;
    for (u32 r = 0; r < U; r++) { const u32 u = W.order[r]; W.rank[u] = r; if (W.staged[u]) { NS++; dfsum += W.dist_df[u]; } }
    if (U - NS > (u32)SA_ST_NT) return SA_OK;                // (one probed term per thread)
    // docs per stage tile: the largest of the sizes below whose expected postings fit the stage with room for the tiles above the
    // mean, and that leave a workgroup of a full device nine tiles or more
    const u32 tmax = T <= 4 ? 4u : 8u;
    const double cap = tmax == 4 ? (double)SaStCap<4>::v : (double)SaStCap<8>::v;
    const double per_doc = (double)dfsum / (double)ix->n_docs;
    u32 docs = 0;
    if (sa_opt_is_set(bt->opts.stage_docs)) docs = (u32)std::min<long long>(1024, std::max<long long>(64, bt->opts.stage_docs)) / 64u * 64u;
    else {
        static const u32 sizes[] = {1024, 768, 512, 384, 256, 128, 64};
        const u64 wgs = (u64)ix->n_cus * (u64)std::max<long long>(1, sa_opt(bt->opts.stage_wgs, 2));
        for (u32 s : sizes) {
            if (per_doc * s + 4.0 * sqrt(per_doc * s) > 0.97 * cap) continue;
            if (!docs) docs = s;                             // (the largest that fits, unless a smaller one spreads the shard better)
            // (measured on 1.25 / 2.5 / 5 M-doc shards, 1024 / 768 / 512 / 384 docs per tile: 0.098 / 0.089 / 0.077 / 0.090, 0.120 / 0.114 / 0.111 /
            //  0.149, 0.161 / 0.169 / 0.186 / 0.278 ms -- the biggest tile from nine tiles per workgroup on, else 512)
            if (ix->n_docs / s >= 9ull * wgs || s <= 512u) { docs = s; break; }
        }
        if (!docs) return SA_OK;                          // (more than ~70 staged postings per doc: not this route)
    }
    std::shared_ptr<sa_stagedir> sd = (bt->st_dir && bt->st_dir->docs == docs) ? bt->st_dir : sa_stagedir_get(ix, im, docs);
    if (!sd) return SA_OK;
    if ((u64)sd->n_rows * (sd->n_st + 1ull) >= (1ull << 29)) return SA_OK;         // (32-bit byte offsets into the directory)
    // the kernel addresses the stream with 32-bit byte offsets from the first staged term
    u64 cell_lo = ~0ull, cell_hi = 0;
    bool all_rowed = true;
    for (u32 r = 0; r < U; r++) {
        const u32 u = W.order[r];
        const u32 term = W.dist_term[u];
        StTerm& x = h_terms[r];
        x.cell0 = sa_imp_base(ix->h_tf_off[term], term);
        x.df = (u32)W.dist_df[u];
        x.row = sd->row[term];
        x.probe = W.staged[u] ? 0xFFFFFFFFu : W.dist_probe[u];
        memcpy(&x.maxf, &im->h_maxf[term], 4);
        x.pad0 = 0; x.pad1 = 0;
        if (!W.staged[u] && (x.row == SA_ST_NOROW || x.probe >= 65536u)) return SA_OK;       // (cannot happen: a probe row means df >= n_docs / 128)
        if (W.staged[u] && x.row == SA_ST_NOROW) all_rowed = false;
        if (W.staged[u]) {
            cell_lo = std::min<u64>(cell_lo, x.cell0);
            cell_hi = std::max<u64>(cell_hi, x.cell0 + (u64)x.df + 4ull);
        }
    }
    if (NS == 0) { cell_lo = h_terms[0].cell0; cell_hi = cell_lo + 4; }
    if (cell_hi - cell_lo >= (1ull << 29)) return SA_OK;              // (32-bit BYTE offsets in the kernel: shards of up to ~18 M docs of this corpus; bigger ones keep the older routes)
    // the queries: staged terms by descending bound, then the probed ones; weights; starting bounds
    for (u32 q = 0; q < B; q++) {
        const size_t qb = (size_t)q * T;
        u32 ord[8];
        for (u32 s = 0; s < T; s++) ord[s] = s;
        auto is_probed = [&](u32 s) { return di[qb + s] != SA_ST_NONE && !W.staged[di[qb + s]]; };
        sort_slots(ord, T, [&](u32 a, u32 c) {
            const bool pa = is_probed(a), pc = is_probed(c);
            if (pa != pc) return !pa;
            return ubs[qb + a] > ubs[qb + c];
        });
        u32 inv = 0;
        for (u32 i = 0; i < T; i++) {
            const u32 s = ord[i];
            h_pu[qb + i] = (unsigned short)(di[qb + s] != SA_ST_NONE ? W.rank[di[qb + s]] : SA_ST_NONE);
            h_pw[qb + i] = row_idf[qb + s];
            inv |= i << (4u * s);
        }
        h_inv[q] = inv;
        u32 sb; memcpy(&sb, &seeds[q], 4);
        h_seed[q] = seeds[q] > 0.f ? sb : 0u;
    }
    // Unless the caller forces the route (option stage = 1): only where it was measured ahead of the grouped overlay / per-query kernels
    // (profiles/route_rule_r06.jsonl, small_set_routes_r06.jsonl; 10 M docs):
    //  * few candidates per document at small k -- the BASELINE set of 256 queries up to k = 32 (1.8 .. 2.1 candidates per doc: 0.24 / 0.28 /
    //    0.36 ms against 0.37 / 0.38 / 0.39), not at k = 100 (2.3: 0.47 against 0.42) or k = 1000, not the pairwise-distinct set (3.7 per doc
    //    and 768 staged terms: 0.97 against 0.46);
    //  * k up to 100 when the set is small -- 16 queries 0.29 against 0.60 ms, 64 queries 0.215 against 0.226 (0.6 candidates per doc);
    //  * a handful of staged terms (the `hot` set, 8 staged terms: ahead up to k = 100);
    //  * NOT a set of one to three queries: the route walks every tile of the shard whatever the number of queries (one query 0.150 ms
    //    against 0.055 for the per-query kernel, four queries 0.165 against 0.150 at k = 10, 0.235 against 0.284 at k = 100).
    const float cpd = (float)(cand_df / (double)ix->n_docs);
    if (sa_opt(bt->opts.stage, -1) != 1) {
        const u32 Bset = bt->B;
        const bool few_cands = bt->k <= 32u && cpd <= 2.5f && NS <= (u32)SA_ST_NT && Bset >= 8u;
        const bool small_set = bt->k > 32u && bt->k <= 100u && cpd <= 0.6f && NS <= (u32)SA_ST_NT && Bset >= 4u;
        const bool few_terms = bt->k <= 100u && NS <= 32u && Bset >= (bt->k <= 32u ? 8u : 4u);
        //  * queries with a SECOND dense term: the overlay route shares a query's first term only, a further dense term sends the
        //    (tile, query) pair to the per-query kernel -- 256 queries of 6 / 8 terms with two terms of ranks 1-10 each: 4.7 / 5.0 ms there,
        //    2.5 / 3.2 ms here (k = 10; 3.3 / 4.1 against 4.8 / 5.1 at k = 100) although 15 candidates per document are expected
        //    (profiles/query_length_routes_r06.jsonl)
        const bool dense_pairs = bt->k <= 100u && dense_further * 2u >= B && Bset >= 8u;
        if (!few_cands && !small_set && !few_terms && !dense_pairs) {
            if (sa_opt(bt->opts.trace, 0)) fprintf(stderr, "sa_stage_plan: not taken (%u queries, k = %u, %u staged terms, %.3f candidates per doc expected)\n", Bset, bt->k, NS, cpd);
            return SA_OK;
        }
    }
    out.q0 = q0; out.nq = B; out.U = U; out.NS = NS; out.docs = docs; out.tmax = tmax; out.all_rowed = all_rowed;
    out.cand_per_doc = cpd;
    if (sa_opt(bt->opts.trace, 0)) fprintf(stderr, "sa_stage_plan: rows %u..%u: %u terms (%u staged, %u probed), %u docs per tile, %.3f candidates per doc expected\n", q0, q0 + B, U, NS, U - NS, docs, cpd);
    out.cell_base = cell_lo;
    out.imp_bytes = (u32)((cell_hi - cell_lo) * 8ull);
    out.dir = sd;
    *ok = true;
    return SA_OK;
}

int sa_stage_plan(sa_batch* bt, char* img, const u32* row_terms, const float* row_idf) {
    bt->stage_ok = false;
    bt->st_slices.clear();
    sa_index* ix = bt->ix;
    const u32 B = bt->B, T = bt->T;
    sa_impacts* im = bt->impacts.get();
    if (!im || im->h_maxf.size() != ix->n_terms || im->h_topf.size() != (size_t)ix->n_terms * SA_TOPF_NR) return SA_OK;
    if (B == 0 || T > 8 || !bt->d_st || ix->n_docs == 0 || ix->n_docs > (1ull << 28)) return SA_OK;
    // Sets of more than 256 queries run slice by slice, one launch after the other, and every launch walks every tile of the shard.
    // Measured against the overlay route, whose items grow with the set (profiles/wide_set_routes_r06*.jsonl, BASELINE-shaped sets,
    // k = 10, staged / overlay): 10 M docs 512 .. 2048 queries 0.86 - 0.93; 5 M docs 0.94 / 1.05 / 1.13; 2.5 M 1.10 / 1.25 / 1.41;
    // 1.25 M 1.31 / 1.48 / 1.80 -- unless forced, only on shards of 8 M docs and more, up to 2048 queries
    if (B > (u32)SA_ST_BMAX && sa_opt(bt->opts.stage, -1) != 1 && (ix->n_docs < 8000000ull || B > 8u * (u32)SA_ST_BMAX)) return SA_OK;
    if (sa_opt(bt->opts.stage_probe, 1) != 0) sa_probe_rows_ensure(ix, im, bt->opts);
    const u32 S = sa_stage_slices(B);
    const size_t SB = sa_stage_slice_bytes(B, T);
    char* base = img + (bt->d_st - bt->d_up);
    u32* h_seed = (u32*)(img + ((char*)bt->d_seed - bt->d_up));
    bt->st_slices.resize(S);
    for (u32 sl = 0; sl < S; sl++) {
        const u32 q0 = sl * (u32)SA_ST_BMAX, nq = std::min<u32>((u32)SA_ST_BMAX, B - q0);
        bool ok = false;
        SA_TRY(sa_stage_plan_slice(bt, base + sl * SB, h_seed + q0, row_terms + (size_t)q0 * T, row_idf + (size_t)q0 * T, q0, nq, bt->st_slices[sl], &ok));
        if (!ok) {                                              // (one slice the route does not take: the whole set runs another route)
            bt->st_slices.clear();
            for (u32 q = 0; q < B; q++) h_seed[q] = 0u;         // (the slice-table kernel forms the starting bounds of that route)
            return SA_OK;
        }
    }
    const sa_stage_slice& f = bt->st_slices[0];
    bt->st_U = f.U; bt->st_NS = f.NS; bt->st_docs = f.docs; bt->st_tmax = f.tmax; bt->st_cand_per_doc = f.cand_per_doc;
    bt->st_cell_base = f.cell_base; bt->st_imp_bytes = f.imp_bytes; bt->st_dir = f.dir;
    bt->stage_ok = true;
    return SA_OK;
}

// ---- the kernel ------------------------------------------------------------------------------------------------
// binary search of term slice `pk` (start << 16 | postings) of the stage for the doc key d4 (= doc * 4, the high word of a
// staged posting); returns the factor, 0 when the doc does not hold the term
__device__ __forceinline__ float sa_st_lookup(const u64* s_post, u32 pk, u32 d4, bool& found) {
    const u32* const w32 = (const u32*)s_post;
    u32 lo = pk >> 16;
    const u32 end = lo + (pk & 0xFFFFu);
    u32 hi = end;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (w32[2u * mid + 1u] < d4) lo = mid + 1u; else hi = mid;
    }
    found = false;
    if (lo < end) {
        const u64 v = s_post[lo];
        if ((u32)(v >> 32) == d4) { found = true; return __uint_as_float((u32)v); }
    }
    return 0.f;
}

// inclusive prefix sum over the 64 lanes of a wave through the DPP data path (no LDS crossbar round trips like the
// ds_bpermute-based __shfl_up): Hillis-Steele inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the rows' totals
__device__ __forceinline__ u32 sa_st_wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}
// Block-wide exclusive scan of TWO u32 per thread behind ONE barrier: the waves' totals go to the half of `red` (2 x 2 NWAVES
// words) that `parity` selects; the caller alternates the parity from call to call, so a wave that is still reading one half
// cannot see the next call's totals (two calls apart there is a barrier every wave has passed).
template <int NWAVES>
__device__ __forceinline__ void sa_block_excl_scan2(u32 a, u32 b, u32* red, u32 parity, u32& ea, u32& eb, u32& ta, u32& tb) {
    const u32 ia = sa_st_wave_incl_scan(a), ib = sa_st_wave_incl_scan(b);
    u32* const r = red + parity * (2u * NWAVES);
    if (sa_lane() == SA_WAVE - 1) { r[2 * sa_wave_id()] = ia; r[2 * sa_wave_id() + 1] = ib; }
    __syncthreads();
    u32 ba = 0, bb = 0; ta = 0; tb = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) {
        const u32 sa = r[2 * w], sb = r[2 * w + 1];
        if (w < sa_wave_id()) { ba += sa; bb += sb; }
        ta += sa; tb += sb;
    }
    ea = ba + ia - a; eb = bb + ib - b;
}

// -DSA_PROBE (scripts/build_probe.sh; never in the product build): cycles a workgroup spends per phase of a tile pass, summed
// over the launch by wave 0 (s_memtime at the phase boundaries), read by sa_debug_stage_probe_read
#ifdef SA_PROBE
__device__ unsigned long long g_sa_stage_probe[32];
#define SA_SPT(i) do { if (tid == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - plast; plast = t_; } } while (0)
extern "C" int sa_debug_stage_probe_read(unsigned long long* out16, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return SA_ERR_HIP;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sa_stage_probe), 32 * 8) != hipSuccess) return SA_ERR_HIP;
    if (clear) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sa_stage_probe), z, 32 * 8) != hipSuccess) return SA_ERR_HIP; }
    return SA_OK;
}
#else
#define SA_SPT(i) do { } while (0)
#endif

typedef unsigned int sa_v2u __attribute__((vector_size(8)));
typedef unsigned int sa_v4u __attribute__((vector_size(16)));
struct alignas(8) StChunk { u32 dc, off; };    // a copy chunk: first stage cell | postings (1 .. 8) << 13; byte offset of its first posting from the stream base

// KT: staged terms per thread (1: up to SA_ST_NT staged terms, the usual case; 2: up to SA_ST_UMAX)
template <int TMAX, int KT>
__global__ void __launch_bounds__(SA_ST_NT, 4) sa_k_bm25_stage(const StageParams sp) {
    constexpr int CAP = SaStCap<TMAX>::v;
    constexpr int NT = SA_ST_NT, NW = NT / SA_WAVE;
    constexpr int NPBMAX = SaStNpb<TMAX>::v, CCAP = SaStCcap<TMAX>::v;
    constexpr int NCH = CAP / 8 + (KT == 1 ? SA_ST_NT : SA_ST_UMAX);   // 8-posting chunks a stage can hold at most (a partly filled one per staged term)
    constexpr int NWL = 1024;                                    // candidates per round of stage A (its survivors fit s_b)
    constexpr int KB = 8;                                        // chunk loads a lane issues before it waits
    static_assert(KT == 1 || KT == 2, "one or two staged terms per thread");
    static_assert(SA_ST_UMAX <= CAP, "a single document's postings must fit the stage");
    static_assert(CAP <= 8192 && SA_ST_UMAX <= 1024, "13-bit stage cells in a chunk descriptor");
    static_assert(SA_ST_BMAX <= 256 && TMAX <= 8, "8-bit query, 3-bit position in a work-list record");
    __shared__ alignas(16) u64 s_post[CAP];                      // the stage: every staged term's slice of this tile, doc-sorted
    __shared__ u32 s_off[SA_ST_UMAX];                           // per distinct term: first cell << 16 | postings; a probed term: probe row << 16 | SA_ST_PROBE
    __shared__ u32 s_tmax[SA_ST_UMAX];                          // bound (fp32 pattern) of its factors in this tile
    __shared__ StChunk s_cd[NCH];                               // the copy's chunk list; then the candidate work list (u32 records)
    __shared__ unsigned short s_pu[SA_ST_BMAX * TMAX];          // [query][position]: distinct-term index (SA_ST_NONE: no term)
    __shared__ alignas(16) float s_pw[SA_ST_BMAX * TMAX];       //   its weight
    __shared__ alignas(16) float s_psfx[SA_ST_BMAX * TMAX];     //   what the positions >= i can add in THIS tile (with the margin)
    __shared__ u32 s_thr[SA_ST_BMAX];
    __shared__ u32 s_b[NWL];                                    // records of the candidates that pass stage A
    __shared__ u32 s_c[CCAP * (2 + TMAX)];                // finalists: query, doc key, contributions by position
    __shared__ unsigned short s_cum[SA_ST_BMAX * TMAX];         // per query: candidates of the essential positions <= i
    __shared__ u32 s_qoff[SA_ST_BMAX + 1];                      // per query: its first candidate
    __shared__ alignas(16) u32 s_bits[NPBMAX * SA_ST_BW];   // presence bitmaps of the probed terms, this tile's docs
    __shared__ u32 s_ref[SA_ST_REF];
    __shared__ u32 s_nref, s_nb, s_nc;
    __shared__ u32 s_red[4 * NW];
    const u32 tid = threadIdx.x, lane = tid & (SA_WAVE - 1), wave = tid / SA_WAVE;
    const u32 T = sp.T, B = sp.B, U = sp.U, NS = sp.NS;
    // The stream and the cm words are read through BUFFER RESOURCES: a scalar base + a 32-bit byte offset per lane (no 64-bit
    // address arithmetic, no address register pairs), and a lane without an item passes an offset past the end and reads 0.
    const __amdgpu_buffer_rsrc_t r_imp = __builtin_amdgcn_make_buffer_rsrc((void*)(sp.imp + sp.cell_base), 0, (int)sp.imp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dir = __builtin_amdgcn_make_buffer_rsrc((void*)sp.dir, 0, (int)sp.dir_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bits = __builtin_amdgcn_make_buffer_rsrc((void*)(sp.pbits ? (const void*)sp.pbits : (const void*)sp.imp), 0, (int)sp.pbits_bytes, 0x00020000);
    auto cell_at = [&](u32 boff) -> u64 { const sa_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r_imp, boff, 0, 0); return ((u64)v[1] << 32) | (u64)v[0]; };
    auto key_at = [&](u32 boff) -> u32 { return __builtin_amdgcn_raw_buffer_load_b32(r_imp, boff + 4u, 0, 0); };      // (the doc key of a cell: its high word)
#ifdef SA_PROBE
    u64 pacc[24];
    for (int i = 0; i < 24; i++) pacc[i] = 0;
    u64 plast = __builtin_amdgcn_s_memtime();
    u32 ptiles = 0, pcand = 0, pfin = 0, pflush = 0;
#endif
    // XCD-aware tile ranges: block b runs on XCD b % 8; an XCD walks a contiguous range of tiles and its workgroups
    // contiguous sub-ranges -- a term's slices of neighbouring tiles are neighbours in memory, so the cache line a slice
    // shares with the next tile's is fetched into ONE L2
    // CO-WALKING (sp.cw > 1): cw workgroups of an XCD share a range of cw * tpw tiles and take every cw-th tile of it -- at any moment they
    // work on NEIGHBOURING tiles, whose slices of a sparse term lie in the same 128-byte line: one fetch into the XCD's L2 serves
    // all of them (with private ranges a workgroup met the line again a whole tile pass later, when ~5 MB of other lines had gone
    // through the 4 MB L2: hit rate 24 %)
    const u32 xcd = blockIdx.x & 7u, wg = blockIdx.x >> 3;
    const u32 t_step = sp.cw;
    const u32 wgrp = wg / t_step, wj = wg - wgrp * t_step;
    const u32 g_begin = xcd * sp.tpx + wgrp * (t_step * sp.tpw);
    u32 t_end = g_begin + t_step * sp.tpw;
    if (t_end > (xcd + 1u) * sp.tpx) t_end = (xcd + 1u) * sp.tpx;
    if (t_end > sp.n_st) t_end = sp.n_st;
    const u32 t_begin = g_begin + wj;
    if (t_begin >= t_end) return;                               // (uniform)

    // the queries' tables, once per workgroup, padded to TMAX positions (a position without a term adds nothing)
    for (u32 i = tid; i < B * (u32)TMAX; i += NT) {
        const u32 q = i / (u32)TMAX, c = i % (u32)TMAX;
        s_pu[i] = c < T ? sp.pu[q * T + c] : (unsigned short)SA_ST_NONE;
        s_pw[i] = c < T ? sp.pw[q * T + c] : 0.f;
    }
    if (tid == 0) { s_nref = 0u; s_nb = 0u; s_nc = 0u; s_qoff[SA_ST_BMAX] = 0xFFFFFFFFu; }
    const bool hasq = tid < B;
    const u32 seed = (hasq && sp.seed) ? sp.seed[tid] : 0u;
    u32 g_raw = 0u;
    u32 parity = 0;
    // This thread's STAGED terms (u = tid, tid + NT).  lo: first posting not yet staged.  A term with a directory row: nx = its cm
    // word (postings | bound of the factors) of the NEXT tile to take, read one tile ahead.  A term without one is WALKED: nx, w1
    // = doc keys of the postings at lo and lo + 1 (read one tile ahead too; the sentinel behind a term's postings has the doc
    // field all ones: a walk stops there); its bound is its largest factor in the shard.  A thread without a term walks a
    // sentinel: nothing ever comes of it.  The PROBED terms (u >= NS) have no owner: their bound is their largest factor in the shard.
    u32 src0[KT], lo[KT], nx[KT], w1[KT], cmi[KT], ab[KT];          // (cmi: the term's directory entry of the next tile to read; ab: that tile's first posting)
    bool rowed[KT];
#pragma unroll
    for (int kx = 0; kx < KT; kx++) {
        const u32 u = tid + (u32)kx * NT;
        const StTerm t0 = sp.terms[0];
        src0[kx] = NS ? (u32)(t0.cell0 - sp.cell_base) + t0.df : 0u;       // (term 0's sentinel)
        lo[kx] = 0; nx[kx] = 0xFFFFFFFFu; w1[kx] = 0xFFFFFFFFu; rowed[kx] = false; cmi[kx] = 0x1FFFFFFFu; ab[kx] = 0;
        if (u < NS) {
            const StTerm t = sp.terms[u];
            src0[kx] = (u32)(t.cell0 - sp.cell_base);
            s_tmax[u] = t.maxf;                                 // (a walked term keeps this bound; a term with a row gets its tile's)
            if (t.row != SA_ST_NOROW) {
                rowed[kx] = true;
                cmi[kx] = t.row * (sp.n_st + 1u) + t_begin;
                const sa_v2u e = __builtin_amdgcn_raw_buffer_load_b64(r_dir, cmi[kx] << 3, 0, 0);
                ab[kx] = e[0]; nx[kx] = e[1];
            } else {
                const u32 key = (u32)((u64)t_begin * sp.docs) << 2;
                u32 a = 0, b = t.df;
                while (a < b) { const u32 mid = a + ((b - a) >> 1); if (key_at((src0[kx] + mid) << 3) < key) a = mid + 1u; else b = mid; }
                lo[kx] = a;
                nx[kx] = key_at((src0[kx] + a) << 3); w1[kx] = key_at((src0[kx] + a + 1u) << 3);
            }
        }
    }
    // a PROBED term (u >= NS; one per thread: the plan sees to it): only the bound of its factors in the tile is taken, from its cm word
    u32 pcmi = 0x1FFFFFFFu, pnx = 0u;
    if (NS + tid < U) {
        const StTerm t = sp.terms[NS + tid];
        s_off[NS + tid] = (t.probe << 16) | SA_ST_PROBE;
        pcmi = t.row * (sp.n_st + 1u) + t_begin;
        pnx = __builtin_amdgcn_raw_buffer_load_b32(r_dir, (pcmi << 3) + 4u, 0, 0);
    }
    // presence bitmaps of the first NPB probed terms: this thread's two 16-byte pieces of a tile's bitmaps (which term, which
    // 128 docs of the tile: the same for every tile)
    u32 bro[2];
    const u32 p4n = sp.docs >> 7;
    auto bdst_of = [&](u32 y) -> u32 { const u32 r = y / (p4n ? p4n : 1u); return r * (u32)(SA_ST_BW / 4) + (y - r * p4n); };   // (16-byte cell of piece y in s_bits)
    {
        const u32 n_bp = sp.NPB * p4n;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const u32 y = (u32)i * NT + tid;
            const bool have = y < n_bp;
            const u32 yy = have ? y : 0u;
            const u32 r = yy / (p4n ? p4n : 1u), c = yy - r * p4n;
            bro[i] = have ? (sp.terms[NS + r].probe * sp.pbits_words + 4u * c) << 2 : 0xFFFFFFF0u;
        }
    }
    // ---- stage C: the finalists waiting in s_c -- the probed terms' factors from their probe rows, the exact score, the query's
    //      candidate list.  Block-uniform; called when the list is nearly full and at the end.
    auto flush_finalists = [&]() {
        SA_SPT(20);
        const u32 nc = s_nc;
        for (u32 z0 = 0; z0 < nc; z0 += NT) {                   // (uniform)
            if (z0 + (tid & ~(u32)(SA_WAVE - 1)) >= nc) continue;   // (a wave without a finalist issues nothing: the VALU is the busiest unit, and the other workgroup of the CU wants it)
            const u32 z = z0 + tid;
            const bool havez = z < nc;
            const u32* const c = s_c + (havez ? z : 0u) * (2u + (u32)TMAX);
            const u32 fq = c[0] & 0xFFu, fmask = c[0] >> 8, fd4 = c[1];
            const u32 qb = fq * (u32)TMAX;
            const u32 doc_l = fd4 >> 2;
            float fx[TMAX], pw[TMAX], pv[TMAX];
            bool isp[TMAX];
            // (every load unconditional and issued before the first is looked at: a position that is not probed reads cell 0)
#pragma unroll
            for (int i = 0; i < TMAX; i++) {
                fx[i] = __uint_as_float(c[2 + i]);
                const u32 u = s_pu[qb + (u32)i];
                const u32 pk = s_off[u != SA_ST_NONE ? u : 0u];
                isp[i] = havez && ((fmask >> i) & 1u) != 0u;
                pw[i] = s_pw[qb + (u32)i];
                pv[i] = sp.probe[isp[i] ? sa_probe_cell(pk >> 16, (u64)doc_l, sp.probe_rows64) : 0ull];
            }
            const u32 inv = sp.inv[fq];
#pragma unroll
            for (int i = 0; i < TMAX; i++) fx[i] = isp[i] ? __fmul_rn(pv[i], pw[i]) : fx[i];
            // the exact score: factor * weight per term, summed in QUERY-TERM order (bm25.pyx:19-23, np.sum over the terms)
            float S = 0.f;
#pragma unroll
            for (int sl = 0; sl < TMAX; sl++) {
                const u32 pos = (u32)sl < T ? (inv >> (4u * (u32)sl)) & 15u : 15u;
                float x = 0.f;
#pragma unroll
                for (int i = 0; i < TMAX; i++) x = pos == (u32)i ? fx[i] : x;
                S = __fadd_rn(S, x);
            }
            const u32 sb = __float_as_uint(S);
            SA_SPT(23);
#ifdef SA_PROBE
            if (tid == 0) { pfin += nc; pflush++; }
#endif
            if (havez && sb >= s_thr[fq]) {
                const u64 doc = sp.doc_base + (u64)doc_l;
                const u32 pos = atomicAdd(&sp.cand_cnt[fq], 1u);
                if (pos < sp.cand_cap) sp.cand[(u64)fq * sp.cand_cap + pos] = ((u64)sb << 32) | (u64)(u32)(~(u32)doc);
                atomicAdd(&sp.hist[(u64)fq * SA_HBINS + sa_score_bin(sb)], 1u);
                if (((pos + 1u) & (u32)SA_ST_REFRESH_MASK) == 0u) { const u32 sl = atomicAdd(&s_nref, 1u); if (sl < (u32)SA_ST_REF) s_ref[sl] = fq; }
            }
        }
        SA_SPT(21);
        __syncthreads();
        if (tid == 0) s_nc = 0u;
        // bounds re-derived for the queries whose candidate list crossed a multiple of 32 entries
        const u32 nref = s_nref < (u32)SA_ST_REF ? s_nref : (u32)SA_ST_REF;
        if (nref) {                                             // (uniform)
            for (u32 r = wave; r < nref; r += NW) {
                const u32 q = s_ref[r];
                sa_hist_refresh(sp.hist + (u64)q * SA_HBINS, &sp.gthr[q], sp.k, lane);
            }
            __syncthreads();
            if (tid == 0) s_nref = 0u;
        }
        __syncthreads();
        SA_SPT(22);
    };
    // whose candidate number x of a pass is: query | position << 8 | posting of the position's slice << 11 (a search of the queries'
    // first candidates, then of the query's essential positions)
    auto owner_of = [&](u32 xx) -> u32 {
        u32 q = 0;
#pragma unroll
        for (int step = SA_ST_BMAX / 2; step >= 1; step >>= 1) q += s_qoff[q + (u32)step] <= xx ? (u32)step : 0u;
        const u32 qb = q * (u32)TMAX;
        const u32 r = xx - s_qoff[q];
        u32 i = 0, base = 0;
#pragma unroll
        for (int c = 0; c < TMAX - 1; c++) { const u32 cc = s_cum[qb + (u32)c]; const bool ge = r >= cc; i = ge ? (u32)c + 1u : i; base = ge ? cc : base; }
        return q | (i << 8) | ((r - base) << 11);
    };
    __syncthreads();

    for (u32 tile = t_begin; tile < t_end; tile += t_step) {
        const u64 tile_d0 = (u64)tile * sp.docs;
        const u64 tile_d1 = tile_d0 + sp.docs < sp.n_docs ? tile_d0 + sp.docs : sp.n_docs;
        SA_SPT(11);
        // postings of this thread's terms in this tile and the bound of their factors there, from what was read a tile ago
        u32 n_t[KT], tm[KT];
        {
            const u32 key = (u32)tile_d1 << 2;
#pragma unroll
            for (int kx = 0; kx < KT; kx++) {
                const u32 cmw = nx[kx];
                u32 nw = 0;
                if (!rowed[kx] && cmw < key) {
                    nw = 1;
                    if (w1[kx] < key) { nw = 2; while (key_at((src0[kx] + lo[kx] + nw) << 3) < key) nw++; }   // (three postings of a rare term in one tile: hardly ever)
                }
                n_t[kx] = rowed[kx] ? cmw & 0xFFFFu : nw;
                tm[kx] = cmw & 0xFFFF0000u;
                lo[kx] = rowed[kx] ? ab[kx] : lo[kx];            // (a term with a row: the tile's first posting comes with its size -- this workgroup's previous tile need not be the one before)
            }
        }
        const u32 ptm = pnx & 0xFFFF0000u;
        if (tile + t_step < t_end) {                              // (uniform) the directory entries the passes below read are the next tile's
#pragma unroll
            for (int kx = 0; kx < KT; kx++) cmi[kx] += rowed[kx] ? t_step : 0u;
            pcmi += NS + tid < U ? t_step : 0u;
        }
        // the queries' bounds (a bound only ever rises: a stale one is valid), read a tile ago
        const u32 g_now = g_raw > seed ? g_raw : seed;
        // A tile whose postings do not fit the stage is taken in doc sub-ranges: halve the range until it fits (a single
        // document holds at most U <= CAP postings), the slices' ends by a search of the posting lists.
        u64 d_s = tile_d0;
        u32 used[KT];                                           // postings of the tile already staged by earlier passes
#pragma unroll
        for (int kx = 0; kx < KT; kx++) used[kx] = 0u;
        while (d_s < tile_d1) {                                 // (uniform)
            u64 d_e = tile_d1;
            u32 n[KT];
#pragma unroll
            for (int kx = 0; kx < KT; kx++) n[kx] = n_t[kx] - used[kx];
            u32 excl, exch, P, NC;
            {
                u32 mine = 0, mch = 0;
#pragma unroll
                for (int kx = 0; kx < KT; kx++) { mine += n[kx]; mch += (n[kx] + 7u) >> 3; }
                sa_block_excl_scan2<NW>(mine, mch, s_red, parity, excl, exch, P, NC);
                parity ^= 1u;
            }
            // (the rare case apart from the common path: a loop with loads in it makes the compiler wait for every load in flight at its head)
            if (P > (u32)CAP && d_e - d_s > 1ull) {
                do {
                    d_e = d_s + ((d_e - d_s) >> 1);
                    const u32 key = (u32)d_e << 2;
#pragma unroll
                    for (int kx = 0; kx < KT; kx++) {
                        u32 a = 0, b = n[kx];
                        const u32 first = src0[kx] + lo[kx];
                        while (a < b) { const u32 mid = a + ((b - a) >> 1); if (key_at((first + mid) << 3) < key) a = mid + 1u; else b = mid; }
                        n[kx] = a;
                    }
                    u32 mine = 0, mch = 0;
#pragma unroll
                    for (int kx = 0; kx < KT; kx++) { mine += n[kx]; mch += (n[kx] + 7u) >> 3; }
                    sa_block_excl_scan2<NW>(mine, mch, s_red, parity, excl, exch, P, NC);
                    parity ^= 1u;
                } while (P > (u32)CAP && d_e - d_s > 1ull);
            }
            SA_SPT(0);
            {
                // stage offsets, bounds and the copy's chunk list: one descriptor per 8 postings
                u32 o = excl, oc = exch;
#pragma unroll
                for (int kx = 0; kx < KT; kx++) {
                    const u32 u = tid + (u32)kx * NT;
                    if (u < NS) {
                        if (rowed[kx]) s_tmax[u] = tm[kx];
                        {
                            s_off[u] = (o << 16) | n[kx];
                            u32 boff = (src0[kx] + lo[kx]) << 3;
                            for (u32 left = n[kx]; left; ) {
                                const u32 c = left < 8u ? left : 8u;
                                StChunk d; d.dc = o | (c << 13); d.off = boff;
                                s_cd[oc] = d;
                                o += c; oc++; boff += 64u; left -= c;
                            }
                        }
                    }
                }
            }
            if (NS + tid < U) s_tmax[NS + tid] = ptm;
            SA_SPT(1);
            __syncthreads();
            SA_SPT(2);
            // ---- the reads for the NEXT tile: cm words, the walked terms' next doc keys, the queries' bounds.  Every load is
            //      UNCONDITIONAL and its result is only looked at a tile later (a load inside a branch makes the compiler wait for
            //      everything in flight where the branch joins); a lane without the case passes an offset past the buffer's end
            //      and reads 0 without a memory access.  Issued in front of the stage's loads, so that they have landed when
            //      those have -- the loops further down contain (rare) loads, and at the head of such a loop the compiler waits
            //      for every load in flight.
            {
#pragma unroll
                for (int kx = 0; kx < KT; kx++) {
                    const u32 kb = rowed[kx] ? 0xFFFFFFE0u : (src0[kx] + lo[kx] + (n_t[kx] - used[kx])) << 3;     // (the cell behind the tile's last posting, whatever the pass)
                    const sa_v2u e = __builtin_amdgcn_raw_buffer_load_b64(r_dir, rowed[kx] ? cmi[kx] << 3 : 0xFFFFFFF0u, 0, 0);
                    const u32 k0 = key_at(kb);
                    nx[kx] = rowed[kx] ? e[1] : k0;
                    ab[kx] = e[0];
                    w1[kx] = key_at(kb + 8u);
                }
                pnx = __builtin_amdgcn_raw_buffer_load_b32(r_dir, NS + tid < U ? (pcmi << 3) + 4u : 0xFFFFFFF0u, 0, 0);
                g_raw = __hip_atomic_load(&sp.gthr[hasq ? tid : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- stage: eight lanes per chunk (and 16 bytes per lane of the probed terms' presence bitmaps).  A lane issues all its
            //      loads before it waits for the first; a lane without an item reads past the end (0) and writes nothing.  While the
            //      loads are in flight the QUERY PHASE runs (it needs the slices' sizes and bounds, not the postings).
            const u32 grp = tid >> 3, sub = tid & 7u;
            sa_v4u bv[2];
            u64 v[KB]; u32 dst[KB];
#pragma unroll
            for (int i = 0; i < 2; i++)
                bv[i] = __builtin_amdgcn_raw_buffer_load_b128(r_bits, bro[i] == 0xFFFFFFF0u ? 0xFFFFFFF0u : bro[i] + ((u32)(tile_d0 >> 5) << 2), 0, 0);
#pragma unroll
            for (int i = 0; i < KB; i++) {
                const u32 x = (u32)i * (NT / 8) + grp;
                const StChunk d = s_cd[x < NC ? x : 0u];
                const bool ok = x < NC && sub < (d.dc >> 13);
                dst[i] = ok ? (d.dc & 0x1FFFu) + sub : 0xFFFFFFFFu;
                v[i] = cell_at(ok ? d.off + (sub << 3) : 0xFFFFFFF0u);
            }
            SA_SPT(3);
            // ---- the queries: bound; what every position can add at most in THIS tile (weight x bound of the term's factors here);
            //      essential positions; their postings are the candidates.  (Branch-free, the LDS reads in batches: a read inside a
            //      branch is waited for on the spot.)
            u32 ncand = 0;
            if (hasq) {
                const u32 qb = tid * (u32)TMAX;
                const u32 thr = g_now > 1u ? g_now : 1u;
                const float thr_f = __uint_as_float(thr);
                u32 uu[TMAX], pk[TMAX]; float ubv[TMAX];
                bool have[TMAX];
#pragma unroll
                for (int i = 0; i < TMAX; i++) { const u32 u = s_pu[qb + (u32)i]; have[i] = u != SA_ST_NONE; uu[i] = have[i] ? u : 0u; }
#pragma unroll
                for (int i = 0; i < TMAX; i++) { pk[i] = s_off[uu[i]]; ubv[i] = __fmul_rn(__uint_as_float(s_tmax[uu[i]]), s_pw[qb + (u32)i]); }
                float sfx = 0.f;
                u32 ness = 0;
#pragma unroll
                for (int i = TMAX - 1; i >= 0; i--) {
                    sfx = __fadd_rn(sfx, have[i] ? ubv[i] : 0.f);
                    const float sm = __fmul_rn(sfx, SA_ST_MARGIN);
                    s_psfx[qb + (u32)i] = sm;
                    ness = (sm >= thr_f && ness == 0u) ? (u32)i + 1u : ness;
                }
                bool bad = false;
                unsigned short cumv[TMAX];
#pragma unroll
                for (int i = 0; i < TMAX; i++) {
                    const u32 nn = pk[i] & 0xFFFFu;
                    const bool ess = have[i] && (u32)i < ness;
                    bad = bad || (ess && nn == SA_ST_PROBE);
                    ncand += ess && nn != SA_ST_PROBE ? nn : 0u;
                    cumv[i] = (unsigned short)ncand;
                }
#pragma unroll
                for (int i = 0; i < TMAX; i++) s_cum[qb + (u32)i] = cumv[i];
                if (bad) *sp.flag = 1u;                             // (a probed term essential: the plan rules it out; the run would be redone)
                s_thr[tid] = thr;
            }
            // the queries' candidates, one after the other: query q's are [s_qoff[q], s_qoff[q + 1])
            u32 C;
            {
                u32 e0, e1, t1;
                sa_block_excl_scan2<NW>(ncand, 0u, s_red, parity, e0, e1, C, t1);
                parity ^= 1u;
                if (tid < (u32)SA_ST_BMAX) s_qoff[tid] = hasq ? e0 : 0xFFFFFFFFu;
            }
            SA_SPT(4);
            // ---- the stage's loads land
#pragma unroll
            for (int i = 0; i < 2; i++) if (bro[i] != 0xFFFFFFF0u) ((sa_v4u*)s_bits)[bdst_of((u32)i * NT + tid)] = bv[i];
#pragma unroll
            for (int i = 0; i < KB; i++) if (dst[i] != 0xFFFFFFFFu) s_post[dst[i]] = v[i];
            for (u32 x0 = (u32)(KB * (NT / 8)); x0 < NC; x0 += (u32)(KB * (NT / 8))) {       // (uniform: a tile with more than KB * 64 chunks, hardly ever)
                u64 v2[KB]; u32 dst2[KB];
#pragma unroll
                for (int i = 0; i < KB; i++) {
                    const u32 x = x0 + (u32)i * (NT / 8) + grp;
                    const StChunk d = s_cd[x < NC ? x : 0u];
                    const bool ok = x < NC && sub < (d.dc >> 13);
                    dst2[i] = ok ? (d.dc & 0x1FFFu) + sub : 0xFFFFFFFFu;
                    v2[i] = cell_at(ok ? d.off + (sub << 3) : 0xFFFFFFF0u);
                }
#pragma unroll
                for (int i = 0; i < KB; i++) if (dst2[i] != 0xFFFFFFFFu) s_post[dst2[i]] = v2[i];
            }
            SA_SPT(5);
            SA_SPT(6);
            __syncthreads();
            SA_SPT(7);
#ifdef SA_PROBE
            pcand += C;
#endif
            const u64 ltmask = (1ull << lane) - 1ull;
            for (u32 c0 = 0; c0 < C; c0 += (u32)NWL) {          // (uniform: rounds of at most NWL candidates)
                const u32 cend = C - c0 < (u32)NWL ? C : c0 + (u32)NWL;
                // ---- stage A, every candidate, one per lane: whose it is (a search of the queries' offsets), then its own
                //      contribution + everything the query's other terms can add in this tile, against the bound.  The survivors'
                //      records -- query | position << 8 | posting << 11 -- are compacted into s_b (one LDS atomic per wave).
                for (u32 x0 = c0; x0 < cend; x0 += NT) {          // (uniform)
                    if (x0 + (tid & ~(u32)(SA_WAVE - 1)) >= cend) continue;   // (a wave without a candidate issues nothing)
                    const u32 x = x0 + tid;
                    const bool valid = x < cend;
                    const u32 xx = valid ? x : c0;
                    const u32 orec = owner_of(xx);
                    SA_SPT(12);
                    const u32 q = orec & 0xFFu, i = (orec >> 8) & 7u, j = orec >> 11;
                    const u32 qb = q * (u32)TMAX;
                    u32 up[TMAX]; float wp[TMAX];
#pragma unroll
                    for (int c = 0; c < TMAX; c++) { up[c] = s_pu[qb + (u32)c]; wp[c] = s_pw[qb + (u32)c]; }
                    u32 u_src = 0; float w_src = 0.f;
#pragma unroll
                    for (int c = 0; c < TMAX; c++) if ((u32)c == i) { u_src = up[c]; w_src = wp[c]; }
                    const u64 vsrc = s_post[(s_off[u_src] >> 16) + j];
                    const u32 od = ((u32)(vsrc >> 32) >> 2) - (u32)tile_d0;
                    // what the other terms can add: all of them in this tile, less the probed terms the doc does not hold (presence bitmaps)
                    float rem = s_psfx[qb] - __fmul_rn(__uint_as_float(s_tmax[u_src]), w_src);
                    u32 tmp[TMAX], bwp[TMAX];
#pragma unroll
                    for (int c = 0; c < TMAX; c++) {
                        const bool pr = up[c] != SA_ST_NONE && up[c] >= NS && up[c] - NS < sp.NPB;
                        const u32 slot = pr ? up[c] - NS : 0u;
                        tmp[c] = s_tmax[pr ? up[c] : 0u];
                        bwp[c] = pr ? s_bits[slot * (u32)SA_ST_BW + (od >> 5)] : 0xFFFFFFFFu;
                    }
#pragma unroll
                    for (int c = 0; c < TMAX; c++) rem -= ((bwp[c] >> (od & 31u)) & 1u) ? 0.f : __fmul_rn(__uint_as_float(tmp[c]), wp[c]);
                    const bool alive = valid && !(__fmul_rn(__fadd_rn(__fmul_rn(__uint_as_float((u32)vsrc), w_src), rem), SA_ST_MARGIN) < __uint_as_float(s_thr[q]));
                    SA_SPT(13);
                    const u64 m = (u64)__builtin_amdgcn_ballot_w64(alive);
                    if (m) {                                        // (wave-uniform)
                        u32 wb = 0;
                        if (lane == (u32)__builtin_ctzll(m)) wb = atomicAdd(&s_nb, (u32)__popcll(m));
                        wb = (u32)__builtin_amdgcn_readlane((int)wb, (int)__builtin_ctzll(m));
                        if (alive) s_b[wb + (u32)__popcll(m & ltmask)] = q | (i << 8) | (j << 11);
                    }
                    SA_SPT(14);
                }
                SA_SPT(8);
                __syncthreads();
                SA_SPT(15);
                const u32 nb = s_nb;
                // ---- stage B, the survivors, one per lane: the document is looked up in the query's other STAGED terms, in
                //      descending-bound order, as long as what is known plus what the remaining terms can add reaches the bound;
                //      the documents that get through are the FINALISTS (s_c: query, doc, the contributions found).  They wait
                //      there -- across rounds and tiles -- until the list is nearly full: stage C (flush_finalists).
                for (u32 y0 = 0; y0 < nb; ) {                    // (uniform)
                    // (a round takes as many survivors as s_c has room for finalists: s_nc only changes between barriers)
                    const u32 want = nb - y0 < (u32)NT ? nb - y0 : (u32)NT;
                    if ((u32)CCAP - s_nc < want && s_nc) flush_finalists();
                    const u32 room = (u32)CCAP - s_nc;
                    const u32 cnt = want < room ? want : room;
                    SA_SPT(16);
                    const u32 y = y0 + tid;
                    if ((tid & ~(u32)(SA_WAVE - 1)) < cnt) {       // (a wave without a survivor issues nothing)
                    // (branch-free, the searches of the query's staged terms IN LOCKSTEP: every LDS read of a step is issued before the
                    //  first is looked at -- one term after the other, with an early exit between them, was a chain of ~35 dependent
                    //  LDS round trips per survivor)
                    const bool act = tid < cnt;
                    const u32 rec = s_b[act ? y : 0u];
                    const u32 q = rec & 0xFFu, i_src = (rec >> 8) & 7u, j = rec >> 11;
                    const u32 qb = q * (u32)TMAX;
                    u32 up[TMAX]; float wp[TMAX], sfxp[TMAX];
#pragma unroll
                    for (int i = 0; i < TMAX; i++) { up[i] = s_pu[qb + (u32)i]; wp[i] = s_pw[qb + (u32)i]; sfxp[i] = s_psfx[qb + (u32)i]; }
                    const float thr_f = __uint_as_float(s_thr[q]);
                    u32 pk[TMAX], tmx[TMAX];
#pragma unroll
                    for (int i = 0; i < TMAX; i++) { const u32 uu = up[i] != SA_ST_NONE ? up[i] : 0u; pk[i] = s_off[uu]; tmx[i] = s_tmax[uu]; }
                    u32 pk_src = 0; float w_src = 0.f, ub_src = 0.f;
#pragma unroll
                    for (int i = 0; i < TMAX; i++) if ((u32)i == i_src) { pk_src = pk[i]; w_src = wp[i]; ub_src = __fmul_rn(__uint_as_float(tmx[i]), wp[i]); }
                    const u64 vsrc = s_post[(pk_src >> 16) + j];
                    const u32 d4 = (u32)(vsrc >> 32);
                    const u32 od = (d4 >> 2) - (u32)tile_d0;
                    const float known0 = __fmul_rn(__uint_as_float((u32)vsrc), w_src);
                    // the staged terms other than the candidate's own: the last posting <= the doc, by halving
                    bool stg[TMAX], prb[TMAX];
                    u32 base[TMAX], len[TMAX];
                    u32 anylen = 0;
#pragma unroll
                    for (int i = 0; i < TMAX; i++) {
                        const bool have = up[i] != SA_ST_NONE && (u32)i != i_src;
                        prb[i] = have && (pk[i] & 0xFFFFu) == SA_ST_PROBE;
                        stg[i] = have && !prb[i] && (pk[i] & 0xFFFFu) != 0u;
                        base[i] = stg[i] ? pk[i] >> 16 : 0u;
                        len[i] = stg[i] ? pk[i] & 0xFFFFu : 0u;
                        anylen |= act ? len[i] : 0u;
                    }
                    const u32* const key32 = (const u32*)s_post;
                    while (__builtin_amdgcn_ballot_w64(anylen > 1u)) {   // (wave-uniform: as many steps as the longest slice needs)
                        u32 kk[TMAX];
#pragma unroll
                        for (int i = 0; i < TMAX; i++) kk[i] = key32[2u * (base[i] + (len[i] >> 1)) + 1u];
                        anylen = 0;
#pragma unroll
                        for (int i = 0; i < TMAX; i++) {
                            const u32 half = len[i] >> 1;
                            base[i] += kk[i] <= d4 ? half : 0u;
                            len[i] -= half;
                            anylen |= len[i];
                        }
                    }
                    u64 cell[TMAX]; u32 bw[TMAX];
#pragma unroll
                    for (int i = 0; i < TMAX; i++) {
                        cell[i] = s_post[base[i]];
                        const u32 slot = prb[i] ? up[i] - NS : 0u;
                        bw[i] = s_bits[(slot < sp.NPB ? slot : 0u) * (u32)SA_ST_BW + (od >> 5)];
                    }
                    // in the terms' order: the bound tests, the duplicates, what is known
                    bool alive = true;
                    float known = known0, pend = 0.f;               // (pend: what the probed terms the doc holds can add at most)
                    u32 pmask = 0;                                  // (the probed positions whose term the doc holds)
                    float xs[TMAX];                                 // the terms' contributions, by position
#pragma unroll
                    for (int i = 0; i < TMAX; i++) {
                        xs[i] = (u32)i == i_src ? known0 : 0.f;
                        const float ubi = __fmul_rn(__uint_as_float(tmx[i]), wp[i]);
                        if (prb[i]) {
                            // a probed term: does the doc hold it (the tile's presence bitmap)?  Then it may add up to its bound, and is probed
                            const bool may = up[i] - NS >= sp.NPB || ((bw[i] >> (od & 31u)) & 1u) != 0u;
                            if (alive && may) { pend = __fadd_rn(pend, ubi); pmask |= 1u << i; }
                        } else if (up[i] != SA_ST_NONE && (u32)i != i_src) {
                            // what the positions from i on can still add (the candidate's own term is already in `known`)
                            const float rem = sfxp[i] - ((u32)i < i_src ? ub_src : 0.f);
                            if (__fmul_rn(__fadd_rn(known, rem), SA_ST_MARGIN) < thr_f) alive = false;
                            const bool found = stg[i] && (u32)(cell[i] >> 32) == d4;
                            if (alive && found) {
                                if ((u32)i < i_src) alive = false;       // the doc is the candidate of that (essential, higher) position
                                else { xs[i] = __fmul_rn(__uint_as_float((u32)cell[i]), wp[i]); known = __fadd_rn(known, xs[i]); }
                            }
                        }
                    }
                    const bool fin = act && alive && !(__fmul_rn(__fadd_rn(known, pend), SA_ST_MARGIN) < thr_f);
                    SA_SPT(17);
                    const u64 m = (u64)__builtin_amdgcn_ballot_w64(fin);
                    if (m) {                                        // (wave-uniform)
                        u32 wb = 0;
                        if (lane == (u32)__builtin_ctzll(m)) wb = atomicAdd(&s_nc, (u32)__popcll(m));
                        wb = (u32)__builtin_amdgcn_readlane((int)wb, (int)__builtin_ctzll(m));
                        if (fin) {
                            u32* const c = s_c + (wb + (u32)__popcll(m & ltmask)) * (2u + (u32)TMAX);
                            c[0] = q | (pmask << 8); c[1] = d4;
#pragma unroll
                            for (int i = 0; i < TMAX; i++) c[2 + i] = __float_as_uint(xs[i]);
                        }
                    }
                    }
                    SA_SPT(18);
                    __syncthreads();
                    SA_SPT(19);
                    y0 += cnt;
                }
                if (tid == 0) s_nb = 0u;
                __syncthreads();
            }
            SA_SPT(9);
            SA_SPT(10);
#pragma unroll
            for (int kx = 0; kx < KT; kx++) { lo[kx] += n[kx]; used[kx] += n[kx]; }
            d_s = d_e;
#ifdef SA_PROBE
            ptiles++;
#endif
        }
    }
    flush_finalists();
#ifdef SA_PROBE
    SA_SPT(11);
    if (tid == 0) {
        for (int i = 0; i < 24; i++) atomicAdd(&g_sa_stage_probe[i], pacc[i]);
        atomicAdd(&g_sa_stage_probe[24], (unsigned long long)ptiles);
        atomicAdd(&g_sa_stage_probe[25], (unsigned long long)pcand);
        atomicAdd(&g_sa_stage_probe[26], 1ull);
        atomicAdd(&g_sa_stage_probe[27], (unsigned long long)pfin);
        atomicAdd(&g_sa_stage_probe[28], (unsigned long long)pflush);
    }
#endif
}

int sa_launch_stage(sa_batch* bt, const Bm25Params& p, hipStream_t st) {
    sa_index* ix = bt->ix;
    if (!bt->stage_ok || bt->st_slices.empty() || !bt->impacts || !p.hist || !p.gthr || !p.imp) { sa_set_error("staged route: no plan"); return SA_ERR_STATE; }
    const size_t SB = sa_stage_slice_bytes(bt->B, bt->T);
    const u32 wgs = (u32)std::min<long long>(8, std::max<long long>(1, sa_opt(bt->opts.stage_wgs, 2)));
    u32 grid = (u32)ix->n_cus * wgs / 8u * 8u;
    if (grid < 8u) grid = 8u;
    // one launch per slice of SA_ST_BMAX device rows, one after the other on the batch's stream: the slice's tables, and the rows'
    // bounds / histograms / candidate lists (the kernel numbers its queries from 0)
    for (size_t sl = 0; sl < bt->st_slices.size(); sl++) {
        const sa_stage_slice& x = bt->st_slices[sl];
        const StLayout L = sa_stage_layout(x.nq, bt->T);
        const char* base = bt->d_st + sl * SB;
        StageParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.imp = p.imp; sp.cell_base = x.cell_base;
        sp.imp_bytes = x.imp_bytes;
        sp.dir_bytes = (u32)std::min<u64>(0xFFFFFFE0ull, (u64)x.dir->n_rows * (x.dir->n_st + 1ull) * 8ull);
        sp.dir = x.dir->d_dir;
        sp.docs = x.docs; sp.n_st = x.dir->n_st;
        sp.n_docs = ix->n_docs; sp.doc_base = ix->doc_base;
        sp.terms = (const StTerm*)(base + L.terms); sp.U = x.U; sp.NS = x.NS;
        sp.probe = bt->impacts->d_probe ? bt->impacts->d_probe : (const float*)p.imp;      // (no probe rows: nothing is probed; the kernel's unconditional loads read a cell nobody uses)
        sp.probe_rows64 = (u64)bt->impacts->n_probe * 64ull;
        sp.pbits = bt->impacts->d_pbits; sp.pbits_words = (u32)bt->impacts->pbits_words;
        sp.pbits_bytes = bt->impacts->d_pbits ? (u32)std::min<u64>(0xFFFFFFE0ull, (u64)bt->impacts->n_probe * bt->impacts->pbits_words * 4ull) : 16u;
        sp.NPB = bt->impacts->d_pbits && x.docs <= 1024u && x.docs % 128u == 0u ? std::min<u32>(x.U - x.NS, x.tmax == 4 ? (u32)SaStNpb<4>::v : (u32)SaStNpb<8>::v) : 0u;
        sp.B = x.nq; sp.T = bt->T; sp.k = bt->k;
        sp.pu = (const unsigned short*)(base + L.pu);
        sp.pw = (const float*)(base + L.pw);
        sp.inv = (const u32*)(base + L.inv);
        sp.seed = p.seed ? p.seed + x.q0 : nullptr;
        sp.gthr = p.gthr + x.q0; sp.hist = p.hist + (size_t)x.q0 * SA_HBINS;
        sp.cand = p.cand + (size_t)x.q0 * p.cand_cap; sp.cand_cap = p.cand_cap; sp.cand_cnt = p.cand_cnt + x.q0;
        sp.flag = bt->d_overflow;
        if (sp.n_st == 0) continue;
        sp.tpx = (sp.n_st + 7u) / 8u;
        const u32 wpx = grid / 8u;
        sp.tpw = (sp.tpx + wpx - 1u) / wpx;
        // co-walking groups (option stage_cw, default 4; 1: private ranges): only when every staged term has a directory row -- a walked
        // term's cursor moves posting by posting and cannot skip the tiles of the group's other workgroups
        // (measured, BASELINE batch at 10 M docs, k = 10, groups of 1 / 2 / 4 / 8 / 16 / 32 / 64: 0.288 / 0.275 / 0.273 / 0.267 / 0.266 / 0.2575 /
        //  0.259 ms; L2 hit rate 6 % -> 75 %, traffic past the L2 1.53 -> 0.39 GB per launch: profiles/stage_kernel_cowalk_ab_r06.jsonl)
        u32 cw = (u32)std::max<long long>(1, sa_opt(bt->opts.stage_cw, 32));
        while (cw > 1u && wpx % cw != 0u) cw >>= 1;              // (the largest power-of-two share of an XCD's workgroups at most the wish)
        sp.cw = (x.all_rowed && cw > 1u && sp.tpw >= 2u) ? cw : 1u;
        if (sa_opt(bt->opts.trace, 0)) fprintf(stderr, "sa_launch_stage: rows %u..%u: %u workgroups, %u tiles of %u docs, %u per workgroup, co-walking groups of %u\n", x.q0, x.q0 + x.nq, grid, sp.n_st, sp.docs, sp.tpw, sp.cw);
        const bool one = x.NS <= (u32)SA_ST_NT;
        if (x.tmax == 4) {
            if (one) hipLaunchKernelGGL((sa_k_bm25_stage<4, 1>), dim3(grid), dim3(SA_ST_NT), 0, st, sp);
            else hipLaunchKernelGGL((sa_k_bm25_stage<4, 2>), dim3(grid), dim3(SA_ST_NT), 0, st, sp);
        } else {
            if (one) hipLaunchKernelGGL((sa_k_bm25_stage<8, 1>), dim3(grid), dim3(SA_ST_NT), 0, st, sp);
            else hipLaunchKernelGGL((sa_k_bm25_stage<8, 2>), dim3(grid), dim3(SA_ST_NT), 0, st, sp);
        }
    }
    return SA_OK;
}

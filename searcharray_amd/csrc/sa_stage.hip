// sa_stage.hip -- the STAGED-TILE route of BM25 top-k batches (round 6).
//
// Replaces, for a batch of B queries, the reference's caller loop
//     np.sum([arr.score(t) for t in query], axis=0)   (test/test_msmarco.py:353-354; score = postings.py:652-680 ->
//     as_dense roaringish_ops.pyx:84-98 -> _bm25_score bm25.pyx:11-25)   ->   np.argpartition (utils/sort.py:24)
// like sa_k_bm25_group_tiles does, with a different DECOMPOSITION:
//
//   * the unit of work is a TILE of `docs` documents (512 by default), not a (tile, query) pair.  A persistent workgroup
//     walks a contiguous range of tiles.  Per tile it STAGES the slice of EVERY DISTINCT TERM of the batch -- the
//     impact-stream postings  doc*4 << 32 | fp32 factor , doc-sorted -- from HBM into LDS, once: the posting lists
//     are streamed from HBM exactly once per batch, whatever the number of queries that share a term, with coalesced
//     8-byte loads (the slice of the stream a tile needs is contiguous);
//   * the queries are then answered FROM LDS.  A query starts with a bound G that at least k documents are known to
//     reach (rank tables of its terms, then the histogram of the documents found so far: sa_topk.hpp).  Its terms are
//     ordered by the most each can contribute (weight x largest factor of the term in this shard); the terms at the
//     end of that order whose bounds TOGETHER stay below G are non-essential: a document that holds only such terms
//     cannot reach G (fp32 sums of non-negatives are monotone; the rounding of the sum is covered by a margin).  Every
//     other document of the tile with a chance appears in the slice of an ESSENTIAL term, so the candidates of a
//     (tile, query) pair are the postings of its essential terms in this tile -- a few documents instead of every
//     posting of every term;
//   * candidates of all queries of the tile are flattened into one work list and taken one per LANE: the lane looks
//     the document up in the query's other terms (binary search of the staged slices) in descending-bound order and
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
// Roofline: HBM-bound streaming of the distinct posting lists (8 bytes per posting, once per batch) -- integer /
// compare work and a scalar fp32 multiply-add per looked-up posting, no MFMA.
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_bm25_params.hpp"
#include "../../include/searcharray_hip.h"

#include <algorithm>
#include <math.h>
#include <new>
#include <unordered_map>

#define SA_ST_NT 512            // threads per workgroup
#define SA_ST_UMAX 1024         // distinct terms of a query set
#define SA_ST_BMAX 256          // queries of a query set
#define SA_ST_WL 768            // candidate records per work-list chunk
#define SA_ST_REF 64            // queries whose bound is re-derived at the end of a tile pass
#define SA_ST_NONE 0xFFFFu      // "no term" in the queries' term tables
#define SA_ST_NOROW 0xFFFFFFFFu
#define SA_ST_MARGIN 1.0000153f // 1 + 2^-16: covers the fp32 roundings of a sum of up to 8 non-negative terms taken in another order

// postings an LDS stage holds (TMAX = 4: the BASELINE shape; 8: wider query tables, smaller stage); two workgroups per CU
template <int TMAX> struct SaStCap { static constexpr int v = TMAX <= 4 ? 6528 : 4608; };

struct alignas(16) StTerm {
    u64 cell0;                  // first cell of the term in the impact stream
    u32 df;
    u32 row;                    // row of the stage directory, or SA_ST_NOROW: the kernel's cursor walks the term
};

struct StageParams {
    const u64* imp;
    const u32* dir; u32 dir_stride;       // stage directory [rows][n_st + 1]
    u32 docs, n_st;                       // docs per stage tile, tiles
    u64 n_docs, doc_base;
    const StTerm* terms; u32 U;
    u32 cb[3];                            // copy classes (sa_batch::st_cb)
    u32 B, T, k;
    const unsigned short* pu;             // [B][T] distinct-term index of the query's term at POSITION i (descending bound), SA_ST_NONE: absent
    const float* pw;                      // [B][T] its weight
    const float* pub;                     // [B][T] its bound: weight x largest factor
    const float* psfx;                    // [B][T+1] bound of the positions >= i together, with the margin (psfx[T] = 0)
    const u32* inv;                       // [B] position of query term s: 4 bits each
    const u32* seed;                      // [B] starting bounds (score bits)
    u32* gthr; u32* hist;                 // [B] cached histogram bounds, [B][SA_HBINS] histograms
    u64* cand; u32 cand_cap; u32* cand_cnt;
    u32 tpx, tpw;                         // tiles per XCD, per workgroup
};

sa_stagedir::~sa_stagedir() {
    if (d_dir) { hipSetDevice(device); hipFree(d_dir); }
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
        dir[e] = sa_lower_bound(tfp + base, 0, cnt, ((u64)j * docs) << SA_KEY_SHIFT, SA_KEY_MASK);
    }
}

// the index's stage directory for tiles of `docs` documents (built on first use; call with the index lock held)
static std::shared_ptr<sa_stagedir> sa_stagedir_get(sa_index* ix, u32 docs) {
    for (auto& d : ix->stagedirs) if (d && d->docs == docs) return d;
    std::shared_ptr<sa_stagedir> sd(new (std::nothrow) sa_stagedir());
    if (!sd) return nullptr;
    sd->device = ix->device; sd->docs = docs;
    sd->n_st = ix->n_docs ? sa_div_up(ix->n_docs, docs) : 0;
    sd->row.assign(ix->n_terms, SA_ST_NOROW);
    std::vector<u32> row_terms;
    const u64 min_df = std::max<u64>(32, sd->n_st / 2);           // at least one posting per two tiles
    for (u32 t = 0; t < ix->n_terms; t++) {
        const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
        if (df >= min_df) { sd->row[t] = (u32)row_terms.size(); row_terms.push_back(t); }
    }
    sd->n_rows = (u32)row_terms.size();
    const u64 entries = (u64)sd->n_rows * (sd->n_st + 1);
    if (hipMalloc(&sd->d_dir, (entries ? entries : 1) * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); sd->d_dir = nullptr; return nullptr; }
    if (entries) {
        u32* d_rt = nullptr;
        if (hipMalloc(&d_rt, row_terms.size() * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        bool ok = hipMemcpyAsync(d_rt, row_terms.data(), row_terms.size() * sizeof(u32), hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        if (ok) {
            const u32 grid = entries / 256 + 1 < 65536 ? (u32)(entries / 256 + 1) : 65536u;
            hipLaunchKernelGGL(sa_k_build_stagedir, dim3(grid), dim3(256), 0, ix->stream, (const u64*)ix->d_tfp, (const u64*)ix->d_tf_off,
                               (const u32*)d_rt, sd->n_rows, sd->n_st, docs, sd->d_dir);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ix->stream) == hipSuccess;
        }
        hipFree(d_rt);
        if (!ok) { (void)hipGetLastError(); return nullptr; }
    }
    if (ix->stagedirs.size() >= 4) ix->stagedirs.erase(ix->stagedirs.begin());     // (batches that still use an old one keep it alive)
    ix->stagedirs.push_back(sd);
    return sd;
}

// ---- the plan of a query set, in the batch's upload block ---------------------------------------------------
struct StLayout { size_t terms, pw, pub, psfx, inv, pu, total; };
static StLayout sa_stage_layout(u32 B, u32 T) {
    StLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    L.terms = take((size_t)B * T * sizeof(StTerm));
    L.pw = take((size_t)B * T * 4); L.pub = take((size_t)B * T * 4); L.psfx = take((size_t)B * (T + 1) * 4);
    L.inv = take((size_t)B * 4); L.pu = take((size_t)B * T * 2);
    L.total = off;
    return L;
}
size_t sa_stage_upload_bytes(u32 B, u32 T) { return sa_stage_layout(B, T).total; }

static float sa_float_up(double x) {                 // the smallest float >= x (x >= 0, finite)
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, INFINITY);
    return f;
}

// Plan the query set (row_terms / row_idf: [B][T] in device-row order) into the upload image: distinct terms (most frequent
// first), per query the terms by descending score bound with their weights, bounds and suffix bounds, the starting
// bounds.  Leaves bt->stage_ok false when the set is not for this route (the caller then takes another one).
int sa_stage_plan(sa_batch* bt, char* img, const u32* row_terms, const float* row_idf) {
    bt->stage_ok = false;
    sa_index* ix = bt->ix;
    const u32 B = bt->B, T = bt->T;
    sa_impacts* im = bt->impacts.get();
    if (!im || im->h_maxf.size() != ix->n_terms || im->h_topf.size() != (size_t)ix->n_terms * SA_TOPF_NR) return SA_OK;
    if (B > SA_ST_BMAX || T > 8 || !bt->d_st || ix->n_docs == 0 || ix->n_docs > (1ull << 28)) return SA_OK;
    const StLayout L = sa_stage_layout(B, T);
    char* base = img + (bt->d_st - bt->d_up);
    StTerm* h_terms = (StTerm*)(base + L.terms);
    float* h_pw = (float*)(base + L.pw); float* h_pub = (float*)(base + L.pub); float* h_psfx = (float*)(base + L.psfx);
    u32* h_inv = (u32*)(base + L.inv);
    unsigned short* h_pu = (unsigned short*)(base + L.pu);
    u32* h_seed = (u32*)(img + ((char*)bt->d_seed - bt->d_up));

    // distinct terms, most frequent first
    std::unordered_map<u32, u32> idx;
    idx.reserve((size_t)B * T * 2);
    std::vector<std::pair<u64, u32>> dist;               // (df, term)
    for (size_t i = 0; i < (size_t)B * T; i++) {
        const u32 t = row_terms[i];
        if (t >= ix->n_terms) continue;
        if (idx.emplace(t, 0u).second) dist.push_back({ix->h_tf_off[t + 1] - ix->h_tf_off[t], t});
    }
    if (dist.empty() || dist.size() > SA_ST_UMAX) return SA_OK;
    std::sort(dist.begin(), dist.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& c) { return a.first > c.first || (a.first == c.first && a.second < c.second); });
    const u32 U = (u32)dist.size();
    u64 dfsum = 0;
    for (u32 u = 0; u < U; u++) { idx[dist[u].second] = u; dfsum += dist[u].first; }
    // docs per stage tile: the largest of the sizes below whose expected postings fit the stage with room for the tiles above the mean
    const u32 tmax = T <= 4 ? 4u : 8u;
    const double cap = tmax == 4 ? (double)SaStCap<4>::v : (double)SaStCap<8>::v;
    const double per_doc = (double)dfsum / (double)ix->n_docs;
    u32 docs = 0;
    if (sa_opt_is_set(bt->opts.stage_docs)) docs = (u32)std::max<long long>(64, bt->opts.stage_docs) / 64u * 64u;
    else {
        static const u32 sizes[] = {1024, 768, 512, 384, 256, 192, 128, 64};
        for (u32 s : sizes) if (per_doc * s + 4.0 * sqrt(per_doc * s) <= 0.97 * cap) { docs = s; break; }
        if (!docs) return SA_OK;                          // (more than ~70 postings per doc over the set's terms: not this route)
    }
    std::shared_ptr<sa_stagedir> sd = sa_stagedir_get(ix, docs);
    if (!sd) return SA_OK;
    for (u32 u = 0; u < U; u++) {
        const u32 t = dist[u].second;
        h_terms[u].cell0 = sa_imp_base(ix->h_tf_off[t], t);
        h_terms[u].df = (u32)dist[u].first;
        h_terms[u].row = sd->row[t];
    }
    // copy classes by the postings a tile is expected to hold
    u32 cb[3] = {0, 0, 0};
    for (u32 u = 0; u < U; u++) {
        const double e = (double)dist[u].first * docs / (double)ix->n_docs;
        if (e >= 40.0) cb[0] = u + 1;
        if (e >= 20.0) cb[1] = u + 1;
        if (e >= 10.0) cb[2] = u + 1;
    }
    // the queries
    u32 rank_idx = SA_TOPF_NR - 1;
    for (int i = SA_TOPF_NR - 1; i >= 0; i--) if (sa_topf_ranks[i] >= bt->k) rank_idx = (u32)i;
    const float seed_scale = (float)sa_opt(bt->opts.seed_scale_pct, 100) / 100.f;
    for (u32 q = 0; q < B; q++) {
        float ub[8]; u32 un[8]; u32 ord[8];
        float seed = 0.f;
        for (u32 s = 0; s < T; s++) {
            const u32 t = row_terms[(size_t)q * T + s];
            const float w = row_idf[(size_t)q * T + s];
            ord[s] = s;
            if (t >= ix->n_terms) { ub[s] = 0.f; un[s] = SA_ST_NONE; continue; }
            un[s] = idx[t];
            ub[s] = im->h_maxf[t] * w;                                   // (fp32 product: factor * w <= maxf * w, rounding is monotone)
            const float sd1 = (im->h_topf[(size_t)t * SA_TOPF_NR + rank_idx] * w) * seed_scale;   // (the arithmetic of sa_k_make_bounds)
            if (sd1 > seed) seed = sd1;
        }
        std::stable_sort(ord, ord + T, [&](u32 a, u32 c) { return ub[a] > ub[c]; });
        u32 inv = 0;
        double sfx = 0.0;
        h_psfx[(size_t)q * (T + 1) + T] = 0.f;
        for (int i = (int)T - 1; i >= 0; i--) {
            const u32 s = ord[i];
            h_pu[(size_t)q * T + i] = (unsigned short)un[s];
            h_pw[(size_t)q * T + i] = row_idf[(size_t)q * T + s];
            h_pub[(size_t)q * T + i] = ub[s];
            sfx += (double)ub[s];
            h_psfx[(size_t)q * (T + 1) + i] = sfx > 0.0 ? sa_float_up(sfx * (double)SA_ST_MARGIN) : 0.f;
            inv |= (u32)i << (4u * s);
        }
        h_inv[q] = inv;
        u32 sb; memcpy(&sb, &seed, 4);
        h_seed[q] = seed > 0.f ? sb : 0u;
    }
    bt->st_U = U; bt->st_docs = docs; bt->st_tmax = tmax;
    bt->st_cb[0] = cb[0]; bt->st_cb[1] = cb[1]; bt->st_cb[2] = cb[2];
    bt->st_dir = sd;
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

template <int TMAX>
__global__ void __launch_bounds__(SA_ST_NT) sa_k_bm25_stage(const StageParams sp) {
    constexpr int CAP = SaStCap<TMAX>::v;
    constexpr int NT = SA_ST_NT, NW = NT / SA_WAVE, KT = SA_ST_UMAX / NT;
    static_assert(SA_ST_UMAX <= CAP, "a single document's postings must fit the stage");
    static_assert(CAP <= 65535, "16-bit stage offsets");
    __shared__ alignas(16) u64 s_post[CAP];                      // the stage: every distinct term's slice of this tile, doc-sorted
    __shared__ u32 s_off[SA_ST_UMAX];                           // per distinct term: start << 16 | postings
    __shared__ u32 s_lo[SA_ST_UMAX];                            // its first posting, relative to the term's base
    __shared__ unsigned short s_pu[SA_ST_BMAX * TMAX];
    __shared__ float s_pw[SA_ST_BMAX * TMAX];
    __shared__ float s_pub[SA_ST_BMAX * TMAX];
    __shared__ float s_psfx[SA_ST_BMAX * (TMAX + 1)];
    __shared__ u32 s_thr[SA_ST_BMAX];
    __shared__ u32 s_wl[SA_ST_WL];
    __shared__ u32 s_ref[SA_ST_REF];
    __shared__ u32 s_nref;
    __shared__ u32 s_red[NW + 1];
    const u32 tid = threadIdx.x, lane = tid & (SA_WAVE - 1), wave = tid / SA_WAVE;
    const u32 T = sp.T, B = sp.B, U = sp.U;
    // XCD-aware tile ranges: block b runs on XCD b % 8; an XCD walks a contiguous range of tiles and its workgroups
    // contiguous sub-ranges -- a term's slices of neighbouring tiles are neighbours in memory, so the cache line a slice
    // shares with the next tile's is fetched into ONE L2
    const u32 xcd = blockIdx.x & 7u, wg = blockIdx.x >> 3;
    const u32 t_begin = xcd * sp.tpx + wg * sp.tpw;
    u32 t_end = t_begin + sp.tpw;
    if (t_end > (xcd + 1u) * sp.tpx) t_end = (xcd + 1u) * sp.tpx;
    if (t_end > sp.n_st) t_end = sp.n_st;
    if (t_begin >= t_end) return;                               // (uniform)

    // the queries' tables, once per workgroup
    for (u32 i = tid; i < B * T; i += NT) { s_pu[i] = sp.pu[i]; s_pw[i] = sp.pw[i]; s_pub[i] = sp.pub[i]; }
    for (u32 i = tid; i < B * (T + 1u); i += NT) s_psfx[i] = sp.psfx[i];
    if (tid == 0) s_nref = 0u;
    const bool hasq = tid < B;
    const u32 seed = (hasq && sp.seed) ? sp.seed[tid] : 0u;
    // this thread's terms: cursor = first posting not yet staged
    u64 cell0[KT]; u32 row[KT], lo[KT];
#pragma unroll
    for (int kx = 0; kx < KT; kx++) {
        const u32 u = tid + (u32)kx * NT;
        cell0[kx] = 0; row[kx] = SA_ST_NOROW; lo[kx] = 0;
        if (u < U) {
            const StTerm t = sp.terms[u];
            cell0[kx] = t.cell0; row[kx] = t.row;
            if (t.row != SA_ST_NOROW) lo[kx] = sp.dir[(u64)t.row * sp.dir_stride + t_begin];
            else {
                const u32 key = (u32)((u64)t_begin * sp.docs) << 2;
                u32 a = 0, b = t.df;
                while (a < b) { const u32 mid = a + ((b - a) >> 1); if ((u32)(sp.imp[t.cell0 + mid] >> 32) < key) a = mid + 1u; else b = mid; }
                lo[kx] = a;
            }
        }
    }
    __syncthreads();

    for (u32 tile = t_begin; tile < t_end; tile++) {
        const u64 tile_d0 = (u64)tile * sp.docs;
        const u64 tile_d1 = tile_d0 + sp.docs < sp.n_docs ? tile_d0 + sp.docs : sp.n_docs;
        // the queries' bounds (a bound only ever rises: a stale one is valid)
        u32 g = hasq ? __hip_atomic_load(&sp.gthr[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        g = g > seed ? g : seed;
        // end of every term's slice of this tile
        u32 hi_t[KT];
#pragma unroll
        for (int kx = 0; kx < KT; kx++) {
            const u32 u = tid + (u32)kx * NT;
            hi_t[kx] = lo[kx];
            if (u < U) {
                if (row[kx] != SA_ST_NOROW) hi_t[kx] = sp.dir[(u64)row[kx] * sp.dir_stride + tile + 1u];
                else {
                    // (the sentinel behind a term's postings has the doc field all ones: the walk stops there)
                    const u32 key = (u32)tile_d1 << 2;
                    u32 c = lo[kx];
                    while ((u32)(sp.imp[cell0[kx] + c] >> 32) < key) c++;
                    hi_t[kx] = c;
                }
            }
        }
        // A tile whose postings do not fit the stage is taken in doc sub-ranges: halve the range until it fits (a single
        // document holds at most U <= CAP postings), the slices' ends by a search of the posting lists.
        u64 d_s = tile_d0;
        while (d_s < tile_d1) {                                 // (uniform)
            u64 d_e = tile_d1;
            u32 hi[KT];
#pragma unroll
            for (int kx = 0; kx < KT; kx++) hi[kx] = hi_t[kx];
            u32 excl, P;
            for (;;) {
                u32 mine = 0;
#pragma unroll
                for (int kx = 0; kx < KT; kx++) mine += hi[kx] - lo[kx];
                excl = sa_block_excl_scan<NW>(mine, s_red, &P);
                if (P <= (u32)CAP || d_e - d_s <= 1ull) break;
                d_e = d_s + ((d_e - d_s) >> 1);
                const u32 key = (u32)d_e << 2;
#pragma unroll
                for (int kx = 0; kx < KT; kx++) {
                    u32 a = lo[kx], b = hi[kx];
                    while (a < b) { const u32 mid = a + ((b - a) >> 1); if ((u32)(sp.imp[cell0[kx] + mid] >> 32) < key) a = mid + 1u; else b = mid; }
                    hi[kx] = a;
                }
            }
            {
                u32 o = excl;
#pragma unroll
                for (int kx = 0; kx < KT; kx++) {
                    const u32 u = tid + (u32)kx * NT;
                    if (u < U) { const u32 n = hi[kx] - lo[kx]; s_off[u] = (o << 16) | n; s_lo[u] = lo[kx]; o += n; }
                }
            }
            __syncthreads();
            // ---- stage: copy the slices.  Terms are ordered by df: the first classes get 64 / 32 / 16 lanes per term, the rest 8
            {
                u32 c0 = 0;
#pragma unroll 1
                for (int c = 0; c < 4; c++) {
                    const u32 c1 = c < 3 ? (sp.cb[c] < U ? sp.cb[c] : U) : U;
                    const u32 gsh = 6u - (u32)c, G = 1u << gsh, per = 64u >> gsh;      // lanes per term, terms per wave step
                    const u32 nsteps = (c1 - c0 + per - 1u) / per;
                    for (u32 step = wave; step < nsteps; step += NW) {
                        const u32 u = c0 + step * per + (lane >> gsh);
                        if (u < c1) {
                            const u32 pk = s_off[u];
                            const u32 start = pk >> 16, n = pk & 0xFFFFu;
                            const u64* src = sp.imp + sp.terms[u].cell0 + s_lo[u];
                            for (u32 j = lane & (G - 1u); j < n; j += 4u * G) {
                                const u32 j1 = j + G, j2 = j + 2u * G, j3 = j + 3u * G;
                                const u64 v0 = src[j];
                                const u64 v1 = j1 < n ? src[j1] : 0ull;
                                const u64 v2 = j2 < n ? src[j2] : 0ull;
                                const u64 v3 = j3 < n ? src[j3] : 0ull;
                                s_post[start + j] = v0;
                                if (j1 < n) s_post[start + j1] = v1;
                                if (j2 < n) s_post[start + j2] = v2;
                                if (j3 < n) s_post[start + j3] = v3;
                            }
                        }
                    }
                    c0 = c1;
                }
            }
            __syncthreads();
            // ---- the queries: bound, essential positions, candidates
            u32 ncand = 0, ness = 0;
            if (hasq) {
                const u32 thr = g > 1u ? g : 1u;
                const float thr_f = __uint_as_float(thr);
                for (u32 i = 0; i < T; i++) if (s_psfx[tid * (T + 1u) + i] >= thr_f) ness = i + 1u;
                for (u32 i = 0; i < ness; i++) ncand += s_off[s_pu[tid * T + i]] & 0xFFFFu;
                s_thr[tid] = thr;
            }
            u32 C;
            const u32 o_q = sa_block_excl_scan<NW>(ncand, s_red, &C);
            for (u32 c0 = 0; c0 < C; c0 += (u32)SA_ST_WL) {      // (uniform)
                // thread q writes the records of its candidates that fall into this chunk: query | position << 9 | posting << 12
                if (ncand) {
                    const u32 a = o_q > c0 ? o_q : c0;
                    const u32 e = o_q + ncand < c0 + (u32)SA_ST_WL ? o_q + ncand : c0 + (u32)SA_ST_WL;
                    if (a < e) {
                        u32 r = a - o_q, i = 0;
                        u32 ni = s_off[s_pu[tid * T]] & 0xFFFFu;
                        while (r >= ni) { r -= ni; i++; ni = s_off[s_pu[tid * T + i]] & 0xFFFFu; }
                        for (u32 x = a; x < e; x++) {
                            s_wl[x - c0] = tid | (i << 9) | (r << 12);
                            r++;
                            while (r >= ni && x + 1u < e) { r = 0; i++; ni = s_off[s_pu[tid * T + i]] & 0xFFFFu; }
                        }
                    }
                }
                __syncthreads();
                const u32 nchunk = C - c0 < (u32)SA_ST_WL ? C - c0 : (u32)SA_ST_WL;
                for (u32 x = tid; x < nchunk; x += NT) {
                    const u32 rec = s_wl[x];
                    const u32 q = rec & 0x1FFu, i_src = (rec >> 9) & 7u, j = rec >> 12;
                    const u32 qb = q * T;
                    const float thr_f = __uint_as_float(s_thr[q]);
                    const u64 v = s_post[(s_off[s_pu[qb + i_src]] >> 16) + j];
                    const u32 d4 = (u32)(v >> 32);
                    float known = __fmul_rn(__uint_as_float((u32)v), s_pw[qb + i_src]);
                    const float ub_src = s_pub[qb + i_src];
                    bool alive = true;
                    for (u32 i = 0; i < T; i++) {
                        if (alive && i != i_src) {
                            // what the positions from i on can still add (the candidate's own term is already in `known`)
                            const float rem = s_psfx[q * (T + 1u) + i] - (i < i_src ? ub_src : 0.f);
                            if (__fmul_rn(__fadd_rn(known, rem), SA_ST_MARGIN) < thr_f) alive = false;
                            else {
                                const u32 u = s_pu[qb + i];
                                if (u != SA_ST_NONE) {
                                    bool found;
                                    const float f = sa_st_lookup(s_post, s_off[u], d4, found);
                                    if (found) {
                                        if (i < i_src) alive = false;    // the doc is the candidate of that (essential, higher) position
                                        else known = __fadd_rn(known, __fmul_rn(f, s_pw[qb + i]));
                                    }
                                }
                            }
                        }
                    }
                    if (alive && __fmul_rn(known, SA_ST_MARGIN) >= thr_f) {
                        // the exact score: factor * weight per term, summed in QUERY-TERM order (bm25.pyx:19-23, np.sum over the terms)
                        const u32 inv = sp.inv[q];
                        float S = 0.f;
                        for (u32 s = 0; s < T; s++) {
                            const u32 i = (inv >> (4u * s)) & 15u;
                            const u32 u = s_pu[qb + i];
                            float xs = 0.f;
                            if (u != SA_ST_NONE) { bool found; xs = __fmul_rn(sa_st_lookup(s_post, s_off[u], d4, found), s_pw[qb + i]); }
                            S = __fadd_rn(S, xs);
                        }
                        const u32 sb = __float_as_uint(S);
                        if (sb >= s_thr[q]) {
                            const u64 doc = sp.doc_base + (u64)(d4 >> 2);
                            const u32 pos = atomicAdd(&sp.cand_cnt[q], 1u);
                            if (pos < sp.cand_cap) sp.cand[(u64)q * sp.cand_cap + pos] = ((u64)sb << 32) | (u64)(u32)(~(u32)doc);
                            atomicAdd(&sp.hist[(u64)q * SA_HBINS + sa_score_bin(sb)], 1u);
                            if (((pos + 1u) & 31u) == 0u) { const u32 sl = atomicAdd(&s_nref, 1u); if (sl < (u32)SA_ST_REF) s_ref[sl] = q; }
                        }
                    }
                }
                __syncthreads();
            }
            // bounds re-derived for the queries whose candidate list crossed a multiple of 32 entries
            __syncthreads();
            const u32 nref = s_nref < (u32)SA_ST_REF ? s_nref : (u32)SA_ST_REF;
            if (nref) {                                         // (uniform)
                for (u32 r = wave; r < nref; r += NW) {
                    const u32 q = s_ref[r];
                    sa_hist_refresh(sp.hist + (u64)q * SA_HBINS, &sp.gthr[q], sp.k, lane);
                }
                __syncthreads();
                if (tid == 0) s_nref = 0u;
            }
#pragma unroll
            for (int kx = 0; kx < KT; kx++) lo[kx] = hi[kx];
            d_s = d_e;
        }
    }
}

int sa_launch_stage(sa_batch* bt, const Bm25Params& p, hipStream_t st) {
    sa_index* ix = bt->ix;
    if (!bt->stage_ok || !bt->st_dir || !p.hist || !p.gthr || !p.imp) { sa_set_error("staged route: no plan"); return SA_ERR_STATE; }
    const StLayout L = sa_stage_layout(bt->B, bt->T);
    StageParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.imp = p.imp;
    sp.dir = bt->st_dir->d_dir; sp.dir_stride = bt->st_dir->n_st + 1u;
    sp.docs = bt->st_docs; sp.n_st = bt->st_dir->n_st;
    sp.n_docs = ix->n_docs; sp.doc_base = ix->doc_base;
    sp.terms = (const StTerm*)(bt->d_st + L.terms); sp.U = bt->st_U;
    sp.cb[0] = bt->st_cb[0]; sp.cb[1] = bt->st_cb[1]; sp.cb[2] = bt->st_cb[2];
    sp.B = bt->B; sp.T = bt->T; sp.k = bt->k;
    sp.pu = (const unsigned short*)(bt->d_st + L.pu);
    sp.pw = (const float*)(bt->d_st + L.pw); sp.pub = (const float*)(bt->d_st + L.pub); sp.psfx = (const float*)(bt->d_st + L.psfx);
    sp.inv = (const u32*)(bt->d_st + L.inv);
    sp.seed = p.seed;
    sp.gthr = p.gthr; sp.hist = p.hist;
    sp.cand = p.cand; sp.cand_cap = p.cand_cap; sp.cand_cnt = p.cand_cnt;
    if (sp.n_st == 0) return SA_OK;
    const u32 wgs = (u32)std::min<long long>(8, std::max<long long>(1, sa_opt(bt->opts.stage_wgs, 2)));
    u32 grid = (u32)ix->n_cus * wgs / 8u * 8u;
    if (grid < 8u) grid = 8u;
    sp.tpx = (sp.n_st + 7u) / 8u;
    const u32 wpx = grid / 8u;
    sp.tpw = (sp.tpx + wpx - 1u) / wpx;
    if (bt->st_tmax == 4) hipLaunchKernelGGL(sa_k_bm25_stage<4>, dim3(grid), dim3(SA_ST_NT), 0, st, sp);
    else hipLaunchKernelGGL(sa_k_bm25_stage<8>, dim3(grid), dim3(SA_ST_NT), 0, st, sp);
    return SA_OK;
}

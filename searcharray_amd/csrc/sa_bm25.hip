// sa_bm25.hip -- term-at-a-time BM25 over the HBM-resident TF postings, with per-tile and
// per-query top-k.
//
// Replaces the reference's per-term dense pipeline
//     as_dense (scatter, roaringish_ops.pyx:84-98)  ->  _bm25_score (bm25.pyx:11-25, O(N) per term)
//     ->  np.sum(axis=0) (test/test_msmarco.py:353-354)  ->  np.argpartition (utils/sort.py:24)
// with one sparse pass: a workgroup owns (query, doc tile); the tile's fp32 score accumulators
// live in LDS; each query term's slice of the posting stream -- the fat TF postings (doc|doc_len|tf
// in one u64) or, for top-k batches, the impact stream derived from them (doc*4 | fp32 factor) -- is
// read once with coalesced 16-byte loads, scored, and added into LDS in QUERY-TERM ORDER (a barrier
// separates terms) so the fp32 sum is bit-identical to the reference's ((s0+s1)+s2)+s3.
// Per-posting arithmetic is the reference's, op for op, each rounded to fp32 (no FMA contraction,
// IEEE division):  tf / (tf + k1 * ((1 - b) + b * (dl / avgdl))) * idf.
//
// Roofline: HBM-bound integer/bitwise + scalar fp32 work (no MFMA).  Algorithmic bytes per query
// = sum_t 8 * df_t (+ 4 * n_docs the reference-style doc_lens pass would read; this layout folds
// doc_len into the posting so the real traffic is lower).
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_bm25_params.hpp"
#include "../../include/searcharray_hip.h"

#include <algorithm>
#include <type_traits>
#include <new>
#include <stdlib.h>
#include <math.h>


// ---- impact stream ---------------------------------------------------------------------------------
// The exhaustive tile kernel is bound by instruction issue per posting, not by posting bytes.
// Everything of a posting's score except the final `* idf` depends only on (tf, doc_len) and the batch
// constants k1, b, avgdl, so it is evaluated ONCE per (k1, b) for the whole shard -- with the reference's
// operation order (bm25.pyx:19-23), so the bits are the ones the per-posting arithmetic would give -- and
// kept beside the TF postings in the layout the tile kernel wants (sa_impacts in sa_index.hpp): per
// posting the kernel is left with a subtract, a compare, a multiply and the LDS read-modify-write.
sa_impacts::~sa_impacts() {
    if (d_imp || d_dense) hipSetDevice(device);
    if (d_imp) hipFree(d_imp);
    if (d_dense) hipFree(d_dense);
    if (d_topf) hipFree(d_topf);
    if (d_maxf) hipFree(d_maxf);
    if (d_probe) hipFree(d_probe);
    if (d_pbits) hipFree(d_pbits);
}

// dense factor row of one term (sa_impacts::d_dense): row[doc] = factor bits of the term's posting of doc; the row is zeroed before
__global__ void __launch_bounds__(256)
sa_k_make_dense_row(const u64* __restrict__ imp, u64 first, u64 df, float* __restrict__ row) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < df; i += (u64)gridDim.x * blockDim.x) {
        const u64 c = imp[first + i];
        row[(u32)(c >> 32) >> 2] = __uint_as_float((u32)c);
    }
}

// (sa_imp_base / sa_imp_sentinel: sa_bm25_params.hpp)

// Rank table of a term's factors (sa_impacts::d_topf): one workgroup per term builds a histogram of its postings' factor
// bits -- 512 bins per octave over [1/16, 1): (bits >> 14) - (123 << 9), clamped -- and reads off, from the top, the bin
// that holds the r-th largest factor for each tabulated rank r; the bin's LOWER edge is what is stored.
#define SA_TOPF_BINS 2048
#define SA_TOPF_LO (123u << 9)
#define SA_TOPF_SLICE 65536ull          // postings per workgroup of a LONG list's histogram (lists of 4 slices and more; option topf_slice: test hook)
struct TopfRanks { u32 r[SA_TOPF_NR]; };
__global__ void __launch_bounds__(256)
sa_k_topf_hist_long(const u64* __restrict__ imp, const u64* __restrict__ tf_off, const u32* __restrict__ slice_term,
                    const u32* __restrict__ slice_first, const u32* __restrict__ long_slot, u32* __restrict__ ghist, u32* __restrict__ gmax, const u64 slice) {
    // one workgroup per SLICE (SA_TOPF_SLICE postings) of a LONG list: its histogram in LDS, added to the term's row of `ghist`
    __shared__ u32 s_h[SA_TOPF_BINS];
    const u32 tid = threadIdx.x, t = slice_term[blockIdx.x];
    const u64 base = tf_off[t], df = tf_off[t + 1] - base;
    const u64 i0 = (u64)slice_first[blockIdx.x] * slice, i1 = i0 + slice < df ? i0 + slice : df;
    for (u32 i = tid; i < (u32)SA_TOPF_BINS; i += 256) s_h[i] = 0;
    __syncthreads();
    const u64* const cells = imp + sa_imp_base(base, t);
    u32 mx = 0u;
    for (u64 i = i0 + tid; i < i1; i += 256) {
        const u32 fb = (u32)cells[i];
        mx = fb > mx ? fb : mx;
        const u32 e = fb >> 14;
        const u32 b = e <= SA_TOPF_LO ? 0u : (e - SA_TOPF_LO > (u32)(SA_TOPF_BINS - 1) ? (u32)(SA_TOPF_BINS - 1) : e - SA_TOPF_LO);
        atomicAdd(&s_h[b], 1u);
    }
    __syncthreads();
    u32* const row = ghist + (u64)long_slot[t] * SA_TOPF_BINS;
    for (u32 i = tid; i < (u32)SA_TOPF_BINS; i += 256) if (s_h[i]) atomicAdd(&row[i], s_h[i]);
    if (mx) atomicMax(&gmax[long_slot[t]], mx);
}

// (long_slot: [n_terms] row of a LONG term in ghist / gmax -- its histogram is there already, sa_k_topf_hist_long -- else all ones; or null)
__global__ void __launch_bounds__(256)
sa_k_make_topf(const u64* __restrict__ imp, const u64* __restrict__ tf_off, u32 n_terms, const TopfRanks ranks, float* __restrict__ topf,
               float* __restrict__ maxf, const u32* __restrict__ long_slot, const u32* __restrict__ ghist, const u32* __restrict__ gmax) {
    __shared__ u32 s_h[SA_TOPF_BINS];
    __shared__ u32 s_part[256];
    __shared__ u32 s_max;
    const u32 tid = threadIdx.x;
    for (u32 t = blockIdx.x; t < n_terms; t += gridDim.x) {
        const u64 base = tf_off[t], df = tf_off[t + 1] - base;
        float* const row = topf + (u64)t * SA_TOPF_NR;
        if (df == 0) {                                           // (uniform)
            if (tid < (u32)SA_TOPF_NR) row[tid] = 0.f;
            if (tid == 0 && maxf) maxf[t] = 0.f;
            continue;
        }
        const u32 ls = long_slot ? long_slot[t] : 0xFFFFFFFFu;     // (uniform)
        for (u32 i = tid; i < (u32)SA_TOPF_BINS; i += 256) s_h[i] = ls != 0xFFFFFFFFu ? ghist[(u64)ls * SA_TOPF_BINS + i] : 0u;
        if (tid == 0) s_max = ls != 0xFFFFFFFFu ? gmax[ls] : 0u;
        __syncthreads();
        const u64* const cells = imp + sa_imp_base(base, t);
        u32 mx = 0u;                                             // (factors are non-negative: their bit patterns order like the values)
        for (u64 i = tid; i < (ls != 0xFFFFFFFFu ? 0ull : df); i += 256) {
            const u32 fb = (u32)cells[i];
            mx = fb > mx ? fb : mx;
            const u32 e = fb >> 14;
            const u32 b = e <= SA_TOPF_LO ? 0u : (e - SA_TOPF_LO > (u32)(SA_TOPF_BINS - 1) ? (u32)(SA_TOPF_BINS - 1) : e - SA_TOPF_LO);
            atomicAdd(&s_h[b], 1u);
        }
        if (mx) atomicMax(&s_max, mx);
        __syncthreads();
        if (tid == 0 && maxf) maxf[t] = __uint_as_float(s_max);
        // thread i owns bins [8 i, 8 i + 8); above[i] = postings in the bins above its eight
        u32 mine = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) mine += s_h[tid * 8u + (u32)j];
        s_part[tid] = mine;
        __syncthreads();
        u32 above = 0;
        for (u32 j = tid + 1; j < 256; j++) above += s_part[j];
        // bin b holds the r-th largest iff  count(bins > b) < r <= count(bins >= b)
        u32 cum = above;
        for (int j = 7; j >= 0; j--) {
            const u32 b = tid * 8u + (u32)j;
            const u32 c = s_h[b];
            for (int x = 0; x < SA_TOPF_NR; x++)
                if (c && cum < ranks.r[x] && ranks.r[x] <= cum + c)
                    row[x] = b == 0u ? 0.f : __uint_as_float((b + SA_TOPF_LO) << 14);
            cum += c;
        }
        // ranks beyond the term's postings: no bound
        if (tid < (u32)SA_TOPF_NR && ranks.r[tid] > df) row[tid] = 0.f;
        __syncthreads();
    }
}


__global__ void __launch_bounds__(256)
sa_k_make_impacts(const u64* __restrict__ tfp, const u64* __restrict__ tf_off, const float* __restrict__ doc_lens,
                  u32 n_terms, u64 n_postings, int dl_packed, float k1, float b, float avgdl, u64* __restrict__ imp) {
    constexpr u32 CHUNK = 2048;
    __shared__ u32 s_t[2];
    const float one_minus_b = 1.0f - b;
    const u64 n_chunks = (n_postings + CHUNK - 1) / CHUNK;
    for (u64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const u64 first = c * CHUNK;
        const u64 last = first + CHUNK <= n_postings ? first + CHUNK - 1 : n_postings - 1;
        // terms of the chunk's first and last posting: largest t with tf_off[t] <= i
        if (threadIdx.x < 2) {
            const u64 i = threadIdx.x == 0 ? first : last;
            u32 lo = 0, hi = n_terms;                  // answer in [lo, hi)
            while (hi - lo > 1) {
                const u32 mid = lo + (hi - lo) / 2;
                if (tf_off[mid] <= i) lo = mid; else hi = mid;
            }
            s_t[threadIdx.x] = lo;
        }
        __syncthreads();
        const u32 t_first = s_t[0], t_last = s_t[1];
        for (u64 i = first + threadIdx.x; i <= last; i += blockDim.x) {
            u32 lo = t_first, hi = t_last + 1;
            while (hi - lo > 1) {
                const u32 mid = lo + (hi - lo) / 2;
                if (tf_off[mid] <= i) lo = mid; else hi = mid;
            }
            const u64 tbase = tf_off[lo];
            const u64 x = tfp[i];
            const u32 doc = (u32)(x >> SA_KEY_SHIFT);
            const float tf = (float)(u32)(x & SA_LSB_MASK);
            const float dl = dl_packed ? (float)(u32)((x >> SA_LSB_BITS) & SA_LSB_MASK) : doc_lens[doc];
            const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl, avgdl))));
            const float sat = __fdiv_rn(tf, __fadd_rn(tf, norm));
            imp[sa_imp_base(tbase, lo) + (i - tbase)] = ((u64)(doc << 2) << 32) | (u64)__float_as_uint(sat);   // doc * 4: the accumulator's byte offset
        }
        __syncthreads();
    }
}

// The impact stream of (k1, b) for this index: the cached one, or a new one (the cache keeps the most
// recent; batches built earlier keep theirs alive).  Null when switched off (option impact = 0), for an empty
// shard, or when HBM is short -- the tile kernel then scores the TF postings.  Call with the index lock held.
static std::shared_ptr<sa_impacts> sa_impacts_get(sa_index* ix, float k1, float b, const sa_options_t& o) {
    if (sa_opt(o.impact, 1) == 0) return nullptr;
    if (ix->n_postings == 0 || ix->n_terms == 0 || ix->avg_doc_len == 0.f) return nullptr;
    // (bit patterns: a NaN parameter must still find its own stream)
    auto same = [](float x, float y) { return memcmp(&x, &y, sizeof(float)) == 0; };
    if (ix->impacts && same(ix->impacts->k1, k1) && same(ix->impacts->b, b) && same(ix->impacts->avgdl, ix->avg_doc_len))
        return ix->impacts;
    std::shared_ptr<sa_impacts> im(new (std::nothrow) sa_impacts());
    if (!im) return nullptr;
    im->device = ix->device; im->k1 = k1; im->b = b; im->avgdl = ix->avg_doc_len;
    im->n = sa_imp_base(ix->n_postings, ix->n_terms) + 4;
    if (hipMalloc(&im->d_imp, im->n * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        im->d_imp = nullptr;
        return nullptr;
    }
    hipStream_t st = ix->stream;
    if (hipMemsetAsync(im->d_imp, 0xFF, im->n * sizeof(u64), st) != hipSuccess) return nullptr;
    const u64 n_chunks = (ix->n_postings + 2047) / 2048;
    const u32 grid = n_chunks < 16384 ? (u32)n_chunks : 16384u;
    hipLaunchKernelGGL(sa_k_make_impacts, dim3(grid), dim3(256), 0, st, ix->d_tfp, ix->d_tf_off, ix->d_doc_lens,
                       ix->n_terms, ix->n_postings, ix->dl_packed ? 1 : 0, k1, b, ix->avg_doc_len, im->d_imp);
    if (hipGetLastError() != hipSuccess) return nullptr;
    // the stream's last pair: a sentinel with the factor 0.0 (doc field all ones like every sentinel) -- what the grouped
    // kernel's empty half entries point at: their lanes add 0 to their spare slots instead of a NaN
    static const u64 tail[2] = {0xFFFFFFFF00000000ull, 0xFFFFFFFF00000000ull};
    if (hipMemcpyAsync(im->d_imp + im->n - 2, tail, sizeof(tail), hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
    // dense factor rows of the terms with df >= n_docs / dense_div (index option; default 4; 0: none), at most 16, most frequent first
    im->dense_slot.assign(ix->n_terms, 0xFFFFFFFFu);
    {
        const int div = (int)sa_opt(ix->opts.dense_div, 4);
        std::vector<std::pair<u64, u32>> cand;
        if (div > 0 && ix->n_tiles > 0)
            for (u32 t = 0; t < ix->n_terms; t++) {
                const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
                if (df * (u64)div >= ix->n_docs && df > 0) cand.push_back({df, t});
            }
        std::sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& c) { return a.first > c.first || (a.first == c.first && a.second < c.second); });
        if (cand.size() > 16) cand.resize(16);
        if (!cand.empty()) {
            im->dense_stride = (u64)ix->n_tiles * ix->tile_docs;
            const size_t bytes = cand.size() * im->dense_stride * sizeof(float);
            if (hipMalloc(&im->d_dense, bytes) == hipSuccess && hipMemsetAsync(im->d_dense, 0, bytes, st) == hipSuccess) {
                for (size_t r = 0; r < cand.size(); r++) {
                    const u32 t = cand[r].second;
                    const u64 df = cand[r].first;
                    const u32 grid = df / 256 + 1 < 8192 ? (u32)(df / 256 + 1) : 8192u;
                    hipLaunchKernelGGL(sa_k_make_dense_row, dim3(grid), dim3(256), 0, st, (const u64*)im->d_imp,
                                       sa_imp_base(ix->h_tf_off[t], t), df, im->d_dense + r * im->dense_stride);
                    im->dense_slot[t] = (u32)r;
                }
                im->n_dense = (u32)cand.size();
            } else {
                (void)hipGetLastError();
                if (im->d_dense) { hipFree(im->d_dense); im->d_dense = nullptr; }   // (HBM short: the postings serve)
            }
        }
    }
    ix->impacts = im;
    return im;
}

// Rank table of every term's factors (sa_impacts::d_topf: 88 bytes per term), built when the FIRST batch that can use starting
// bounds asks for it -- not with the stream: a batch with term_seed = 0, with k > 1024 or on the pruning route never reads it.
// Tried once per stream; when HBM is short the batches start from 0 as before round 4.  Call with the index lock held.
static void sa_impacts_ensure_topf(sa_index* ix, sa_impacts* im) {
    if (!im || im->d_topf || im->topf_tried) return;
    im->topf_tried = true;
    const size_t bytes = (size_t)ix->n_terms * SA_TOPF_NR * sizeof(float);
    if (hipMalloc(&im->d_topf, bytes) != hipSuccess) { (void)hipGetLastError(); im->d_topf = nullptr; return; }
    TopfRanks rk;
    for (int i = 0; i < SA_TOPF_NR; i++) rk.r[i] = sa_topf_ranks[i];
    if (hipMalloc(&im->d_maxf, (size_t)ix->n_terms * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); im->d_maxf = nullptr; }
    // Long lists (4 slices of 64 K postings and more) are histogrammed slice by slice first: with one workgroup per TERM the ten longest
    // lists of a 10 M-doc shard were 16 ms of the table's build (round 5 review, item 8)
    u32* d_long = nullptr;                                      // [n_terms] slot | [slices] term | [slices] slice | [n_long][BINS] hist | [n_long] max
    const u32 *d_slot = nullptr, *d_ghist = nullptr, *d_gmax = nullptr;
    {
        std::vector<u32> slot(ix->n_terms, 0xFFFFFFFFu), sl_term, sl_first;
        u32 n_long = 0;
        const u64 slice = (u64)std::max<long long>(64, sa_opt(ix->opts.topf_slice, (long long)SA_TOPF_SLICE));
        for (u32 t = 0; t < ix->n_terms; t++) {
            const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
            if (df < 4ull * slice) continue;
            slot[t] = n_long++;
            for (u64 sl = 0; sl * slice < df; sl++) { sl_term.push_back(t); sl_first.push_back((u32)sl); }
        }
        if (n_long) {
            const size_t words = (size_t)ix->n_terms + 2 * sl_term.size() + (size_t)n_long * SA_TOPF_BINS + n_long;
            if (hipMalloc(&d_long, words * sizeof(u32)) == hipSuccess) {
                u32* const d_st = d_long + ix->n_terms;
                u32* const d_sf = d_st + sl_term.size();
                u32* const d_h = d_sf + sl_term.size();
                bool ok = hipMemcpyAsync(d_long, slot.data(), slot.size() * sizeof(u32), hipMemcpyHostToDevice, ix->stream) == hipSuccess &&
                          hipMemcpyAsync(d_st, sl_term.data(), sl_term.size() * sizeof(u32), hipMemcpyHostToDevice, ix->stream) == hipSuccess &&
                          hipMemcpyAsync(d_sf, sl_first.data(), sl_first.size() * sizeof(u32), hipMemcpyHostToDevice, ix->stream) == hipSuccess &&
                          hipMemsetAsync(d_h, 0, ((size_t)n_long * SA_TOPF_BINS + n_long) * sizeof(u32), ix->stream) == hipSuccess;
                if (ok) {
                    hipLaunchKernelGGL(sa_k_topf_hist_long, dim3((u32)sl_term.size()), dim3(256), 0, ix->stream, (const u64*)im->d_imp, (const u64*)ix->d_tf_off,
                                       (const u32*)d_st, (const u32*)d_sf, (const u32*)d_long, d_h, d_h + (size_t)n_long * SA_TOPF_BINS, slice);
                    // (the copies above read pageable host vectors that die with this scope)
                    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ix->stream) == hipSuccess;
                }
                if (ok) { d_slot = d_long; d_ghist = d_h; d_gmax = d_h + (size_t)n_long * SA_TOPF_BINS; }
                else { (void)hipGetLastError(); hipFree(d_long); d_long = nullptr; }
            } else (void)hipGetLastError();
        }
    }
    const u32 grid = ix->n_terms < 16384u ? ix->n_terms : 16384u;
    hipLaunchKernelGGL(sa_k_make_topf, dim3(grid), dim3(256), 0, ix->stream, (const u64*)im->d_imp, (const u64*)ix->d_tf_off, ix->n_terms, rk,
                       im->d_topf, im->d_maxf, d_slot, d_ghist, d_gmax);
    const bool topf_ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ix->stream) == hipSuccess;
    if (d_long) hipFree(d_long);
    if (!topf_ok) {
        (void)hipGetLastError();
        hipFree(im->d_topf); im->d_topf = nullptr;
        if (im->d_maxf) { hipFree(im->d_maxf); im->d_maxf = nullptr; }
        return;
    }
    // host copies (the staged-tile route plans a query set on the host: starting bounds, score bounds of its terms)
    if (im->d_maxf) {
        im->h_topf.resize((size_t)ix->n_terms * SA_TOPF_NR);
        im->h_maxf.resize(ix->n_terms);
        if (hipMemcpy(im->h_topf.data(), im->d_topf, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(im->h_maxf.data(), im->d_maxf, (size_t)ix->n_terms * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            im->h_topf.clear(); im->h_maxf.clear();
        }
    }
}
static bool sa_batch_wants_seed(const sa_batch* bt) {
    return bt->impacts && bt->k <= 1024u && sa_opt(bt->opts.term_seed, 1) != 0 && sa_opt(bt->opts.sparse, -1) != 1;
}
// the staged-tile route (sa_stage.hip) is asked for: option `stage` = 1, or unset while `sparse` is unset too (a caller that
// sets `sparse` chooses between the two older routes)
static bool sa_batch_stage_wanted(const sa_batch* bt) {
    const long long s = sa_opt(bt->opts.stage, -1);
    if (s >= 0) return s != 0;
    return !sa_opt_is_set(bt->opts.sparse);
}

// Saturation table of a batch, laid out [dl][tf - 1].  For integer doc lengths dl < tab_w and term frequencies
// 1 <= tf <= SA_SAT_NTF the per-posting factor  tf / (tf + k1 * ((1 - b) + b * (dl / avgdl)))
// depends only on (tf, dl) and the batch constants, so it is tabulated once per batch with the
// reference's exact operation order (bm25.pyx:19-23: every op rounded to fp32, IEEE division) and
// copied into LDS by each workgroup: scoring a posting becomes one LDS read and one multiply by
// idf instead of two divisions.  Postings outside the table take the arithmetic path.
#define SA_SAT_NTF 8
#define SA_SAT_WMAX 128
#define SA_GRID_Y 32768u      // tiles per grid.y slab of the tile kernel's (queries, tiles) grid

__global__ void sa_k_make_sattab(float* __restrict__ tab, u32 tab_w, float k1, float b, float avgdl) {
    const float one_minus_b = 1.0f - b;
    const u32 n = SA_SAT_NTF * tab_w;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float tf = (float)(i % SA_SAT_NTF + 1), dl = (float)(i / SA_SAT_NTF);   // [dl][tf - 1]: index = dl << 3 | tf - 1
        const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(b, __fdiv_rn(dl, avgdl))));
        tab[i] = __fdiv_rn(tf, __fadd_rn(tf, norm));
    }
}

// Per-batch slice table: for every (query, term, tile) the first posting of that term whose
// doc id is >= tile * TILE, relative to the term's posting base (qbase).  Built once per batch
// (the index is immutable) from the tile directory -- or by a lower-bound search for terms too
// rare to have a directory row -- so the scoring kernel finds its slices with ONE dependent load.
__global__ void __launch_bounds__(256)
sa_k_make_bounds(const u64* __restrict__ tfp, const u64* __restrict__ tf_off, const u32* __restrict__ dir_slot,
                 const u32* __restrict__ tile_dir, u32 n_terms, u32 n_tiles, u32 tile_docs,
                 const u32* __restrict__ terms, u32 BT, u32* __restrict__ bounds, u64* __restrict__ qbase,
                 u64* __restrict__ qbase_imp, const float* __restrict__ topf, const float* __restrict__ idf, u32 T, u32 rank_idx,
                 u32* __restrict__ seed, float seed_scale) {
    const u64 total = (u64)BT * (n_tiles + 1);
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 qt = (u32)(e / (n_tiles + 1)), tile = (u32)(e % (n_tiles + 1));
        const u32 term = terms[qt];
        u32 rel = 0;
        u64 base = 0, ibase = 0, isent = 0;
        if (term < n_terms) {
            base = tf_off[term];
            ibase = sa_imp_base(base, term);
            isent = sa_imp_sentinel(ibase, tf_off[term + 1] - base);
            const u32 cnt = (u32)(tf_off[term + 1] - base);
            const u32 slot = dir_slot[term];
            if (slot != 0xFFFFFFFFu) rel = tile_dir[(u64)slot * (n_tiles + 1) + tile];
            else rel = sa_lower_bound(tfp + base, 0, cnt, ((u64)tile * tile_docs) << SA_KEY_SHIFT, SA_KEY_MASK);
        }
        bounds[e] = rel;
        if (tile == 0) {
            qbase[qt] = base;
            if (qbase_imp) { qbase_imp[2 * (u64)qt] = ibase; qbase_imp[2 * (u64)qt + 1] = isent; }
            // the query's starting bound: the best of  weight x (k-th largest factor of the term, or of a rank beyond k)
            // over its terms (sa_impacts::d_topf; seed[] arrives zeroed with the upload)
            if (seed && term < n_terms) {
                // (seed_scale: 1.0 -- x * 1.0 = x bit for bit -- except under the test hook that forces bounds which are too high)
                const float s = __fmul_rn(__fmul_rn(topf[(u64)term * SA_TOPF_NR + rank_idx], idf[qt]), seed_scale);
                if (s > 0.f) atomicMax(&seed[qt / T], __float_as_uint(s));
            }
        }
    }
}

// MODE 0: dense output and/or block-level threshold top-k (any k <= 1024).
// MODE 1: pruned wave-level top-k against the query's global bound (any k <= 1024), the batch fast path.
// One work item = one (tile, query) pair; `qi` indexes the queries in play (p.qlist maps it, if set).
// IMP: read the impact stream (p.imp / p.qbase_imp) instead of the TF postings.  A batch of 8 postings per
// lane then updates its accumulators together -- 8 LDS reads in flight, the adds, 8 writes -- and a posting
// outside the tile is steered to a spare slot behind the tile instead of being branched around.
// LDS of one tile item: the accumulators (+ one spare slot per lane: acc[TILE + lane]) aliased with the
// MODE 0 selection lists, and the saturation table (TF route) / the selection's per-wave histogram scratch
// (impact route).  Declared by the kernels and handed to the item, so that a kernel which does other work
// on the same tile (sa_k_bm25_group_tiles) can lend the item its own accumulators.
template <int TILE, int MODE>
__host__ __device__ constexpr size_t sa_tile_smem_u64() {
    constexpr int CAP = (TILE >= 8192) ? 2048 : TILE / 4;
    constexpr size_t ACC_BYTES = (size_t)TILE * 4;
    constexpr size_t SEL_BYTES = MODE == 0 ? (size_t)(CAP + SA_KMAX) * 8 : 0;
    return (ACC_BYTES > SEL_BYTES ? ACC_BYTES : SEL_BYTES) / 8 + SA_WAVE / 2;
}
template <int THREADS, bool IMP>
__host__ __device__ constexpr int sa_tile_tab_floats() {
    constexpr int NW = THREADS / SA_WAVE;
    return IMP ? (NW * SA_HBINS / 2 > 64 ? NW * SA_HBINS / 2 : 64) : SA_SAT_NTF * SA_SAT_WMAX;
}

template <int TILE, int THREADS, int MODE, bool IMP>
__device__ __forceinline__ void sa_bm25_tile_item(const Bm25Params& p, const u32 tile, const u32 qi, u64* smem, float* s_tab) {
    constexpr int NW = THREADS / SA_WAVE;
    constexpr int E = TILE / THREADS;
    constexpr int CAP = (TILE >= 8192) ? 2048 : TILE / 4;      // candidate list capacity (MODE 0)
    constexpr int LE = (CAP + THREADS - 1) / THREADS;
    constexpr int PF = 4;                                       // 16-byte loads in flight per lane
    __shared__ u64 red64[NW + 1];
    __shared__ u32 red[NW + 1];
    __shared__ u32 s_cnt[2];
    float* acc = (float*)smem;

    // Work items (tile, query) are dispatched tile-major, one workgroup per item: the items in
    // flight at any moment are the same few tiles across many queries, so posting slices shared
    // by queries are served from L2.  (Resident grids were measured slower on MI355X, twice: walking
    // the items with a static stride -- 3.3-3.7 ms vs 2.7 ms per 256-query batch at 10M docs -- and
    // pulling them from an atomic work queue with the next item's loads requested ahead -- 3.1 ms vs
    // 1.6 ms: DESIGN 3.1.)
    const u32 tid = threadIdx.x;
    const u32 T = p.T;
    const u32 tab_w = p.tab_w;
    if constexpr (!IMP) {
        for (u32 i = tid; i < SA_SAT_NTF * tab_w; i += THREADS) s_tab[i] = p.sattab[i];
    }
    // (queries answered by the sparse candidate path, sa_sparse.hip, are not in the list)
    const u32 q = p.qlist ? p.qlist[qi] : qi;
    const u64 tile_base = (u64)tile * TILE;
    // pruning slots of this query (see the top-k section); loaded first so the L2 latency hides
    // behind the posting stream.  L1-bypassing load: a fresher bound prunes more.
    // (k > 32 on tiles of <= 4 waves: the cached histogram bound instead, see sa_tile_topk_hist)
    constexpr bool HIST_OK = (size_t)NW * SA_HBINS * 2 <= sizeof(float) * SA_SAT_NTF * SA_SAT_WMAX;   // 16-bit bins (<= 8 waves)
    const bool use_hist = MODE == 1 && HIST_OK && p.hist != nullptr;
    u32 slot_val = 0xFFFFFFFFu;
    if (MODE == 1 && !use_hist && (tid & (SA_WAVE - 1)) < 32u)
        slot_val = __hip_atomic_load(&p.slots[q * 32u + (tid & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (use_hist) {
        slot_val = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.seed) { const u32 sd = p.seed[q]; slot_val = sd > slot_val ? sd : slot_val; }
    }

    // 1. this tile's slice [lo, hi) of every query term: one dependent load from the batch's slice
    //    table.  Lane t of EVERY wave requests term t's entries (the same few cache lines), so the
    //    slices reach the scalar registers by readlane -- no LDS staging, no barrier between the lookup
    //    and the first posting loads; the accumulators are cleared while those are in flight.
    const u32 lane = tid & (u32)(SA_WAVE - 1);
    u64 r_lo = 0, r_hi = 0, r_send = 0;                         // r_send: the sentinel pair behind the term (impact stream)
    float r_idf = 0.f;
    if (lane < T) {
        const u32 qt = q * T + lane;
        const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
        u64 base;
        if constexpr (IMP) {
            const sa_u64x2 bs = ((const sa_u64x2*)p.qbase_imp)[qt];
            base = bs.x;
            r_send = bs.y;
        } else {
            base = p.qbase[qt];
        }
        const u32 r0 = row[0], r1 = row[1];
        r_lo = base + r0;
        r_hi = base + r1;
        r_idf = p.idf[qt];
    }
    // the query terms with postings in this tile: phases of the others are skipped altogether
    u32 todo = (u32)__ballot(r_hi > r_lo);
    auto lane64 = [](u64 x, u32 l) -> u64 {                      // value of lane l (wave-uniform l) in scalar registers
        const u32 a = (u32)__builtin_amdgcn_readlane((int)(u32)x, (int)l), b = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), (int)l);
        return ((u64)b << 32) | a;
    };

    // 2. term-at-a-time accumulation.  A slice is read as 16-byte pairs from its 16-byte-aligned
    //    hull (one global_load_dwordx4 per lane, PF in flight); a pair element outside [lo, hi)
    //    belongs to a neighbouring tile and is masked.  The first PF loads of term t+1 are issued
    //    before term t is scored, so short slices cost no extra memory round trip.
    const float k1 = p.k1, bb = p.b, avgdl = p.avgdl;
    const float one_minus_b = 1.0f - bb;
    // Per posting: unpack, saturation-table lookup (or the arithmetic path for out-of-table
    // tf / dl), multiply by idf, and a read-modify-write of the doc's LDS accumulator.  One
    // posting per (term, doc): no two lanes of a term phase touch the same slot.
    auto score_into = [&](u64 x, bool valid, float idf) {
        const u32 d = (u32)((x >> SA_KEY_SHIFT) - tile_base);
        const u32 tfi = (u32)(x & SA_LSB_MASK);
        const u32 dli = (u32)((x >> SA_LSB_BITS) & SA_LSB_MASK);
        if (valid) {
            float sat;
            if (tfi - 1u < (u32)SA_SAT_NTF && dli < tab_w) {
                sat = s_tab[(dli << 3) + tfi - 1u];              // [dl][tf - 1] (SA_SAT_NTF == 8): shift + add, no multiply
            } else {
                const float tf = (float)tfi;
                const float dl = p.dl_packed ? (float)dli : p.doc_lens[tile_base + d];
                const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(bb, __fdiv_rn(dl, avgdl))));
                sat = __fdiv_rn(tf, __fadd_rn(tf, norm));
            }
            acc[d] = __fadd_rn(acc[d], __fmul_rn(sat, idf));
        }
    };
    // Impact stream: the factor is in the posting and a posting is valid iff its doc lies in this tile
    // (terms start on even indices and gaps hold an all-ones sentinel, so whatever else a pair load of
    // the hull brings along -- the previous / next tile's posting, padding -- fails the same test).
    const u32 tile_base_b = (u32)tile_base * 4u;                // the stream holds doc * 4: byte offsets into the accumulators
    auto acc_at = [&](u32 byte_off) -> float& { return *(float*)((char*)acc + byte_off); };
    const u64* const stream = IMP ? p.imp : p.tfp;
    struct Batch { sa_u64x2 v[PF]; };
    // All indices below are relative to a0, the 16-byte-aligned start of the slice's hull: a slice is
    // far shorter than 2^32 postings, so the per-posting bookkeeping is 32-bit.
    // pairs [first, first + PF*THREADS) of the hull of [lo, hi)
    // `send` (impact stream): first cell of the sentinel pair behind the slice's term.  Every load of the
    // impact stream is unconditional and clamped to that pair: what lies between the end of the slice and
    // it are the term's postings of later tiles, which fail the doc test like the sentinels themselves --
    // no per-lane bounds test, no fill value, nothing to invalidate.
    auto load_batch = [&](u64 lo, u64 hi, u64 send, u32 first) -> Batch {
        Batch b;
        const u64 a0 = lo & ~1ull;
        const sa_u64x2* pairs = (const sa_u64x2*)(stream + a0);
        if constexpr (IMP) {
            const u32 jc = (u32)((send - a0) >> 1);
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const u32 j = first + (u32)u * THREADS + tid;
                b.v[u] = pairs[j < jc ? j : jc];
            }
        } else {
            const u32 npairs = (hi > a0) ? (u32)((hi - a0 + 1) >> 1) : 0u;
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const u32 j = first + (u32)u * THREADS + tid;
                if (j < npairs) b.v[u] = pairs[j];
                else { b.v[u].x = 0; b.v[u].y = 0; }
            }
        }
        return b;
    };
    // impact stream: one pair per lane
    auto score_pair = [&](const sa_u64x2& v, float idf) {
        const u32 spare = ((u32)TILE + (tid & (u32)(SA_WAVE - 1))) * 4u;
        const u32 d0 = (u32)(v.x >> 32) - tile_base_b, d1 = (u32)(v.y >> 32) - tile_base_b;
        const u32 s0 = d0 < (u32)TILE * 4u ? d0 : spare, s1 = d1 < (u32)TILE * 4u ? d1 : spare;
        const float v0 = acc_at(s0), v1 = acc_at(s1);
        const float w0 = __fadd_rn(v0, __fmul_rn(__uint_as_float((u32)v.x), idf));
        const float w1 = __fadd_rn(v1, __fmul_rn(__uint_as_float((u32)v.y), idf));
        acc_at(s0) = w0;
        acc_at(s1) = w1;
    };
    auto score_batch = [&](const Batch& b, u64 lo, u64 hi, u32 first, float idf) {
        const u64 a0 = lo & ~1ull;
        const u32 npairs = (hi > a0) ? (u32)((hi - a0 + 1) >> 1) : 0u;
        if constexpr (IMP) {
            const u32 spare = ((u32)TILE + (tid & (u32)(SA_WAVE - 1))) * 4u;
            if (first + (tid & ~(u32)(SA_WAVE - 1)) >= npairs) return;     // wave-uniform: no pair of this batch is this wave's
            if (npairs - first <= (u32)THREADS) {               // the last step of a slice (a short slice's only one)
                score_pair(b.v[0], idf);
                return;
            }
            // lanes / steps past the end of the slice hold sentinels
            u32 slot[2 * PF];
            float val[2 * PF];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const u32 d0 = (u32)(b.v[u].x >> 32) - tile_base_b, d1 = (u32)(b.v[u].y >> 32) - tile_base_b;
                slot[2 * u] = d0 < (u32)TILE * 4u ? d0 : spare;
                slot[2 * u + 1] = d1 < (u32)TILE * 4u ? d1 : spare;
            }
#pragma unroll
            for (int i = 0; i < 2 * PF; i++) val[i] = acc_at(slot[i]);
#pragma unroll
            for (int u = 0; u < PF; u++) {
                val[2 * u] = __fadd_rn(val[2 * u], __fmul_rn(__uint_as_float((u32)b.v[u].x), idf));
                val[2 * u + 1] = __fadd_rn(val[2 * u + 1], __fmul_rn(__uint_as_float((u32)b.v[u].y), idf));
            }
#pragma unroll
            for (int i = 0; i < 2 * PF; i++) acc_at(slot[i]) = val[i];
            return;
        }
        const u32 r_lo = (u32)(lo - a0), r_hi = (u32)(hi - a0);     // the slice inside its hull
#pragma unroll
        for (int u = 0; u < PF; u++) {
            // wave-uniform skip: no lane of this wave has a pair at this step
            const u32 jw = first + (u32)u * THREADS + (tid & ~(u32)(SA_WAVE - 1));
            if (jw >= npairs) continue;
            const u32 r0 = 2u * (first + (u32)u * THREADS + tid);
            score_into(b.v[u].x, r0 >= r_lo && r0 < r_hi, idf);
            score_into(b.v[u].y, r0 + 1u < r_hi, idf);           // r0 + 1 >= r_lo always holds
        }
    };
    // Only terms with postings in this tile get a phase (and its barrier): an empty phase still costs
    // address arithmetic and a barrier -- measured 0.15 ms per term and launch at 10 M docs x 256
    // queries, more than the postings of the rare terms themselves.
    if (MODE == 1 && todo == 0u) return;                        // nothing scored: no doc can enter the top-k
    // Phases are taken in groups of four.  A group's prologue moves the slices (lo, hi, idf) of its
    // terms into scalar registers and requests the first 16 bytes per lane of ALL its phases (P[0..3]);
    // the rest of a phase's first batch -- only slices longer than one step have one -- is requested
    // one phase ahead (nxt).  Inside the group a phase starts from registers: a phase with few
    // postings (the rare terms of a query) otherwise spends far longer on address arithmetic and its
    // own memory round trip than on its postings (measured: 0.2 ms per term and
    // launch at 10 M docs x 256 queries, whatever the number of postings).
    auto load_step = [&](u64 lo, u64 hi, u64 send, u32 step) -> sa_u64x2 {     // step `step` of the first batch
        const u64 a0 = lo & ~1ull;
        const sa_u64x2* pairs = (const sa_u64x2*)(stream + a0);
        const u32 j = step * THREADS + tid;
        if constexpr (IMP) {
            const u32 jc = (u32)((send - a0) >> 1);
            return pairs[j < jc ? j : jc];
        } else {
            const u32 npairs = (hi > a0) ? (u32)((hi - a0 + 1) >> 1) : 0u;
            sa_u64x2 v;
            if (j < npairs) v = pairs[j];
            else { v.x = 0; v.y = 0; }
            return v;
        }
    };
    auto pairs_of = [](u64 lo, u64 hi) -> u32 {
        const u64 a0 = lo & ~1ull;
        return (hi > a0) ? (u32)((hi - a0 + 1) >> 1) : 0u;
    };
    // steps 1 .. PF-1 of a slice's first batch (nothing to fetch for a slice of one step)
    auto load_rest = [&](Batch& b, u64 lo, u64 hi, u64 send) {
        if (pairs_of(lo, hi) > (u32)THREADS) {
#pragma unroll
            for (int u = 1; u < PF; u++) b.v[u] = load_step(lo, hi, send, (u32)u);
        } else {
            const u64 fill = IMP ? ~0ull : 0ull;
#pragma unroll
            for (int u = 1; u < PF; u++) { b.v[u].x = fill; b.v[u].y = fill; }
        }
    };
    // The first loads of a group are UNCONDITIONAL (TF postings: lanes past the end of a slice re-read its last
    // pair and are zeroed when the pair is consumed) and issued in the order  first batch of phase 0, second
    // batch of phase 0 if its slice is long, first step of phases 1..3 : the compiler can then count them,
    // and phase 0 -- peeled out of the loop below -- waits with vmcnt(3) for its own data only.  The slices of
    // the rare terms (never shared between queries, so they come from HBM, not L2) arrive while the
    // frequent first term is being scored instead of holding the workgroup up before it starts.
    auto load_step_u = [&](u64 lo, u64 hi, u64 send, u32 step) -> sa_u64x2 {
        if constexpr (IMP) {
            return load_step(lo, hi, send, step);
        } else {
            const u64 a0 = lo & ~1ull;
            const u32 npairs = pairs_of(lo, hi);
            const sa_u64x2* pairs = (const sa_u64x2*)(stream + a0);
            const u32 j = step * THREADS + tid;
            const u32 last = npairs ? npairs - 1u : 0u;
            return pairs[j < last ? j : last];
        }
    };
    auto invalidate = [&](sa_u64x2& v, u32 j, u32 npairs) {
        if constexpr (!IMP) {
            if (j >= npairs) { v.x = 0; v.y = 0; }
        }
    };
    // one term phase: the first batch `cur`, then -- long slices -- the rest, `b` being the second batch (already requested)
    auto run_phase = [&](const Batch& cur, Batch& b, u64 lo, u64 hi, u64 send, float idf) {
        const u32 npairs = pairs_of(lo, hi);
        score_batch(cur, lo, hi, 0, idf);
        if (npairs > (u32)PF * THREADS) {
            u32 first = (u32)PF * THREADS;
            while (first < npairs) {
                const u32 nf = first + (u32)PF * THREADS;
                Batch b2;
                if (nf < npairs) b2 = load_batch(lo, hi, send, nf);
                score_batch(b, lo, hi, first, idf);
                if (nf < npairs) b = b2;
                first = nf;
            }
        }
    };
    bool cleared = false;
    while (todo) {
        u64 L[4], H[4], S[4];
        float W[4];
        sa_u64x2 P[4];
        {
            u32 g = todo;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool have = g != 0u;
                const u32 ti = have ? (u32)__builtin_ctz(g) : 0u;
                g &= g - 1u;                                     // (0 stays 0)
                L[i] = have ? lane64(r_lo, ti) : 0ull;
                H[i] = have ? lane64(r_hi, ti) : 0ull;
                S[i] = (IMP && have) ? lane64(r_send, ti) : 0ull;
                W[i] = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(r_idf), (int)ti));
            }
        }
        // ---- phase 0 of the group
        {
            const u64 lo = L[0], hi = H[0];
            const u32 npairs = pairs_of(lo, hi);
            Batch cur, b;
#pragma unroll
            for (int u = 0; u < PF; u++) cur.v[u] = load_step_u(lo, hi, S[0], (u32)u);
            if (npairs > (u32)PF * THREADS) b = load_batch(lo, hi, S[0], (u32)PF * THREADS);
#pragma unroll
            for (int i = 1; i < 4; i++) P[i] = load_step_u(L[i], H[i], S[i], 0);
            if (!cleared) {                                     // first group: clear the accumulators behind the loads
#pragma unroll
                for (int j = 0; j < E; j++) acc[j * THREADS + tid] = 0.f;
                __syncthreads();
                cleared = true;
            }
            todo &= todo - 1u;
#pragma unroll
            for (int u = 0; u < PF; u++) invalidate(cur.v[u], (u32)u * THREADS + tid, npairs);
            run_phase(cur, b, lo, hi, S[0], W[0]);
            __syncthreads();
        }
        // ---- phases 1..3
        if constexpr (IMP) {
            // Usual case: every further term of the group has a slice of one step (at most THREADS pairs).
            // Such a phase is its pair from P[k], one LDS update and the barrier -- none of the generic
            // phase's bookkeeping, which costs more than the update itself (measured: 0.14 ms per phase and
            // launch at 10 M docs x 256 queries for the generic loop).
            const u32 np1 = pairs_of(L[1], H[1]), np2 = pairs_of(L[2], H[2]), np3 = pairs_of(L[3], H[3]);
            if (np1 <= (u32)THREADS && np2 <= (u32)THREADS && np3 <= (u32)THREADS) {
                const u32 npk[4] = {0u, np1, np2, np3};
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    if (npk[k] == 0u) break;                    // (absent terms are the last ones of a group)
                    todo &= todo - 1u;
                    if ((tid & ~(u32)(SA_WAVE - 1)) < npk[k]) {
                        sa_u64x2 v = P[k];
                        invalidate(v, tid, npk[k]);
                        score_pair(v, W[k]);
                    }
                    __syncthreads();
                }
                continue;
            }
        }
        Batch nxt;
        load_rest(nxt, L[1], H[1], S[1]);
#pragma unroll 1
        for (int k = 1; k < 4 && todo; k++) {
            todo &= todo - 1u;
            L[0] = L[1]; L[1] = L[2]; L[2] = L[3]; L[3] = 0;
            H[0] = H[1]; H[1] = H[2]; H[2] = H[3]; H[3] = 0;
            S[0] = S[1]; S[1] = S[2]; S[2] = S[3]; S[3] = 0;
            W[0] = W[1]; W[1] = W[2]; W[2] = W[3];
            P[0] = P[1]; P[1] = P[2]; P[2] = P[3];
            const u64 lo = L[0], hi = H[0];
            const u32 npairs = pairs_of(lo, hi);
            Batch cur = nxt, b;
            cur.v[0] = P[0];
            invalidate(cur.v[0], tid, npairs);
            load_rest(nxt, L[1], H[1], S[1]);                   // (the next group's first phase: nothing, L[1] = H[1] = 0)
            if (npairs > (u32)PF * THREADS) b = load_batch(lo, hi, S[0], (u32)PF * THREADS);   // requested before the first batch is scored
            run_phase(cur, b, lo, hi, S[0], W[0]);
            __syncthreads();
        }
    }
    if (!cleared) {                                             // MODE 0 without any posting in the tile: all zero
#pragma unroll
        for (int j = 0; j < E; j++) acc[j * THREADS + tid] = 0.f;
        __syncthreads();
    }

    const u64 remain = p.n_docs - tile_base;
    const u32 tile_n = remain < (u64)TILE ? (u32)remain : (u32)TILE;

    // 3. dense drop-in output (SearchArray.score): coalesced tile store
    if (MODE == 0 && p.dense_out) {
        float* out = p.dense_out + (u64)q * p.n_docs + tile_base;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u32 e = j * THREADS + tid;
            if (e < tile_n) out[e] = acc[e];
        }
    }
    const u32 k = p.k;
    if (p.cand && !p.no_topk) {
    if constexpr (MODE == 1) {
        if constexpr (HIST_OK) {
            if (use_hist) {
                // the saturation table is dead (barrier after the last term): its LDS is the waves' scratch
                sa_tile_topk_hist<TILE, THREADS>(acc, slot_val, q, tile, p.doc_base + tile_base, k, p.hist, p.gthr, p.cand,
                                                 p.cand_cap, p.cand_cnt, (u32*)s_tab);
                return;
            }
        }
        sa_tile_topk_pruned<TILE, THREADS>(acc, slot_val, q, tile, p.doc_base + tile_base, k, p.slots, p.cand,
                                           p.cand_cap, p.cand_cnt);
    } else {
    // 4. per-tile top-k -> composite keys  score_bits<<32 | ~global_doc
    u64* cand = p.cand + ((u64)q * p.n_tiles + tile) * p.cand_per_tile;
    u64* sel = smem + CAP;                    // selected keys (aliases acc once keys are in registers)
    u32 nsel = 0;
    // general k: threshold selection.
    u32 key32[E];
    u32 lmax = 0, nnz_local = 0;
#pragma unroll
    for (int j = 0; j < E; j++) {
        key32[j] = __float_as_uint(acc[j * THREADS + tid]);
        lmax = key32[j] > lmax ? key32[j] : lmax;
        nnz_local += key32[j] != 0 ? 1u : 0u;
    }
    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    const u32 nnz = sa_block_sum<NW>(nnz_local, red);            // barriers: acc is dead from here on
    if (nnz > 0) {
        u32 theta = 1;
        if (nnz > k && k <= (u32)THREADS) {
            // k-th largest per-thread maximum is a lower bound of the tile's k-th largest score
            const u64 th = sa_block_kth_largest<1, NW>([&](int) { return (u64)lmax; }, k, red64);
            theta = th > 1 ? (u32)th : 1u;
        }
        u32 c_local = 0;
#pragma unroll
        for (int j = 0; j < E; j++) c_local += key32[j] >= theta ? 1u : 0u;
        const u32 C = sa_block_sum<NW>(c_local, red);
        if (C <= (u32)CAP) {
            // gather the survivors into an LDS list, then select exactly among them
            u64* list = smem;
#pragma unroll
            for (int j = 0; j < E; j++) {
                if (key32[j] >= theta) {
                    const u32 pos = atomicAdd(&s_cnt[0], 1u);
                    list[pos] = ((u64)key32[j] << 32) | (u64)(0xFFFFu - (u32)(j * THREADS + tid));
                }
            }
            __syncthreads();
            u64 lk[LE];
#pragma unroll
            for (int e = 0; e < LE; e++) {
                const u32 idx = e * THREADS + tid;
                lk[e] = idx < C ? list[idx] : 0ull;
            }
            u64 kth = 1;
            if (C > k) kth = sa_block_kth_largest<LE, NW>([&](int e) { return lk[e]; }, k, red64);
#pragma unroll
            for (int e = 0; e < LE; e++) {
                if (lk[e] != 0 && lk[e] >= kth) {
                    const u32 pos = atomicAdd(&s_cnt[1], 1u);
                    if (pos < (u32)SA_KMAX) sel[pos] = lk[e];
                }
            }
        } else {
            // Too many survivors (massive ties).  Rare: bisect over the score bits still sitting in
            // the LDS tile (re-read every step -- slow but register-free), composite keys formed on
            // the fly; `sel` aliases part of the tile, so winners are staged through `list` slots
            // only after the last read.
            const u32* tile_bits = (const u32*)smem;
            u64 m = 0;
            for (int j = 0; j < E; j++) {
                const u32 e = j * THREADS + tid;
                const u32 kb = tile_bits[e];
                const u64 c = kb ? (((u64)kb << 32) | (u64)(0xFFFFu - e)) : 0ull;
                m = c > m ? c : m;
            }
            m = sa_block_max64<NW>(m, red64);
            int top = 63 - __clzll((long long)m);
            if ((top & 1) == 0) top++;
            u64 prefix = 0;
            for (int bit = top; bit >= 1; bit -= 2) {
                const u64 c1 = prefix | (1ull << (bit - 1)), c2 = prefix | (2ull << (bit - 1)), c3 = prefix | (3ull << (bit - 1));
                u64 packed = 0;
                for (int j = 0; j < E; j++) {
                    const u32 e = j * THREADS + tid;
                    const u32 kb = tile_bits[e];
                    const u64 x = kb ? (((u64)kb << 32) | (u64)(0xFFFFu - e)) : 0ull;
                    packed += (x >= c1 ? 1ull : 0ull) + (x >= c2 ? (1ull << 21) : 0ull) + (x >= c3 ? (1ull << 42) : 0ull);
                }
#pragma unroll
                for (int o = SA_WAVE / 2; o > 0; o >>= 1) packed += __shfl_xor(packed, o, SA_WAVE);
                if (sa_lane() == 0) red64[sa_wave_id()] = packed;
                __syncthreads();
                u64 tot = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) tot += red64[w];
                __syncthreads();
                const u32 n1 = (u32)(tot & 0x1FFFFF), n2 = (u32)((tot >> 21) & 0x1FFFFF), n3 = (u32)((tot >> 42) & 0x1FFFFF);
                if (n3 >= k) prefix = c3; else if (n2 >= k) prefix = c2; else if (n1 >= k) prefix = c1;
            }
            const u64 thr = prefix > 1 ? prefix : 1;
            // every thread finished reading the tile (barrier above); now it may be overwritten
#pragma unroll
            for (int j = 0; j < E; j++) {
                const u32 e = j * THREADS + tid;
                const u64 c = key32[j] ? (((u64)key32[j] << 32) | (u64)(0xFFFFu - e)) : 0ull;
                if (c >= thr) {
                    const u32 pos = atomicAdd(&s_cnt[1], 1u);
                    if (pos < (u32)SA_KMAX) sel[pos] = c;
                }
            }
        }
    }
    __syncthreads();
    nsel = s_cnt[1] < k ? s_cnt[1] : k;
    for (u32 i = tid; i < p.cand_per_tile; i += THREADS) {
        u64 o = 0;
        if (i < nsel) {
            const u64 c = sel[i];
            const u64 doc = p.doc_base + tile_base + (0xFFFFu - (u32)(c & 0xFFFFu));
            o = (c & 0xFFFFFFFF00000000ull) | (u64)(u32)(~(u32)doc);
        }
        cand[i] = o;
    }
    }
    }   // top-k
}

template <int TILE, int THREADS, int MODE, bool IMP>
__global__ void __launch_bounds__(THREADS) sa_k_bm25_tiles(const Bm25Params p) {
    // grid (queries, tiles): x runs fastest, so the dispatch order is tile-major without a division
    __shared__ alignas(16) u64 smem[sa_tile_smem_u64<TILE, MODE>()];
    __shared__ float s_tab[sa_tile_tab_floats<THREADS, IMP>()];
    const u32 tile = p.tile0 + blockIdx.z * SA_GRID_Y + blockIdx.y;
    if (tile >= p.tile_end) return;
    sa_bm25_tile_item<TILE, THREADS, MODE, IMP>(p, tile, blockIdx.x, smem, s_tab);
}

// The queries the sparse candidate path handed back (usually none): their number is only known on the
// device, so a resident grid walks the (tile, query) items of the list -- no host round trip to size a
// launch.  Slower per item than one workgroup per item, which does not matter for a rare fallback.
template <int TILE, int THREADS, bool IMP>
__global__ void __launch_bounds__(THREADS) sa_k_bm25_tiles_list(const Bm25Params p) {
    __shared__ alignas(16) u64 smem[sa_tile_smem_u64<TILE, 1>()];
    __shared__ float s_tab[sa_tile_tab_floats<THREADS, IMP>()];
    const u32 nq = *p.nq_dev;
    const u64 n_items = (u64)nq * p.n_tiles;
    for (u64 item = blockIdx.x; item < n_items; item += gridDim.x) {
        sa_bm25_tile_item<TILE, THREADS, 1, IMP>(p, (u32)(item / nq), (u32)(item % nq), smem, s_tab);
        __syncthreads();                              // LDS is reused by the next item
    }
}

// ---------------------------------------------------------------------------------------------
// Grouped exhaustive scoring: queries of a batch that share their FIRST term.
//
// Real batches repeat their frequent terms (the BASELINE query set draws its first term from ten terms),
// and the frequent term is where the postings are: per (tile, query) item of sa_k_bm25_tiles ~87 % of the
// posting loads and LDS updates belong to a term that 25 other queries of the batch score identically --
// same impact stream, same idf, and, the term being FIRST in query order, the same partial sum
// 0 + s0 = s0.  Here ONE wave owns a (tile, group) item:
//   base      the shared first term is scored once into the tile's accumulators ("base");
//   overlay   the queries of the group are taken one after the other: the postings of a query's further
//             terms are added IN PLACE in query-term order (so a doc's sum is ((s0 + s1) + s2) + s3 bit for
//             bit), the written value carrying its sign bit as a "touched by this query" mark; the lane that
//             touched a doc first evaluates the doc's final score against the query's bound, appends it
//             if it survives, and puts the base value back.  Docs a query does not touch score exactly the
//             base value: they need no per-query work at all as long as the tile's largest base value is
//             below the query's bound (else the general path below runs);
//   general   a query whose further terms have more postings in the tile than the overlay holds in registers,
//             or whose bound is not above the base values yet, is left to the per-query item: the (tile, query)
//             pair goes on a work list that sa_k_bm25_tiles_wl walks right after this kernel.
// Every posting of every query term is still read and scored -- nothing is skipped on a score bound; the
// shared term is read once per group instead of once per query.  Results are bit-identical to the per-query
// kernel (same candidates above the same bounds, same merge).  The next query's postings and bound are
// requested before the current one is processed (fixed number of loads per query, so the compiler can
// count them and wait for the current query's data only).
// ---------------------------------------------------------------------------------------------
#define SA_GRP_NH 10        // 64-posting halves of a query's further terms held in registers (640 postings per tile and query)
#define SA_GRP_MAXQ 16      // queries per table pass of a group item
#define SA_GRP_ITEM_ROUNDS 16    // items of 32 queries (two table passes) from this many rounds of 4096 one-pass items on; option group_item

struct GroupParams {
    const u32* grp;         // [n_groups][3]: first device row, number of rows (rows of a group are contiguous), dense factor row of the shared first term or 0xFFFFFFFF
    const float* dense;     // dense factor rows (sa_impacts::d_dense), or null
    u64 dense_stride;
    u32 n_groups;
    u32 tile0, n_tiles_run; // tiles [tile0, tile0 + n_tiles_run)
    u32 tpx;                // tiles per XCD: XCD x takes the RANGE [x * tpx, (x + 1) * tpx) of the run's tiles (0: tiles t = x mod 8)
    u32 tt, tt_shift;       // lanes per query while the step tables are built: power of two >= max(T - 1, 1)
    u32 cq;                 // queries per table pass: an item of more queries builds its tables pass after pass over ONE base
    u64* wl;                // work list of (tile << 32 | device row) items left to the per-query kernel
    u32* wl_cnt;
};

// One HALF = up to 64 postings of ONE term of one query in this tile, one per lane (8-byte loads): every LDS
// instruction of the overlay touches the postings of a single term, i.e. pairwise distinct docs.  Its descriptor is
// ONE 8-byte LDS cell:
//   [47:0]  address of the half's first posting (impact stream)
//   [53:48] lim: the half's last posting; lane L loads posting min(L, lim).  The lanes past the half's postings so
//           hold COPIES of its last posting, and copies are harmless in a read-modify-write of one LDS instruction:
//           all of them read the same value, write the same value and later put the same value back -- the overlay
//           needs no per-lane validity test (the rare evaluation below does: a doc must be reported once)
//   [58:54] which of the query's overlaid terms (its weight sits in the item's weight table at [query][term])
//   [63:60] entry 0 of a query only: its number of halves (15: more than the table holds -> per-query kernel)
// (round 2 kept 16-byte entries with the weight inside: 3 KiB of table; with 8-byte entries, a 256 / 512-byte weight
//  table and a 32-entry survivor buffer an item needs 10 KiB of LDS -- 16 waves per CU instead of 12.)
#define SA_GRPH_LIM_SHIFT 48
#define SA_GRPH_TERM_SHIFT 54
#define SA_GRPH_NH_SHIFT 60
#define SA_GRPH_OVER 15u

// f(0), f(STEP), f(2 STEP), ... while the index is below n (< NH): nested tests, so the chain is left at the first
// index that is not, and every index is a compile-time constant (arrays indexed by it stay in registers)
template <int H, int NH, int STEP, class F>
__device__ __forceinline__ void sa_static_while_below(u32 n, F&& f) {
    if constexpr (H < NH) {
        if ((u32)H < n) {
            f(std::integral_constant<int, H>{});
            sa_static_while_below<H + STEP, NH, STEP>(n, f);
        }
    }
}

// s_waitcnt with vmcnt = 0 and the other counters open (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
#define SA_WAIT_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)

#define SA_GRP_REFRESH_STEP 32
#define SA_GRP_SURV_CAP 32          // survivors an item buffers before it writes them out (a pair has at most 16)
#define SA_GRP_LOOSE_POSTINGS 400   // loose groups: expected postings of a query per tile, all terms together (SA_LOOSE_POSTINGS).  Round 3,
                                    // distinct-terms batch, 10 M docs: 64 / 96 / 128 / 192 / 256 / 384 / 512 -> 0.82 / 0.74 / 0.74 / 0.73 / 0.75 /
                                    // 0.80 / 0.81 ms at k = 10, hence 128.  Round 4, with the starting bounds (the per-query kernel AND the
                                    // overlay both start with a bound): 32 / 64 / 96 / 128 / 192 / 256 / 320 / 400 / 500 / 640 / 900 / all ->
                                    // 1.00 / 0.80 / 0.62 / 0.54 / 0.50 / 0.47 / 0.457 / 0.459 / 0.468 / 0.476 / 0.483 / 0.485 ms at k = 10;
                                    // k = 1000: 128 / 320 / 400 / 500 / 640 -> 0.89 / 0.84 / 0.81 / 0.77 / 0.81

// -DSA_PROBE (scripts/build_probe.sh; never in the product build): where a wave of the grouped kernel spends its cycles --
// s_memtime at the section boundaries, summed over the items that reach the overlay, read by sa_debug_probe_read.  Reading
// the clock waits for the wave's LDS operations (lgkmcnt(0)): sections that end with LDS writes are charged their drain.
#ifdef SA_PROBE
#define SA_PROBE_SLOTS 4096
__device__ unsigned long long g_sa_probe[SA_PROBE_SLOTS * 16];       // (slot = block mod SLOTS: no two waves in flight share cells)
#define SA_PT(i) do { const u64 t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - plast; plast = t_; } while (0)
#ifdef SA_PROBE_FINE      // sub-sections of the item's preamble: vector loads are waited for where they are charged
#define SA_PTV(i) do { __builtin_amdgcn_s_waitcnt(0x0F70); SA_PT(i); } while (0)
#else
#define SA_PTV(i) do { } while (0)
#endif
extern "C" int sa_debug_probe_read(unsigned long long* out16, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return SA_ERR_HIP;
    std::vector<unsigned long long> h((size_t)SA_PROBE_SLOTS * 16, 0ull);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_sa_probe), h.size() * 8) != hipSuccess) return SA_ERR_HIP;
    for (int i = 0; i < 16; i++) { out16[i] = 0; for (int sl = 0; sl < SA_PROBE_SLOTS; sl++) out16[i] += h[(size_t)sl * 16 + i]; }
    if (clear) { std::fill(h.begin(), h.end(), 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_sa_probe), h.data(), h.size() * 8) != hipSuccess) return SA_ERR_HIP; }
    return SA_OK;
}
#else
#define SA_PT(i) do { } while (0)
#define SA_PTV(i) do { } while (0)
#endif

// IDFN: cells of the item's weight table = queries x lanes-per-query of the table build (64: up to 4 overlaid terms per
// query -- the BASELINE shape --, 128: anything else the host admits, n * tt <= 128)
template <int TILE, int IDFN>
__global__ void __launch_bounds__(64, 4) sa_k_bm25_group_tiles(const Bm25Params p, const GroupParams gp) {
    constexpr int NH = SA_GRP_NH;
    static_assert(NH < (int)SA_GRPH_OVER && NH >= 8, "half count field");
    __shared__ alignas(16) u64 smem[sa_tile_smem_u64<TILE, 1>()];
    __shared__ u64 s_half[SA_GRP_MAXQ][NH];
    __shared__ float s_idf[IDFN];
    __shared__ u64 s_surv[SA_GRP_SURV_CAP];                     // survivors waiting for their places: score bits << 32 | accumulator offset << 6 | query of the item
    u32* const accu = (u32*)smem;
    const u32 lane = threadIdx.x;
#ifdef SA_PROBE
    u64 pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 plast = __builtin_amdgcn_s_memtime();
#endif
    // XCD-aware item order: consecutive blocks go to consecutive XCDs (block b runs on XCD b % 8), so the
    // eight tiles of a chunk sit on eight XCDs and ALL groups of a tile follow each other on the same XCD:
    // the slices of the further terms that queries of different groups share are fetched into one L2 only.
    const u32 per = 8u * gp.n_groups;
    const u32 chunk = blockIdx.x / per, r = blockIdx.x % per;
    const u32 g = r >> 3;
    // ... and an XCD walks a RANGE of consecutive tiles: a term's slices of neighbouring tiles are neighbours in memory, and
    // with tiles dealt round-robin the cache line a tile's slice shares with the next tile's was fetched into two L2s -- for a
    // sparse term (4 postings = 32 bytes per tile) every line into four.
    const u32 trel = gp.tpx ? (r & 7u) * gp.tpx + chunk : chunk * 8u + (r & 7u);
    if (trel >= gp.n_tiles_run) return;
    const u32 tile = gp.tile0 + trel;
    // a LOOSE group (bit 31 of the size): queries that share nothing -- no base, ALL their terms are overlaid on
    // cleared accumulators (0 + s0 = s0, so the sums are the same); what they share is the item's fixed cost
    const u32 row0 = gp.grp[3 * g], n_raw = gp.grp[3 * g + 1], n = n_raw & 0x7FFFFFFFu;
    const u32 dslot = gp.dense ? gp.grp[3 * g + 2] : 0xFFFFFFFFu;
    const bool loose = (n_raw >> 31) != 0u;
    const u32 T = p.T;
    const u64 tile_base = (u64)tile * TILE;
    const u32 tile_base_b = (u32)tile_base * 4u;
    const u32 spare = ((u32)TILE + lane) * 4u;                  // byte offset of this lane's spare slot
    const u64* const stream = p.imp;
    auto at = [&](u32 byte_off) -> u32& { return *(u32*)((char*)accu + byte_off); };
    auto ballot = [](bool c) -> u64 { return (u64)__builtin_amdgcn_ballot_w64(c); };

    // ---- the shared first term
    const u32 qt0 = row0 * T;
    const u32* hrow = p.bounds + (u64)qt0 * (p.n_tiles + 1) + tile;
    const u32 h0 = loose ? 0u : hrow[0], h1 = loose ? 0u : hrow[1];
    const sa_u64x2 hbs = ((const sa_u64x2*)p.qbase_imp)[qt0];
    const float hidf = p.idf[qt0];

    // ---- half tables of the queries' further terms: lane (qi, t) looks up term t's slice of query qi in this
    //      tile and writes one entry per 64 postings, in query-term order
    const u32 TT = gp.tt, tsh = gp.tt_shift, QPP = 64u >> tsh;
    // round 5: an item holds up to 64 queries of its group and takes them in PASSES of gp.cq (<= 16): the tables are built per
    // pass, the base -- and everything in front of it: block -> (tile, group), group entry, the shared term's slice and dense row --
    // once per item (cq0: first query of the pass, nq: its queries)
    u32 cq0 = 0u, nq = n < gp.cq ? n : gp.cq;
    struct Pre { u32 r0, r1; u64 base, send; float idf; };
    auto pre_load = [&](u32 ps) -> Pre {
        Pre x; x.r0 = 0; x.r1 = 0; x.base = 0; x.send = 0; x.idf = 0.f;
        const u32 qi = ps * QPP + (lane >> tsh), t = (loose ? 0u : 1u) + (lane & (TT - 1u));
        if (qi < nq && t < T) {
            const u32 qt = (row0 + cq0 + qi) * T + t;
            const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
            const sa_u64x2 bs = ((const sa_u64x2*)p.qbase_imp)[qt];
            x.base = bs.x; x.send = bs.y;
            x.r0 = row[0]; x.r1 = row[1]; x.idf = p.idf[qt];
        }
        return x;
    };
    auto pre_store = [&](u32 ps, const Pre& x) {
        const u32 qi = ps * QPP + (lane >> tsh), tl = lane & (TT - 1u);
        const u32 np = x.r1 - x.r0;                             // postings of this term in this tile
        const u32 halves = (np + 63u) >> 6;
        u32 incl = halves;                                      // inclusive scan over the TT lanes of a query
        for (u32 o = 1; o < TT; o <<= 1) {
            const u32 up = __shfl_up(incl, o, SA_WAVE);
            if (tl >= o) incl += up;
        }
        const u32 excl = incl - halves;
        const u32 total = (u32)__shfl((int)incl, (int)(lane | (TT - 1u)), SA_WAVE);
        if (qi < nq) {
            s_idf[qi * TT + tl] = x.idf;
            const u64 c0 = x.base + x.r0;                       // first cell of the slice
            const u64 nhf = (u64)(total <= (u32)NH ? total : SA_GRPH_OVER) << SA_GRPH_NH_SHIFT;
            for (u32 j = 0; j < halves && excl + j < (u32)NH; j++) {
                const u64 c = c0 + (u64)j * 64ull;
                const u64 lim = (np - j * 64u < 64u ? np - j * 64u : 64u) - 1u;     // last posting of the half
                s_half[qi][excl + j] = (u64)(stream + c) | (lim << SA_GRPH_LIM_SHIFT) | ((u64)tl << SA_GRPH_TERM_SHIFT) |
                                       (excl + j == 0u ? nhf : 0ull);
            }
            // a query without postings: an entry 0 that says "0 halves"
            if (tl == 0u && total == 0u) s_half[qi][0] = (u64)(stream + p.imp_tail);
        }
    };
    // the shared term's dense factor row, requested BEFORE the tables are built (round 5: its latency used to start after them)
    float4 dv[TILE / 256];
    if (dslot != 0xFFFFFFFFu) {
        const float4* row4 = (const float4*)(gp.dense + (u64)dslot * gp.dense_stride + tile_base);
#pragma unroll
        for (int j = 0; j < TILE / 256; j++) dv[j] = row4[j * 64 + (int)lane];
    }
    SA_PTV(9);                                                  // (fine: group entry, first term's slice bounds, stream base, weight)
    auto build_tables = [&]() {
        const u32 NP = (nq + QPP - 1u) / QPP;                   // 1 or 2 lane passes (host: cq * TT <= 128)
        __builtin_amdgcn_wave_barrier();
        const Pre x0 = pre_load(0);
        Pre x1 = x0;
        if (NP > 1u) x1 = pre_load(1);
        SA_PTV(10);                                             // (fine: the further terms' slice bounds)
        pre_store(0, x0);
        if (NP > 1u) pre_store(1, x1);
        // One wave per workgroup: its LDS instructions execute in program order for all lanes, so lanes hand data
        // to each other through LDS without s_barrier; the wave barriers only pin the order of the accesses
        // for the compiler (they emit no instruction).
        __builtin_amdgcn_wave_barrier();
    };
    build_tables();
    if (n <= gp.cq) {
        // nothing to score in this tile at all?
        const u32 mine = lane < n ? (u32)(s_half[lane][0] >> SA_GRPH_NH_SHIFT) : 0u;
        if (h1 == h0 && ballot(mine != 0u) == 0ull) return;
    }
    SA_PT(0);                                                   // tables of the item built

    // The half descriptors of a query reach the scalar registers with ONE LDS instruction: lane h reads entry h
    // (8 bytes) and v_readlane hands the fields out -- no LDS round trip per half; a second one brings the query's
    // weights (lane t: term t).
    struct Q { u64 v[NH]; u32 dlo, dhi; float w; };
    typedef const __attribute__((address_space(1))) u64* gptr_u64;
    // request the postings of query qi's halves (dynamic number of loads: the caller has made sure that no
    // older load is outstanding, so "everything landed" is an exact wait for them later) and its bound
    auto prefetch = [&](u32 qi, Q& X) {
        const u64 d = s_half[qi][lane < (u32)NH ? lane : 0u];
        X.dlo = (u32)d; X.dhi = (u32)(d >> 32);
        X.w = s_idf[qi * TT + ((X.dhi >> (SA_GRPH_TERM_SHIFT - 32)) & 0x1Fu)];     // lane h: the weight of half h's term
        const u32 nh_raw = (u32)__builtin_amdgcn_readfirstlane((int)X.dhi) >> (SA_GRPH_NH_SHIFT - 32);
        // (too many: the pair goes to the per-query kernel; an odd count: the empty partner entry is loaded too --
        //  nothing masks a half's lanes, its postings must be the sentinel's)
        const u32 nh = nh_raw <= (u32)NH ? nh_raw : 0u;
        sa_static_while_below<0, NH, 1>(nh, [&](auto hc) {       // (a query has ~3 halves on average)
            constexpr int h = decltype(hc)::value;
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)X.dhi, h);
            const u64 a = (u64)(u32)__builtin_amdgcn_readlane((int)X.dlo, h) | ((u64)(hi & 0xFFFFu) << 32);
            const u32 lim = (hi >> (SA_GRPH_LIM_SHIFT - 32)) & 0x3Fu;
            X.v[h] = ((gptr_u64)a)[lane < lim ? lane : lim];     // lanes past the end: copies of the last posting
        });
    };
    // the queries' bounds, one per lane, read once per item (a bound only ever rises: a stale one is valid)
    u32 thr_all = lane < n ? __hip_atomic_load(&p.gthr[row0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (p.seed && lane < n) { const u32 sd = p.seed[row0 + lane]; thr_all = sd > thr_all ? sd : thr_all; }
    Q A, B;
    prefetch(0, A);                                             // in flight while the base is built
    SA_PTV(11);                                                 // (fine: bounds + query 0's postings)

    // ---- base: clear, then the first term's slice scored once (write-only: 0 + s0 = s0)
    u32 base_max;
    if (dslot != 0xFFFFFFFFu) {
        // the shared term has a dense factor row: lane = doc, four docs per 16-byte load and LDS store, no unpacking and no
        // scatter (docs without the term hold 0.0 -> 0.0 * idf = +0.0, what the cleared accumulator holds)
        float4* a4 = (float4*)accu;
        u32 lmax = 0;
#pragma unroll
        for (int j = 0; j < TILE / 256; j++) {
            float4 w;
            w.x = __fmul_rn(dv[j].x, hidf); w.y = __fmul_rn(dv[j].y, hidf); w.z = __fmul_rn(dv[j].z, hidf); w.w = __fmul_rn(dv[j].w, hidf);
            a4[j * 64 + (int)lane] = w;
            const u32 m0 = __float_as_uint(w.x) > __float_as_uint(w.y) ? __float_as_uint(w.x) : __float_as_uint(w.y);
            const u32 m1 = __float_as_uint(w.z) > __float_as_uint(w.w) ? __float_as_uint(w.z) : __float_as_uint(w.w);
            const u32 m = m0 > m1 ? m0 : m1;
            lmax = m > lmax ? m : lmax;
        }
        __builtin_amdgcn_wave_barrier();
        at(spare) = 0u;
        __builtin_amdgcn_wave_barrier();
        base_max = sa_wave_max_u32(lmax);
    } else {
        float4* a4 = (float4*)accu;
#pragma unroll
        for (int j = 0; j < TILE / 256; j++) a4[j * 64 + (int)lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_wave_barrier();
        const u64 lo = hbs.x + h0, hi = hbs.x + h1, a0 = lo & ~1ull;
        const u32 npairs = hi > lo ? (u32)((hi - a0 + 1) >> 1) : 0u;
        const u32 jc = (u32)((hbs.y - a0) >> 1);
        const sa_u64x2* pairs = (const sa_u64x2*)(stream + a0);
        u32 lmax = 0;
        constexpr int PF = 4;
        sa_u64x2 cur[PF], nxt[PF];
        if (npairs) {
#pragma unroll
            for (int u = 0; u < PF; u++) { const u32 j = (u32)u * 64u + lane; cur[u] = pairs[j < jc ? j : jc]; }
        }
        for (u32 first = 0; first < npairs; first += (u32)PF * 64u) {
            const bool more = first + (u32)PF * 64u < npairs;
            if (more) {
#pragma unroll
                for (int u = 0; u < PF; u++) { const u32 j = first + (u32)(PF + u) * 64u + lane; nxt[u] = pairs[j < jc ? j : jc]; }
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const u32 d0 = (u32)(cur[u].x >> 32) - tile_base_b, d1 = (u32)(cur[u].y >> 32) - tile_base_b;
                const bool in0 = d0 < (u32)TILE * 4u, in1 = d1 < (u32)TILE * 4u;
                const u32 w0 = __float_as_uint(__fmul_rn(__uint_as_float((u32)cur[u].x), hidf));
                const u32 w1 = __float_as_uint(__fmul_rn(__uint_as_float((u32)cur[u].y), hidf));
                at(in0 ? d0 : spare) = w0;
                at(in1 ? d1 : spare) = w1;
                const u32 m0 = in0 ? w0 : 0u, m1 = in1 ? w1 : 0u;
                lmax = m0 > lmax ? m0 : lmax;
                lmax = m1 > lmax ? m1 : lmax;
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < PF; u++) cur[u] = nxt[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        at(spare) = 0u;                                         // (took the base's out-of-tile postings; the overlay's spare values start at 0)
        __builtin_amdgcn_wave_barrier();
        base_max = sa_wave_max_u32(lmax);
    }

    SA_PT(1);                                                   // first query requested, base built
    // queries left to the per-query kernel (work list at the end)
    u64 deferred = 0ull;
    // survivors out: one reservation per query with survivors (lane qi reserves for query qi: the atomics travel
    // together), every buffered entry to its place (the buffer is in query order: an entry's offset inside its query
    // is its index minus the exclusive prefix of the per-query counts), bounds re-derived where a list crossed a
    // multiple of SA_GRP_REFRESH_STEP entries
    u32 nsurv = 0, my_surv = 0;                                  // buffered (wave-uniform); lane qi: those of query qi
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        u32 cbase = 0;
        if (my_surv) cbase = atomicAdd(&p.cand_cnt[row0 + lane], my_surv);
        u32 start = my_surv;
#pragma unroll
        for (int o = 1; o < SA_WAVE; o <<= 1) {
            const u32 up = (u32)__shfl_up((int)start, (unsigned)o, SA_WAVE);
            if (lane >= (u32)o) start += up;
        }
        start -= my_surv;
        for (u32 e0 = 0; e0 < nsurv; e0 += (u32)SA_WAVE) {
            const u32 e = e0 + lane;
            const bool have = e < nsurv;
            const u64 ent = s_surv[have ? e : 0u];
            const u32 eq = (u32)ent & 0x3Fu, sl = (u32)ent >> 6, fin = (u32)(ent >> 32);
            const u32 qb = (u32)__shfl((int)cbase, (int)eq, SA_WAVE), qs = (u32)__shfl((int)start, (int)eq, SA_WAVE);
            if (have) {
                const u32 q = row0 + eq;
                atomicAdd(&p.hist[(u64)q * SA_HBINS + sa_score_bin(fin)], 1u);
                const u32 pos = qb + (e - qs);
                const u64 doc = p.doc_base + tile_base + (u64)(sl >> 2);
                if (pos < p.cand_cap) p.cand[(u64)q * p.cand_cap + pos] = ((u64)fin << 32) | (u64)(u32)(~(u32)doc);
            }
        }
        const bool crossed = my_surv != 0u && cbase / (u32)SA_GRP_REFRESH_STEP != (cbase + my_surv) / (u32)SA_GRP_REFRESH_STEP;
        for (u64 m = ballot(crossed); m; m &= m - 1ull) {
            const u32 q = row0 + (u32)__builtin_ctzll(m);
            sa_hist_refresh(p.hist + (u64)q * SA_HBINS, &p.gthr[q], p.k, lane);
        }
        __builtin_amdgcn_wave_barrier();
        nsurv = 0; my_surv = 0;
    };
    // One query of the item.  Per half, IN QUERY-TERM ORDER: read the docs' accumulators, add, write back with the sign
    // bit set ("touched by this query"; written as -(|o| + s), one instruction), remember what was read.  Nothing is
    // decided per lane: a lane outside the half's postings holds a copy of its last posting (an empty partner half:
    // the zero sentinel, which min(d, spare) sends to the lane's spare slot).  Then the base values go back -- every lane rewrites what it READ, the halves
    // in REVERSE order, so that for a doc touched by several terms the value read first (the base) is written last:
    // no lane needs to know whether it was the first to touch its doc.  The final scores are only looked at when the
    // largest value any lane wrote reaches the query's bound (one compare per query instead of a read, a mask and
    // two compares per half): then every doc's final value is read before the restore and the first touchers
    // (sign bit of what they read) report the docs that reach the bound.
    auto process = [&](u32 qi, const Q& X) {
        const u32 nh_raw = (u32)__builtin_amdgcn_readfirstlane((int)X.dhi) >> (SA_GRPH_NH_SHIFT - 32);
        const u32 gq = cq0 + qi;                                // the query's place in the item (its lane of thr_all / my_surv, its bit of deferred)
        const u32 thr_q = (u32)__builtin_amdgcn_readlane((int)thr_all, (int)gq);
        const u32 thr = thr_q > 1u ? thr_q : 1u;
        if (nh_raw > (u32)NH || base_max >= thr) { deferred |= 1ull << gq; return; }
        if (nh_raw == 0u) return;                               // the query scores exactly the base here: all below its bound
        const u32 nh = nh_raw;                                  // (round 4: an odd count no longer gets an empty partner half)
        u32 rs[NH], ro[NH];
        u32 wmax = 0u;                                          // largest value written (sign bit set)
        sa_static_while_below<0, NH, 1>(nh, [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            {
                __builtin_amdgcn_wave_barrier();                // a half sees the previous half's (other lanes') writes
                const float w = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(X.w), h));
                const u64 v = X.v[h];
                const u32 d = (u32)(v >> 32) - tile_base_b;
                const u32 sl = d < spare ? d : spare;           // (a doc outside the tile: d >= TILE * 4)
                const u32 o = at(sl);
                __builtin_amdgcn_wave_barrier();                // (all lanes have read -- copies of a posting read the same value -- before any writes)
                const u32 nw = __float_as_uint(__fsub_rn(-fabsf(__uint_as_float(o)), __fmul_rn(__uint_as_float((u32)v), w)));
                at(sl) = nw;
                wmax = nw > wmax ? nw : wmax;
                rs[h] = sl;
                ro[h] = o;
            }
        });
        __builtin_amdgcn_wave_barrier();
        const bool look = ballot(wmax >= (thr | 0x80000000u)) != 0ull;
        u32 fin[NH];
        if (look) {                                             // (rare once the bound stands)
#pragma unroll
            for (int h = 0; h < NH; h++) fin[h] = (u32)h < nh ? at(rs[h]) & 0x7FFFFFFFu : 0u;
            __builtin_amdgcn_wave_barrier();
        }
        // base values back, last half first
        auto back = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            at(rs[h]) = ro[h];
            __builtin_amdgcn_wave_barrier();
        };
        // (nh = 1 .. NH; the usual 2 or 3 halves cost three or four scalar tests -- as a switch with fall-through the compiler
        //  built a chain of ~25 scalar flag operations in front of the stores of the usual case)
        static_assert(NH == 10, "restore chain");
        if (nh > 4u) {
            if (nh > 8u) { if (nh > 9u) back(std::integral_constant<int, 9>{}); back(std::integral_constant<int, 8>{}); }
            if (nh > 6u) { if (nh > 7u) back(std::integral_constant<int, 7>{}); back(std::integral_constant<int, 6>{}); }
            if (nh > 5u) back(std::integral_constant<int, 5>{});
            back(std::integral_constant<int, 4>{});
        }
        if (nh > 2u) { if (nh > 3u) back(std::integral_constant<int, 3>{}); back(std::integral_constant<int, 2>{}); }
        if (nh > 1u) back(std::integral_constant<int, 1>{});
        back(std::integral_constant<int, 0>{});
        if (!look) return;
        u64 kb[NH];
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            kb[i] = 0ull;
            if ((u32)i < nh) {
                const u32 lim = ((u32)__builtin_amdgcn_readlane((int)X.dhi, i) >> (SA_GRPH_LIM_SHIFT - 32)) & 0x3Fu;
                kb[i] = ballot(lane <= lim && rs[i] < (u32)TILE * 4u && !(ro[i] >> 31) && fin[i] >= thr);
            }
            c += (u32)__popcll(kb[i]);
        }
        if (c == 0u) return;
        if (c > 16u) { deferred |= 1ull << gq; return; }        // bound still far off: the per-query item's histogram path refines it first
        // Survivors are buffered in LDS and written out together (flush below): reserving places in a query's
        // candidate list is an atomic WITH a return value -- a round trip to L2 the wave sits out; per surviving pair
        // that was most of the 0.45 ms the candidates cost at k = 1000.  Flushed, the reservations of all queries of
        // the item travel together.
        if (nsurv + c > (u32)SA_GRP_SURV_CAP) flush();
        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (kb[i]) {                                        // (uniform: halves without survivors cost a scalar test)
                if ((kb[i] >> lane) & 1ull) s_surv[nsurv + (u32)__popcll(kb[i] & lt)] = ((u64)fin[i] << 32) | (u64)(rs[i] << 6) | (u64)gq;
                nsurv += (u32)__popcll(kb[i]);
            }
        }
        if (lane == gq) my_surv = c;
    };

    // ---- the queries of the group, two per round so that the prefetch buffers swap without copies.  Before the
    //      next query's loads are issued everything older has landed (an exact wait: only the current query's
    //      loads are outstanding), so the current query is processed while exactly the next one's loads fly.
    for (;;) {
        for (u32 qi = 0; qi < nq; qi += 2) {
            SA_WAIT_VMCNT0();
            SA_PT(2);
            if (qi + 1u < nq) prefetch(qi + 1u, B);
            SA_PT(3);
            process(qi, A);
            SA_PT(4);
            if (qi + 1u < nq) {
                SA_WAIT_VMCNT0();
                SA_PT(2);
                if (qi + 2u < nq) prefetch(qi + 2u, A);
                SA_PT(3);
                process(qi + 1u, B);
                SA_PT(4);
            }
        }
        // the item's next pass: tables of its next queries over the same base
        cq0 += nq;
        if (cq0 >= n) break;
        nq = n - cq0 < gp.cq ? n - cq0 : gp.cq;
        build_tables();
        prefetch(0, A);
        SA_PT(0);
    }
    if (nsurv) flush();
    // ---- general path: hand the (tile, query) pairs to the per-query kernel that follows (sa_k_bm25_tiles_wl)
    if (deferred) {
        const u32 c = (u32)__popcll(deferred);
        u32 wbase = 0;
        if (lane == 0) wbase = atomicAdd(gp.wl_cnt, c);
        wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
        if ((deferred >> lane) & 1ull)
            gp.wl[wbase + (u32)__popcll(deferred & ((1ull << lane) - 1ull))] = ((u64)tile << 32) | (u64)(row0 + lane);
    }
#ifdef SA_PROBE
    SA_PT(5);
    if (lane == 0) {
        unsigned long long* const ps = g_sa_probe + (size_t)(blockIdx.x % SA_PROBE_SLOTS) * 16;
        for (int i = 0; i < 6; i++) atomicAdd(&ps[i], pacc[i]);
        for (int i = 9; i < 12; i++) atomicAdd(&ps[i], pacc[i]);
        atomicAdd(&ps[6], 1ull);
        atomicAdd(&ps[7], (unsigned long long)n);
        atomicAdd(&ps[8], (unsigned long long)__popcll(deferred));
    }
#endif
}

// The (tile, query) items the grouped kernel left to the per-query path (usually none once the bounds stand):
// their number is only known on the device, so a resident grid walks the work list.
template <int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS) sa_k_bm25_tiles_wl(const Bm25Params p, const u64* __restrict__ wl,
                                                               const u32* __restrict__ wl_cnt) {
    __shared__ alignas(16) u64 smem[sa_tile_smem_u64<TILE, 1>()];
    __shared__ float s_tab[sa_tile_tab_floats<THREADS, true>()];
    const u32 nitems = *wl_cnt;
    for (u32 i = blockIdx.x; i < nitems; i += gridDim.x) {
        const u64 it = wl[i];
        sa_bm25_tile_item<TILE, THREADS, 1, true>(p, (u32)(it >> 32), (u32)it, smem, s_tab);
        __syncthreads();                              // LDS is reused by the next item
    }
}

// Merge n_cand candidate keys per query into the k best, sorted descending.
// One workgroup of 1024 threads per query.
//
// Small rows (<= SA_MERGE_LIST keys: a cross-rank merge, or a well-pruned append list) go straight
// into LDS and a bitonic sort finishes.  Larger rows are first cut by a lower bound of the k-th
// largest key that is already known:
//   gthr / slots       the bound the tile kernel left behind (>= k docs score at or above it)
//   rank_stride > 0    rows made of sorted groups of rank_stride keys: the k-th largest GROUP
//                      LEADER bounds the k-th largest key (k distinct groups own a key >= it)
// One pass keeps the keys above the bound in LDS.  If they overflow the LDS list (large k, ties)
// the survivors are compacted in place at the front of the row and the exact k-th largest is
// found by MSB-first bisection over the survivors only.
#define SA_MERGE_LIST 2048

// clears `words` u32 at `slots` and `n8` 8-byte cells at `bloom`
__global__ void __launch_bounds__(256)
sa_k_run_reset(u32* __restrict__ slots, u64 words, u64* __restrict__ bloom, u64 n8, u32* __restrict__ one_more) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && one_more) *one_more = 0u;
    for (u64 i = t; i < words; i += stride) slots[i] = 0u;
    for (u64 i = t; i < n8; i += stride) bloom[i] = 0ull;
}

// (THREADS: 1024, or 256 for small k -- a workgroup of 16 waves needs 16 free wave slots and 17 KiB of LDS on ONE CU at once, which a
//  device full of one-wave scoring workgroups hands out slowly; for k <= 64 a query has a few dozen candidates and four waves do)
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
sa_k_topk_merge(u64* __restrict__ cand, u32 n_cand_max, u32 k, u64* __restrict__ out,
                const u32* __restrict__ out_row, u32 rank_stride, const u32* __restrict__ cnt,
                const u32* __restrict__ slots, const u32* __restrict__ gthr, u32* __restrict__ overflow,
                u32* __restrict__ clr_state, u32 clr_B, u32 clr_hist, u32* __restrict__ clr_one,
                u32 gather_stride, u32* __restrict__ xflag, const u32* __restrict__ seed) {
    constexpr int NW = THREADS / SA_WAVE;
    __shared__ u64 red64[NW + 1];
    __shared__ u64 sel[SA_MERGE_LIST];
    __shared__ u32 s_n;
    const u32 q = blockIdx.x, tid = threadIdx.x;
    if (gather_stride) {
        // The row is read straight out of an all-gather result [rank][B*k + 1] (no regroup launch): key i of query q is
        // key i % k of rank i / k; the extra cell of every rank's block is its overflow flag, OR-ed into *xflag.
        // Only rows that fit the LDS list take this route (the host checks), so nothing is compacted in place.
        if (tid == 0) s_n = 0;
        for (u32 i = tid; i < SA_MERGE_LIST; i += THREADS) sel[i] = 0;
        __syncthreads();
        const u32 B_ = gridDim.x;
        for (u32 i = tid; i < n_cand_max; i += THREADS) {
            const u32 r = i / rank_stride, j = i % rank_stride;
            const u64 x = cand[(u64)r * gather_stride + (u64)q * rank_stride + j];
            if (x) { const u32 pos = atomicAdd(&s_n, 1u); sel[pos] = x; }
        }
        if (q == 0 && tid == 0 && xflag) {
            u32 any = 0;
            for (u32 r = 0; r < n_cand_max / rank_stride; r++) any |= cand[(u64)r * gather_stride + (u64)B_ * rank_stride] != 0ull ? 1u : 0u;
            if (any) *xflag = 1u;
        }
        __syncthreads();
        const u32 n_sel = s_n;
        u32 np2 = 2;
        while (np2 < n_sel) np2 <<= 1;
        sa_block_bitonic_desc(sel, np2);
        const u32 row_ = out_row ? out_row[q] : q;
        for (u32 i = tid; i < k; i += THREADS) out[(u64)row_ * k + i] = (i < SA_MERGE_LIST) ? sel[i] : 0ull;
        return;
    }
    u64* c = cand + (u64)q * n_cand_max;
    // cnt != null: the row is an append list holding cnt[q] keys (pruned tile selection)
    u32 n_cand = n_cand_max;
    if (cnt) {
        const u32 have = cnt[q];
        n_cand = have < n_cand_max ? have : n_cand_max;
        if (have > n_cand_max && overflow && tid == 0) atomicMax(overflow, 1u);   // the list ran over: see sa_batch_fetch
    }
    const u32 row = out_row ? out_row[q] : q;       // device row q holds caller query out_row[q]
    for (u32 i = tid; i < SA_MERGE_LIST; i += THREADS) sel[i] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    // MSB-first bisection (2 bits per step) for the k-th largest of c[0 : n : stride]
    auto kth_largest = [&](u32 n, u32 stride) -> u64 {
        u64 m = 0;
        for (u32 i = tid; i < n; i += THREADS) { const u64 x = c[(u64)i * stride]; m = x > m ? x : m; }
        m = sa_block_max64<NW>(m, red64);
        if (m == 0 || n < k) return 0;
        int top = 63 - __clzll((long long)m);
        if ((top & 1) == 0) top++;
        u64 prefix = 0;
        for (int bit = top; bit >= 1; bit -= 2) {
            const u64 c1 = prefix | (1ull << (bit - 1)), c2 = prefix | (2ull << (bit - 1)), c3 = prefix | (3ull << (bit - 1));
            u64 packed = 0;
            for (u32 i = tid; i < n; i += THREADS) {
                const u64 x = c[(u64)i * stride];
                packed += (x >= c1 ? 1ull : 0ull) + (x >= c2 ? (1ull << 21) : 0ull) + (x >= c3 ? (1ull << 42) : 0ull);
            }
#pragma unroll
            for (int o = SA_WAVE / 2; o > 0; o >>= 1) packed += __shfl_xor(packed, o, SA_WAVE);
            if (sa_lane() == 0) red64[sa_wave_id()] = packed;
            __syncthreads();
            u64 tot = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) tot += red64[w];
            __syncthreads();
            const u32 n1 = (u32)(tot & 0x1FFFFF), n2 = (u32)((tot >> 21) & 0x1FFFFF), n3 = (u32)((tot >> 42) & 0x1FFFFF);
            if (n3 >= k) prefix = c3; else if (n2 >= k) prefix = c2; else if (n1 >= k) prefix = c1;
        }
        return prefix;
    };

    u64 thr = 1;
    if (gthr) {
        // histogram bound left behind by the scoring kernels: at least k docs score >= it, so only
        // keys at or above it matter -- usually a few dozen, sorted by one wave below
        thr = (u64)gthr[q] << 32;
        if (thr < 1) thr = 1;
    } else if (n_cand > SA_MERGE_LIST) {
        if (slots) {
            u32 g = slots[(u64)q * 32 + (tid & 31)];
            g = sa_wave_min_u32(g);                   // every wave computes the same minimum
            thr = (u64)g << 32;
        } else if (rank_stride > 0 && n_cand / rank_stride >= k) {
            thr = kth_largest(n_cand / rank_stride, rank_stride);
        }
        if (thr < 1) thr = 1;
    }
    for (u32 i = tid; i < n_cand; i += THREADS) {
        const u64 x = c[i];
        if (x >= thr) {
            const u32 pos = atomicAdd(&s_n, 1u);
            if (pos < SA_MERGE_LIST) sel[pos] = x;
        }
    }
    __syncthreads();
    u32 n_sel = s_n;
    __syncthreads();
    if (n_sel > SA_MERGE_LIST) {                      // uniform
        // compact the survivors to the front of the row (a chunk's writes land below its own
        // end: at most as many survivors as keys read so far), then bisect over them only
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (u32 base = 0; base < n_cand; base += 4 * THREADS) {
            u64 x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = base + j * THREADS + tid;
                x[j] = i < n_cand ? c[i] : 0ull;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x[j] >= thr) c[atomicAdd(&s_n, 1u)] = x[j];
        }
        __syncthreads();
        const u32 n_surv = s_n;
        __syncthreads();
        const u64 g = kth_largest(n_surv, 1);         // exact: keys are distinct, so exactly k keys are >= g
        thr = g > 1 ? g : 1;
        for (u32 i = tid; i < SA_MERGE_LIST; i += THREADS) sel[i] = 0;
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (u32 i = tid; i < n_surv; i += THREADS) {
            const u64 x = c[i];
            if (x >= thr) {
                const u32 pos = atomicAdd(&s_n, 1u);
                if (pos < SA_MERGE_LIST) sel[pos] = x;
            }
        }
        __syncthreads();
        n_sel = s_n;
    }
    // Fewer than k keys at or above a bound: with bounds derived from k counted docs, or from a term's k-th largest factor,
    // that cannot happen -- a safety net under the starting bounds (a rank table out of step with the impact stream, a
    // scoring path whose rounding broke the monotonicity argument): the run is flagged and redone without bounds.
    if (gthr && overflow && n_sel < k && tid == 0 && (gthr[q] != 0u || (seed && seed[q] != 0u))) atomicMax(overflow, 2u);
    n_sel = n_sel < SA_MERGE_LIST ? n_sel : SA_MERGE_LIST;
    if (n_sel <= SA_WAVE) {                           // uniform
        // short list: one wave sorts it in registers (bitonic over lanes, no barriers)
        if (tid < SA_WAVE) {
            u64 x = sel[tid];                         // slots past n_sel are 0
#pragma unroll
            for (u32 size = 2; size <= SA_WAVE; size <<= 1) {
#pragma unroll
                for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
                    const u64 y = __shfl_xor(x, (int)stride, SA_WAVE);
                    const bool lower = (tid & stride) == 0;           // this lane holds the pair's lower index
                    const bool desc = (tid & size) == 0;
                    const bool take_max = lower == desc;
                    x = take_max ? (x > y ? x : y) : (x < y ? x : y);
                }
            }
            sel[tid] = x;
        }
        __syncthreads();
    } else {
        u32 np2 = 2;
        while (np2 < n_sel) np2 <<= 1;
        sa_block_bitonic_desc(sel, np2);
    }
    for (u32 i = tid; i < k; i += THREADS) out[(u64)row * k + i] = (i < SA_MERGE_LIST) ? sel[i] : 0ull;
    // The merge is the last reader of the run's per-query state, so it leaves it zeroed for the next run of the
    // batch -- bound slots [B][32], cursors [B], cached bounds [B], histograms [B][SA_HBINS] (layout: sa_batch_alloc_topk),
    // the grouped kernel's work-list cursor -- instead of a reset launch in front of every run.
    if (clr_state) {
        if (tid < 32u) clr_state[(u64)q * 32u + tid] = 0u;
        if (tid == 32u) clr_state[(u64)clr_B * 32u + q] = 0u;
        if (tid == 33u) clr_state[(u64)clr_B * 33u + q] = 0u;
        if (clr_hist) for (u32 i = tid; i < (u32)SA_HBINS; i += THREADS) clr_state[(u64)clr_B * 34u + (u64)q * SA_HBINS + i] = 0u;
        if (clr_one && q == 0u && tid == 34u) *clr_one = 0u;
    }
}

#define SA_MERGE_LAUNCH(K_, B_, ST_, ...)                                                                                  \
    do {                                                                                                               \
        if ((K_) <= 64u && sa_opt(bt->opts.merge_small, 1) != 0)                                                          \
            hipLaunchKernelGGL(sa_k_topk_merge<256>, dim3(B_), dim3(256), 0, ST_, __VA_ARGS__);                          \
        else                                                                                                           \
            hipLaunchKernelGGL(sa_k_topk_merge<1024>, dim3(B_), dim3(1024), 0, ST_, __VA_ARGS__);                        \
    } while (0)

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------



static void sa_fill_params(const sa_index* ix, Bm25Params& p) {
    p.tfp = ix->d_tfp; p.tf_off = ix->d_tf_off; p.dir_slot = ix->d_dir_slot; p.tile_dir = ix->d_tile_dir;
    p.doc_lens = ix->d_doc_lens; p.n_terms = ix->n_terms; p.n_tiles = ix->n_tiles;
    p.n_docs = ix->n_docs; p.doc_base = ix->doc_base; p.dl_packed = ix->dl_packed ? 1 : 0;
    p.avgdl = ix->avg_doc_len;
    p.qlist = nullptr; p.nq = 0;                     // callers set nq (= B) after filling B
    p.tile0 = 0; p.tile_end = ix->n_tiles;
}

static u32 sa_tile_waves(u32 tile_docs) {
    switch (tile_docs) {
        case 1024: return 2; case 2048: return 1; case 4096: return 2; case 8192: return 4;
        case 16384: return 8; case 32768: return 16; default: return 1;
    }
}

static int sa_launch_make_sattab(sa_index* ix, float* d_tab, u32* tab_w_out, float k1, float b, hipStream_t st) {
    u32 w = 0;
    if (ix->dl_packed) {
        w = 64;
        while (w < SA_SAT_WMAX && w <= ix->max_doc_len) w <<= 1;
    }
    *tab_w_out = w;
    if (w) hipLaunchKernelGGL(sa_k_make_sattab, dim3(sa_div_up((u64)SA_SAT_NTF * w, 256)), dim3(256), 0, st, d_tab, w, k1, b,
                              ix->avg_doc_len);
    return SA_OK;
}

static int sa_launch_make_bounds(sa_index* ix, const u32* d_terms, u32 BT, u32* d_bounds, u64* d_qbase, hipStream_t st,
                                 u64* d_qbase_imp = nullptr, const float* d_topf = nullptr, const float* d_idf = nullptr, u32 T = 1,
                                 u32 k = 1, u32* d_seed = nullptr, float seed_scale = 1.f) {
    const u64 total = (u64)BT * (ix->n_tiles + 1);
    if (total == 0) return SA_OK;
    const u32 grid = total / 256 + 1 < 8192 ? (u32)(total / 256 + 1) : 8192;
    u32 rank_idx = SA_TOPF_NR - 1;                               // the smallest tabulated rank >= k
    for (int i = SA_TOPF_NR - 1; i >= 0; i--) if (sa_topf_ranks[i] >= k) rank_idx = (u32)i;
    hipLaunchKernelGGL(sa_k_make_bounds, dim3(grid), dim3(256), 0, st, ix->d_tfp, ix->d_tf_off, ix->d_dir_slot,
                       ix->d_tile_dir, ix->n_terms, ix->n_tiles, ix->tile_docs, d_terms, BT, d_bounds, d_qbase, d_qbase_imp,
                       d_topf, d_idf, T, rank_idx, (d_topf && d_idf) ? d_seed : (u32*)nullptr, seed_scale);
    return SA_OK;
}

// the slice table of the batch's current query set (the staged-tile route leaves it out of the step: built when another route runs)
static int sa_batch_ensure_bounds(sa_batch* bt, hipStream_t st) {
    if (bt->bounds_valid) return SA_OK;
    SA_TRY(sa_launch_make_bounds(bt->ix, bt->d_terms, bt->B * bt->T, bt->d_bounds, bt->d_qbase, st, bt->d_qbase_imp,
                                 bt->seed_on ? bt->impacts->d_topf : nullptr, bt->d_idf, bt->T, bt->k, bt->d_seed,
                                 (float)sa_opt(bt->opts.seed_scale_pct, 100) / 100.f));
    bt->bounds_valid = true;
    return SA_OK;
}

#define SA_LAUNCH_TILE(TILE, THREADS)                                                              \
    {                                                                                              \
        if (MODE == 1 && p.imp)                                                                    \
            hipLaunchKernelGGL((sa_k_bm25_tiles<TILE, THREADS, MODE, MODE == 1>), grid, dim3(THREADS), 0, st, p); \
        else                                                                                       \
            hipLaunchKernelGGL((sa_k_bm25_tiles<TILE, THREADS, MODE, false>), grid, dim3(THREADS), 0, st, p); \
    }                                                                                              \
    break

template <int MODE>
static int sa_launch_bm25_mode(sa_index* ix, const Bm25Params& p, hipStream_t st) {
    const u32 nt = p.tile_end - p.tile0;
    const u32 gy = nt < SA_GRID_Y ? nt : SA_GRID_Y;
    const dim3 grid(p.nq, gy, (nt + SA_GRID_Y - 1) / SA_GRID_Y);
    switch (ix->tile_docs) {
        case 1024: SA_LAUNCH_TILE(1024, 128);
        case 2048: SA_LAUNCH_TILE(2048, 64);
        case 4096: SA_LAUNCH_TILE(4096, 128);
        case 8192: SA_LAUNCH_TILE(8192, 256);
        case 16384: SA_LAUNCH_TILE(16384, 512);
        case 32768: SA_LAUNCH_TILE(32768, 1024);
        default:
            sa_set_error("unsupported tile_docs %u", ix->tile_docs);
            return SA_ERR_STATE;
    }
    return SA_OK;
}

#define SA_LAUNCH_LIST(TILE, THREADS)                                                              \
    {                                                                                              \
        if (p.imp)                                                                                 \
            hipLaunchKernelGGL((sa_k_bm25_tiles_list<TILE, THREADS, true>), dim3(grid), dim3(THREADS), 0, st, p); \
        else                                                                                       \
            hipLaunchKernelGGL((sa_k_bm25_tiles_list<TILE, THREADS, false>), dim3(grid), dim3(THREADS), 0, st, p); \
    }                                                                                              \
    break

static int sa_launch_bm25_list(sa_index* ix, const Bm25Params& p, hipStream_t st) {
    if (ix->n_tiles == 0 || p.B == 0) return SA_OK;
    const u64 worst = (u64)p.B * ix->n_tiles;
    const u32 grid = worst < 4096 ? (u32)worst : 4096u;
    switch (ix->tile_docs) {
        case 1024: SA_LAUNCH_LIST(1024, 128);
        case 2048: SA_LAUNCH_LIST(2048, 64);
        case 4096: SA_LAUNCH_LIST(4096, 128);
        case 8192: SA_LAUNCH_LIST(8192, 256);
        default:
            sa_set_error("unsupported tile_docs %u for the query-list scan", ix->tile_docs);
            return SA_ERR_STATE;
    }
    return SA_OK;
}

static int sa_launch_bm25_groups(sa_index* ix, const sa_batch* bt, const Bm25Params& p, u32 tile0, hipStream_t st) {
    GroupParams gp;
    gp.grp = bt->d_grp; gp.n_groups = bt->n_groups;
    gp.tile0 = tile0; gp.n_tiles_run = ix->n_tiles - tile0;
    gp.tt = bt->grp_tt; gp.tt_shift = bt->grp_tt_shift; gp.cq = bt->grp_cq;
    gp.dense = (bt->impacts && sa_opt(bt->opts.group_dense, 1) != 0) ? bt->impacts->d_dense : nullptr;
    gp.dense_stride = bt->impacts ? bt->impacts->dense_stride : 0;
    const u64 blocks = (u64)((gp.n_tiles_run + 7u) / 8u) * 8u * gp.n_groups;
    gp.tpx = sa_opt(bt->opts.xcd_range, 1) != 0 ? (gp.n_tiles_run + 7u) / 8u : 0u;
    if (blocks > 0x7FFFFFFFull) { sa_set_error("grouped launch: bad grid"); return SA_ERR_STATE; }
    gp.wl = bt->d_wl; gp.wl_cnt = bt->d_wl_cnt;
    const u64 worst = (u64)gp.n_tiles_run * bt->n_grouped_rows;
    const u32 wgrid = worst < 2048 ? (u32)worst : 2048u;
    if (wgrid == 0) return SA_OK;
    // (weight table: n * tt cells; 64 cover up to 4 overlaid terms per query at 16 queries per item)
    const bool small = (u32)SA_GRP_MAXQ * gp.tt <= 64u;
#ifdef SA_PROBE
    // (measurement build: unused dynamic LDS per workgroup lowers the resident waves per CU -- the occupancy experiment of DESIGN 3.1a)
    const u32 lds_pad = getenv("SA_PROBE_LDS_PAD") ? (u32)atoi(getenv("SA_PROBE_LDS_PAD")) : 0u;
#else
    const u32 lds_pad = 0u;
#endif
#define SA_LAUNCH_GROUP(TILE, THREADS)                                                                                     \
    {                                                                                                                      \
        if (blocks && small) hipLaunchKernelGGL((sa_k_bm25_group_tiles<TILE, 64>), dim3((u32)blocks), dim3(64), lds_pad, st, p, gp);       \
        else if (blocks) hipLaunchKernelGGL((sa_k_bm25_group_tiles<TILE, 128>), dim3((u32)blocks), dim3(64), lds_pad, st, p, gp);            \
        hipLaunchKernelGGL((sa_k_bm25_tiles_wl<TILE, THREADS>), dim3(wgrid), dim3(THREADS), 0, st, p, (const u64*)gp.wl,   \
                           (const u32*)gp.wl_cnt);                                                                         \
    }                                                                                                                      \
    break
    switch (ix->tile_docs) {
        case 1024: SA_LAUNCH_GROUP(1024, 128);
        case 2048: SA_LAUNCH_GROUP(2048, 64);
        case 4096: SA_LAUNCH_GROUP(4096, 128);
        default:
            sa_set_error("unsupported tile_docs %u for the grouped kernel", ix->tile_docs);
            return SA_ERR_STATE;
    }
#undef SA_LAUNCH_GROUP
    return SA_OK;
}

static int sa_launch_bm25(sa_index* ix, const Bm25Params& p, hipStream_t st) {
    if (ix->n_tiles == 0 || p.nq == 0 || p.tile_end <= p.tile0) return SA_OK;
    return p.pruned ? sa_launch_bm25_mode<1>(ix, p, st) : sa_launch_bm25_mode<0>(ix, p, st);
}


// ---- BASELINE config 2 as one launch (reference postings.py:652-680 score(), bm25.pyx:11-25; summed over the query terms as
// test/test_msmarco.py:353-354 does): the dense float32[n_docs] BM25 vector of T terms, written ONCE -- straight into the caller's
// device vector (times its boost) or into the scratch vector the host copy starts from.  A workgroup owns a tile of docs: zeroes
// its accumulators in LDS, adds factor x idf for every posting of every term in the tile (term after term, so a doc's sum is formed
// in query-term order exactly like the tile kernel's), stores the tile.  Bytes per call: 8 x (postings of the terms) + 4 x n_docs.
// The terms and weights travel in the kernel arguments: no upload, no slice-table launch (a slice's ends come from the index's own
// tile directory, or a search of the term's postings), no fill, no scale / copy pass.
struct DenseQuery { u32 term[SA_MAX_QTERMS]; float w[SA_MAX_QTERMS]; };

template <int TILE>
__global__ void __launch_bounds__(256)
sa_k_bm25_dense_direct(const u64* __restrict__ tfp, const u64* __restrict__ tf_off, const u32* __restrict__ dir_slot,
                       const u32* __restrict__ tile_dir, const u64* __restrict__ imp, u32 n_terms, u32 n_tiles, u64 n_docs,
                       const DenseQuery q, u32 T, float boost, int has_boost, float* __restrict__ out) {
    __shared__ alignas(16) float acc[TILE];
    __shared__ u64 s_first[SA_MAX_QTERMS];
    __shared__ u32 s_cnt[SA_MAX_QTERMS];
    const u32 tile = blockIdx.x, tid = threadIdx.x;
    const u64 d0 = (u64)tile * TILE;
    for (u32 i = tid; i < (u32)TILE / 4u; i += 256u) ((float4*)acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < T) {
        const u32 term = q.term[tid];
        u64 first = 0; u32 cnt = 0;
        if (term < n_terms) {
            const u64 base = tf_off[term];
            const u32 df = (u32)(tf_off[term + 1] - base);
            const u32 slot = dir_slot[term];
            u32 lo, hi;
            if (slot != 0xFFFFFFFFu) {
                lo = tile_dir[(u64)slot * (n_tiles + 1) + tile];
                hi = tile_dir[(u64)slot * (n_tiles + 1) + tile + 1u];
            } else {
                lo = sa_lower_bound(tfp + base, 0, df, d0 << SA_KEY_SHIFT, SA_KEY_MASK);
                hi = sa_lower_bound(tfp + base, lo, df, (d0 + (u64)TILE) << SA_KEY_SHIFT, SA_KEY_MASK);
            }
            first = sa_imp_base(base, term) + lo;
            cnt = hi - lo;
        }
        s_first[tid] = first; s_cnt[tid] = cnt;
    }
    __syncthreads();
    for (u32 t = 0; t < T; t++) {                               // (uniform)
        const u64 first = s_first[t];
        const u32 cnt = s_cnt[t];
        const float w = q.w[t];
        for (u32 j = tid; j < cnt; j += 256u) {
            const u64 cell = imp[first + j];
            const u32 d = ((u32)(cell >> 32) >> 2) - (u32)d0;   // (a term holds a doc once: no two lanes meet on an accumulator)
            acc[d] = __fadd_rn(acc[d], __fmul_rn(__uint_as_float((u32)cell), w));
        }
        if (T > 1u) __syncthreads();
    }
    __syncthreads();
    const u64 left = n_docs - d0;
    const u32 n = left < (u64)TILE ? (u32)left : (u32)TILE;
    if (n == (u32)TILE && (n_docs & 3ull) == 0ull) {
        for (u32 i = tid; i < (u32)TILE / 4u; i += 256u) {
            float4 v = ((const float4*)acc)[i];
            if (has_boost) { v.x = __fmul_rn(v.x, boost); v.y = __fmul_rn(v.y, boost); v.z = __fmul_rn(v.z, boost); v.w = __fmul_rn(v.w, boost); }
            ((float4*)(out + d0))[i] = v;
        }
    } else {
        for (u32 i = tid; i < n; i += 256u) out[d0 + i] = has_boost ? __fmul_rn(acc[i], boost) : acc[i];
    }
}

extern "C" int sa_index_bm25_dense(sa_index_t* ix, const uint32_t* terms, const float* idf,
                                   int n_query_terms, float k1, float b, float* out) {
    SA_ARG(ix && out, "null argument");
    SA_ARG(n_query_terms >= 0, "n_query_terms < 0");
    SA_ARG(n_query_terms == 0 || (terms && idf), "terms/idf null");
    std::unique_lock<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SaDenseLaneScope lane(ix, g);                               // (enqueue under the lock, wait outside it: sa_index.hpp, DenseLane)
    SA_TRY(lane.rc);
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    // reference similarity.py:31-32: avg_doc_lens == 0 -> zeros
    if (N == 0 || n_query_terms == 0 || ix->avg_doc_len == 0.f) {
        sa_emit_zeros(ix, out);
        return SA_OK;
    }
    const int T = n_query_terms;
    SA_ARG(T <= SA_MAX_QTERMS, "more than 32 query terms per call is not supported");
    // One launch over the impact stream of (k1, b) -- the stream the index already holds for these parameters, or a new one when
    // it holds none (built once: the batches of the same parameters share it); an index whose stream belongs to OTHER parameters
    // keeps it and scores the TF postings as before.
    if (sa_opt(ix->opts.dense_direct, 1) != 0 && (ix->tile_docs == 1024 || ix->tile_docs == 2048 || ix->tile_docs == 4096) && ix->n_tiles > 0) {
        auto same = [](float x, float y) { return memcmp(&x, &y, sizeof(float)) == 0; };
        std::shared_ptr<sa_impacts> im;
        if (!ix->impacts || (same(ix->impacts->k1, k1) && same(ix->impacts->b, b) && same(ix->impacts->avgdl, ix->avg_doc_len)))
            im = sa_impacts_get(ix, k1, b, ix->opts);
        if (im && im->d_imp) {
            DenseQuery dq;
            memset(&dq, 0, sizeof(dq));
            for (int i = 0; i < T; i++) { dq.term[i] = terms[i]; dq.w[i] = idf[i]; }
            float* dst = nullptr; float boost = 1.f; int has_boost = 0;
            const bool to_vec = sa_vec_target_take(ix, &dst, &boost, &has_boost);
            if (!to_vec) {
                void* scratch;
                SA_TRY(sa_index_scratch(ix, N * sizeof(float) + 1024, &scratch));
                dst = (float*)scratch;
            }
#define SA_LAUNCH_DENSE(TILE)                                                                                                      \
    hipLaunchKernelGGL((sa_k_bm25_dense_direct<TILE>), dim3(ix->n_tiles), dim3(256), 0, st, (const u64*)ix->d_tfp, (const u64*)ix->d_tf_off, \
                       (const u32*)ix->d_dir_slot, (const u32*)ix->d_tile_dir, (const u64*)im->d_imp, ix->n_terms, ix->n_tiles, N, dq, (u32)T,  \
                       boost, has_boost, dst)
            if (ix->tile_docs == 1024) SA_LAUNCH_DENSE(1024);
            else if (ix->tile_docs == 2048) SA_LAUNCH_DENSE(2048);
            else SA_LAUNCH_DENSE(4096);
#undef SA_LAUNCH_DENSE
            SA_HIP(hipGetLastError());
            if (!to_vec) SA_TRY(sa_emit_dense(ix, dst, out));
            SA_TRY(lane.finish());
            return SA_OK;
        }
    }
    // scratch: [out chunk accumulators N floats][terms][idf]
    void* scratch;
    const size_t nb = (size_t)T * (ix->n_tiles + 1);
    SA_TRY(sa_index_scratch(ix, N * sizeof(float) + (size_t)T * 16 + nb * 4 + SA_SAT_NTF * SA_SAT_WMAX * 4 + 1024, &scratch));
    float* d_out = (float*)scratch;
    u64* d_qbase = (u64*)(d_out + ((N + 1) & ~(u64)1));
    u32* d_terms = (u32*)(d_qbase + T);
    float* d_idf = (float*)(d_terms + T);
    float* d_tab = d_idf + T;
    u32* d_bounds = (u32*)(d_tab + SA_SAT_NTF * SA_SAT_WMAX);
    SA_HIP(hipMemcpyAsync(d_terms, terms, (size_t)T * sizeof(u32), hipMemcpyHostToDevice, st));
    SA_HIP(hipMemcpyAsync(d_idf, idf, (size_t)T * sizeof(float), hipMemcpyHostToDevice, st));
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    sa_fill_params(ix, p);
    p.terms = d_terms; p.idf = d_idf; p.B = 1; p.nq = 1; p.T = (u32)T; p.k = 0;
    p.k1 = k1; p.b = b; p.pruned = 0;
    p.dense_out = d_out; p.cand = nullptr;
    p.bounds = d_bounds; p.qbase = d_qbase;
    p.sattab = d_tab;
    SA_TRY(sa_launch_make_sattab(ix, d_tab, &p.tab_w, k1, b, st));
    SA_TRY(sa_launch_make_bounds(ix, d_terms, (u32)T, d_bounds, d_qbase, st));
    SA_TRY(sa_launch_bm25(ix, p, st));
    SA_TRY(sa_emit_dense(ix, d_out, out));
    SA_TRY(lane.finish());
    SA_HIP(hipGetLastError());
    return SA_OK;
}

void sa_batch_free(sa_batch* bt) {
    if (!bt) return;
    if (bt->ix) {
        // every stream a run of this batch may have used: the index stream, the side stream (ungrouped rows), the
        // exchange stream (all-gather, cross-rank merge, result copies) and the lanes of dense-route phrases
        hipSetDevice(bt->ix->device);
        hipStreamSynchronize(bt->ix->stream);
        if (bt->st) hipStreamSynchronize(bt->st);
        if (bt->ix->sstream) hipStreamSynchronize(bt->ix->sstream);
        if (bt->ix->xstream) hipStreamSynchronize(bt->ix->xstream);
        for (int j = 0; j < 3; j++)
            if (bt->ix->lane_stream[j]) hipStreamSynchronize(bt->ix->lane_stream[j]);
    }
    // (d_terms, d_idf, d_perm, d_grp, d_ub, d_ub_order, d_lead, d_p1_off, d_qdf, d_qrow8, d_bloom_off, d_bloom_shift,
    //  d_plan live inside the upload block)
    if (bt->own_stream && bt->st) { hipStreamSynchronize(bt->st); hipStreamDestroy(bt->st); }
    if (bt->d_up) hipFree(bt->d_up);
    for (int i = 0; i < 2; i++) {
        if (bt->h_up[i]) hipHostFree(bt->h_up[i]);
        if (bt->ev_up[i]) hipEventDestroy(bt->ev_up[i]);
    }
    if (bt->h_res) hipHostFree(bt->h_res);
    if (bt->ev_res) hipEventDestroy(bt->ev_res);
    if (bt->ev_final) hipEventDestroy(bt->ev_final);
    if (bt->d_cand) hipFree(bt->d_cand);
    if (bt->d_bounds) hipFree(bt->d_bounds);
    if (bt->d_sattab) hipFree(bt->d_sattab);
    if (bt->d_qbase) hipFree(bt->d_qbase);
    if (bt->d_qbase_imp) hipFree(bt->d_qbase_imp);
    bt->impacts.reset();
    if (bt->d_slots) hipFree(bt->d_slots);
    if (bt->d_local) hipFree(bt->d_local);
    if (bt->d_gather) hipFree(bt->d_gather);
    if (bt->d_xlocal) hipFree(bt->d_xlocal);
    for (int i = 0; i < 2; i++) {
        if (bt->ev_side[i]) hipEventDestroy(bt->ev_side[i]);
        if (bt->ev_scored[i]) hipEventDestroy(bt->ev_scored[i]);
        if (bt->ev_exchanged[i]) hipEventDestroy(bt->ev_exchanged[i]);
    }
    if (bt->d_final) hipFree(bt->d_final);
    if (bt->d_xcand) hipFree(bt->d_xcand);
    if (bt->d_wbounds) hipFree(bt->d_wbounds);
    if (bt->d_wbase) hipFree(bt->d_wbase);
    if (bt->d_wlen) hipFree(bt->d_wlen);
    if (bt->d_stats) hipFree(bt->d_stats);
    if (bt->d_wl) hipFree(bt->d_wl);
    if (bt->d_wl_cnt) hipFree(bt->d_wl_cnt);
    if (bt->d_iota) hipFree(bt->d_iota);
    if (bt->d_route) hipFree(bt->d_route);
    if (bt->d_emask) hipFree(bt->d_emask);
    if (bt->d_p2_off) hipFree(bt->d_p2_off);
    if (bt->d_tile_q) hipFree(bt->d_tile_q);
    if (bt->d_surv) hipFree(bt->d_surv);
    if (bt->d_bloom) hipFree(bt->d_bloom);
    for (hipEvent_t e : bt->ev0) hipEventDestroy(e);
    for (hipEvent_t e : bt->ev1) hipEventDestroy(e);
    delete bt;
}

// The upload block of a batch: `bytes` on the device + two page-locked host images (see sa_batch.hpp).
int sa_batch_alloc_upload(sa_batch* bt, size_t bytes) {
    bt->up_bytes = (bytes + 15) & ~(size_t)15;
    SA_HIP(hipMalloc(&bt->d_up, bt->up_bytes));
    for (int i = 0; i < 2; i++) {
        SA_HIP(hipHostMalloc(&bt->h_up[i], bt->up_bytes, 0));
        memset(bt->h_up[i], 0, bt->up_bytes);
        SA_HIP(hipEventCreateWithFlags(&bt->ev_up[i], hipEventDisableTiming));
    }
    return SA_OK;
}

// The host image the next reset fills (waits, if it must, until the copy that last used it is done -- two resets
// ago, so in a steady stream this never blocks).
int sa_batch_upload_begin(sa_batch* bt, char** image) {
    const u32 i = bt->up_n & 1u;
    if (bt->up_used[i]) SA_HIP(hipEventSynchronize(bt->ev_up[i]));
    *image = bt->h_up[i];
    return SA_OK;
}

// one async copy of the whole image, on the index stream: ordered behind the runs that still read the old tables
int sa_batch_upload_commit(sa_batch* bt) {
    const u32 i = bt->up_n & 1u;
    SA_HIP(hipMemcpyAsync(bt->d_up, bt->h_up[i], bt->up_bytes, hipMemcpyHostToDevice, bt->st));
    SA_HIP(hipEventRecord(bt->ev_up[i], bt->st));
    bt->up_used[i] = true;
    bt->up_n++;
    return SA_OK;
}

// Candidate lists, pruning slots, result buffers and the timing-event ring of a batch whose tile
// kernel runs n_tiles tiles of `waves` waves per query.
int sa_batch_alloc_topk(sa_batch* bt, u32 n_tiles, u32 waves) {
    const u32 B = bt->B;
    // candidate storage per query: worst case every wave appends k keys; capped at 1 Mi keys per
    // query (8 MiB) -- with the cap an overflow is theoretically possible and is detected at run
    // time (sa_batch_run_shard re-runs a BM25 batch with the unpruned block-level selection).
    const u64 worst = (u64)(n_tiles ? n_tiles : 1) * bt->k * waves;
    const u64 mode0 = (u64)(n_tiles ? n_tiles : 1) * bt->k;
    u64 cap = worst < (1ull << 20) ? worst : (1ull << 20);
    if (cap < mode0) cap = mode0;                      // the unpruned layout [n_tiles][k] must fit too
    // the sparse candidate path appends every doc of a lead term that is scored before the bound exists
    if (bt->kind == 0 && cap < (1ull << 17)) cap = 1ull << 17;
    if (sa_opt_is_set(bt->opts.cand_cap)) {            // tests: force the overflow handling
        const u64 forced = (u64)bt->opts.cand_cap;
        cap = forced > mode0 ? forced : mode0;
    }
    bt->cand_cap = (u32)cap;
    bt->cap_limited = cap < worst;                     // (sa_tile_topk_pruned appends at most k keys per wave)
    const size_t ncand = (size_t)B * cap;
    SA_HIP(hipMalloc(&bt->d_cand, ncand * sizeof(u64)));
    // slots + cursors (+ cached bounds + score histograms for k > 32): one memset per run
    SA_HIP(hipMalloc(&bt->d_slots, ((size_t)B * (34 + SA_HBINS) + 1) * sizeof(u32)));
    bt->d_cand_cnt = bt->d_slots + (size_t)B * 32;
    bt->d_gthr = bt->d_slots + (size_t)B * 33;
    bt->d_hist = bt->d_slots + (size_t)B * 34;
    SA_HIP(hipMalloc(&bt->d_local, (size_t)B * bt->k * sizeof(u64)));
    // the final keys, and behind them two flag cells that travel to the host with them in ONE copy: [B*k] "a candidate
    // list of this shard ran over" (set by the merge kernel), [B*k + 1] the same over all ranks (set by the exchange)
    SA_HIP(hipMalloc(&bt->d_final, ((size_t)B * bt->k + 2) * sizeof(u64)));
    SA_HIP(hipMemset(bt->d_final, 0, ((size_t)B * bt->k + 2) * sizeof(u64)));
    bt->d_overflow = (u32*)(bt->d_final + (size_t)B * bt->k);
    bt->d_xflag = (u32*)(bt->d_final + (size_t)B * bt->k + 1);
    SA_HIP(hipMemset(bt->d_local, 0, (size_t)B * bt->k * sizeof(u64)));
    SA_HIP(hipHostMalloc(&bt->h_res, ((size_t)B * bt->k + 2) * sizeof(u64), 0));
    SA_HIP(hipEventCreateWithFlags(&bt->ev_res, hipEventDisableTiming));
    SA_HIP(hipEventCreateWithFlags(&bt->ev_final, hipEventDisableTiming));
    for (int i = 0; i < SA_EVENT_RING; i++) {
        hipEvent_t a = nullptr, c = nullptr;
        SA_HIP(hipEventCreate(&a));
        bt->ev0.push_back(a);
        SA_HIP(hipEventCreate(&c));
        bt->ev1.push_back(c);
    }
    return SA_OK;
}

static u64 sa_pow2_cells(u64 df) {                     // Bloom cells of a lead term: the power of two in [8, 16) x df, at least 1024
    u32 bits = 10;
    while ((1ull << bits) < 8 * df && bits < 30) bits++;
    return 1ull << bits;
}

// ---- a BM25 batch in two steps: sa_batch_alloc_bm25 sizes every device buffer ONCE from (B, T, k, the shard's
//      tiles), sa_batch_fill computes everything that depends on the queries into the upload image and enqueues the
//      copy and the slice-table kernel.  sa_batch_create = alloc + fill + one synchronisation; sa_batch_reset = fill.
static int sa_batch_alloc_bm25(sa_batch* bt) {
    sa_index* ix = bt->ix;
    const size_t B = bt->B, T = bt->T;
    // A BM25 batch has a stream of its own: batches of one index share nothing but the (read-only) index, so two
    // batches used alternately by a query stream overlap -- the tail of one batch's scoring kernels (the last, partly
    // filled round of workgroups) and its merge run beside the head of the next.  Option batch_stream = 0: the index stream.
    bt->st = ix->stream;
    if (sa_opt(bt->opts.batch_stream, 1) != 0) {
        SA_HIP(hipStreamCreateWithFlags(&bt->st, hipStreamNonBlocking));
        bt->own_stream = true;
    }
    // upload block (8-byte fields first)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 7) & ~(size_t)7; return o; };
    const size_t o_p1 = take((B + 1) * 8), o_boff = take(B * 8), o_terms = take(B * T * 4), o_idf = take(B * T * 4),
                 o_perm = take(B * 4), o_grp = take(3 * B * 4), o_ub = take(B * (T + 1) * 4), o_ord = take(B * T * 4),
                 o_lead = take(B * 4), o_qdf = take(B * T * 4), o_row8 = take(B * T * 4), o_bsh = take(B * 4),
                 o_seed = take(B * 4);
    bt->st_bytes = sa_stage_upload_bytes((u32)B, (u32)T);
    off = (off + 15) & ~(size_t)15;
    const size_t o_st = take(bt->st_bytes);
    SA_TRY(sa_batch_alloc_upload(bt, off));
    char* u = bt->d_up;
    bt->d_p1_off = (u64*)(u + o_p1); bt->d_bloom_off = (u64*)(u + o_boff);
    bt->d_terms = (u32*)(u + o_terms); bt->d_idf = (float*)(u + o_idf); bt->d_perm = (u32*)(u + o_perm);
    bt->d_grp = (u32*)(u + o_grp); bt->d_ub = (float*)(u + o_ub); bt->d_ub_order = (u32*)(u + o_ord);
    bt->d_lead = (u32*)(u + o_lead); bt->d_qdf = (u32*)(u + o_qdf); bt->d_qrow8 = (u32*)(u + o_row8);
    bt->d_bloom_shift = (u32*)(u + o_bsh);
    bt->d_seed = (u32*)(u + o_seed);
    bt->d_st = u + o_st;
    {
        std::vector<u32> iota(B);
        for (u32 i = 0; i < B; i++) iota[i] = i;
        SA_HIP(hipMalloc(&bt->d_iota, B * sizeof(u32)));
        SA_HIP(hipMemcpy(bt->d_iota, iota.data(), B * sizeof(u32), hipMemcpyHostToDevice));
    }
    // work list of the grouped kernel: at most one entry per (tile, row)
    bt->wl_cap = (u32)std::max<size_t>(1, (size_t)ix->n_tiles * B);
    SA_HIP(hipMalloc(&bt->d_wl, (size_t)bt->wl_cap * sizeof(u64)));
    SA_HIP(hipMalloc(&bt->d_wl_cnt, sizeof(u32)));
    SA_HIP(hipMemset(bt->d_wl_cnt, 0, sizeof(u32)));
    SA_TRY(sa_batch_alloc_topk(bt, ix->n_tiles, sa_tile_waves(ix->tile_docs)));
    SA_HIP(hipMalloc(&bt->d_bounds, (B * T * (ix->n_tiles + 1) + 1) * sizeof(u32)));
    SA_HIP(hipMalloc(&bt->d_qbase, B * T * sizeof(u64)));
    SA_HIP(hipMalloc(&bt->d_sattab, SA_SAT_NTF * SA_SAT_WMAX * sizeof(float)));
    // dynamic pruning (sa_sparse.hip): device-only tables
    SA_HIP(hipMalloc(&bt->d_route, B * sizeof(u32)));
    SA_HIP(hipMalloc(&bt->d_emask, B * sizeof(u32)));
    SA_HIP(hipMalloc(&bt->d_p2_off, (B + 1) * sizeof(u64)));
    SA_HIP(hipMalloc(&bt->d_tile_q, (B + 2) * sizeof(u32)));
    SA_HIP(hipMemset(bt->d_route, 0xFF, B * sizeof(u32)));
    {
        // survivors of the phase-2 bound check: a few percent of the candidates; capped, the rest is scored in place
        u64 cap = (u64)B * 65536;
        if (cap > (16u << 20)) cap = 16u << 20;
        bt->surv_cap = (u32)cap;
        SA_HIP(hipMalloc(&bt->d_surv, (size_t)cap * 2 * sizeof(u64)));
    }
    SA_TRY(sa_launch_make_sattab(ix, bt->d_sattab, &bt->tab_w, bt->k1, bt->b, bt->st));
    // the impact stream of this (k1, b): shared through the index, built on first use (on the index stream: done
    // before this batch's stream goes on)
    bt->impacts = sa_impacts_get(ix, bt->k1, bt->b, bt->opts);
    SA_HIP(hipStreamSynchronize(ix->stream));
    if (bt->impacts) SA_HIP(hipMalloc(&bt->d_qbase_imp, B * T * 2 * sizeof(u64)));
    return SA_OK;
}

// lead terms of dynamic pruning: up to 1/64 of the shard's docs (phase 1 scores every one of them), and never more
// postings than fit the candidate list while the bound is still unknown
static u64 sa_batch_lead_limit(const sa_batch* bt) {
    const sa_index* ix = bt->ix;
    u64 limit1 = ix->n_docs / 64 > 4096 ? ix->n_docs / 64 : 4096;
    if (limit1 > (u64)bt->cand_cap * 3 / 4) limit1 = (u64)bt->cand_cap * 3 / 4;
    return limit1;
}

// The Bloom filters of the lead terms are sized per query set (bt->bloom_bytes, sa_batch_fill).  The buffer holds what the
// query sets seen so far needed, with half as much again: a run whose set needs more waits for the batch's stream (older runs
// of this batch read the old buffer), frees it and allocates the larger one -- after the first few sets of a stream never
// again.  (Round 3 allocated the worst case of the shard -- B lead terms of lead-limit postings each: 512 MiB per 256-query
// batch at 10 M docs where a BASELINE set needs ~20 MiB.)
int sa_batch_ensure_bloom(sa_batch* bt) {
    const size_t need = bt->bloom_bytes;
    if (bt->d_bloom && bt->bloom_cap >= need) return SA_OK;
    if (bt->d_bloom) {
        SA_HIP(hipStreamSynchronize(bt->st));
        SA_HIP(hipFree(bt->d_bloom));
        bt->d_bloom = nullptr; bt->bloom_cap = 0;
    }
    size_t cap = need + need / 2;
    const size_t floor_bytes = (size_t)std::max<long long>(1024, sa_opt(bt->opts.bloom_floor, 1 << 20));      // (tests: small, so that the buffer has to grow)
    if (cap < floor_bytes) cap = floor_bytes;
    SA_HIP(hipMalloc(&bt->d_bloom, cap));
    bt->bloom_cap = cap;
    return SA_OK;
}

// Dynamic pruning tables of the current query set (sa_sparse.hip), derived into the upload image `img` from the rows already
// placed there (terms / idf in device-row order).  ~9 us for 256 x 4 terms: sa_batch_fill only calls it when the run may take
// the pruning path (SA_SPARSE, or the default rule with few grouped rows); a run that wants the path after all -- the
// switch changed between reset and run -- derives them then and uploads the image again (sa_batch_ensure_sparse_tables).
static void sa_batch_fill_prune_tables(sa_batch* bt, char* img) {
    sa_index* ix = bt->ix;
    const u32 B = bt->B, T = bt->T;
    const float k1 = bt->k1, b = bt->b;
    auto at = [&](const void* dptr) { return img + ((const char*)dptr - bt->d_up); };
    u64* h_p1 = (u64*)at(bt->d_p1_off);
    u64* h_boff = (u64*)at(bt->d_bloom_off);
    u32* h_terms = (u32*)at(bt->d_terms);
    float* h_idf = (float*)at(bt->d_idf);
    float* h_ub = (float*)at(bt->d_ub);
    u32* h_ord = (u32*)at(bt->d_ub_order);
    u32* h_lead = (u32*)at(bt->d_lead);
    u32* h_qdf = (u32*)at(bt->d_qdf);
    u32* h_row8 = (u32*)at(bt->d_qrow8);
    u32* h_bshift = (u32*)at(bt->d_bloom_shift);
        // Dynamic pruning tables (sa_sparse.hip).  Per query: the terms in ascending idf order with the
        // prefix sums of their idf -- what the j cheapest terms can add to a score at most, since
        // tf/(tf+norm) <= 1 (needs k1 >= 0 and 0 <= b <= 1; a negative or non-finite idf switches the
        // pruning off) -- and the LEAD term: the highest-idf term with postings in this shard.
        const bool formula_ok = k1 >= 0.f && b >= 0.f && b <= 1.f;
        const u64 limit1 = sa_batch_lead_limit(bt);
        const u64 sp_div = (u64)std::max<long long>(1, sa_opt(bt->opts.sparse_div, 8));
        bt->sparse_limit2 = ix->n_docs / sp_div > 4096 ? ix->n_docs / sp_div : 4096;
        std::pair<float, u32> v[SA_MAX_QTERMS];
        h_p1[0] = 0;
        for (u32 r = 0; r < B; r++) {
            bool ok = formula_ok;
            for (u32 t = 0; t < T; t++) {
                const bool known = h_terms[(size_t)r * T + t] < ix->n_terms;
                const float w = known ? h_idf[(size_t)r * T + t] : 0.f;
                if (!(w >= 0.f) || w > 3.0e38f) ok = false;
                v[t] = {w, t};
            }
            std::stable_sort(v, v + T, [](const std::pair<float, u32>& a, const std::pair<float, u32>& c) { return a.first < c.first; });
            double acc = 0.0;
            h_ub[(size_t)r * (T + 1)] = 0.f;
            for (u32 j = 0; j < T; j++) {
                h_ord[(size_t)r * T + j] = v[j].second;
                acc += (double)v[j].first;
                // rounded up: the fp32 sum the kernels form can exceed the exact sum by a few ulps
                float ubf = (float)(acc * (1.0 + 1e-5));
                ubf = nextafterf(ubf, INFINITY);
                h_ub[(size_t)r * (T + 1) + j + 1] = ok ? ubf : INFINITY;
            }
            // lead: highest idf among the terms with postings here; too frequent -> scan the tiles
            h_lead[r] = 0xFFFFFFFFu;
            h_p1[r + 1] = 0;
            if (ok) {
                for (int j = (int)T - 1; j >= 0; j--) {
                    const u32 t = v[(size_t)j].second;
                    const u32 term = h_terms[(size_t)r * T + t];
                    if (term >= ix->n_terms) continue;
                    const u64 df = ix->h_tf_off[term + 1] - ix->h_tf_off[term];
                    if (df == 0) continue;
                    if (df <= limit1) { h_lead[r] = t; h_p1[r + 1] = df; }          // postings; turned into items below
                    break;
                }
            }
        }
        {
            // Lead-phase work items: 1024 postings each when there is plenty of work (measured best at
            // 10 M docs: 0.66 vs 0.79 ms per step with 256), 256 when the shard is small and the phase
            // would otherwise not fill the GPU (1.25 M docs: 0.157 vs 0.169 ms).
            u64 lead_postings = 0;
            for (u32 r = 0; r < B; r++) lead_postings += h_p1[r + 1];
            bt->sparse_chunk1 = lead_postings >= (1ull << 19) ? SA_SP_CHUNK : SA_SP_CHUNK_LEAD;
            if (sa_opt(bt->opts.sp_chunk1, 0) >= 64) bt->sparse_chunk1 = (u32)bt->opts.sp_chunk1;
            for (u32 r = 0; r < B; r++) h_p1[r + 1] = (h_p1[r + 1] + bt->sparse_chunk1 - 1) / bt->sparse_chunk1;
        }
        for (u32 r = 0; r < B; r++) h_p1[r + 1] += h_p1[r];
        bt->sparse_p1_total = h_p1[B];
        bt->sparse_ok = true;
        bt->sparse_p2_max = 0;
        for (size_t i = 0; i < (size_t)B * T; i++) {
            const u32 term = h_terms[i];
            h_qdf[i] = 0; h_row8[i] = SA_DD_NONE;
            if (term >= ix->n_terms) continue;
            h_qdf[i] = (u32)(ix->h_tf_off[term + 1] - ix->h_tf_off[term]);
            bt->sparse_p2_max += ((u64)h_qdf[i] + SA_SP_CHUNK - 1) / SA_SP_CHUNK;
            if (!ix->h_tf8_slot.empty()) h_row8[i] = ix->h_tf8_slot[term];
        }
        // Bloom filter of every lead term: the power of two in [8, 16) x df cells, at least 1024
        size_t bytes = 0;
        for (u32 r = 0; r < B; r++) {
            const u64 df = h_lead[r] == 0xFFFFFFFFu ? 0 : h_qdf[(size_t)r * T + h_lead[r]];
            const u64 cells = sa_pow2_cells(df);
            h_boff[r] = bytes;
            h_bshift[r] = 32u - (u32)__builtin_ctzll(cells);
            bytes += (size_t)cells;
        }
        bt->bloom_bytes = bytes;
        bt->sparse_lazy = false;
}

static int sa_batch_fill(sa_batch* bt, const uint32_t* terms, const float* idf) {
    sa_index* ix = bt->ix;
    const u32 B = bt->B, T = bt->T;
    const float k1 = bt->k1, b = bt->b;
    char* img = nullptr;
    const u64 t_begin = sa_now_ns();
    SA_TRY(sa_batch_upload_begin(bt, &img));
    auto at = [&](const void* dptr) { return img + ((const char*)dptr - bt->d_up); };
    u32* h_terms = (u32*)at(bt->d_terms);
    float* h_idf = (float*)at(bt->d_idf);
    u32* h_perm = (u32*)at(bt->d_perm);
    u32* h_grpd = (u32*)at(bt->d_grp);

    // Order queries by their most frequent term so XCD groups share posting tiles in L2.
    bt->perm.resize(B);
    for (u32 i = 0; i < B; i++) bt->perm[i] = i;
    std::vector<u64> heavy(B, 0);
    std::vector<u32> heavy_term(B, SA_NO_TERM);
    bt->alg_bytes = 0; bt->postings_bytes = 0;
    for (u32 i = 0; i < B; i++) {
        for (u32 t = 0; t < T; t++) {
            const u32 term = terms[(size_t)i * T + t];
            if (term >= ix->n_terms) continue;
            const u64 df = ix->h_tf_off[term + 1] - ix->h_tf_off[term];
            bt->postings_bytes += 8 * df;
            if (df > heavy[i]) { heavy[i] = df; heavy_term[i] = term; }
        }
        bt->alg_bytes += 4 * ix->n_docs;
    }
    bt->alg_bytes += bt->postings_bytes;
    std::stable_sort(bt->perm.begin(), bt->perm.end(), [&](u32 a, u32 c) {
        if (heavy[a] != heavy[c]) return heavy[a] > heavy[c];
        return heavy_term[a] < heavy_term[c];
    });
    // Groups of queries that share their FIRST term (same term, same idf bits): sa_k_bm25_group_tiles scores the
    // shared term once per (tile, group).  Grouped queries take the first device rows, group by group (big
    // groups are cut into balanced pieces of at most `maxq` queries), the others keep their order behind them.
    std::vector<u32> h_grp;
    bt->n_groups = 0; bt->n_grouped_rows = 0; bt->n_shared_rows = 0;
    {
        // lanes per query while the half tables are built: a power of two >= the terms overlaid -- T - 1 for groups
        // with a shared first term, T for loose groups (decided below; loose groups are only formed if the wider
        // table still takes SA_GRP_MAXQ queries)
        u32 tt = 1, tsh = 0;
        while (tt + 1u < T) { tt <<= 1; tsh++; }               // power of two >= max(T - 1, 1)
        u32 tt_loose = 1, tsh_loose = 0;
        while (tt_loose < T) { tt_loose <<= 1; tsh_loose++; }
        const bool loose_on = sa_opt(bt->opts.group_loose, 1) != 0 && 128u / tt_loose >= SA_GRP_MAXQ;
        u32 maxq = std::min<u32>(SA_GRP_MAXQ, 128u / tt);
        // (a shard whose (tile, group) items do not fill the device for many rounds is better cut into more, shorter items:
        //  SA_GROUP_MAXQ; measured on a 1.25 M-doc shard below)
        maxq = std::min<u32>(maxq, (u32)std::max<long long>(1, sa_opt(bt->opts.group_maxq, SA_GRP_MAXQ)));
        // queries per ITEM: an item takes its queries in passes of maxq over one base, so a group of 25 is ONE item (two passes), not
        // two items that each pay the item's fixed cost (block -> tile, group entry, slice of the shared term, dense row, base:
        // ~6 K of an item's ~34 K cycles)
        //  Two passes per item are worth it where the launch has many rounds of items to run (same box, BASELINE batch, items of 16 vs
        //  32 queries: 10 M docs 0.402 -> 0.381 ms, 7.5 M 0.311 -> 0.302, 5 M 0.2185 -> 0.218, 2.5 M 0.124 -> 0.134, 1.25 M 0.085 -> 0.094:
        //  fewer, longer items lengthen the launch's tail) -- from 16 rounds of the device's 4096 resident waves on; loose groups have
        //  no base to share and keep 16 (measured: 0.464 -> 0.483 ms with 32).  2048 queries per batch (groups of ~200, 155 rounds): items of
        //  16 / 32 / 48 / 64 queries 2.339 / 2.213 / 2.172 / 2.168 ms -- four passes from 64 rounds on.
        u32 item_q = maxq;                                      // (set below, once the groups are known)
        bt->grp_cq = maxq;
        const u32 gmin = (u32)std::max<long long>(1, sa_opt(bt->opts.group_min, 2));
        bool idf_ok = k1 >= 0.f && b >= 0.f && b <= 1.f;       // scores must be non-negative (the sign bit is a mark)
        for (size_t i = 0; i < (size_t)B * T && idf_ok; i++) idf_ok = idf[i] >= 0.f && idf[i] <= 3.0e38f;
        const bool on = sa_opt(bt->opts.group, 1) != 0 && idf_ok && maxq >= 1 &&
                        (ix->tile_docs == 1024 || ix->tile_docs == 2048 || ix->tile_docs == 4096);
        if (on) {
            std::vector<std::vector<u32>> members;              // in order of first appearance
            std::vector<std::pair<u32, u32>> keys;
            std::vector<u32> rest;
            for (u32 r = 0; r < B; r++) {
                const u32 q = bt->perm[r];
                const u32 t0 = terms[(size_t)q * T];
                u32 ib; memcpy(&ib, &idf[(size_t)q * T], 4);
                if (t0 >= ix->n_terms) { rest.push_back(q); continue; }
                size_t gi = 0;
                for (; gi < keys.size(); gi++) if (keys[gi].first == t0 && keys[gi].second == ib) break;
                if (gi == keys.size()) { keys.push_back({t0, ib}); members.emplace_back(); }
                members[gi].push_back(q);
            }
            {
                u64 items16 = 0;
                for (auto& m : members) if (m.size() >= gmin) items16 += (m.size() + maxq - 1) / maxq;
                const u64 rounds = items16 * (u64)ix->n_tiles / 4096u;          // of one-pass items over the device's resident waves
                const long long dflt = rounds >= 4u * SA_GRP_ITEM_ROUNDS ? 64 : rounds >= (u64)SA_GRP_ITEM_ROUNDS ? 32 : 16;
                item_q = std::max<u32>(maxq, std::min<u32>(64u, (u32)std::max<long long>(1, sa_opt(bt->opts.group_item, dflt))) / maxq * maxq);
            }
            std::vector<u32> order;
            for (auto& m : members) {
                if (m.size() < gmin) { rest.insert(rest.end(), m.begin(), m.end()); continue; }
                const u32 pieces = ((u32)m.size() + item_q - 1) / item_q;
                u32 done = 0;
                for (u32 pc = 0; pc < pieces; pc++) {
                    const u32 sz = ((u32)m.size() - done + (pieces - pc) - 1) / (pieces - pc);
                    h_grp.push_back((u32)order.size()); h_grp.push_back(sz);
                    {
                        const u32 t0 = terms[(size_t)m[done] * T];
                        const bool have = bt->impacts && bt->impacts->d_dense && t0 < bt->impacts->dense_slot.size();
                        h_grp.push_back(have ? bt->impacts->dense_slot[t0] : 0xFFFFFFFFu);
                    }
                    for (u32 i = 0; i < sz; i++) order.push_back(m[done + i]);
                    done += sz;
                }
            }
            bt->n_shared_rows = (u32)order.size();                   // rows in groups with a shared first term
            // Loose groups: queries left over whose terms are all sparse per tile (what a (tile, query) pair of the
            // per-query kernel costs is clearing and scanning the tile's accumulators, 2048 slots for ~130 postings;
            // as an overlay on accumulators that are cleared once per 16 queries and touched only where the postings
            // are, it costs the postings).  Eligible: an expected sum of postings per tile that fits the overlay's
            // half table with room to spare; the others keep the per-query kernel (their tiles are dense).
            if (loose_on && ix->n_tiles > 0) {
                std::vector<u32> sparse_rows, dense_rows;
                for (u32 q : rest) {
                    u64 dfsum = 0;
                    for (u32 t = 0; t < T; t++) {
                        const u32 term = terms[(size_t)q * T + t];
                        if (term < ix->n_terms) dfsum += ix->h_tf_off[term + 1] - ix->h_tf_off[term];
                    }
                    if (dfsum > 0 && dfsum / ix->n_tiles <= (u64)std::max<long long>(1, sa_opt(bt->opts.loose_postings, SA_GRP_LOOSE_POSTINGS))) sparse_rows.push_back(q);
                    else dense_rows.push_back(q);
                }
                if (sparse_rows.size() >= 2) {
                    const u32 item_l = sa_opt_is_set(bt->opts.group_item) ? std::max<u32>(SA_GRP_MAXQ, item_q / SA_GRP_MAXQ * SA_GRP_MAXQ) : (u32)SA_GRP_MAXQ;
                    const u32 pieces = ((u32)sparse_rows.size() + item_l - 1) / item_l;
                    u32 done = 0;
                    for (u32 pc = 0; pc < pieces; pc++) {
                        const u32 sz = ((u32)sparse_rows.size() - done + (pieces - pc) - 1) / (pieces - pc);
                        h_grp.push_back((u32)order.size()); h_grp.push_back(sz | 0x80000000u); h_grp.push_back(0xFFFFFFFFu);
                        for (u32 i = 0; i < sz; i++) order.push_back(sparse_rows[done + i]);
                        done += sz;
                    }
                    rest = dense_rows;
                    tt = tt_loose; tsh = tsh_loose;
                }
            }
            // Groups of ONE (round 6, option group_one; round 5 review item 4): a query left over -- too dense for a loose group, its first
            // term shared with nobody -- is an item of its own: one WAVE per (tile, query) whose base comes from the first term's dense factor
            // row with 16-byte loads where it has one (no scatter of ~1900 postings per tile), from its postings otherwise, the other terms
            // overlaid.  Without it such queries run the per-query kernel (a workgroup per pair) on the side stream.  Measured
            // (profiles/group_of_one_dense_row_ab_r06.jsonl; 256 pairwise-distinct queries, 13 of them left over, 10 with a dense row):
            // 10 M docs k = 10 / 1000: none 0.465 / 0.859 ms, rows only (group_one = 1) 0.431 / 0.725, all (2, the default) 0.408 / 0.648;
            // 1.25 M docs k = 10 / 100: 0.138 / 0.180 -> 0.106 / 0.132; never slower on the BASELINE / hot sets.
            const long long g1 = sa_opt(bt->opts.group_one, 2);
            if (g1 != 0 && bt->impacts && sa_opt(bt->opts.group_dense, 1) != 0) {
                std::vector<u32> still;
                for (u32 q : rest) {
                    const u32 t0 = terms[(size_t)q * T];
                    const u32 slot = (bt->impacts->d_dense && t0 < ix->n_terms && t0 < bt->impacts->dense_slot.size()) ? bt->impacts->dense_slot[t0] : 0xFFFFFFFFu;
                    if (slot != 0xFFFFFFFFu || (g1 >= 2 && t0 < ix->n_terms)) { h_grp.push_back((u32)order.size()); h_grp.push_back(1u); h_grp.push_back(slot); order.push_back(q); }
                    else still.push_back(q);
                }
                rest = still;
            }
            bt->n_grouped_rows = (u32)order.size();
            bt->n_groups = (u32)(h_grp.size() / 3);
            order.insert(order.end(), rest.begin(), rest.end());
            bt->perm = order;
        }
        bt->grp_tt = tt; bt->grp_tt_shift = tsh;
    }
    for (u32 r = 0; r < B; r++) {
        memcpy(&h_terms[(size_t)r * T], &terms[(size_t)bt->perm[r] * T], T * sizeof(u32));
        memcpy(&h_idf[(size_t)r * T], &idf[(size_t)bt->perm[r] * T], T * sizeof(float));
        h_perm[r] = bt->perm[r];
    }
    memset(h_grpd, 0, (size_t)3 * B * sizeof(u32));
    if (!h_grp.empty()) memcpy(h_grpd, h_grp.data(), h_grp.size() * sizeof(u32));     // (at most B groups)
    {
        // the pruning tables: now, if the run may prune (the rule of sa_batch_run_shard, with what is known here); else on demand
        const int sp_env = (int)sa_opt(bt->opts.sparse, -1);
        const bool impact_default = bt->impacts != nullptr && sa_opt(bt->opts.impact, 1) != 0;     // (the run's rule, with what is known here)
        const bool maybe_sparse = sp_env >= 0 ? sp_env != 0 : !impact_default;
        if (maybe_sparse || sa_opt(bt->opts.sparse_lazy, 1) == 0) sa_batch_fill_prune_tables(bt, img);
        else { bt->sparse_ok = false; bt->sparse_lazy = true; bt->bloom_bytes = 0; bt->sparse_p1_total = 0; bt->sparse_p2_max = 0; }
    }
    memset(at(bt->d_seed), 0, (size_t)B * sizeof(u32));
    {
        bool w_ok = k1 >= 0.f && b >= 0.f && b <= 1.f;
        for (size_t i = 0; i < (size_t)B * T && w_ok; i++) w_ok = idf[i] >= 0.f && idf[i] <= 3.0e38f;
        const bool wanted = w_ok && sa_batch_wants_seed(bt);
        if (wanted) sa_impacts_ensure_topf(ix, bt->impacts.get());
        bt->seed_on = wanted && bt->impacts->d_topf;
    }
    // the staged-tile route's plan (sa_stage.hip): distinct terms, per-query bound tables and the starting bounds, formed on the
    // host into the same upload.  A set that has one does not need the slice table: sa_k_make_bounds is left out of the step and
    // only runs if the run takes another route after all (sa_batch_ensure_bounds)
    bt->stage_ok = false;
    bt->st_dir.reset();
    bt->st_slices.clear();
    if (bt->seed_on && sa_batch_stage_wanted(bt)) SA_TRY(sa_stage_plan(bt, img, h_terms, h_idf));
    const u64 t_host = sa_now_ns();
    SA_TRY(sa_batch_upload_commit(bt));
    bt->bounds_valid = false;
    if (!bt->stage_ok) SA_TRY(sa_batch_ensure_bounds(bt, bt->st));
    SA_HIP(hipGetLastError());
    const u64 t_end = sa_now_ns();
    bt->host_ns[0] += t_host - t_begin; bt->host_ns[1] += t_end - t_host; bt->host_ns[3]++;
    return SA_OK;
}

extern "C" int sa_batch_create(sa_index_t* ix, const uint32_t* terms, const float* idf, int n_queries,
                               int n_query_terms, int k, float k1, float b, sa_batch_t** out) {
    SA_ARG(ix && out && terms && idf, "null argument");
    SA_ARG(n_queries > 0 && n_query_terms > 0, "empty batch");
    SA_ARG(n_query_terms <= SA_MAX_QTERMS, "more than 32 terms per query is not supported");
    SA_ARG(k > 0 && k <= SA_KMAX, "k must be in [1, 1024]");
    SA_ARG(ix->doc_base + ix->n_docs <= 0xFFFFFFFFull, "global doc ids must fit 32 bits for top-k");
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    sa_batch* bt = new (std::nothrow) sa_batch();
    if (!bt) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    bt->opts = sa_options_for_new_handle(&ix->opts);
    bt->ix = ix; bt->B = (u32)n_queries; bt->T = (u32)n_query_terms; bt->k = (u32)k; bt->k1 = k1; bt->b = b;
    int rc = sa_batch_alloc_bm25(bt);
    if (rc == SA_OK) rc = sa_batch_fill(bt, terms, idf);
    if (rc == SA_OK && hipStreamSynchronize(bt->st) != hipSuccess) {
        sa_set_error("sa_batch_create: hipStreamSynchronize failed");
        rc = SA_ERR_HIP;
    }
    if (rc != SA_OK) { sa_batch_free(bt); return rc; }
    *out = bt;
    return SA_OK;
}

// A NEW set of queries in an existing batch (same B, T, k, k1, b): the host derives grouping, pruning tables and
// statistics into a page-locked image, ONE hipMemcpyAsync replaces the device tables and sa_k_make_bounds rebuilds
// the slice table -- all enqueued on the index stream behind the runs still in flight, nothing allocated, nothing
// waited for.  The caller idiom it serves is the reference's score() on a fresh query (postings.py:652-680; timed
// as test/test_msmarco.py:345-395 times it): two batches used alternately keep the device busy while the host
// prepares the next query set.
static int sa_batch_redo_if_flagged(sa_batch* bt);

extern "C" int sa_batch_reset(sa_batch_t* bt, const uint32_t* terms, const float* idf) {
    SA_ARG(bt && bt->ix && terms && idf, "null argument");
    SA_ARG(bt->kind == 0, "sa_batch_reset takes a BM25 batch (phrase batches: sa_phrase_batch_reset)");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    // a run whose results have not been fetched: its merge (on the exchange stream when sharded) still reads the row order this
    // reset replaces -- wait for that run's result copy first (landed long ago in the run / fetch / reset idiom: no cost there)
    if (bt->res_pending && bt->unfetched) {
        SA_HIP(hipEventSynchronize(bt->ev_res));
        SA_TRY(sa_batch_redo_if_flagged(bt));                   // (a flagged run is redone while the tables still hold ITS query set)
    }
    return sa_batch_fill(bt, terms, idf);
}

extern "C" int sa_index_set_idf_table(sa_index_t* ix, const float* idf_per_term, uint32_t n_terms) {
    SA_ARG(ix && (idf_per_term || n_terms == 0), "null argument");
    SA_ARG(n_terms == ix->n_terms, "one idf per term of the index");
    // (weights that are negative or not finite keep their batches off the grouped kernel, the starting bounds and the pruning --
    //  every step, silently: the table is checked once, here)
    for (uint32_t t = 0; t < n_terms; t++) SA_ARG(idf_per_term[t] >= 0.f && idf_per_term[t] <= 3.0e38f, "idf table: weights must be finite and >= 0");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->h_idf.assign(idf_per_term, idf_per_term + n_terms);
    return SA_OK;
}

// regroup an all-gather result [rank][B*k (+ extra)] into per-query candidate rows [B][rank*k]; `extra` = 1: every
// rank's block ends with its overflow flag, OR-ed into *xflag (a candidate list ran over on SOME rank: all ranks
// learn it from the exchange itself and redo the batch together at fetch)
__global__ void sa_k_regroup(const u64* __restrict__ gathered, u32 nranks, u32 B, u32 k, u32 extra, u64* __restrict__ out,
                             u32* __restrict__ xflag) {
    const u64 per = (u64)B * k;
    const u64 total = (u64)nranks * per;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 j = (u32)(i % k);
        const u32 q = (u32)((i / k) % B);
        const u32 r = (u32)(i / per);
        out[((u64)q * nranks + r) * k + j] = gathered[(u64)r * (per + extra) + (i - (u64)r * per)];
    }
    if (extra && xflag && blockIdx.x == 0 && threadIdx.x == 0) {
        u32 any = 0;
        for (u32 r = 0; r < nranks; r++) any |= gathered[(u64)r * (per + extra) + per] != 0ull ? 1u : 0u;
        if (any) *xflag = 1u;
    }
}

// stage 1 (tile scoring + per-tile top-k) and stage 2 (per-shard merge) on the index stream
// defer_check: an overflowing candidate list is only flagged on the device; sa_batch_fetch re-runs the
// batch unpruned before handing out results (no host round trip between the tile kernel and the merge)
static int sa_batch_run_shard(sa_batch* bt, u64* shard_out, bool defer_check, bool force_unpruned = false, u32* overflow_cell = nullptr) {
    sa_index* ix = bt->ix;
    hipStream_t st = bt->st;
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    sa_fill_params(ix, p);
    p.terms = bt->d_terms; p.idf = bt->d_idf; p.B = bt->B; p.T = bt->T; p.k = bt->k;
    p.k1 = bt->k1; p.b = bt->b;
    p.bounds = bt->d_bounds; p.qbase = bt->d_qbase;
    if (bt->impacts && bt->kind == 0 && sa_opt(bt->opts.impact, 1) != 0) {
        p.imp = bt->impacts->d_imp; p.qbase_imp = bt->d_qbase_imp;
        p.imp_tail = bt->impacts->n - 2;
    }
    p.sattab = bt->d_sattab; p.tab_w = bt->tab_w;
    p.pruned = (sa_opt(bt->opts.pruned_topk, 1) && !force_unpruned) ? 1 : 0;   // pruned wave-level selection (any k <= 1024)
    p.dense_out = nullptr; p.cand = bt->d_cand;
    p.no_topk = (int)sa_opt(bt->opts.no_topk, 0);
    p.cand_per_tile = bt->k;
    p.cand_cap = bt->cand_cap;
    p.cand_cnt = bt->d_cand_cnt;
    p.slots = bt->d_slots;
    if (bt->kind == 1) { p.pruned = 1; defer_check = false; }   // phrase tiles: pruned selection only
    // k > 32: histogram bound (BM25 tiles of <= 4 waves); SA_TOPK_HIST=0 keeps the slot bound
    // dynamic pruning (sa_sparse.hip; SA_SPARSE=0: score every posting, the exhaustive reference
    // behaviour): needs the histogram bound for every k
    const bool hist_possible = p.pruned && bt->kind == 0 && sa_tile_waves(ix->tile_docs) <= 8 &&
                               sa_opt(bt->opts.topk_hist, 1) != 0;
    // Unset, SA_SPARSE follows the measurements: pruning pays while the shard holds many docs per requested
    // result (10 M docs: 2.2x at k = 10, 1.9x at k = 100, but the exhaustive kernel is 1.2x faster at k = 1000;
    // 1.25 M docs, k = 1000: exhaustive 1.8x faster) -- on from 32768 docs per result (8192 since round 6, below).  SA_SPARSE=1 / 0 force it.
    const int sparse_env = (int)sa_opt(bt->opts.sparse, -1);
    // Round 2: when most queries of the batch share their first terms, the grouped exhaustive kernel is as fast at
    // k = 10 and faster above (10 M docs, BASELINE batch: 369 K vs 341 K queries/s at k = 100, 236 K vs 114 K at
    // k = 1000) -- such batches score every posting from k = 32 on.
    // (only where the grouped kernel can actually run: impact stream, histogram bound, its tile sizes -- otherwise the
    //  batch would fall to the per-query exhaustive kernel, which pruning beats 2x)
    const bool group_can_run = bt->n_groups && p.pruned && hist_possible && p.imp && !p.no_topk && sa_opt(bt->opts.group, 1) != 0 &&
                               (ix->tile_docs == 1024 || ix->tile_docs == 2048 || ix->tile_docs == 4096);
    // Round 3: the grouped kernel (4 waves per SIMD, ~100 VALU instructions per (tile, query) pair) beats pruning on such
    // batches from k = 10 on (10 M docs, BASELINE batch, k = 10: 0.46 vs 0.63 ms per step) -- no lower limit on k any more.
    // ... and so do the loose groups on batches WITHOUT shared terms (256 x 4 pairwise-distinct terms of ranks 1 .. 1024,
    // all of them frequent: 0.62 ms exhaustive vs 1.68 ms pruned at k = 10): the exhaustive path is the default whenever
    // at least half of the batch's queries are in groups of either kind.
    // Round 5, measured instead of argued (scripts/route_rule.py, profiles/route_rule_r05*.jsonl: the share of the batch the grouped
    // kernel can take x k, 10 M and 1.25 M docs): with the impact stream and the starting bounds the exhaustive path is within 5 % of
    // pruning in EVERY cell and up to 2 x faster (k = 100, less than half of the batch groupable: 1.59 vs 2.92 ms -- round 4's rule
    // picked pruning there); pruning keeps the batches the impact route cannot take (no impact stream, the histogram bound off,
    // other tile sizes), where the per-query TF kernel is what it competes with.
    const bool impact_route = p.pruned && hist_possible && p.imp && !p.no_topk &&
                              (ix->tile_docs == 1024 || ix->tile_docs == 2048 || ix->tile_docs == 4096);
    // Round 6, the batches that remain (no impact stream), measured by shard size (profiles/route_rule_r06_no_impact_stream*.jsonl: 1.25 / 2.5 /
    // 5 / 10 M docs x k = 10 / 100 / 1000 x two query sets): pruning beats the TF kernels 1.4 - 5.7 x from 10 000 docs per requested result
    // on (1.0 - 1.9 x at 10 000 - 12 500), loses up to 1.4 x at 5 000 and below -- on from 8192 docs per result (rounds 3-5: 32768).
    const bool sparse_wanted = sparse_env >= 0 ? sparse_env != 0 : (ix->n_docs >= (u64)bt->k * 8192ull && !impact_route);
    if (sparse_wanted && bt->sparse_lazy && bt->kind == 0 &&
        !(bt->stage_ok && sa_batch_stage_wanted(bt))) {
        // the tables were left out at reset (the run was expected to score every posting): derive them into the image of the
        // current query set -- once its upload has left the host buffer -- and upload it again, behind everything on this stream
        const u32 last = (bt->up_n - 1u) & 1u;
        SA_HIP(hipEventSynchronize(bt->ev_up[last]));
        sa_batch_fill_prune_tables(bt, bt->h_up[last]);
        SA_HIP(hipMemcpyAsync(bt->d_up, bt->h_up[last], bt->up_bytes, hipMemcpyHostToDevice, st));
        SA_HIP(hipEventRecord(bt->ev_up[last], st));
        // (the image carries the starting bounds as the host left them -- zeros: the slice-table kernel forms them again)
        bt->bounds_valid = false;
        SA_TRY(sa_batch_ensure_bounds(bt, st));
    }
    // Round 6: the staged-tile route (sa_stage.hip) takes every query set it has a plan for -- distinct terms staged in LDS once
    // per tile, the queries answered from there; it needs the histogram bound and the impact stream like the grouped kernel
    const bool stage = bt->kind == 0 && bt->stage_ok && sa_batch_stage_wanted(bt) && p.pruned && hist_possible && p.imp && !p.no_topk &&
                       ix->avg_doc_len != 0.f;
    bt->last_route_stage = stage;
    if (!stage && bt->kind == 0) SA_TRY(sa_batch_ensure_bounds(bt, st));
    const bool sparse = hist_possible && ix->tile_docs <= 8192 && bt->sparse_ok && ix->avg_doc_len != 0.f && ix->n_tiles > 0 &&
                        sparse_wanted && !stage;
    bt->last_route_sparse = sparse;
    const bool use_hist = hist_possible &&
                          (sparse || stage || bt->k >= (u32)sa_opt(bt->opts.topk_hist_mink, defer_check ? 1 : 33));
    p.hist = use_hist ? bt->d_hist : nullptr;
    p.gthr = use_hist ? bt->d_gthr : nullptr;
    // the bounds the queries start with (exhaustive kernels only: the pruning path derives its own from the lead terms)
    p.seed = (use_hist && !sparse && bt->kind == 0 && bt->seed_on && p.imp) ? bt->d_seed : nullptr;
    p.qlist = nullptr; p.nq = bt->B;
    if (sparse) SA_TRY(sa_batch_ensure_bloom(bt));
    if (p.pruned && (!bt->state_clean || sparse)) {
        // one launch clears the per-run state -- bound slots / cursors / histograms; only the first run of a batch (or
        // the one after a failed run) needs it, every merge leaves the state zeroed -- and, for dynamic pruning, the
        // lead terms' Bloom filters
        const size_t words = bt->state_clean ? 0 : (size_t)bt->B * (34 + SA_HBINS);
        const size_t bloom8 = sparse ? bt->bloom_bytes / 8 : 0;            // bloom_bytes is a multiple of 1024
        const size_t work = words + bloom8 + 1;
        const u32 grid = work / 256 + 1 < 2048 ? (u32)(work / 256 + 1) : 2048u;
        hipLaunchKernelGGL(sa_k_run_reset, dim3(grid), dim3(256), 0, st, bt->d_slots, (u64)words,
                           (u64*)(sparse ? bt->d_bloom : nullptr), (u64)bloom8, bt->state_clean ? (u32*)nullptr : bt->d_wl_cnt);
    }
    bt->state_clean = false;                           // (until this run's merge is enqueued)
    const u32 slot = bt->ev_n % SA_EVENT_RING;
    SA_HIP(hipEventRecord(bt->ev0[slot], st));
    const u32 n_tiles = bt->kind == 1 ? bt->pn_tiles : ix->n_tiles;
    if (ix->avg_doc_len != 0.f && n_tiles > 0) {
        if (bt->kind == 1) SA_TRY(sa_launch_phrase_tiles(bt, st));
        else {
            if (stage) {
                SA_TRY(sa_launch_stage(bt, p, st));
            } else if (sparse) {
                // candidates first; a resident grid then scans the queries the sparse path gave back
                SA_TRY(sa_launch_sparse(bt, st));
                p.qlist = bt->d_tile_q; p.nq_dev = bt->d_tile_q + bt->B;
                SA_TRY(sa_launch_bm25_list(ix, p, st));
            } else if (group_can_run && p.hist) {
                // Queries that share their first term: the first tiles through the per-query kernel, which
                // establishes every query's bound (k-th best score so far), then one wave per (tile, group).
                // Queries without a group go through the per-query kernel over all tiles.
                u32 warm = std::max<u32>(16u, bt->k / 8u);     // (k = 1000, 10 M docs: 250 / 128 / 64 warm-up tiles -> 1.07 / 1.02 / 1.05 ms per step)
                // the queries start with bounds from their terms' rank tables (p.seed): no warm-up tiles at all -- the grouped
                // kernel finds bounds above the base values from its first item
                // (10 M docs, k = 10: 16 / 4 / 1 / 0 warm-up tiles 0.394 / 0.396 / 0.388 / 0.383 ms of kernels, without the
                //  starting bounds 0.420; 1.25 M-doc shard: 0.096 / 0.094 / - / 0.0825 against 0.126)
                if (p.seed) warm = 0;
                if (sa_opt_is_set(bt->opts.group_warm)) warm = (u32)std::max<long long>(0, bt->opts.group_warm);
                warm = std::min(warm, ix->n_tiles);
                // The ungrouped rows (per-query kernel over all tiles) share nothing with the grouped ones -- not a
                // query, not a counter -- so they run on the side stream BESIDE the warm-up tiles and the grouped kernel
                // (a few dense queries are a small grid of long workgroups: alone on the device they took 0.28 ms of a
                // 0.82 ms step on a batch without shared terms); the merge waits for both.
                bool side = false;
                int rc_side = SA_OK;
                if (bt->n_grouped_rows < bt->B) {
                    Bm25Params pu = p;
                    pu.qlist = bt->d_iota + bt->n_grouped_rows; pu.nq = bt->B - bt->n_grouped_rows;
                    side = sa_opt(bt->opts.group_side, 1) != 0;
                    if (side) {
                        if (!ix->sstream) SA_HIP(hipStreamCreateWithFlags(&ix->sstream, hipStreamNonBlocking));
                        for (int i = 0; i < 2; i++)
                            if (!bt->ev_side[i]) SA_HIP(hipEventCreateWithFlags(&bt->ev_side[i], hipEventDisableTiming));
                        SA_HIP(hipEventRecord(bt->ev_side[0], st));               // (after the reset: the launches above)
                        SA_HIP(hipStreamWaitEvent(ix->sstream, bt->ev_side[0], 0));
                        rc_side = sa_launch_bm25(ix, pu, ix->sstream);
                        SA_HIP(hipEventRecord(bt->ev_side[1], ix->sstream));
                    } else {
                        SA_TRY(sa_launch_bm25(ix, pu, st));
                    }
                }
                // (whatever fails from here on, the index stream still joins the side stream: nothing of this run
                //  may be in flight on a stream that sa_batch_free does not wait for in order)
                int rc_main = rc_side;
                if (rc_main == SA_OK) {
                    Bm25Params pa = p;
                    pa.qlist = bt->d_iota; pa.nq = bt->n_grouped_rows; pa.tile0 = 0; pa.tile_end = warm;
                    rc_main = sa_launch_bm25(ix, pa, st);
                }
                if (rc_main == SA_OK && ix->n_tiles > warm) rc_main = sa_launch_bm25_groups(ix, bt, p, warm, st);
                if (side) SA_HIP(hipStreamWaitEvent(st, bt->ev_side[1], 0));
                SA_TRY(rc_main);
            } else {
                SA_TRY(sa_launch_bm25(ix, p, st));
            }
        }
    } else {
        SA_HIP(hipMemsetAsync(bt->d_cand, 0, (size_t)bt->B * p.cand_cap * sizeof(u64), st));
    }
    SA_HIP(hipEventRecord(bt->ev1[slot], st));
    bt->ev_n++;
    // (with the histogram bound a wave appends all its survivors, so the worst case is not bounded by k)
    const bool may_overflow = p.pruned && (bt->cap_limited || use_hist) && n_tiles > 0;
    if (may_overflow && !defer_check) {
        // the candidate lists are smaller than the worst case: make sure no query ran over
        std::vector<u32> h_cnt(bt->B);
        SA_HIP(hipMemcpyAsync(h_cnt.data(), bt->d_cand_cnt, (size_t)bt->B * sizeof(u32), hipMemcpyDeviceToHost, st));
        SA_HIP(hipStreamSynchronize(st));
        bool over = false;
        for (u32 i = 0; i < bt->B; i++) over |= h_cnt[i] > bt->cand_cap;
        if (over && bt->kind == 1) {
            sa_set_error("phrase batch: candidate list overflow (k too large for this many matching tiles)");
            return SA_ERR_UNSUPPORTED;
        }
        if (over) {
            p.pruned = 0;
            p.cand_per_tile = bt->k;
            p.hist = nullptr; p.gthr = nullptr; p.qlist = nullptr; p.nq = bt->B;
            SA_TRY(sa_launch_bm25(ix, p, st));
        }
    }
    const u32 n_cand = p.pruned ? p.cand_cap : (n_tiles ? n_tiles : 1) * p.cand_per_tile;
    SA_MERGE_LAUNCH(bt->k, bt->B, st, bt->d_cand, n_cand, bt->k, shard_out,
                       (const u32*)bt->d_perm, 0u, (const u32*)(p.pruned ? bt->d_cand_cnt : nullptr),
                       (const u32*)(p.pruned && !p.hist ? bt->d_slots : nullptr),
                       (const u32*)(p.pruned && p.hist ? bt->d_gthr : nullptr),
                       (may_overflow && defer_check) ? (overflow_cell ? overflow_cell : bt->d_overflow) : (u32*)nullptr,
                       bt->d_slots, bt->B, 1u, bt->d_wl_cnt, 0u, (u32*)nullptr, (const u32*)p.seed);
    bt->state_clean = true;
    bt->ran = true;
    return SA_OK;
}

// stage 3: merge the per-rank top-k lists [nranks][B*k (+ extra)] (device memory) into d_final
static int sa_batch_merge_ranks(sa_batch* bt, const u64* d_gathered, int nranks, hipStream_t st, u32 extra = 0) {
    const size_t count = (size_t)bt->B * bt->k;
    if ((u64)nranks * bt->k <= (u64)SA_MERGE_LIST) {
        // the usual case: the gathered keys of a query fit the merge's LDS list -- one launch reads them where they are
        SA_MERGE_LAUNCH(bt->k, bt->B, st, (u64*)d_gathered, (u32)nranks * bt->k, bt->k, bt->d_final,
                           (const u32*)nullptr, bt->k, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr,
                           (u32*)nullptr, (u32*)nullptr, 0u, 0u, (u32*)nullptr, (u32)(count + extra), extra ? bt->d_xflag : (u32*)nullptr,
                           (const u32*)nullptr);
        return SA_OK;
    }
    if (!bt->d_xcand || bt->xcand_ranks < nranks) {
        if (bt->d_xcand) SA_HIP(hipFree(bt->d_xcand));
        bt->d_xcand = nullptr;
        SA_HIP(hipMalloc(&bt->d_xcand, (size_t)nranks * count * sizeof(u64)));
        bt->xcand_ranks = nranks;
    }
    const u64 total = (u64)nranks * count;
    const u32 grid = total / 256 + 1 < 4096 ? (u32)(total / 256 + 1) : 4096;
    hipLaunchKernelGGL(sa_k_regroup, dim3(grid), dim3(256), 0, st, d_gathered, (u32)nranks, bt->B, bt->k, extra, bt->d_xcand,
                       bt->d_xflag);
    // every rank's block is its sorted top-k: group leaders = rank maxima
    SA_MERGE_LAUNCH(bt->k, bt->B, st, bt->d_xcand, (u32)nranks * bt->k, bt->k, bt->d_final,
                       (const u32*)nullptr, bt->k, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr,
                       (u32*)nullptr, (u32*)nullptr, 0u, 0u, (u32*)nullptr, 0u, (u32*)nullptr, (const u32*)nullptr);
    return SA_OK;
}

// A run whose result copy has landed (bt->h_res): was it flagged?  1: a candidate list overflowed (only possible when the
// bound could not rise: degenerate score distributions); 2: fewer than k keys at or above a bound (a starting bound was too
// high).  The batch is then REDONE with the unpruned selection -- against the device tables of the query set that run scored,
// which is why sa_batch_reset / sa_batch_step call this for a run nobody has fetched yet BEFORE they replace those tables
// (round 4 redid at fetch time only: a flagged run followed by reset + fetch returned the new set's results as the old set's).
// Sharded: every rank saw the same flag (it travelled with the all-gather) and every rank makes the same calls, so all of
// them redo the exchange.  Call with the index lock held, after ev_res has fired.
static int sa_batch_redo_if_flagged(sa_batch* bt) {
    sa_index* ix = bt->ix;
    const size_t n = (size_t)bt->B * bt->k;
    const u32 over = (u32)bt->h_res[n + (ix->comm ? 1 : 0)];
    if (!over) return SA_OK;
    if (sa_opt(bt->opts.trace, 0)) fprintf(stderr, "sa_batch: run redone without bounds (flag %u)\n", over);
    SA_HIP(hipStreamSynchronize(bt->st));
    if (ix->xstream) SA_HIP(hipStreamSynchronize(ix->xstream));
    SA_HIP(hipMemset(bt->d_overflow, 0, sizeof(u32)));
    SA_HIP(hipMemset(bt->d_xflag, 0, sizeof(u32)));
    if (bt->d_xlocal) {
        SA_HIP(hipMemset(bt->d_xlocal + n, 0, sizeof(u64)));
        SA_HIP(hipMemset(bt->d_xlocal + 2 * n + 1, 0, sizeof(u64)));
    }
    if (ix->comm) {
        const size_t count = n;
        int nranks = 1;
        SA_TRY(sa_comm_allgather_topk(ix, nullptr, nullptr, 0, &nranks, ix->xstream));
        SA_TRY(sa_batch_run_shard(bt, bt->d_xlocal, false, true));
        SA_HIP(hipStreamSynchronize(bt->st));
        SA_TRY(sa_comm_allgather_topk(ix, bt->d_xlocal, bt->d_gather, count, &nranks, ix->xstream));
        SA_TRY(sa_batch_merge_ranks(bt, bt->d_gather, nranks, ix->xstream));
        SA_HIP(hipStreamSynchronize(ix->xstream));
    } else {
        SA_TRY(sa_batch_run_shard(bt, bt->d_final, false, true));
        SA_HIP(hipStreamSynchronize(bt->st));
    }
    SA_HIP(hipGetLastError());
    SA_HIP(hipMemcpy(bt->h_res, bt->d_final, (n + 2) * sizeof(u64), hipMemcpyDeviceToHost));
    bt->h_res[n] = 0; bt->h_res[n + 1] = 0;                     // (the image now holds the redone results: nothing left to flag)
    return SA_OK;
}

// Results to the host without synchronising a stream: the keys and the two overflow flags behind them are copied into
// the batch's page-locked result buffer and sa_batch_fetch waits for the copy's event only.  ALL result copies of an
// index go through its one exchange stream, behind an event of the stream that wrote the keys: device-to-host copies
// issued on the batches' own streams hold those streams up -- measured with 4 batches in flight on a 1.25 M-doc shard:
// 0.200 ms per step with the copies on the batch streams, 0.113 ms on the exchange stream (SA_RES_XS=0 switches back).
static int sa_batch_queue_result_copy(sa_batch* bt, hipStream_t src_stream) {
    const size_t n = (size_t)bt->B * bt->k + 2;
    sa_index* ix = bt->ix;
    if (sa_opt(bt->opts.res_xs, 1) != 0) {
        if (!ix->xstream) SA_HIP(hipStreamCreateWithFlags(&ix->xstream, hipStreamNonBlocking));
        if (src_stream != ix->xstream) {
            SA_HIP(hipEventRecord(bt->ev_final, src_stream));
            SA_HIP(hipStreamWaitEvent(ix->xstream, bt->ev_final, 0));
            src_stream = ix->xstream;
        }
    }
    SA_HIP(hipMemcpyAsync(bt->h_res, bt->d_final, n * sizeof(u64), hipMemcpyDeviceToHost, src_stream));
    SA_HIP(hipEventRecord(bt->ev_res, src_stream));
    bt->res_pending = true;
    bt->unfetched = true;
    return SA_OK;
}

// sa_batch_run with the index lock held (sa_batch_step fills and runs under ONE lock: no other thread's call on the batch
// or the index lands between the two)
static int sa_batch_run_locked(sa_batch* bt, int sync) {
    sa_index* ix = bt->ix;
    SA_HIP(hipSetDevice(ix->device));
    hipStream_t st = bt->st;
    const u64 t_run = sa_now_ns();
    struct RunClock { sa_batch* b; u64 t0; ~RunClock() { b->host_ns[2] += sa_now_ns() - t0; } } run_clock{bt, t_run};
    if (ix->comm) {
        // Scoring runs on the index stream; the all-gather of the per-shard top-k and the
        // cross-rank merge run on the exchange stream, double-buffered, so they overlap the next
        // run's scoring kernels (the exchange is latency-bound: B*k*8 bytes per rank).  Every rank's block ends
        // with one extra cell, its overflow flag: the ranks learn from the exchange itself whether any of them
        // has to redo the batch unpruned (sa_batch_fetch) -- no separate collective, no host round trip.
        int nranks = 1;
        const size_t count = (size_t)bt->B * bt->k;
        const size_t cell = count + 1;
        hipStream_t xs = ix->xstream;
        SA_TRY(sa_comm_allgather_topk(ix, nullptr, nullptr, 0, &nranks, xs));
        if (!bt->d_xlocal) {
            SA_HIP(hipMalloc(&bt->d_xlocal, 2 * cell * sizeof(u64)));
            SA_HIP(hipMemset(bt->d_xlocal, 0, 2 * cell * sizeof(u64)));
            for (int i = 0; i < 2; i++) {
                SA_HIP(hipEventCreateWithFlags(&bt->ev_scored[i], hipEventDisableTiming));
                SA_HIP(hipEventCreateWithFlags(&bt->ev_exchanged[i], hipEventDisableTiming));
            }
        }
        if (!bt->d_gather || bt->gather_ranks < nranks) {
            SA_HIP(hipStreamSynchronize(xs));
            if (bt->d_gather) SA_HIP(hipFree(bt->d_gather));
            bt->d_gather = nullptr;
            SA_HIP(hipMalloc(&bt->d_gather, 2 * (size_t)nranks * cell * sizeof(u64)));
            bt->gather_ranks = nranks;
        }
        if ((u64)nranks * bt->k > (u64)SA_MERGE_LIST && (!bt->d_xcand || bt->xcand_ranks < nranks))
            SA_HIP(hipStreamSynchronize(xs));                                             // merge_ranks reallocates
        const u32 bsel = bt->xstep & 1;
        bt->xstep++;
        u64* xl = bt->d_xlocal + bsel * cell;
        u64* xg = bt->d_gather + bsel * (size_t)nranks * cell;
        if (bt->exchanged_valid[bsel]) SA_HIP(hipStreamWaitEvent(st, bt->ev_exchanged[bsel], 0));
        SA_TRY(sa_batch_run_shard(bt, xl, true, false, (u32*)(xl + count)));      // (the flag cell: set by the shard merge, sticky until fetch)
        SA_HIP(hipEventRecord(bt->ev_scored[bsel], st));
        SA_HIP(hipStreamWaitEvent(xs, bt->ev_scored[bsel], 0));
        SA_TRY(sa_comm_allgather_topk(ix, xl, xg, cell, &nranks, xs));
        SA_TRY(sa_batch_merge_ranks(bt, xg, nranks, xs, 1u));
        SA_TRY(sa_batch_queue_result_copy(bt, xs));
        SA_HIP(hipEventRecord(bt->ev_exchanged[bsel], xs));
        bt->exchanged_valid[bsel] = true;
        if (sync) SA_HIP(hipStreamSynchronize(xs));
    } else {
        SA_TRY(sa_batch_run_shard(bt, bt->d_final, true));
        SA_TRY(sa_batch_queue_result_copy(bt, st));
    }
    if (sync) {
        SA_HIP(hipStreamSynchronize(st));
        SA_HIP(hipGetLastError());
    }
    return SA_OK;
}

extern "C" int sa_batch_run(sa_batch_t* bt, int sync) {
    SA_ARG(bt && bt->ix, "null batch");
    std::lock_guard<std::mutex> g(bt->ix->mu);
    return sa_batch_run_locked(bt, sync);
}

// One step of a query stream in one call: idf gathered from the index's table, reset, run (header, Part 2) -- under one lock.
extern "C" int sa_batch_step(sa_batch_t* bt, const uint32_t* terms) {
    SA_ARG(bt && bt->ix && terms, "null argument");
    SA_ARG(bt->kind == 0, "sa_batch_step takes a BM25 batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_ARG(ix->h_idf.size() == ix->n_terms, "sa_index_set_idf_table has not been called");
    SA_HIP(hipSetDevice(ix->device));
    const size_t n = (size_t)bt->B * bt->T;
    bt->step_idf.resize(n);
    for (size_t i = 0; i < n; i++) bt->step_idf[i] = terms[i] < ix->n_terms ? ix->h_idf[terms[i]] : 0.f;
    if (bt->res_pending && bt->unfetched) {
        SA_HIP(hipEventSynchronize(bt->ev_res));
        SA_TRY(sa_batch_redo_if_flagged(bt));                   // (a flagged run is redone while the tables still hold ITS query set)
    }
    SA_TRY(sa_batch_fill(bt, terms, bt->step_idf.data()));
    return sa_batch_run_locked(bt, 0);
}

extern "C" int sa_batch_seeds(sa_batch_t* bt, float* out) {
    SA_ARG(bt && bt->ix && out, "null argument");
    SA_ARG(bt->kind == 0, "sa_batch_seeds takes a BM25 batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_HIP(hipStreamSynchronize(bt->st));
    std::vector<u32> rows(bt->B);
    SA_HIP(hipMemcpy(rows.data(), bt->d_seed, (size_t)bt->B * sizeof(u32), hipMemcpyDeviceToHost));
    for (u32 r = 0; r < bt->B; r++) {                           // device row r holds caller query perm[r]
        float f;
        memcpy(&f, &rows[r], 4);
        out[bt->perm[r]] = bt->seed_on ? f : 0.f;
    }
    return SA_OK;
}

extern "C" int sa_batch_host_times(sa_batch_t* bt, uint64_t* out4) {
    SA_ARG(bt && out4, "null argument");
    for (int i = 0; i < 4; i++) out4[i] = bt->host_ns[i];
    return SA_OK;
}

// External-collective variant (the caller owns the exchange, e.g. torch.distributed over RCCL,
// or gloo in the CPU tests): run this shard, hand out its top-k keys, merge gathered keys.
extern "C" int sa_batch_run_local(sa_batch_t* bt, void* local_keys_out_device, int sync) {
    SA_ARG(bt && bt->ix, "null batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    bt->res_pending = false;                           // (the external-collective route: fetch reads d_final, not the page-locked copy of an earlier sa_batch_run)
    SA_TRY(sa_batch_run_shard(bt, bt->d_local, false));
    if (local_keys_out_device)
        SA_HIP(hipMemcpyAsync(local_keys_out_device, bt->d_local, (size_t)bt->B * bt->k * sizeof(u64),
                              hipMemcpyDefault, bt->st));      // (the caller's buffer: device memory, or page-locked host memory)
    if (sync) {
        SA_HIP(hipStreamSynchronize(bt->st));
        SA_HIP(hipGetLastError());
    }
    return SA_OK;
}

extern "C" int sa_batch_merge_gathered(sa_batch_t* bt, const void* gathered_keys_device, int nranks, int sync) {
    SA_ARG(bt && bt->ix && gathered_keys_device, "null argument");
    SA_ARG(nranks >= 1, "nranks < 1");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    bt->res_pending = false;
    SA_TRY(sa_batch_merge_ranks(bt, (const u64*)gathered_keys_device, nranks, bt->st));
    if (sync) {
        SA_HIP(hipStreamSynchronize(bt->st));
        SA_HIP(hipGetLastError());
    }
    return SA_OK;
}

extern "C" int sa_batch_fetch(sa_batch_t* bt, float* scores_out, uint64_t* docs_out) {
    SA_ARG(bt && bt->ix && scores_out && docs_out, "null argument");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    const size_t n = (size_t)bt->B * bt->k;
    const u64* keys = nullptr;
    std::vector<u64> legacy;
    if (bt->res_pending) {
        // the usual route (sa_batch_run): wait for THIS batch's result copy, nothing else -- other batches of the
        // index may be in flight behind it
        SA_HIP(hipEventSynchronize(bt->ev_res));
        bt->unfetched = false;
        SA_TRY(sa_batch_redo_if_flagged(bt));
        keys = bt->h_res;
    } else {
        // external-collective route (sa_batch_run_local / sa_batch_merge_gathered): the caller's last call decides
        legacy.resize(n);
        SA_HIP(hipStreamSynchronize(bt->st));
        if (ix->xstream) SA_HIP(hipStreamSynchronize(ix->xstream));
        SA_HIP(hipGetLastError());
        SA_HIP(hipMemcpy(legacy.data(), bt->d_final, n * sizeof(u64), hipMemcpyDeviceToHost));
        keys = legacy.data();
    }
    for (u32 r = 0; r < bt->B; r++) {
        const u32 qi = r;                        // results are stored in caller order
        for (u32 j = 0; j < bt->k; j++) {
            const u64 key = keys[(size_t)r * bt->k + j];
            const u32 sb = (u32)(key >> 32);
            float s;
            memcpy(&s, &sb, 4);
            scores_out[(size_t)qi * bt->k + j] = key ? s : 0.f;
            docs_out[(size_t)qi * bt->k + j] = key ? (u64)(u32)(~(u32)(key & 0xFFFFFFFFull)) : SA_NO_DOC;
        }
    }
    return SA_OK;
}

// Diagnostics: switch on per-query counting of the candidate docs the sparse path scores (an extra
// atomic per candidate, so not for timed runs) and read the totals of the runs since the previous call.
extern "C" int sa_batch_stats(sa_batch_t* bt, int enable, uint64_t* sparse_candidates_out, uint64_t* sparse_queries_out) {
    SA_ARG(bt && bt->ix, "null batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_HIP(hipStreamSynchronize(bt->st));
    u64 total = 0, nq = 0;
    if (bt->d_stats) {
        std::vector<u32> h(bt->B);
        SA_HIP(hipMemcpy(h.data(), bt->d_stats, (size_t)bt->B * sizeof(u32), hipMemcpyDeviceToHost));
        for (u32 v : h) total += v;
        SA_HIP(hipMemset(bt->d_stats, 0, (size_t)bt->B * sizeof(u32)));
    }
    if (bt->d_route) {
        std::vector<u32> h(bt->B);
        SA_HIP(hipMemcpy(h.data(), bt->d_route, (size_t)bt->B * sizeof(u32), hipMemcpyDeviceToHost));
        for (u32 v : h) nq += v == 0 ? 1 : 0;
    }
    if (enable && !bt->d_stats) {
        SA_HIP(hipMalloc(&bt->d_stats, (size_t)bt->B * sizeof(u32)));
        SA_HIP(hipMemset(bt->d_stats, 0, (size_t)bt->B * sizeof(u32)));
    } else if (!enable && bt->d_stats) {
        SA_HIP(hipFree(bt->d_stats));
        bt->d_stats = nullptr;
    }
    if (sparse_candidates_out) *sparse_candidates_out = total;
    if (sparse_queries_out) *sparse_queries_out = nq;
    return SA_OK;
}

extern "C" int sa_batch_profile(sa_batch_t* bt, double* kernel_ms_out, uint64_t* alg_bytes_out,
                                uint64_t* postings_bytes_out) {
    SA_ARG(bt && bt->ix, "null batch");
    SA_ARG(bt->ran && bt->ev_n > 0, "batch has not been run since the last profile call");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_HIP(hipStreamSynchronize(bt->st));
    // mean over the runs since the previous call (at most the last SA_EVENT_RING of them)
    const u32 n = bt->ev_n < SA_EVENT_RING ? bt->ev_n : SA_EVENT_RING;
    double sum = 0.0;
    for (u32 i = 0; i < n; i++) {
        float ms = 0.f;
        SA_HIP(hipEventElapsedTime(&ms, bt->ev0[i], bt->ev1[i]));
        sum += ms;
    }
    bt->ev_n = 0;
    if (kernel_ms_out) *kernel_ms_out = n ? sum / n : 0.0;
    if (alg_bytes_out) *alg_bytes_out = bt->alg_bytes;
    if (postings_bytes_out) *postings_bytes_out = bt->postings_bytes;
    return SA_OK;
}

extern "C" int sa_batch_last_route(sa_batch_t* bt, int* pruned_out) {
    SA_ARG(bt && bt->ix && pruned_out, "null argument");
    std::lock_guard<std::mutex> g(bt->ix->mu);
    *pruned_out = bt->last_route_stage ? 2 : bt->last_route_sparse ? 1 : 0;
    return SA_OK;
}

extern "C" int sa_batch_group_info(sa_batch_t* bt, uint32_t out[4]) {
    SA_ARG(bt && out, "null argument");
    out[0] = bt->n_groups; out[1] = bt->n_grouped_rows; out[2] = bt->n_shared_rows; out[3] = bt->B - bt->n_grouped_rows;
    return SA_OK;
}

// A batch's switches after its creation: effective from its next reset / run (what was sized at creation -- candidate
// capacity, streams, the phrase tile size -- keeps what it was).
extern "C" int sa_batch_set_options(sa_batch_t* bt, const sa_options_t* o) {
    SA_ARG(bt && bt->ix && o, "null argument");
    SA_ARG(o->struct_size == sizeof(sa_options_t), "sa_options_t: wrong struct_size (fill it with sa_options_init)");
    std::lock_guard<std::mutex> g(bt->ix->mu);
    bt->opts = *o;
    return SA_OK;
}
extern "C" int sa_batch_get_options(sa_batch_t* bt, sa_options_t* out) {
    SA_ARG(bt && bt->ix && out, "null argument");
    std::lock_guard<std::mutex> g(bt->ix->mu);
    *out = bt->opts;
    return SA_OK;
}

// Diagnostics: the rank table of a term in this batch's impact stream (22 lower bounds of its r-th largest factor, sa_topf_ranks) and its
// exact largest factor -- what the starting bounds are formed from.  Tests compare the tables of differently built streams.
extern "C" int sa_batch_debug_rank_table(sa_batch_t* bt, uint32_t term, float* ranks22_out, float* maxf_out) {
    SA_ARG(bt && bt->ix && ranks22_out && maxf_out, "null argument");
    std::lock_guard<std::mutex> g(bt->ix->mu);
    sa_impacts* im = bt->impacts.get();
    if (!im || term >= bt->ix->n_terms || im->h_topf.size() != (size_t)bt->ix->n_terms * SA_TOPF_NR || im->h_maxf.size() != bt->ix->n_terms) {
        sa_set_error("sa_batch_debug_rank_table: the batch has no rank tables (no impact stream, term_seed = 0, or an unknown term)");
        return SA_ERR_STATE;
    }
    memcpy(ranks22_out, &im->h_topf[(size_t)term * SA_TOPF_NR], SA_TOPF_NR * sizeof(float));
    *maxf_out = im->h_maxf[term];
    return SA_OK;
}

extern "C" int sa_batch_destroy(sa_batch_t* bt) {
    if (!bt) return SA_OK;
    // (under the index lock: a dense call on another thread swaps ix->stream / scratch to its lane while it holds the lock --
    //  sa_batch_free must synchronise the index's own stream, not a lane's)
    if (bt->ix) { std::lock_guard<std::mutex> g(bt->ix->mu); sa_batch_free(bt); }
    else sa_batch_free(bt);
    return SA_OK;
}

// sa_sparse.hip -- dynamic pruning for BM25 top-k batches: score only the docs that can still win.
//
// The tile kernel (sa_bm25.hip) is exhaustive: every posting of every query term is scored, like the
// reference (`np.sum([arr.score(t) for t in q])` + argpartition, reference test_msmarco.py:353,
// utils/sort.py:24).  For top-k that is far more than needed.  With G a score that at least k docs
// are known to reach, a doc containing only "non-essential" terms -- the lowest-idf terms whose idf
// sum stays below G; tf/(tf+norm) <= 1, so a term never adds more than its idf -- cannot enter the
// top-k (MaxScore, Turtle & Flood 1995).  The frequent terms are exactly the low-idf ones, so the long
// posting lists never have to be read:
//
//   phase 1   every doc of the query's LEAD term (highest idf = rarest) is scored against all query
//             terms and counted in the query's score histogram      -> G1, a bound from real docs
//   routing   essential terms = those not covered by G1.  Few essential postings: sparse route.
//             Otherwise (all terms frequent, or G1 too low) the query's state is reset and the tile
//             kernel scans it exhaustively.
//   phase 2   every posting of the other essential terms is a candidate doc; it is scored by the
//             thread that finds it unless a higher-priority essential term also occurs in the doc
//             (then that term's posting owns it), and kept if it reaches G1.
//
// Scoring a candidate = one lookup per query term, in QUERY ORDER with the reference's operation
// order, so scores are bit-identical to the exhaustive path: a frequent term is one byte load from
// its dense tf row (sa_index.hpp) plus the doc length; other terms are a lower-bound search inside the
// doc's tile slice (slice table of the batch).  Integer / scalar-fp32 work, latency-bound gathers;
// no LDS, no MFMA.  The merge (sa_k_topk_merge) is unchanged.
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "../../include/searcharray_hip.h"
#include <stdlib.h>

#define SA_NO_LEAD 0xFFFFFFFFu
// Per-query Bloom filter (one hash) of the lead term's docs: one-byte cells, so phase 1 fills it with
// plain stores (setting a bit would take an atomic per doc); 8..16 cells per lead doc (sized at
// batch creation), i.e. 6..12 % false positives.

struct SparseParams {
    const u64* tfp;
    const float* doc_lens;
    const unsigned char* tf8;
    const u32* tfbits;     // presence bitmaps of the dense-row terms
    u64 tfbits_words;
    const u32* tf8_slot;
    u32 n_terms, n_tiles, tile_docs;
    u64 n_docs, doc_base;
    int dl_packed;
    const u32* terms;      // [B][T]
    const float* idf;      // [B][T]
    const u32* bounds;     // [B][T][n_tiles+1]
    const u64* qbase;      // [B][T]
    u32 B, T, k;
    float k1, b, avgdl;
    const float* ub;       // [B][T+1]
    const u32* ub_order;   // [B][T]
    const u32* qdf;        // [B][T] postings of each query term in this shard
    const u32* qrow8;      // [B][T] dense tf row of each query term, or SA_DD_NONE
    const u32* lead;       // [B] query-term index of the lead term, or SA_NO_LEAD (scan the tiles)
    u32 chunk1;            // postings per lead-phase work item
    const u64* p1_off;     // [B+1] prefix sums of the lead terms' work items (chunks of SA_SP_CHUNK_LEAD postings)
    u32* route;            // [B] 0: sparse, 1: tiles
    u32* emask;            // [B] essential query terms (bit t)
    u64* p2_off;           // [B+1] prefix sums of the phase-2 work items
    u32* tile_q;           // [B] out: queries the tile kernel must scan
    u32* tile_cnt;         // out: their number
    u64 limit2;            // most phase-2 candidates a sparse-route query may have
    u32* hist;
    u32* gthr;
    u64* cand;
    u32 cand_cap;
    u32* cand_cnt;
    unsigned char* bloom;  // Bloom filters of the lead docs, query q at bloom_off[q], 2^(32 - bloom_shift[q]) cells
    const u64* bloom_off;  // [B]
    const u32* bloom_shift;// [B]
    u32 tile_shift;        // log2(tile_docs)
    u32* stats;
    // phase-2 survivors of the bound check, scored by their own kernel (no divergence with the filter)
    u64* surv;             // [surv_cap][2]: doc << 32 | q,  tc << 40 | tf << 20 | dl  (fits: 18-bit tf / dl)
    u32 surv_cap;
    u32* surv_cnt;
};

__device__ __forceinline__ u64 sa_sp_df(const SparseParams& p, u32 q, u32 t) { return p.qdf[q * p.T + t]; }

__device__ __forceinline__ u32 sa_bloom_cell(u64 doc, u32 shift) { return ((u32)doc * 2654435761u) >> shift; }

// tf (and the posting's doc-length field) of query term t in `doc`; 0 when the doc lacks the term.
// Frequent terms: one byte of their dense tf row.  Others: a lower-bound search of the doc's tile slice.
__device__ __forceinline__ u32 sa_sparse_tf(const SparseParams& p, u32 qt, u32 term, u64 doc, u32 tile, u32& dli) {
    const u32 row8 = p.qrow8[qt];
    if (row8 != SA_DD_NONE) {
        const u32 tfi = p.tf8[(u64)row8 * p.n_docs + doc];
        if (tfi != 255u) {                                          // 255: saturated byte, take the posting
            dli = (tfi && p.dl_packed) ? (u32)p.doc_lens[doc] : 0u;
            return tfi;
        }
    }
    const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
    const u32 lo = row[0], hi = row[1];
    const u64* a = p.tfp + p.qbase[qt];
    const u32 j = sa_lower_bound(a, lo, hi, doc << SA_KEY_SHIFT, SA_KEY_MASK);
    if (j < hi) {
        const u64 x = a[j];
        if ((x >> SA_KEY_SHIFT) == doc) { dli = (u32)((x >> SA_LSB_BITS) & SA_LSB_MASK); return (u32)(x & SA_LSB_MASK); }
    }
    dli = 0;
    return 0;
}

// presence only (no doc length needed)
__device__ __forceinline__ bool sa_sparse_has(const SparseParams& p, u32 qt, u32 term, u64 doc, u32 tile) {
    const u32 row8 = p.qrow8[qt];
    if (row8 != SA_DD_NONE) return (p.tfbits[(u64)row8 * p.tfbits_words + (doc >> 5)] >> (doc & 31u)) & 1u;
    const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
    const u32 lo = row[0], hi = row[1];
    const u64* a = p.tfp + p.qbase[qt];
    const u32 j = sa_lower_bound(a, lo, hi, doc << SA_KEY_SHIFT, SA_KEY_MASK);
    return j < hi && (a[j] >> SA_KEY_SHIFT) == doc;
}

__device__ __forceinline__ float sa_sparse_term_score(const SparseParams& p, u32 tfi, u32 dli, u64 doc, float idf) {
    const float one_minus_b = 1.0f - p.b;
    const float tf = (float)tfi;
    const float dl = p.dl_packed ? (float)dli : p.doc_lens[doc];
    const float norm = __fmul_rn(p.k1, __fadd_rn(one_minus_b, __fmul_rn(p.b, __fdiv_rn(dl, p.avgdl))));
    const float sat = __fdiv_rn(tf, __fadd_rn(tf, norm));
    return __fmul_rn(sat, idf);
}

// BM25 of `doc` for query q: sum over the query terms in query order (bm25.pyx:19-23 per term,
// every op rounded to fp32).  The term `known_t` (the one whose posting produced the candidate)
// comes with its tf / doc length.
__device__ __forceinline__ float sa_sparse_score(const SparseParams& p, u32 q, u64 doc, u32 known_t, u32 known_tf,
                                                 u32 known_dl) {
    const u32 tile = (u32)(doc >> p.tile_shift);
    float s = 0.f;
    if (p.T <= 4 && p.dl_packed) {
        // Up to four terms (the common query shape): fetch every term's tf FIRST -- the probes are
        // independent random accesses (dense tf rows, tile slices), mostly L2 misses, and issued back to
        // back they overlap instead of paying one memory round trip per term.  The doc length is the
        // doc's, not the term's: the known posting carries it, no doc_lens load.
        u32 tfv[4];
        bool look[4];                                               // term present in the query and not the known one
#pragma unroll
        for (u32 t = 0; t < 4; t++) {                               // pass 1: the dense-row bytes, no dependent branches
            tfv[t] = 0; look[t] = false;
            if (t >= p.T) continue;
            const u32 qt = q * p.T + t;
            if (p.terms[qt] >= p.n_terms) continue;
            if (t == known_t) { tfv[t] = known_tf; continue; }
            look[t] = true;
            const u32 row8 = p.qrow8[qt];
            tfv[t] = row8 != SA_DD_NONE ? (u32)p.tf8[(u64)row8 * p.n_docs + doc] : 255u;
        }
#pragma unroll
        for (u32 t = 0; t < 4; t++) {                               // pass 2: terms without a row / saturated bytes
            if (look[t] && tfv[t] == 255u) {
                u32 dl_unused;
                tfv[t] = sa_sparse_tf(p, q * p.T + t, p.terms[q * p.T + t], doc, tile, dl_unused);
            }
        }
#pragma unroll
        for (u32 t = 0; t < 4; t++) {
            if (t >= p.T || tfv[t] == 0) continue;
            s = __fadd_rn(s, sa_sparse_term_score(p, tfv[t], known_dl, doc, p.idf[q * p.T + t]));
        }
        return s;
    }
    for (u32 t = 0; t < p.T; t++) {
        const u32 qt = q * p.T + t;
        const u32 term = p.terms[qt];
        if (term >= p.n_terms) continue;
        u32 tfi, dli;
        if (t == known_t) { tfi = known_tf; dli = known_dl; }
        else tfi = sa_sparse_tf(p, qt, term, doc, tile, dli);
        if (tfi == 0) continue;
        s = __fadd_rn(s, sa_sparse_term_score(p, tfi, dli, doc, p.idf[qt]));
    }
    return s;
}

// Keep a scored doc: count it in the query's histogram and append it to the candidate list.  Lanes
// of a wave mostly work for the same query, so the list cursor is bumped once per wave when they do.
__device__ __forceinline__ void sa_sparse_keep(const SparseParams& p, u32 q, u64 doc, float s, u32 thr, bool active) {
    const u32 x = __float_as_uint(s);
    const bool keep = active && x >= thr;
    const u64 kmask = __ballot(keep);
    if (kmask == 0) return;                                         // wave-uniform
    {
        // lanes of a wave mostly hit the same few bins of the same query: one atomic per distinct bin
        const u32 key = keep ? q * (u32)SA_HBINS + sa_score_bin(x) : 0xFFFFFFFFu;
        u64 todo = kmask;
        while (todo) {                                              // wave-uniform
            const u32 src = (u32)__ffsll((long long)todo) - 1u;
            const u32 k0 = (u32)__shfl((int)key, (int)src, SA_WAVE);
            const u64 same = __ballot(key == k0);
            if (sa_lane() == src) atomicAdd(&p.hist[k0], (u32)__popcll(same));
            todo &= ~same;
        }
    }
    const u32 q0 = (u32)__builtin_amdgcn_readfirstlane((int)(keep ? q : 0xFFFFFFFFu));
    u32 pos;
    if (__ballot(keep && q != q0) == 0 && q0 != 0xFFFFFFFFu) {
        // every keeping lane has the query of the first active lane (which keeps): one atomic
        const u32 lane = sa_lane();
        u32 base = 0;
        const u32 first = (u32)__ffsll((long long)kmask) - 1u;
        if (lane == first) base = atomicAdd(&p.cand_cnt[q], (u32)__popcll(kmask));
        base = (u32)__shfl((int)base, (int)first, SA_WAVE);
        pos = base + (u32)__popcll(kmask & ((1ull << lane) - 1ull));
    } else {
        pos = keep ? atomicAdd(&p.cand_cnt[q], 1u) : 0u;
    }
    if (keep && pos < p.cand_cap) p.cand[(u64)q * p.cand_cap + pos] = ((u64)x << 32) | (u64)(u32)(~(u32)(p.doc_base + doc));
}


// largest q with off[q] <= g
__device__ __forceinline__ u32 sa_sp_find(const u64* __restrict__ off, u32 n, u64 g) {
    u32 lo = 0, hi = n;
    while (hi - lo > 1) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// phase 1: work items = chunks of the lead terms' postings; everything that depends on the query only
// is uniform across the workgroup.  Every lead doc is scored; it is kept (counted + appended) if it
// reaches the query's bound so far, which the first wave refreshes after every item.
__global__ void __launch_bounds__(256) sa_k_sparse_lead(const SparseParams p) {
    const u64 n_items = p.p1_off[p.B];
    for (u64 item = blockIdx.x; item < n_items; item += gridDim.x) {
        const u32 q = sa_sp_find(p.p1_off, p.B, item);
        const u32 lt = p.lead[q];
        const u32 qt = q * p.T + lt;
        const u64 start = (item - p.p1_off[q]) * p.chunk1;
        const u64 df = p.qdf[qt];
        const u32 n = df - start < (u64)p.chunk1 ? (u32)(df - start) : p.chunk1;
        const u64* post = p.tfp + p.qbase[qt] + start;
        const u32 gq = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 thr = gq > 1u ? gq : 1u;
        unsigned char* bloom = p.bloom + p.bloom_off[q];
        const u32 bshift = p.bloom_shift[q];
        for (u32 i0 = 0; i0 < n; i0 += 256) {                       // uniform trip count
            const u32 i = i0 + threadIdx.x;
            const bool active = i < n;
            u64 doc = 0;
            float s = 0.f;
            if (active) {
                const u64 x = post[i];
                doc = x >> SA_KEY_SHIFT;
                bloom[sa_bloom_cell(doc, bshift)] = 1;
                s = sa_sparse_score(p, q, doc, lt, (u32)(x & SA_LSB_MASK), (u32)((x >> SA_LSB_BITS) & SA_LSB_MASK));
            }
            sa_sparse_keep(p, q, doc, s, thr, active);
        }
        if (p.stats && threadIdx.x == 0) atomicAdd(&p.stats[q], n);
        __syncthreads();                                             // this item's counts are out
        if (threadIdx.x < SA_WAVE) sa_hist_refresh(p.hist + (u64)q * SA_HBINS, p.gthr + q, p.k, threadIdx.x);
    }
}

// routing: one wave per query -- bound from the histogram, essential terms, phase-2 work
__global__ void __launch_bounds__(256) sa_k_sparse_route(const SparseParams p) {
    const u32 lane = threadIdx.x & (SA_WAVE - 1);
    const u32 q = blockIdx.x * (blockDim.x / SA_WAVE) + threadIdx.x / SA_WAVE;
    if (q >= p.B) return;                                             // wave-uniform
    const u32 lt = p.lead[q];
    u32 g = 0;
    if (lt != SA_NO_LEAD) g = sa_hist_refresh(p.hist + (u64)q * SA_HBINS, p.gthr + q, p.k, lane);
    const float gf = __uint_as_float(g);
    const float* ubq = p.ub + (u64)q * (p.T + 1);
    u32 ne = 0;
    for (u32 j = 1; j <= p.T; j++) if (ubq[j] < gf) ne = j;           // ub ascending in j
    u32 em = p.T >= 32 ? 0xFFFFFFFFu : ((1u << p.T) - 1u);
    for (u32 j = 0; j < ne; j++) em &= ~(1u << p.ub_order[q * p.T + j]);
    u64 c2 = 0, items = 0;
    for (u32 t = 0; t < p.T; t++)
        if (((em >> t) & 1u) && t != lt && p.terms[q * p.T + t] < p.n_terms) {
            const u64 df = sa_sp_df(p, q, t);
            c2 += df;
            items += (df + SA_SP_CHUNK - 1) / SA_SP_CHUNK;
        }
    // (no bound yet: phase 2 would keep every candidate -- only if they fit the list)
    const bool sparse = lt != SA_NO_LEAD && c2 <= p.limit2 && (g != 0 || c2 <= (u64)p.cand_cap / 2);
    if (!sparse) {
        // the tiles will count every doc of this query again: forget phase 1
        for (u32 i = lane; i < (u32)SA_HBINS; i += SA_WAVE) p.hist[(u64)q * SA_HBINS + i] = 0;
        items = 0;
    }
    if (lane == 0) {
        p.route[q] = sparse ? 0u : 1u;
        p.emask[q] = em;
        p.p2_off[q + 1] = items;                                      // work items, scanned by sa_k_sparse_scan
        if (!sparse) { p.cand_cnt[q] = 0; p.gthr[q] = 0; }
    }
}

// prefix sums of the phase-2 work items and the list of queries left to the tile kernel: one wave
__global__ void __launch_bounds__(64) sa_k_sparse_scan(const SparseParams p) {
    if (blockIdx.x != 0) return;
    const u32 lane = threadIdx.x;
    u64 carry = 0;
    u32 nt = 0;
    for (u32 base = 0; base < p.B; base += SA_WAVE) {                 // uniform
        const u32 q = base + lane;
        const u64 v = q < p.B ? p.p2_off[q + 1] : 0;
        const bool scan_it = q < p.B && p.route[q] != 0u;
        u64 inc = v;                                                  // inclusive prefix over the wave
#pragma unroll
        for (int o = 1; o < SA_WAVE; o <<= 1) {
            const u64 up = __shfl_up(inc, (unsigned)o, SA_WAVE);
            if (lane >= (u32)o) inc += up;
        }
        if (q < p.B) p.p2_off[q + 1] = carry + inc;
        carry += __shfl(inc, SA_WAVE - 1, SA_WAVE);
        const u64 m = __ballot(scan_it);
        if (scan_it) p.tile_q[nt + (u32)__popcll(m & ((1ull << lane) - 1ull))] = q;
        nt += (u32)__popcll(m);
    }
    if (lane == 0) { p.p2_off[0] = 0; *p.tile_cnt = nt; *p.surv_cnt = 0; }
}

// phase 2: work items = chunks of the non-lead essential terms' postings of the sparse-route queries.
// Most candidates are dismissed without being scored: the doc is skipped if a higher-priority
// essential term owns it, and as soon as what it has plus everything it could still get (the idf of
// the terms not yet probed, highest first) falls short of the bound.  Survivors go to a list.
__global__ void __launch_bounds__(256) sa_k_sparse_rest(const SparseParams p) {
    __shared__ u64 s_surv[2 * SA_SP_CHUNK];
    __shared__ u32 s_n, s_base;
    const u64 n_items = p.p2_off[p.B];
    for (u64 item = blockIdx.x; item < n_items; item += gridDim.x) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        // ---- uniform per item
        const u32 q = sa_sp_find(p.p2_off, p.B, item);
        u64 r = item - p.p2_off[q];
        const u32 em = p.emask[q], lt = p.lead[q];
        u32 tc = 0;
        for (; tc < p.T; tc++) {
            if (!((em >> tc) & 1u) || tc == lt || p.terms[q * p.T + tc] >= p.n_terms) continue;
            const u64 nch = ((u64)p.qdf[q * p.T + tc] + SA_SP_CHUNK - 1) / SA_SP_CHUNK;
            if (r < nch) break;
            r -= nch;
        }
        const u32 qt = q * p.T + tc;
        const u64 start = r * SA_SP_CHUNK;
        const u64 df = p.qdf[qt];
        const u32 n = df - start < (u64)SA_SP_CHUNK ? (u32)(df - start) : (u32)SA_SP_CHUNK;
        const u64* post = p.tfp + p.qbase[qt] + start;
        const u32 gq = p.gthr[q];
        const u32 thr = gq > 1u ? gq : 1u;
        const float gf = __uint_as_float(thr);
        const unsigned char* bloom = p.bloom + p.bloom_off[q];
        const u32 bshift = p.bloom_shift[q];
        const float idf_c = p.idf[qt];
        const float rest0 = p.ub[(u64)q * (p.T + 1) + p.T] - idf_c;
        // ---- per posting
        for (u32 i0 = 0; i0 < n; i0 += 256) {                       // uniform trip count
            const u32 i = i0 + threadIdx.x;
            bool alive = i < n;
            u64 doc = 0;
            u32 ktf = 0, kdl = 0;
            if (alive) {
                const u64 x = post[i];
                doc = x >> SA_KEY_SHIFT;
                ktf = (u32)(x & SA_LSB_MASK); kdl = (u32)((x >> SA_LSB_BITS) & SA_LSB_MASK);
                const u32 tile = (u32)(doc >> p.tile_shift);
                // what the doc has for sure, and the most the other terms could add
                float have = sa_sparse_term_score(p, ktf, kdl, doc, idf_c);
                float rest = rest0;
                // probe the other terms from the most valuable down (ub_order is ascending in idf)
                for (int j = (int)p.T - 1; j >= 0 && alive; j--) {
                    const u32 t = p.ub_order[q * p.T + (u32)j];
                    if (t == tc) continue;
                    const u32 qt2 = q * p.T + t;
                    const u32 term = p.terms[qt2];
                    if (term >= p.n_terms) continue;
                    if ((have + rest) * 1.00001f < gf) { alive = false; break; }
                    bool has;
                    if (t == lt) {
                        // the lead's docs are in the query's Bloom filter: most candidates are cleared by one
                        // load, the rest by the exact search
                        has = bloom[sa_bloom_cell(doc, bshift)] != 0 && sa_sparse_has(p, qt2, term, doc, tile);
                    } else {
                        has = sa_sparse_has(p, qt2, term, doc, tile);
                    }
                    const float w = p.idf[qt2];
                    rest -= w;
                    if (has) {
                        // the doc belongs to its lead posting, else to its lowest-numbered essential term
                        if (((em >> t) & 1u) && (t == lt || t < tc)) { alive = false; break; }
                        have += w;
                    }
                }
                if (alive && (have + rest) * 1.00001f < gf) alive = false;
            }
            // survivors wait in LDS until the item is done
            if (alive) {
                const u32 pos = atomicAdd(&s_n, 1u);
                s_surv[2 * pos] = (doc << 32) | (u64)q;
                s_surv[2 * pos + 1] = ((u64)tc << 40) | ((u64)ktf << 20) | (u64)kdl;
            }
        }
        // The item's survivors go to the scoring kernel's list (scoring them here would stall whole
        // waves behind a few lanes' lookups): ONE global cursor bump per item.  A full list is scored
        // in place.
        __syncthreads();
        const u32 ns = s_n;
        if (ns) {                                                    // uniform
            if (threadIdx.x == 0) s_base = atomicAdd(p.surv_cnt, ns);
            __syncthreads();
            const u32 base = s_base;
            const u32 n_fit = base >= p.surv_cap ? 0u : (p.surv_cap - base < ns ? p.surv_cap - base : ns);
            for (u32 i = threadIdx.x; i < 2 * n_fit; i += 256) p.surv[2 * (u64)base + i] = s_surv[i];
            for (u32 i0 = n_fit; i0 < ns; i0 += 256) {               // uniform trip count; list full
                const u32 i = i0 + threadIdx.x;
                const bool active = i < ns;
                u64 doc = 0;
                float s = 0.f;
                if (active) {
                    const u64 b2 = s_surv[2 * i + 1];
                    doc = s_surv[2 * i] >> 32;
                    s = sa_sparse_score(p, q, doc, (u32)(b2 >> 40), (u32)((b2 >> 20) & 0xFFFFFu), (u32)(b2 & 0xFFFFFu));
                }
                sa_sparse_keep(p, q, doc, s, thr, active);
            }
        }
        __syncthreads();                                             // s_n / s_surv are reused by the next item
        if (p.stats && threadIdx.x == 0) atomicAdd(&p.stats[q], n);
    }
}

// phase 2b: exact scores of the survivors
__global__ void __launch_bounds__(256) sa_k_sparse_score(const SparseParams p) {
    const u32 have = *p.surv_cnt;
    const u64 total = have < p.surv_cap ? have : p.surv_cap;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 g0 = (u64)blockIdx.x * blockDim.x; g0 < total; g0 += stride) {       // wave-uniform trip count
        const u64 g = g0 + threadIdx.x;
        const bool active = g < total;
        u32 q = 0, thr = 1;
        u64 doc = 0;
        float s = 0.f;
        if (active) {
            const u64 a = p.surv[2 * g], b = p.surv[2 * g + 1];
            q = (u32)(a & 0xFFFFFFFFull);
            doc = a >> 32;
            const u32 gq = p.gthr[q];
            thr = gq > 1u ? gq : 1u;
            s = sa_sparse_score(p, q, doc, (u32)(b >> 40), (u32)((b >> 20) & 0xFFFFFu), (u32)(b & 0xFFFFFu));
        }
        sa_sparse_keep(p, q, doc, s, thr, active);
    }
}

int sa_launch_sparse(sa_batch* bt, hipStream_t st) {
    sa_index* ix = bt->ix;
    SparseParams p;
    memset(&p, 0, sizeof(p));
    p.tfp = ix->d_tfp; p.doc_lens = ix->d_doc_lens; p.tf8 = ix->d_tf8; p.tfbits = ix->d_tfbits; p.tfbits_words = ix->tfbits_words;
    p.tf8_slot = ix->n_tf8_terms ? ix->d_tf8_slot : nullptr;
    p.n_terms = ix->n_terms; p.n_tiles = ix->n_tiles; p.tile_docs = ix->tile_docs;
    p.n_docs = ix->n_docs; p.doc_base = ix->doc_base; p.dl_packed = ix->dl_packed ? 1 : 0;
    p.terms = bt->d_terms; p.idf = bt->d_idf; p.bounds = bt->d_bounds; p.qbase = bt->d_qbase;
    p.B = bt->B; p.T = bt->T; p.k = bt->k; p.k1 = bt->k1; p.b = bt->b; p.avgdl = ix->avg_doc_len;
    p.ub = bt->d_ub; p.ub_order = bt->d_ub_order; p.lead = bt->d_lead; p.p1_off = bt->d_p1_off;
    p.route = bt->d_route; p.emask = bt->d_emask; p.p2_off = bt->d_p2_off; p.limit2 = bt->sparse_limit2;
    p.tile_q = bt->d_tile_q; p.tile_cnt = bt->d_tile_q + bt->B;
    p.hist = bt->d_hist; p.gthr = bt->d_gthr; p.cand = bt->d_cand; p.cand_cap = bt->cand_cap; p.cand_cnt = bt->d_cand_cnt;
    p.stats = bt->d_stats;
    p.chunk1 = bt->sparse_chunk1;
    p.bloom = (unsigned char*)bt->d_bloom; p.bloom_off = bt->d_bloom_off; p.bloom_shift = bt->d_bloom_shift;
    p.tile_shift = 0;
    while ((1u << p.tile_shift) < ix->tile_docs) p.tile_shift++;
    // (the Bloom filters were cleared by sa_k_run_reset together with the bound state)
    p.qdf = bt->d_qdf; p.qrow8 = bt->d_qrow8;
    p.surv = bt->d_surv; p.surv_cap = bt->surv_cap; p.surv_cnt = bt->d_tile_q + bt->B + 1;
    const u64 n1 = bt->sparse_p1_total;                            // work items of phase 1
    if (n1) {
        const u32 grid = n1 < 16384 ? (u32)n1 : 16384u;
        hipLaunchKernelGGL(sa_k_sparse_lead, dim3(grid), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(sa_k_sparse_route, dim3((bt->B + 3) / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sa_k_sparse_scan, dim3(1), dim3(64), 0, st, p);
    // (the phase-2 item count lives on the device; the grid is sized by its static upper bound)
    const u64 n2 = bt->sparse_p2_max;
    const u32 grid2 = n2 < 16384 ? (u32)(n2 ? n2 : 1) : 16384u;
    hipLaunchKernelGGL(sa_k_sparse_rest, dim3(grid2), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sa_k_sparse_score, dim3(grid2 < 2048 ? grid2 : 2048), dim3(256), 0, st, p);
    return SA_OK;
}

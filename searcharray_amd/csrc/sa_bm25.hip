// sa_bm25.hip -- term-at-a-time BM25 over the HBM-resident TF postings, with per-tile and
// per-query top-k.
//
// Replaces the reference's per-term dense pipeline
//     as_dense (scatter, roaringish_ops.pyx:84-98)  ->  _bm25_score (bm25.pyx:11-25, O(N) per term)
//     ->  np.sum(axis=0) (test/test_msmarco.py:353-354)  ->  np.argpartition (utils/sort.py:24)
// with one sparse pass: a workgroup owns (query, doc tile); the tile's fp32 score accumulators
// live in LDS; each query term's slice of the fat posting stream (doc|doc_len|tf in one u64) is
// read once with coalesced 64-bit loads, scored, and added into LDS in QUERY-TERM ORDER (a barrier
// separates terms) so the fp32 sum is bit-identical to the reference's ((s0+s1)+s2)+s3.
// Per-posting arithmetic is the reference's, op for op, each rounded to fp32 (no FMA contraction,
// IEEE division):  tf / (tf + k1 * ((1 - b) + b * (dl / avgdl))) * idf.
//
// Roofline: HBM-bound integer/bitwise + scalar fp32 work (no MFMA).  Algorithmic bytes per query
// = sum_t 8 * df_t (+ 4 * n_docs the reference-style doc_lens pass would read; this layout folds
// doc_len into the posting so the real traffic is lower).
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "../../include/searcharray_hip.h"

#include <algorithm>
#include <new>
#include <stdlib.h>

#define SA_MAX_QTERMS 32
#define SA_KMAX 1024
#define SA_EVENT_RING 128

struct alignas(16) sa_u64x2 { u64 x, y; };

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
    u32 B, T, k;
    float k1, b, avgdl;
    u32 q_per_xcd;         // >0: XCD-grouped block mapping, 0: plain tile-major
    int small_k_argmax;    // k <= 32: iterative block arg-max instead of threshold selection
    int no_topk;           // timing experiments only: skip the per-tile selection
    u32 cand_per_tile;     // general mode: candidate slots per (query, tile) = k
    u32 cand_cap;          // pruned mode: capacity of each query's append list
    u32* cand_cnt;         // pruned mode: [B] append cursors
    u32* slots;            // pruned mode: [B][32] pruning slots (score bits)
    // outputs
    float* dense_out;      // [B][n_docs] or null
    u64* cand;             // [B][n_tiles][k] composite keys (global doc ids) or null
};

// block -> (tile, query).  Hardware places consecutive workgroups on consecutive XCDs
// (block b -> XCD b % 8, each XCD with a private 4 MiB L2).  In grouped mode XCD x serves
// queries [x*q_per_xcd, (x+1)*q_per_xcd) and walks the tiles in order, so the queries that
// stream the same tile of the same frequent term hit that XCD's L2.  Speed only.
__device__ __forceinline__ bool sa_map_block(const Bm25Params& p, u32& tile, u32& q) {
    const u32 b = blockIdx.x;
    if (p.q_per_xcd) {
        const u32 xcd = b & 7u, j = b >> 3;
        tile = j / p.q_per_xcd;
        q = xcd * p.q_per_xcd + (j % p.q_per_xcd);
        return q < p.B && tile < p.n_tiles;
    }
    tile = b / p.B;
    q = b % p.B;
    return tile < p.n_tiles;
}

template <int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS) sa_k_bm25_tiles(const Bm25Params p) {
    constexpr int NW = THREADS / SA_WAVE;
    constexpr int E = TILE / THREADS;
    constexpr int CAP = (TILE >= 8192) ? 2048 : TILE / 4;      // candidate list capacity
    constexpr int LE = (CAP + THREADS - 1) / THREADS;
    constexpr size_t ACC_BYTES = (size_t)TILE * 4;
    constexpr size_t SEL_BYTES = (size_t)(CAP + SA_KMAX) * 8;
    constexpr size_t SMEM_U64 = (ACC_BYTES > SEL_BYTES ? ACC_BYTES : SEL_BYTES) / 8;
    __shared__ u64 smem[SMEM_U64];
    __shared__ u64 s_lo[SA_MAX_QTERMS], s_hi[SA_MAX_QTERMS];
    __shared__ u64 red64[NW + 1];
    __shared__ u32 red[NW + 1];
    __shared__ u32 s_cnt[2];
    float* acc = (float*)smem;

    u32 tile, q;
    if (!sa_map_block(p, tile, q)) return;              // uniform per block
    const u32 tid = threadIdx.x;
    const u64 tile_base = (u64)tile * TILE;
    const u32 T = p.T;
    // pruning slots of this query (see the top-k section); loaded first so the L2 latency hides
    // behind the posting stream.  L1-bypassing load: a fresher bound prunes more.
    u32 slot_val = 0xFFFFFFFFu;
    if (p.small_k_argmax && (tid & (SA_WAVE - 1)) < 32u)
        slot_val = __hip_atomic_load(&p.slots[q * 32u + (tid & 31u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // 1. clear accumulators, locate this tile's slice of every query term
#pragma unroll
    for (int j = 0; j < E; j++) acc[j * THREADS + tid] = 0.f;
    if (tid < T) {
        const u32 term = p.terms[q * T + tid];
        u64 lo = 0, hi = 0;
        if (term < p.n_terms) {
            const u64 base = p.tf_off[term];
            const u32 cnt = (u32)(p.tf_off[term + 1] - base);
            const u32 slot = p.dir_slot[term];
            if (slot != 0xFFFFFFFFu) {
                const u32* row = p.tile_dir + (u64)slot * (p.n_tiles + 1);
                lo = base + row[tile];
                hi = base + row[tile + 1];
            } else {
                const u32 a = sa_lower_bound(p.tfp + base, 0, cnt, tile_base << SA_KEY_SHIFT, SA_KEY_MASK);
                const u32 z = sa_lower_bound(p.tfp + base, a, cnt, (tile_base + TILE) << SA_KEY_SHIFT, SA_KEY_MASK);
                lo = base + a;
                hi = base + z;
            }
        }
        s_lo[tid] = lo;
        s_hi[tid] = hi;
    }
    __syncthreads();

    // 2. term-at-a-time accumulation.  Postings are streamed as 16-byte pairs (one
    // global_load_dwordx4 per lane, 4 in flight) from the 16-byte-aligned part of the slice;
    // the unaligned head/tail posting is handled by lane 0.
    const float k1 = p.k1, bb = p.b, avgdl = p.avgdl;
    const float one_minus_b = 1.0f - bb;
    auto score_into = [&](u64 x, float idf) {
        const u32 d = (u32)((x >> SA_KEY_SHIFT) - tile_base);
        const float tf = (float)(u32)(x & SA_LSB_MASK);
        const float dl = p.dl_packed ? (float)(u32)((x >> SA_LSB_BITS) & SA_LSB_MASK)
                                     : p.doc_lens[tile_base + d];
        const float norm = __fmul_rn(k1, __fadd_rn(one_minus_b, __fmul_rn(bb, __fdiv_rn(dl, avgdl))));
        const float s = __fmul_rn(__fdiv_rn(tf, __fadd_rn(tf, norm)), idf);
        acc[d] = __fadd_rn(acc[d], s);
    };
    for (u32 t = 0; t < T; t++) {
        const u64 lo = s_lo[t], hi = s_hi[t];
        const float idf = p.idf[q * T + t];
        u64 a = (lo + 1ull) & ~1ull;
        if (a > hi) a = hi;
        if (tid == 0 && lo < a) score_into(p.tfp[lo], idf);
        const u64 npairs = (hi - a) >> 1;
        const sa_u64x2* pairs = (const sa_u64x2*)(p.tfp + a);
        u64 j = tid;
        for (; j + 3ull * THREADS < npairs; j += 4ull * THREADS) {
            sa_u64x2 pp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pp[u] = pairs[j + (u64)u * THREADS];
#pragma unroll
            for (int u = 0; u < 4; u++) { score_into(pp[u].x, idf); score_into(pp[u].y, idf); }
        }
        for (; j < npairs; j += THREADS) {
            const sa_u64x2 pq = pairs[j];
            score_into(pq.x, idf);
            score_into(pq.y, idf);
        }
        if (tid == 0 && ((hi - a) & 1ull)) score_into(p.tfp[hi - 1], idf);
        __syncthreads();
    }

    const u64 remain = p.n_docs - tile_base;
    const u32 tile_n = remain < (u64)TILE ? (u32)remain : (u32)TILE;

    // 3. dense drop-in output (SearchArray.score): coalesced tile store
    if (p.dense_out) {
        float* out = p.dense_out + (u64)q * p.n_docs + tile_base;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u32 e = j * THREADS + tid;
            if (e < tile_n) out[e] = acc[e];
        }
    }
    if (!p.cand || p.no_topk) return;

    // 4. per-tile top-k -> composite keys  score_bits<<32 | ~global_doc
    const u32 k = p.k;
    u64* cand = p.cand + (p.small_k_argmax ? 0ull : ((u64)q * p.n_tiles + tile) * p.cand_per_tile);
    u64* sel = smem + CAP;                    // selected keys (aliases acc once keys are in registers)
    u32 nsel = 0;

    if (p.small_k_argmax) {
        // k <= 32: PRUNED selection.  Per query, 32 global slots hold the best score seen by 32
        // disjoint families of waves (slot = wave index mod 32), so G = min(slots) is a score that
        // at least 32 distinct docs reach: nothing below G can enter the top-k.  A wave whose
        // maximum is below G (almost every wave once the first tiles have run) is done after one
        // DPP reduction.  Otherwise it appends its elements >= G -- all of them when there are at
        // most k, else its exact top-k by k rounds of a wave-wide arg-max -- to the query's
        // candidate list.  Stale slot reads only weaken the pruning (slots grow monotonically);
        // the final top-k is exact and deterministic.
        const u32 lane = tid & (SA_WAVE - 1), wave = tid / SA_WAVE;
        u32 lmax = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u32 x = __float_as_uint(acc[j * THREADS + tid]);
            lmax = x > lmax ? x : lmax;
        }
        const u32 wmax = sa_wave_max_u32(lmax);
        const u32 g = sa_wave_min_u32(slot_val);
        const u32 thr = g > 1u ? g : 1u;
        if (wmax < thr) return;                                   // wave-uniform
        const u32 widx = tile * NW + wave;
        if (lane == 0) atomicMax(&p.slots[q * 32u + (widx & 31u)], wmax);
        u64* qcand = p.cand + (u64)q * p.cand_cap;
        const u64 lt = (1ull << lane) - 1ull;
        u32 c = 0;
#pragma unroll
        for (int j = 0; j < E; j++)
            c += (u32)__popcll(__ballot(__float_as_uint(acc[j * THREADS + tid]) >= thr));
        if (c <= k) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&p.cand_cnt[q], c);
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int j = 0; j < E; j++) {
                const u32 e = j * THREADS + tid;
                const u32 x = __float_as_uint(acc[e]);
                const bool keep = x >= thr;
                const u64 b = __ballot(keep);
                if (keep) {
                    const u32 pos = base + (u32)__popcll(b & lt);
                    const u64 doc = p.doc_base + tile_base + e;
                    if (pos < p.cand_cap) qcand[pos] = ((u64)x << 32) | (u64)(u32)(~(u32)doc);
                }
                base += (u32)__popcll(b);
            }
            return;
        }
        // more than k survivors (first tiles of a query, or heavy ties): exact top-k of this wave.
        // Each lane tracks its best and second best element in registers (branch-free); the tile
        // is rescanned only when one lane wins twice in a row of promotions.
        u32 b1k, b1j, b2k, b2j;
#define SA_RESCAN()                                                                   \
        do {                                                                          \
            b1k = 0; b1j = 0; b2k = 0; b2j = 0;                                       \
            _Pragma("unroll") for (int j = 0; j < E; j++) {                           \
                const u32 x = __float_as_uint(acc[j * THREADS + tid]);                \
                const bool g1 = x > b1k, g2 = x > b2k;                                \
                const u32 n2k = g1 ? b1k : (g2 ? x : b2k);                            \
                const u32 n2j = g1 ? b1j : (g2 ? (u32)j : b2j);                       \
                b1k = g1 ? x : b1k; b1j = g1 ? (u32)j : b1j; b2k = n2k; b2j = n2j;    \
            }                                                                         \
        } while (0)
        SA_RESCAN();
        bool stale = false;                         // true: b2 already promoted, next best unknown
        u64 mine_out = 0;
        u32 found = 0;
        for (u32 r = 0; r < k; r++) {
            const u32 m = sa_wave_max_u32(b1k);
            if (m == 0) break;                       // wave-uniform
            const u32 e1 = (b1k == m) ? (b1j * THREADS + tid) : 0xFFFFFFFFu;
            const u32 emin = sa_wave_min_u32(e1);    // ties -> smallest doc id
            if (lane == r) mine_out = ((u64)m << 32) | (u64)emin;
            found = r + 1;
            const bool owner = (e1 == emin);
            if (owner) acc[emin] = 0.f;
            const bool need = owner && stale;
            if (owner && !stale) { b1k = b2k; b1j = b2j; b2k = 0; stale = true; }
            if (__any(need)) { SA_RESCAN(); stale = false; }
        }
#undef SA_RESCAN
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&p.cand_cnt[q], found);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < found && base + lane < p.cand_cap) {
            const u64 doc = p.doc_base + tile_base + (u32)(mine_out & 0xFFFFFFFFull);
            qcand[base + lane] = (mine_out & 0xFFFFFFFF00000000ull) | (u64)(u32)(~(u32)doc);
        }
        return;
    }

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

// Merge n_cand candidate keys per query into the k best, sorted descending.
// One workgroup of 1024 threads per query.
//
// rank_stride > 0: the candidates come in groups of rank_stride keys sorted by rank (a wave's
// arg-max output, or a rank's sorted top-k), so the k-th largest GROUP LEADER is a lower bound G
// of the k-th largest key overall (k distinct groups own a key >= G).  One pass then keeps only
// keys >= G -- a few dozen -- in LDS and a bitonic sort finishes.  If the survivors overflow the
// LDS list (pathological ties), or rank_stride == 0, the k-th largest is found by MSB-first
// bisection over the whole candidate array instead.
#define SA_MERGE_LIST 2048

__global__ void __launch_bounds__(1024)
sa_k_topk_merge(const u64* __restrict__ cand, u32 n_cand_max, u32 k, u64* __restrict__ out,
                const u32* __restrict__ out_row, u32 rank_stride, const u32* __restrict__ cnt) {
    constexpr int NW = 1024 / SA_WAVE;
    __shared__ u64 red64[NW + 1];
    __shared__ u64 sel[SA_MERGE_LIST];
    __shared__ u32 s_n;
    const u32 q = blockIdx.x, tid = threadIdx.x;
    const u64* c = cand + (u64)q * n_cand_max;
    // cnt != null: the row is an append list holding cnt[q] keys (pruned tile selection)
    u32 n_cand = n_cand_max;
    if (cnt) { const u32 have = cnt[q]; n_cand = have < n_cand_max ? have : n_cand_max; }
    const u32 row = out_row ? out_row[q] : q;       // device row q holds caller query out_row[q]
    for (u32 i = tid; i < SA_MERGE_LIST; i += 1024) sel[i] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    // MSB-first bisection (2 bits per step) for the k-th largest of c[0 : n : stride]
    auto kth_largest = [&](u32 n, u32 stride) -> u64 {
        u64 m = 0;
        for (u32 i = tid; i < n; i += 1024) { const u64 x = c[(u64)i * stride]; m = x > m ? x : m; }
        m = sa_block_max64<NW>(m, red64);
        if (m == 0 || n < k) return 0;
        int top = 63 - __clzll((long long)m);
        if ((top & 1) == 0) top++;
        u64 prefix = 0;
        for (int bit = top; bit >= 1; bit -= 2) {
            const u64 c1 = prefix | (1ull << (bit - 1)), c2 = prefix | (2ull << (bit - 1)), c3 = prefix | (3ull << (bit - 1));
            u64 packed = 0;
            for (u32 i = tid; i < n; i += 1024) {
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
    bool exact = false;                               // thr is the exact k-th largest key
    if (rank_stride > 0 && n_cand / rank_stride >= k) {
        const u64 g = kth_largest(n_cand / rank_stride, rank_stride);
        thr = g > 1 ? g : 1;
    } else if (n_cand > SA_MERGE_LIST) {
        const u64 g = kth_largest(n_cand, 1);
        thr = g > 1 ? g : 1;
        exact = true;
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        for (u32 i = tid; i < n_cand; i += 1024) {
            const u64 x = c[i];
            if (x >= thr) {
                const u32 pos = atomicAdd(&s_n, 1u);
                if (pos < SA_MERGE_LIST) sel[pos] = x;
            }
        }
        __syncthreads();
        const u32 n_sel = s_n;
        __syncthreads();
        if (n_sel <= SA_MERGE_LIST || exact) break;   // uniform
        // survivors overflowed the list: fall back to the exact threshold and gather again
        for (u32 i = tid; i < SA_MERGE_LIST; i += 1024) sel[i] = 0;
        if (tid == 0) s_n = 0;
        const u64 g = kth_largest(n_cand, 1);
        thr = g > 1 ? g : 1;
        exact = true;
        __syncthreads();
    }
    u32 n_sel = s_n < SA_MERGE_LIST ? s_n : SA_MERGE_LIST;
    u32 np2 = 2;
    while (np2 < n_sel) np2 <<= 1;
    sa_block_bitonic_desc(sel, np2);
    for (u32 i = tid; i < k; i += 1024) out[(u64)row * k + i] = (i < SA_MERGE_LIST) ? sel[i] : 0ull;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct sa_batch {
    sa_index* ix = nullptr;
    u32 B = 0, T = 0, k = 0;
    float k1 = 1.2f, b = 0.75f;
    std::vector<u32> perm;          // device row r holds caller query perm[r] (results: caller order)
    u32* d_terms = nullptr;
    u32* d_perm = nullptr;
    float* d_idf = nullptr;
    u64* d_cand = nullptr;          // [B][n_tiles][waves*k]: per-tile blocks, or per-query append lists
    u32* d_cand_cnt = nullptr;      // [B] append cursors (pruned selection)
    u32* d_slots = nullptr;         // [B][32] pruning slots
    u64* d_local = nullptr;         // [B][k] per-shard result
    u64* d_gather = nullptr;        // [nranks][B][k] (multi-GPU)
    u64* d_final = nullptr;         // [B][k]
    u64* d_xcand = nullptr;         // [B][nranks*k] regrouped gather
    int xcand_ranks = 0;
    std::vector<hipEvent_t> ev0, ev1;   // ring of (start, stop) events around the scoring kernel
    u32 ev_n = 0;                       // runs recorded since the last sa_batch_profile
    u64 alg_bytes = 0, postings_bytes = 0;
    bool ran = false;
};

int sa_comm_allgather_topk(sa_index* ix, const u64* d_local, u64* d_gather, size_t count, int* nranks_out);

static int sa_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static void sa_fill_params(const sa_index* ix, Bm25Params& p) {
    p.tfp = ix->d_tfp; p.tf_off = ix->d_tf_off; p.dir_slot = ix->d_dir_slot; p.tile_dir = ix->d_tile_dir;
    p.doc_lens = ix->d_doc_lens; p.n_terms = ix->n_terms; p.n_tiles = ix->n_tiles;
    p.n_docs = ix->n_docs; p.doc_base = ix->doc_base; p.dl_packed = ix->dl_packed ? 1 : 0;
    p.avgdl = ix->avg_doc_len;
}

static u32 sa_tile_waves(u32 tile_docs) {
    switch (tile_docs) {
        case 1024: return 2; case 2048: return 1; case 4096: return 2; case 8192: return 4;
        case 16384: return 8; case 32768: return 16; default: return 1;
    }
}

static int sa_launch_bm25(sa_index* ix, const Bm25Params& p, hipStream_t st) {
    if (ix->n_tiles == 0 || p.B == 0) return SA_OK;
    const u32 grid = p.q_per_xcd ? 8u * p.q_per_xcd * ix->n_tiles : p.B * ix->n_tiles;
    switch (ix->tile_docs) {
        case 1024:
            hipLaunchKernelGGL((sa_k_bm25_tiles<1024, 128>), dim3(grid), dim3(128), 0, st, p); break;
        case 2048:
            hipLaunchKernelGGL((sa_k_bm25_tiles<2048, 64>), dim3(grid), dim3(64), 0, st, p); break;
        case 4096:
            hipLaunchKernelGGL((sa_k_bm25_tiles<4096, 128>), dim3(grid), dim3(128), 0, st, p); break;
        case 8192:
            hipLaunchKernelGGL((sa_k_bm25_tiles<8192, 256>), dim3(grid), dim3(256), 0, st, p); break;
        case 16384:
            hipLaunchKernelGGL((sa_k_bm25_tiles<16384, 512>), dim3(grid), dim3(512), 0, st, p); break;
        case 32768:
            hipLaunchKernelGGL((sa_k_bm25_tiles<32768, 1024>), dim3(grid), dim3(1024), 0, st, p); break;
        default:
            sa_set_error("unsupported tile_docs %u", ix->tile_docs);
            return SA_ERR_STATE;
    }
    return SA_OK;
}

extern "C" int sa_index_bm25_dense(sa_index_t* ix, const uint32_t* terms, const float* idf,
                                   int n_query_terms, float k1, float b, float* out) {
    SA_ARG(ix && out, "null argument");
    SA_ARG(n_query_terms >= 0, "n_query_terms < 0");
    SA_ARG(n_query_terms == 0 || (terms && idf), "terms/idf null");
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    const u64 N = ix->n_docs;
    // reference similarity.py:31-32: avg_doc_lens == 0 -> zeros
    if (N == 0) return SA_OK;
    if (n_query_terms == 0 || ix->avg_doc_len == 0.f) {
        memset(out, 0, N * sizeof(float));
        return SA_OK;
    }
    // scratch: [out chunk accumulators N floats][terms][idf]
    const int T = n_query_terms;
    SA_ARG(T <= SA_MAX_QTERMS, "more than 32 query terms per call is not supported");
    void* scratch;
    SA_TRY(sa_index_scratch(ix, N * sizeof(float) * 2 + (size_t)T * 8 + 256, &scratch));
    float* d_out = (float*)scratch;
    float* d_tmp = d_out + N;
    u32* d_terms = (u32*)(d_tmp + N);
    float* d_idf = (float*)(d_terms + T);
    SA_HIP(hipMemcpyAsync(d_terms, terms, (size_t)T * sizeof(u32), hipMemcpyHostToDevice, st));
    SA_HIP(hipMemcpyAsync(d_idf, idf, (size_t)T * sizeof(float), hipMemcpyHostToDevice, st));
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    sa_fill_params(ix, p);
    p.terms = d_terms; p.idf = d_idf; p.B = 1; p.T = (u32)T; p.k = 0;
    p.k1 = k1; p.b = b; p.q_per_xcd = 0; p.small_k_argmax = 0;
    p.dense_out = d_out; p.cand = nullptr;
    (void)d_tmp;
    SA_TRY(sa_launch_bm25(ix, p, st));
    SA_HIP(hipMemcpyAsync(out, d_out, N * sizeof(float), hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

static void sa_batch_free(sa_batch* bt) {
    if (!bt) return;
    if (bt->ix) hipSetDevice(bt->ix->device);
    if (bt->d_terms) hipFree(bt->d_terms);
    if (bt->d_perm) hipFree(bt->d_perm);
    if (bt->d_idf) hipFree(bt->d_idf);
    if (bt->d_cand) hipFree(bt->d_cand);
    if (bt->d_cand_cnt) hipFree(bt->d_cand_cnt);
    if (bt->d_slots) hipFree(bt->d_slots);
    if (bt->d_local) hipFree(bt->d_local);
    if (bt->d_gather) hipFree(bt->d_gather);
    if (bt->d_final) hipFree(bt->d_final);
    if (bt->d_xcand) hipFree(bt->d_xcand);
    for (hipEvent_t e : bt->ev0) hipEventDestroy(e);
    for (hipEvent_t e : bt->ev1) hipEventDestroy(e);
    delete bt;
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
    bt->ix = ix; bt->B = (u32)n_queries; bt->T = (u32)n_query_terms; bt->k = (u32)k; bt->k1 = k1; bt->b = b;
    const u32 B = bt->B, T = bt->T;
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
    std::vector<u32> h_terms((size_t)B * T);
    std::vector<float> h_idf((size_t)B * T);
    for (u32 r = 0; r < B; r++) {
        memcpy(&h_terms[(size_t)r * T], &terms[(size_t)bt->perm[r] * T], T * sizeof(u32));
        memcpy(&h_idf[(size_t)r * T], &idf[(size_t)bt->perm[r] * T], T * sizeof(float));
    }
    int rc = SA_OK;
    auto fail = [&](int code) { sa_batch_free(bt); return code; };
#define SA_HIP_B(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { sa_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return fail(SA_ERR_HIP); } } while (0)
    SA_HIP_B(hipMalloc(&bt->d_terms, h_terms.size() * sizeof(u32)));
    SA_HIP_B(hipMalloc(&bt->d_idf, h_idf.size() * sizeof(float)));
    SA_HIP_B(hipMalloc(&bt->d_perm, (size_t)B * sizeof(u32)));
    SA_HIP_B(hipMemcpy(bt->d_perm, bt->perm.data(), (size_t)B * sizeof(u32), hipMemcpyHostToDevice));
    const size_t ncand = (size_t)B * (ix->n_tiles ? ix->n_tiles : 1) * bt->k * sa_tile_waves(ix->tile_docs);
    SA_HIP_B(hipMalloc(&bt->d_cand, ncand * sizeof(u64)));
    SA_HIP_B(hipMalloc(&bt->d_cand_cnt, (size_t)B * sizeof(u32)));
    SA_HIP_B(hipMalloc(&bt->d_slots, (size_t)B * 32 * sizeof(u32)));
    SA_HIP_B(hipMalloc(&bt->d_local, (size_t)B * bt->k * sizeof(u64)));
    SA_HIP_B(hipMalloc(&bt->d_final, (size_t)B * bt->k * sizeof(u64)));
    SA_HIP_B(hipMemset(bt->d_final, 0, (size_t)B * bt->k * sizeof(u64)));
    SA_HIP_B(hipMemset(bt->d_local, 0, (size_t)B * bt->k * sizeof(u64)));
    SA_HIP_B(hipMemcpy(bt->d_terms, h_terms.data(), h_terms.size() * sizeof(u32), hipMemcpyHostToDevice));
    SA_HIP_B(hipMemcpy(bt->d_idf, h_idf.data(), h_idf.size() * sizeof(float), hipMemcpyHostToDevice));
    for (int i = 0; i < SA_EVENT_RING; i++) {
        hipEvent_t a = nullptr, c = nullptr;
        SA_HIP_B(hipEventCreate(&a));
        bt->ev0.push_back(a);
        SA_HIP_B(hipEventCreate(&c));
        bt->ev1.push_back(c);
    }
#undef SA_HIP_B
    (void)rc;
    *out = bt;
    return SA_OK;
}

// regroup an all-gather result [rank][B][k] into per-query candidate rows [B][rank*k]
__global__ void sa_k_regroup(const u64* __restrict__ gathered, u32 nranks, u32 B, u32 k, u64* __restrict__ out) {
    const u64 total = (u64)nranks * B * k;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 j = (u32)(i % k);
        const u32 q = (u32)((i / k) % B);
        const u32 r = (u32)(i / ((u64)k * B));
        out[((u64)q * nranks + r) * k + j] = gathered[i];
    }
}

// stage 1 (tile scoring + per-tile top-k) and stage 2 (per-shard merge) on the index stream
static int sa_batch_run_shard(sa_batch* bt, u64* shard_out) {
    sa_index* ix = bt->ix;
    hipStream_t st = ix->stream;
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    sa_fill_params(ix, p);
    p.terms = bt->d_terms; p.idf = bt->d_idf; p.B = bt->B; p.T = bt->T; p.k = bt->k;
    p.k1 = bt->k1; p.b = bt->b;
    const int xcd_mode = sa_env_int("SA_XCD_MODE", 0);
    p.q_per_xcd = (xcd_mode && bt->B >= 8) ? (bt->B + 7) / 8 : 0;
    p.small_k_argmax = (bt->k <= 32 && sa_env_int("SA_SMALLK_ARGMAX", 1)) ? 1 : 0;
    p.dense_out = nullptr; p.cand = bt->d_cand;
    p.no_topk = sa_env_int("SA_NO_TOPK", 0);
    p.cand_per_tile = bt->k;
    p.cand_cap = (ix->n_tiles ? ix->n_tiles : 1) * bt->k * sa_tile_waves(ix->tile_docs);
    p.cand_cnt = bt->d_cand_cnt;
    p.slots = bt->d_slots;
    if (p.small_k_argmax) {
        SA_HIP(hipMemsetAsync(bt->d_cand_cnt, 0, (size_t)bt->B * sizeof(u32), st));
        SA_HIP(hipMemsetAsync(bt->d_slots, 0, (size_t)bt->B * 32 * sizeof(u32), st));
    }
    const u32 slot = bt->ev_n % SA_EVENT_RING;
    SA_HIP(hipEventRecord(bt->ev0[slot], st));
    if (ix->avg_doc_len != 0.f && ix->n_tiles > 0) {
        SA_TRY(sa_launch_bm25(ix, p, st));
    } else {
        SA_HIP(hipMemsetAsync(bt->d_cand, 0, (size_t)bt->B * p.cand_cap * sizeof(u64), st));
    }
    SA_HIP(hipEventRecord(bt->ev1[slot], st));
    bt->ev_n++;
    const u32 n_cand = p.small_k_argmax ? p.cand_cap : (ix->n_tiles ? ix->n_tiles : 1) * p.cand_per_tile;
    hipLaunchKernelGGL(sa_k_topk_merge, dim3(bt->B), dim3(1024), 0, st, bt->d_cand, n_cand, bt->k, shard_out,
                       (const u32*)bt->d_perm, 0u, (const u32*)(p.small_k_argmax ? bt->d_cand_cnt : nullptr));
    bt->ran = true;
    return SA_OK;
}

// stage 3: merge the per-rank top-k lists [nranks][B][k] (device memory) into d_final
static int sa_batch_merge_ranks(sa_batch* bt, const u64* d_gathered, int nranks) {
    sa_index* ix = bt->ix;
    hipStream_t st = ix->stream;
    const size_t count = (size_t)bt->B * bt->k;
    if (!bt->d_xcand || bt->xcand_ranks < nranks) {
        if (bt->d_xcand) SA_HIP(hipFree(bt->d_xcand));
        bt->d_xcand = nullptr;
        SA_HIP(hipMalloc(&bt->d_xcand, (size_t)nranks * count * sizeof(u64)));
        bt->xcand_ranks = nranks;
    }
    const u64 total = (u64)nranks * count;
    const u32 grid = total / 256 + 1 < 4096 ? (u32)(total / 256 + 1) : 4096;
    hipLaunchKernelGGL(sa_k_regroup, dim3(grid), dim3(256), 0, st, d_gathered, (u32)nranks, bt->B, bt->k, bt->d_xcand);
    // every rank's block is its sorted top-k: group leaders = rank maxima
    hipLaunchKernelGGL(sa_k_topk_merge, dim3(bt->B), dim3(1024), 0, st, bt->d_xcand, (u32)nranks * bt->k, bt->k, bt->d_final,
                       (const u32*)nullptr, bt->k, (const u32*)nullptr);
    return SA_OK;
}

extern "C" int sa_batch_run(sa_batch_t* bt, int sync) {
    SA_ARG(bt && bt->ix, "null batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    if (ix->comm) {
        SA_TRY(sa_batch_run_shard(bt, bt->d_local));
        int nranks = 1;
        const size_t count = (size_t)bt->B * bt->k;
        SA_TRY(sa_comm_allgather_topk(ix, nullptr, nullptr, 0, &nranks));
        if (!bt->d_gather) SA_HIP(hipMalloc(&bt->d_gather, (size_t)nranks * count * sizeof(u64)));
        SA_TRY(sa_comm_allgather_topk(ix, bt->d_local, bt->d_gather, count, &nranks));
        SA_TRY(sa_batch_merge_ranks(bt, bt->d_gather, nranks));
    } else {
        SA_TRY(sa_batch_run_shard(bt, bt->d_final));
    }
    if (sync) {
        SA_HIP(hipStreamSynchronize(st));
        SA_HIP(hipGetLastError());
    }
    return SA_OK;
}

// External-collective variant (the caller owns the exchange, e.g. torch.distributed over RCCL,
// or gloo in the CPU tests): run this shard, hand out its top-k keys, merge gathered keys.
extern "C" int sa_batch_run_local(sa_batch_t* bt, void* local_keys_out_device, int sync) {
    SA_ARG(bt && bt->ix, "null batch");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_TRY(sa_batch_run_shard(bt, bt->d_local));
    if (local_keys_out_device)
        SA_HIP(hipMemcpyAsync(local_keys_out_device, bt->d_local, (size_t)bt->B * bt->k * sizeof(u64),
                              hipMemcpyDeviceToDevice, ix->stream));
    if (sync) {
        SA_HIP(hipStreamSynchronize(ix->stream));
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
    SA_TRY(sa_batch_merge_ranks(bt, (const u64*)gathered_keys_device, nranks));
    if (sync) {
        SA_HIP(hipStreamSynchronize(ix->stream));
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
    std::vector<u64> keys(n);
    SA_HIP(hipStreamSynchronize(ix->stream));
    SA_HIP(hipGetLastError());
    SA_HIP(hipMemcpy(keys.data(), bt->d_final, n * sizeof(u64), hipMemcpyDeviceToHost));
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

extern "C" int sa_batch_profile(sa_batch_t* bt, double* kernel_ms_out, uint64_t* alg_bytes_out,
                                uint64_t* postings_bytes_out) {
    SA_ARG(bt && bt->ix, "null batch");
    SA_ARG(bt->ran && bt->ev_n > 0, "batch has not been run since the last profile call");
    sa_index* ix = bt->ix;
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    SA_HIP(hipStreamSynchronize(ix->stream));
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

extern "C" int sa_batch_destroy(sa_batch_t* bt) {
    if (!bt) return SA_OK;
    sa_batch_free(bt);
    return SA_OK;
}

// sa_topk.hpp -- block-cooperative top-k selection primitives (device).
//
// Ranking key: a 64-bit composite  score_bits << 32 | tie , where score_bits is the IEEE
// pattern of a non-negative fp32 score (monotone as an unsigned integer) and `tie` grows as the
// doc id shrinks, so "larger key" == "higher score, then smaller doc id".  Keys are unique per
// doc, which makes the k-th largest key a clean threshold: exactly k keys are >= it.  Key 0
// means "no candidate" (score 0 never ranks; the reference's argpartition would return
// arbitrary zero-score docs there, reference utils/sort.py:24).
#pragma once
#include "sa_common.hpp"

template <int NW>
__device__ __forceinline__ u64 sa_block_max64(u64 v, u64* red64) {
    v = sa_wave_max64(v);
    if (sa_lane() == 0) red64[sa_wave_id()] = v;
    __syncthreads();
    u64 m = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { u64 x = red64[w]; m = x > m ? x : m; }
    __syncthreads();
    return m;
}

// Largest value v such that at least k of the block's keys are >= v (0 when fewer than k keys
// are non-zero).  Each thread contributes EPT keys held in registers.  MSB-first bisection,
// two bits per step: three counts per step travel packed in one 64-bit block reduction.
//
// `key(e)` yields the e-th key of the calling thread (e in [0, EPT)); it is re-evaluated every
// step, so composite keys can be formed on the fly from narrower registers.
template <int EPT, int NW, class KeyFn>
__device__ __forceinline__ u64 sa_block_kth_largest(KeyFn key, u32 k, u64* red64) {
    u64 m = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) { const u64 x = key(e); m = x > m ? x : m; }
    m = sa_block_max64<NW>(m, red64);
    if (m == 0 || k == 0) return 0;
    int top = 63 - __clzll((long long)m);
    if ((top & 1) == 0) top++;            // pairs of bits: (top, top-1)
    u64 prefix = 0;
    for (int bit = top; bit >= 1; bit -= 2) {
        const u64 c1 = prefix | (1ull << (bit - 1));
        const u64 c2 = prefix | (2ull << (bit - 1));
        const u64 c3 = prefix | (3ull << (bit - 1));
        u64 packed = 0;                    // 21 bits per count (block holds < 2^21 keys)
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const u64 x = key(e);
            packed += (x >= c1 ? 1ull : 0ull) + (x >= c2 ? (1ull << 21) : 0ull) + (x >= c3 ? (1ull << 42) : 0ull);
        }
        // block sum of a u64
#pragma unroll
        for (int o = SA_WAVE / 2; o > 0; o >>= 1) packed += __shfl_xor(packed, o, SA_WAVE);
        if (sa_lane() == 0) red64[sa_wave_id()] = packed;
        __syncthreads();
        u64 tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) tot += red64[w];
        __syncthreads();
        const u32 n1 = (u32)(tot & 0x1FFFFF), n2 = (u32)((tot >> 21) & 0x1FFFFF), n3 = (u32)((tot >> 42) & 0x1FFFFF);
        if (n3 >= k) prefix = c3;
        else if (n2 >= k) prefix = c2;
        else if (n1 >= k) prefix = c1;
    }
    return prefix;
}

// In-LDS bitonic sort, descending, of n_pow2 (power of two, <= 2 * blockDim) u64 keys.
__device__ __forceinline__ void sa_block_bitonic_desc(u64* a, u32 n_pow2) {
    for (u32 size = 2; size <= n_pow2; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (u32 t = threadIdx.x; t < (n_pow2 >> 1); t += blockDim.x) {
                const u32 lo = 2 * t - (t & (stride - 1));
                const u32 hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const u64 x = a[lo], y = a[hi];
                if (desc ? (x < y) : (x > y)) { a[lo] = y; a[hi] = x; }
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Pruned wave-level top-k of one scored tile (shared by the BM25 and the phrase tile kernels).
//
// `acc` holds the TILE fp32 scores of this workgroup's tile in LDS (thread tid owns elements
// j * THREADS + tid); `doc0` is the global doc id of element 0.  Per query, 32 global slots each
// hold a score that `rr = ceil(k/32)` distinct docs of ONE wave reach (the rr-th largest per-lane
// maximum of that wave), taken over 32 disjoint families of waves (slot = wave index mod 32).  So
// G = min(slots) is a score at least 32 * rr >= k distinct docs reach: nothing below G can enter
// the top-k.  A wave whose maximum is below G (almost every wave once the first tiles have run) is
// done after one DPP reduction.  Otherwise it appends its elements >= G -- all of them when there
// are at most k, else its exact top-k by k rounds of a wave-wide arg-max -- to the query's
// candidate list (cand[q * cand_cap ...], cursor cand_cnt[q]).  Stale slot reads only weaken the
// pruning (slots grow monotonically); the final top-k is exact and deterministic.
// `slot_val`: lanes 0..31 of every wave hold slots[q*32 + lane] (loaded early by the caller to
// hide the latency), the other lanes 0xFFFFFFFF.  No barriers inside: waves finish independently.
// `docs` (LDS) given: element e is the doc doc0 + docs[e] instead of doc0 + e (a block of documents that are not neighbours).
template <int TILE, int THREADS>
__device__ __forceinline__ void sa_tile_topk_pruned(float* acc, u32 slot_val, u32 q, u32 tile, u64 doc0, u32 k,
                                                    u32* __restrict__ slots, u64* __restrict__ cand, u32 cand_cap,
                                                    u32* __restrict__ cand_cnt, const u32* docs = nullptr, const u32 wave_stride = 0u) {
    constexpr int NW = THREADS / SA_WAVE;
    constexpr int E = TILE / THREADS;
    const u32 tid = threadIdx.x;
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
    if (wmax < thr) return;                                    // wave-uniform
#define SA_ELEM(j) ((u32)(j) * THREADS + tid)
    // (the slot this (tile, wave) raises.  wave_stride: for tiles whose elements are COMPACTED to the front -- their first wave holds
    //  most of them, the last ones often nothing -- the slots go by tile first, so that the first waves of consecutive tiles reach all 32)
    const u32 widx = wave_stride ? tile + wave * wave_stride : tile * NW + wave;
    {
        // slot update: the rr-th largest lane maximum (lanes counted individually)
        const u32 my_slot = (u32)__shfl((int)slot_val, (int)(widx & 31u), SA_WAVE);
        if (wmax > my_slot) {                                 // wave-uniform
            const u32 rr = (k + 31u) / 32u;
            u32 v = lmax, cnt = 0, mr = 0;
            for (u32 it = 0; it < rr; it++) {
                const u32 m = it == 0 ? wmax : sa_wave_max_u32(v);
                if (m == 0) break;
                cnt += (u32)__popcll(__ballot(v == m));
                if (cnt >= rr) { mr = m; break; }
                v = (v == m) ? 0u : v;
            }
            if (lane == 0 && mr > my_slot) atomicMax(&slots[q * 32u + (widx & 31u)], mr);
        }
    }
    if (wave_stride && k <= 32u) {
        // A compacted tile's first wave also raises the slots of the tile's other waves -- mostly empty, they would leave their slots
        // at 0 and the bound (the minimum over the slots) with them -- with its NEXT largest lane maxima: other lanes, other documents,
        // so the slots stay backed by pairwise distinct documents (k <= 32: one document per slot).
        u32 v = lmax;
        { const u64 eq = __ballot(v == wmax); if (lane == (u32)__builtin_ctzll(eq)) v = 0u; }
        for (u32 i = 1; i < (u32)NW; i++) {
            const u32 m = sa_wave_max_u32(v);
            if (m < thr) break;                                // (wave-uniform: every slot is at thr or above already)
            const u32 si = (widx + i * wave_stride) & 31u;
            const u32 cur = (u32)__shfl((int)slot_val, (int)si, SA_WAVE);
            if (lane == 0 && m > cur) atomicMax(&slots[q * 32u + si], m);
            const u64 eq = __ballot(v == m);
            if (lane == (u32)__builtin_ctzll(eq)) v = 0u;
        }
    }
    u64* qcand = cand + (u64)q * cand_cap;
    const u64 lt = (1ull << lane) - 1ull;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < E; j++)
        c += (u32)__popcll(__ballot(__float_as_uint(acc[SA_ELEM(j)]) >= thr));
    if (c <= k) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&cand_cnt[q], c);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u32 e = SA_ELEM(j);
            const u32 x = __float_as_uint(acc[e]);
            const bool keep = x >= thr;
            const u64 b = __ballot(keep);
            if (keep) {
                const u32 pos = base + (u32)__popcll(b & lt);
                const u64 doc = doc0 + (docs ? (u64)docs[e] : (u64)e);
                if (pos < cand_cap) qcand[pos] = ((u64)x << 32) | (u64)(u32)(~(u32)doc);
            }
            base += (u32)__popcll(b);
        }
        return;
    }
    // more than k survivors (first tiles of a query, or heavy ties): exact top-k of this wave.
    // Each lane tracks its best and second best element in registers (branch-free); the tile
    // is rescanned only when one lane wins twice in a row of promotions.
    u32 cbase = 0;
    if (lane == 0) cbase = atomicAdd(&cand_cnt[q], k);
    cbase = (u32)__builtin_amdgcn_readfirstlane((int)cbase);
    u32 b1k, b1j, b2k, b2j;
#define SA_RESCAN()                                                                   \
    do {                                                                          \
        b1k = 0; b1j = 0; b2k = 0; b2j = 0;                                       \
        _Pragma("unroll") for (int j = 0; j < E; j++) {                           \
            const u32 x = __float_as_uint(acc[SA_ELEM(j)]);                \
            const bool g1 = x > b1k, g2 = x > b2k;                                \
            const u32 n2k = g1 ? b1k : (g2 ? x : b2k);                            \
            const u32 n2j = g1 ? b1j : (g2 ? (u32)j : b2j);                       \
            b1k = g1 ? x : b1k; b1j = g1 ? (u32)j : b1j; b2k = n2k; b2j = n2j;    \
        }                                                                         \
    } while (0)
    SA_RESCAN();
    bool stale = false;                         // true: b2 already promoted, next best unknown
    u32 found = 0;
    for (u32 r = 0; r < k; r++) {
        const u32 m = sa_wave_max_u32(b1k);
        if (m == 0) break;                       // wave-uniform
        const u32 e1 = (b1k == m) ? SA_ELEM(b1j) : 0xFFFFFFFFu;
        const u32 emin = sa_wave_min_u32(e1);    // ties -> smallest doc id
        found = r + 1;
        const bool owner = (e1 == emin);
        if (owner) {
            acc[emin] = 0.f;
            const u64 doc = doc0 + (docs ? (u64)docs[emin] : (u64)emin);
            if (cbase + r < cand_cap) qcand[cbase + r] = ((u64)m << 32) | (u64)(u32)(~(u32)doc);
        }
        const bool need = owner && stale;
        if (owner && !stale) { b1k = b2k; b1j = b2j; b2k = 0; stale = true; }
        if (__any(need)) { SA_RESCAN(); stale = false; }
    }
#undef SA_RESCAN
#undef SA_ELEM
    for (u32 r = found + lane; r < k; r += SA_WAVE)           // unused reserved slots
        if (cbase + r < cand_cap) qcand[cbase + r] = 0ull;
}

// ---------------------------------------------------------------------------------------
// Pruned top-k for large k (32 < k <= 1024): the bound comes from a per-query score HISTOGRAM.
//
// The slot bound above is loose for k >> 32 (a slot holds a score that only ceil(k/32) docs of one
// wave reach).  Here every wave that survives the bound adds its surviving docs to a global
// 256-bin histogram of the query (bins = the top bits of the fp32 pattern: 16 bins per octave,
// 4.4 % wide, scores 2^-6 .. 2^10), and the bound is the lower edge of the highest bin above which
// at least k docs have been counted -- the k-th best score seen so far, to bin resolution.  Every
// doc is counted at most once (by the wave that owns it) and only docs that exist are counted, so
// at least k docs reach the bound: nothing below it can enter the top-k.  Waves whose maximum is
// below the cached bound gthr[q] are done after one reduction; the others count their survivors
// (per-wave histogram in LDS, flushed with one global atomic per non-empty bin), refresh the bound
// and append ALL their docs at or above it (no per-wave exact top-k: the bound is tight enough
// that appending is cheaper).  Stale reads only weaken the bound.
// lds_hist: 128 u32 per wave of scratch LDS (two 16-bit bins per word: a wave owns < 2^16 docs).
// ---------------------------------------------------------------------------------------
#define SA_HBINS 256
#define SA_HBIN_SHIFT 19
#define SA_HBIN_BASE ((127u - 6u) << 4)

__device__ __forceinline__ u32 sa_score_bin(u32 x) {
    const u32 e = x >> SA_HBIN_SHIFT;
    return e <= SA_HBIN_BASE ? 0u : (e - SA_HBIN_BASE > (u32)(SA_HBINS - 1) ? (u32)(SA_HBINS - 1) : e - SA_HBIN_BASE);
}
__device__ __forceinline__ u32 sa_bin_edge(u32 b) { return b == 0 ? 0u : (b + SA_HBIN_BASE) << SA_HBIN_SHIFT; }

// The bound of a query's histogram: lower edge of the highest bin with at least k docs at or above
// it (0 if fewer than k docs are counted).  Wave-cooperative: lane L passes the totals of bins
// 4L .. 4L+3; every lane returns the same value.
__device__ __forceinline__ u32 sa_hist_bound(const u32 (&tot)[SA_HBINS / SA_WAVE], u32 k, u32 lane) {
    u32 lane_sum = 0;
#pragma unroll
    for (int i = 0; i < SA_HBINS / SA_WAVE; i++) lane_sum += tot[i];
    u32 suf = lane_sum;                                         // inclusive suffix sum over lanes
#pragma unroll
    for (int o = 1; o < SA_WAVE; o <<= 1) {
        const u32 up = __shfl_down(suf, (unsigned)o, SA_WAVE);
        if (lane + (u32)o < (u32)SA_WAVE) suf += up;
    }
    const u64 ok = __ballot(suf >= k);
    u32 g = 0;
    if (ok) {
        const u32 ls = 63u - (u32)__clzll((long long)ok);       // highest lane whose suffix reaches k
        u32 above = __shfl_down(suf, 1u, SA_WAVE);              // docs in the lanes above mine
        if (lane == (u32)SA_WAVE - 1) above = 0;
        u32 bsel = 0;
#pragma unroll
        for (int i = SA_HBINS / SA_WAVE - 1; i >= 0; i--) {
            above += tot[i];
            if (bsel == 0 && above >= k) bsel = lane * (SA_HBINS / SA_WAVE) + (u32)i + 1u;   // +1: 0 means none
        }
        const u32 b = (u32)__shfl((int)bsel, (int)ls, SA_WAVE);
        g = b ? sa_bin_edge(b - 1u) : 0u;
    }
    return g;
}

// read a query's histogram and raise its cached bound (one wave; every lane must call)
__device__ __forceinline__ u32 sa_hist_refresh(u32* __restrict__ qh, u32* __restrict__ gthr_q, u32 k, u32 lane) {
    u32 tot[SA_HBINS / SA_WAVE];
#pragma unroll
    for (int i = 0; i < SA_HBINS / SA_WAVE; i++)
        tot[i] = __hip_atomic_load(&qh[lane * (SA_HBINS / SA_WAVE) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 g = sa_hist_bound(tot, k, lane);
    if (lane == 0 && g) atomicMax(gthr_q, g);
    return g;
}

template <int TILE, int THREADS>
__device__ __forceinline__ void sa_tile_topk_hist(float* acc, u32 gc, u32 q, u32 tile, u64 doc0, u32 k,
                                                  u32* __restrict__ hist, u32* __restrict__ gthr,
                                                  u64* __restrict__ cand, u32 cand_cap, u32* __restrict__ cand_cnt,
                                                  u32* lds_hist) {
    constexpr int E = TILE / THREADS;
    const u32 tid = threadIdx.x;
    const u32 lane = tid & (SA_WAVE - 1), wave = tid / SA_WAVE;
    u32 lmax = 0;
#pragma unroll
    for (int j = 0; j < E; j++) {
        const u32 x = __float_as_uint(acc[j * THREADS + tid]);
        lmax = x > lmax ? x : lmax;
    }
    const u32 wmax = sa_wave_max_u32(lmax);
    const u32 thr0 = gc > 1u ? gc : 1u;
    if (wmax < thr0) return;                                   // wave-uniform
    u32* qh = hist + (u64)q * SA_HBINS;
    u64* qcand = cand + (u64)q * cand_cap;
    const u64 lt = (1ull << lane) - 1ull;
    u32 c0 = 0;
#pragma unroll
    for (int j = 0; j < E; j++)
        c0 += (u32)__popcll(__ballot(__float_as_uint(acc[j * THREADS + tid]) >= thr0));
    // Steady state: a handful of docs reach the cached bound.  Count them with one global atomic
    // each and append them; the bound itself is refreshed by every 8th surviving wave (and by any
    // wave with many survivors, i.e. while the bound is still far off) -- a stale bound is valid.
    constexpr int NW = THREADS / SA_WAVE;
    if (c0 <= 16u && ((tile * NW + wave) & 7u) != 0u) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&cand_cnt[q], c0);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u32 e = (u32)j * THREADS + tid;
            const u32 x = __float_as_uint(acc[e]);
            const bool keep = x >= thr0;
            const u64 b = __ballot(keep);
            if (keep) {
                atomicAdd(&qh[sa_score_bin(x)], 1u);
                const u32 pos = base + (u32)__popcll(b & lt);
                if (pos < cand_cap) qcand[pos] = ((u64)x << 32) | (u64)(u32)(~(u32)(doc0 + e));
            }
            base += (u32)__popcll(b);
        }
        return;
    }
    // 1. count this wave's docs >= the cached bound, per bin (bin b = half b & 1 of word b >> 1)
    static_assert(E * SA_WAVE < 65536, "16-bit bins");
    u32* wh = lds_hist + wave * (SA_HBINS / 2);
#pragma unroll
    for (int i = 0; i < SA_HBINS / 2 / SA_WAVE; i++) wh[i * SA_WAVE + lane] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < E; j++) {
        const u32 x = __float_as_uint(acc[j * THREADS + tid]);
        if (x >= thr0) { const u32 b = sa_score_bin(x); atomicAdd(&wh[b >> 1], 1u << (16u * (b & 1u))); }
    }
    __builtin_amdgcn_wave_barrier();
    // lane L owns bins 4L .. 4L+3: flush, then read the query's totals back
    u32 tot[SA_HBINS / SA_WAVE];
#pragma unroll
    for (int i = 0; i < SA_HBINS / SA_WAVE; i++) {
        const u32 b = lane * (SA_HBINS / SA_WAVE) + i;
        const u32 v = (wh[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu;
        u32 old = 0;
        if (v) old = atomicAdd(&qh[b], v);
        else old = __hip_atomic_load(&qh[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[i] = old + v;
    }
    // 2. the bound: highest bin with at least k docs at or above it
    const u32 g = sa_hist_bound(tot, k, lane);
    if (lane == 0 && g > gc) atomicMax(&gthr[q], g);
    const u32 thr = g > thr0 ? g : thr0;
    // 3. append every doc of this wave at or above the bound
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < E; j++)
        c += (u32)__popcll(__ballot(__float_as_uint(acc[j * THREADS + tid]) >= thr));
    if (c == 0) return;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&cand_cnt[q], c);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
    for (int j = 0; j < E; j++) {
        const u32 e = (u32)j * THREADS + tid;
        const u32 x = __float_as_uint(acc[e]);
        const bool keep = x >= thr;
        const u64 b = __ballot(keep);
        if (keep) {
            const u32 pos = base + (u32)__popcll(b & lt);
            if (pos < cand_cap) qcand[pos] = ((u64)x << 32) | (u64)(u32)(~(u32)(doc0 + e));
        }
        base += (u32)__popcll(b);
    }
}


// sa_bm25_group.hpp -- sa_k_bm25_group_fx: the grouped exhaustive BM25 kernel (round 5).  Included by sa_bm25.hip
// (needs Bm25Params, GroupParams, the half descriptors and the top-k helpers defined there).
//
// Reference path: the caller idiom `np.sum([sa.score(t) for t in q], axis=0)` + top-k of /root/reference/test/test_msmarco.py:345-395
// (scores: searcharray/similarity.py:19-38 + searcharray/bm25.pyx:11-25; top-k: searcharray/utils/sort.py:24), for a batch of
// queries that share their first term.
//
// One WORKGROUP of SA_GFX_NW waves owns a (tile, group) item; a group is up to 16 x SA_GFX_NW queries with the same
// (first term, weight).  What changed against round 2-4's one-wave item (sa_k_bm25_group_tiles):
//
//  * FILTER in fixed point, DECIDE in fp32.  The tile's accumulators hold an integer image of the scores: every
//    contribution s = fl(factor * w) enters as trunc(s * 2^F) (the product with the pre-scaled weight w * 2^F has the same
//    mantissa: scaling by a power of two is exact).  Integer addition is associative and commutative, so
//      - a query's postings are ADDED with LDS atomics (ds_add_rtn_u32) in any order -- all halves of a query are in
//        flight together and waited for ONCE; the fp32 overlay had to read-modify-write half after half, one LDS round
//        trip each, because `((s0 + s1) + s2) + s3` must be formed in query-term order (measured: profiles/issue_probe_r05.jsonl,
//        a dependent random-address LDS round trip is ~100 cycles, and the kernel ran at half the issue rate a 1 : 1
//        VALU / SALU mix reaches);
//      - the base comes back by SUBTRACTING the same integers (ds_sub_u32): nothing to remember, nothing read;
//      - an accumulator is TWO 16-bit fields, one per wave of the item: each wave overlays its own queries in its own
//        field (the base sits in both), so two waves share 8 KiB of accumulators without seeing each other's sums --
//        28 waves per CU instead of 16 (the fp32 overlay owned its tile).  (First version, measured: four waves adding
//        into ONE 32-bit sum -- conservative too, what another wave has added only makes a sum larger -- but a
//        neighbour's rare-term contribution is as large as the bound's margin: about one false candidate per (tile,
//        query) pair, 30 x the kernel time in list traffic.)
//    The value an atomic returns plus the lane's own contribution is an UPPER bound of the integer image of the doc's
//    score as far as the query's postings have been added; the LAST posting of a doc (LDS executes a wave's instructions
//    in order) sees all of them.  A doc whose exact fp32 score E reaches the query's bound G satisfies
//      trunc(G * 2^F) <= E * 2^F <= U + T + T * 2^-23 * S * 2^F   (U the integer sum; each of the T truncations loses < 1; the
//    fp32 sum of T non-negative terms exceeds the real sum by at most (T - 1) relative roundings; S = the largest
//    weight sum of a query), so the test `U + slack >= trunc(G * 2^F)` never loses a doc.  The postings that pass it -- a
//    handful per query and shard once the bound stands -- are written to a list as (query row, term position, doc), and
//    sa_k_bm25_fx_rescore forms the exact score of every listed doc: one thread per record finds the doc's posting of every
//    query term in its tile slice and adds `fl(factor * w)` in query-term order with the reference's fp32 operations
//    (bm25.pyx:19-23 through the impact stream; np.sum's row order); the record of the doc's LAST term position reports
//    it (a doc is listed once per posting that passed), if the exact score reaches the bound.  What reaches the candidate
//    lists, the histogram and the merge is bit for bit what the fp32 overlay produced.
//  * Every posting of every query term is still read and enters the sum; nothing is skipped on a score bound.
//
// A (tile, query) pair whose bound is not above the base values yet (or that has no bound), or whose further terms have more
// than SA_GRP_NH halves in the tile, goes on the work list of the per-query kernel (sa_k_bm25_tiles_wl), as before.
#pragma once

#define SA_GFX_NW 2            // waves per (tile, group) item: one 16-bit field of the accumulators each
#ifndef SA_GFX_NHP
#define SA_GFX_NHP 8           // halves of a query a wave holds in registers at once (one pass)
#endif
#ifndef SA_GFX_MINW
#define SA_GFX_MINW 8          // waves per SIMD the register allocation aims at
#endif
#define SA_GFX_MAXQ 16         // queries per wave

__device__ __forceinline__ u32 sa_f2u_rz(float x) { return __float2uint_rz(x); }     // v_cvt_u32_f32: truncates, saturates, NaN -> 0


// record of a posting that passed the filter: doc (local to the shard, 28 bits) | term position << 28 | device row << 33
#define SA_FXC_TERM_SHIFT 28
#define SA_FXC_ROW_SHIFT 33
#define SA_GFX_DES_SHIFT 59    // half descriptor: the half belongs to the query's designated term position (its postings count docs for the bound)
#define SA_FXC_LISTS 64        // sub-lists, each with its own cursor (cursor i at fxc_cnt[i * SA_FXC_CNT_STRIDE]: a cache line each)
#define SA_FXC_CNT_STRIDE 32

// TILE: docs per index tile (the granularity of the batch's slice table); ST: index tiles per item -- an item scores TILE * ST docs
template <int TILE, int IDFN, int ST>
__global__ void __launch_bounds__(SA_GFX_NW * 64, (ST > 1 ? 4 : SA_GFX_MINW)) sa_k_bm25_group_fx(const Bm25Params p, const GroupParams gp) {
    constexpr int NH = ST > 1 ? 14 : SA_GRP_NH, NHP = SA_GFX_NHP, NW = SA_GFX_NW, ITEM = TILE * ST;     // (NH: halves of a query the table holds)
    static_assert(NH < (int)SA_GRPH_OVER && NHP <= NH, "half counts");
    __shared__ alignas(16) u32 accu[ITEM];                      // the item's integer accumulators, shared by its waves
    __shared__ u64 s_half_all[NW][SA_GFX_MAXQ][NH];
    __shared__ float s_idf_all[NW][IDFN];
    __shared__ u32 s_bmax[NW];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u64 (*const s_half)[NH] = s_half_all[wave];
    float* const s_idf = s_idf_all[wave];
    // item order: as sa_k_bm25_group_tiles (block b runs on XCD b % 8; an XCD walks a RANGE of consecutive tiles)
    const u32 per = 8u * gp.n_groups;
    const u32 chunk = blockIdx.x / per, r = blockIdx.x % per;
    const u32 g = r >> 3;
    const u32 trel = gp.tpx ? (r & 7u) * gp.tpx + chunk : chunk * 8u + (r & 7u);
    if (trel >= gp.n_items_run) return;
    const u32 tile = gp.tile0 + trel * (u32)ST;                 // the item's first index tile
    const u32 tspan = p.n_tiles - tile < (u32)ST ? p.n_tiles - tile : (u32)ST;   // its index tiles (the last item may hold fewer)
    const u32 row0 = gp.grp[3 * g], n_raw = gp.grp[3 * g + 1], n_all = n_raw & 0x7FFFFFFFu;
    const u32 dslot = gp.dense ? gp.grp[3 * g + 2] : 0xFFFFFFFFu;
    const bool loose = (n_raw >> 31) != 0u;
    const u32 T = p.T;
    const u64 tile_base = (u64)tile * TILE;
    const u32 tile_base_b = (u32)tile_base * 4u;
    const u64* const stream = p.imp;
    const float scale = gp.fx_scale;
    auto at = [&](u32 byte_off) -> u32& { return *(u32*)((char*)accu + byte_off); };
    auto ballot = [](bool c) -> u64 { return (u64)__builtin_amdgcn_ballot_w64(c); };

    // this wave's queries: rows [row0 + q0, row0 + q0 + n) of the group
    const u32 per_wave = (n_all + (u32)NW - 1u) / (u32)NW;
    const u32 q0 = wave * per_wave;
    const u32 n = q0 < n_all ? (n_all - q0 < per_wave ? n_all - q0 : per_wave) : 0u;
    const u32 wrow0 = row0 + q0;

    // ---- the shared first term
    const u32 qt0 = row0 * T;
    const u32* hrow = p.bounds + (u64)qt0 * (p.n_tiles + 1) + tile;
    const u32 h0 = loose ? 0u : hrow[0], h1 = loose ? 0u : hrow[tspan];
    const sa_u64x2 hbs = ((const sa_u64x2*)p.qbase_imp)[qt0];
    const float hidf = p.idf[qt0];
    const float hidf_s = __fmul_rn(hidf, scale);

    // ---- half tables of this wave's queries (as sa_k_bm25_group_tiles: lane (qi, t) looks up term t's slice of query qi)
    const u32 TT = gp.tt, tsh = gp.tt_shift, QPP = 64u >> tsh;
    struct Pre { u32 r0, r1; u64 base; float idf; u32 des; };
    auto pre_load = [&](u32 ps) -> Pre {
        Pre x; x.r0 = 0; x.r1 = 0; x.base = 0; x.idf = 0.f; x.des = 0u;
        const u32 qi = ps * QPP + (lane >> tsh), t = (loose ? 0u : 1u) + (lane & (TT - 1u));
        if (qi < n && t < T) {
            const u32 qt = (wrow0 + qi) * T + t;
            const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
            x.base = ((const sa_u64x2*)p.qbase_imp)[qt].x;
            x.r0 = row[0]; x.r1 = row[tspan]; x.idf = p.idf[qt];
            x.des = gp.qdes[wrow0 + qi] == t ? 1u : 0u;
        }
        return x;
    };
    auto pre_store = [&](u32 ps, const Pre& x) {
        const u32 qi = ps * QPP + (lane >> tsh), tl = lane & (TT - 1u);
        const u32 np = x.r1 - x.r0;
        const u32 halves = (np + 63u) >> 6;
        u32 incl = halves;
        for (u32 o = 1; o < TT; o <<= 1) {
            const u32 up = __shfl_up(incl, o, SA_WAVE);
            if (tl >= o) incl += up;
        }
        const u32 excl = incl - halves;
        const u32 total = (u32)__shfl((int)incl, (int)(lane | (TT - 1u)), SA_WAVE);
        if (qi < n) {
            s_idf[qi * TT + tl] = x.idf;
            const u64 c0 = x.base + x.r0;
            const u64 nhf = (u64)(total <= (u32)NH ? total : SA_GRPH_OVER) << SA_GRPH_NH_SHIFT;
            for (u32 j = 0; j < halves && excl + j < (u32)NH; j++) {
                const u64 c = c0 + (u64)j * 64ull;
                const u64 lim = (np - j * 64u < 64u ? np - j * 64u : 64u) - 1u;
                s_half[qi][excl + j] = (u64)(stream + c) | (lim << SA_GRPH_LIM_SHIFT) | ((u64)tl << SA_GRPH_TERM_SHIFT) |
                                       ((u64)x.des << SA_GFX_DES_SHIFT) | (excl + j == 0u ? nhf : 0ull);
            }
            if (tl == 0u && total == 0u) s_half[qi][0] = (u64)(stream + p.imp_tail);
        }
    };
    const u32 NP = (n + QPP - 1u) / QPP;                        // 0, 1 or 2 passes (host: 16 * TT <= 128)
    if (NP) {
        const Pre x0 = pre_load(0);
        Pre x1 = x0;
        if (NP > 1u) x1 = pre_load(1);
        pre_store(0, x0);
        if (NP > 1u) pre_store(1, x1);
    }

    // ---- base: the first term's integer image, built by all waves of the item (write-only: 0 + s0 = s0)
    u32 lmax = 0;
    if (dslot != 0xFFFFFFFFu) {
        // dense factor row: thread = 4 docs per 16-byte load and LDS store (docs without the term hold 0.0 -> 0)
        const float4* row4 = (const float4*)(gp.dense + (u64)dslot * gp.dense_stride + tile_base);
        uint4* a4 = (uint4*)accu;
        float4 v[ITEM / (256 * NW)];
#pragma unroll
        for (int j = 0; j < ITEM / (256 * NW); j++) v[j] = row4[j * 64 * NW + (int)threadIdx.x];
#pragma unroll
        for (int j = 0; j < ITEM / (256 * NW); j++) {
            uint4 w;
            w.x = sa_f2u_rz(__fmul_rn(v[j].x, hidf_s)); w.y = sa_f2u_rz(__fmul_rn(v[j].y, hidf_s));
            w.z = sa_f2u_rz(__fmul_rn(v[j].z, hidf_s)); w.w = sa_f2u_rz(__fmul_rn(v[j].w, hidf_s));
            const u32 m0 = w.x > w.y ? w.x : w.y, m1 = w.z > w.w ? w.z : w.w;
            const u32 m = m0 > m1 ? m0 : m1;
            lmax = m > lmax ? m : lmax;
            w.x *= 0x10001u; w.y *= 0x10001u; w.z *= 0x10001u; w.w *= 0x10001u;      // (the base in both fields)
            a4[j * 64 * NW + (int)threadIdx.x] = w;
        }
    } else {
        uint4* a4 = (uint4*)accu;
#pragma unroll
        for (int j = 0; j < ITEM / (256 * NW); j++) a4[j * 64 * NW + (int)threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        // (one posting per thread and step: the slice's docs are pairwise distinct, all inside the tile)
        const u64 lo = hbs.x + h0;
        const u32 np = h1 - h0;
        for (u32 i = threadIdx.x; i < np; i += 64u * NW) {
            const u64 c = stream[lo + i];
            const u32 w0 = sa_f2u_rz(__fmul_rn(__uint_as_float((u32)c), hidf_s));
            at((u32)(c >> 32) - tile_base_b) = w0 * 0x10001u;
            lmax = w0 > lmax ? w0 : lmax;
        }
    }
    {
        const u32 wm = sa_wave_max_u32(lmax);
        if (lane == 0) s_bmax[wave] = wm;
    }
    __syncthreads();
    u32 base_max = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) base_max = s_bmax[w] > base_max ? s_bmax[w] : base_max;
    if (n == 0u) return;                                        // (a group smaller than the item's waves)
    // From here on the waves of the item do not wait for each other; lanes of ONE wave hand data to each other through
    // its own LDS tables without s_barrier (a wave's LDS instructions execute in program order).
    {
        const u32 mine = lane < n ? (u32)(s_half[lane][0] >> SA_GRPH_NH_SHIFT) : 0u;
        if (base_max == 0u && ballot(mine != 0u) == 0ull) return;   // nothing to score in this tile at all
    }

    // the queries' bounds, one per lane (a bound only ever rises: a stale one is valid), and their integer images
    u32 thr_all = lane < n ? __hip_atomic_load(&p.gthr[wrow0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (p.seed && lane < n) { const u32 sd = p.seed[wrow0 + lane]; thr_all = sd > thr_all ? sd : thr_all; }
    u32 thr_fx_all;
    {
        const u32 tf = sa_f2u_rz(__fmul_rn(__uint_as_float(thr_all), scale));
        thr_fx_all = tf > gp.fx_slack ? tf - gp.fx_slack : 0u;   // a doc at or above the bound has U >= this
    }

    typedef const __attribute__((address_space(1))) u64* gptr_u64;
    const u32 lane8 = lane * 8u;
    const u32 fsh = wave * 16u;                                 // this wave's field of the accumulators
    u64 deferred = 0ull;
    // ---- the wave's queries, one after the other
    for (u32 qi = 0; qi < n; qi++) {
        // descriptors: lane h holds half h's cell, the weight of its term (exact, and pre-scaled by 2^F)
        const u64 dsc = s_half[qi][lane < (u32)NH ? lane : 0u];
        const u32 dlo = (u32)dsc, dhi = (u32)(dsc >> 32);
        const float wx = s_idf[qi * TT + ((dhi >> (SA_GRPH_TERM_SHIFT - 32)) & 0x1Fu)];
        const float ws = __fmul_rn(wx, scale);
        const u32 nh_raw = (u32)__builtin_amdgcn_readfirstlane((int)dhi) >> (SA_GRPH_NH_SHIFT - 32);
        const u32 thr_q = (u32)__builtin_amdgcn_readlane((int)thr_all, (int)qi);
        const u32 thr_fx = (u32)__builtin_amdgcn_readlane((int)thr_fx_all, (int)qi);
        if (nh_raw > (u32)NH || thr_q == 0u || base_max >= thr_fx) { deferred |= 1ull << qi; continue; }
        if (nh_raw == 0u) continue;                             // the query scores exactly the base here: all below its bound
        // Passes of up to NHP halves (a query of the BASELINE shape: one).  All passes ADD before anything is taken back, so the
        // last posting of a doc sees all of the query's contributions whatever pass holds it; the registers of the last pass
        // take their contributions back, earlier passes (rare: more than 512 postings of one query in one tile) are read again.
        for (u32 hb = 0; hb < nh_raw; hb += (u32)NHP) {
            const u32 nh = nh_raw - hb < (u32)NHP ? nh_raw - hb : (u32)NHP;
            const bool last_pass = hb + (u32)NHP >= nh_raw;
            // one straight-line body per number of halves (N = 1 .. NHP): no test per half and phase
            auto pass = [&](auto nc) {
                constexpr int N = decltype(nc)::value;
                u64 v[N];
                bool ok[N];                                     // this lane holds a posting of the half
                u32 sl[N], cc[N], sum[N];                       // the docs' accumulators (byte offsets), the contributions, the integer sums
#pragma unroll
                for (int h = 0; h < N; h++) {
                    const u32 hi = (u32)__builtin_amdgcn_readlane((int)dhi, (int)hb + h);
                    const u64 a = (u64)(u32)__builtin_amdgcn_readlane((int)dlo, (int)hb + h) | ((u64)(hi & 0xFFFFu) << 32);
                    const u32 lim8 = ((hi >> (SA_GRPH_LIM_SHIFT - 32)) & 0x3Fu) << 3;
                    ok[h] = lane8 <= lim8;
                    v[h] = *(gptr_u64)((const __attribute__((address_space(1))) char*)a + (lane8 < lim8 ? lane8 : lim8));   // lanes past the end: copies of the last posting (they add 0)
                }
                __builtin_amdgcn_sched_barrier(0);              // (all of the pass's loads are requested before the first one is waited for)
#pragma unroll
                for (int h = 0; h < N; h++) {
                    const float w = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ws), (int)hb + h));
                    const u32 c = sa_f2u_rz(__fmul_rn(__uint_as_float((u32)v[h]), w));
                    sl[h] = (u32)(v[h] >> 32) - tile_base_b;
                    cc[h] = c << fsh;
                }
                // (lanes past a half's end sit the atomics out: their copies of the last posting would all hit ONE address, and
                //  an LDS atomic takes same-address lanes one after the other)
#pragma unroll
                for (int h = 0; h < N; h++) {
                    sum[h] = 0u;
                    if (ok[h]) sum[h] = atomicAdd(&at(sl[h]), cc[h]);    // (... and all of the pass's atomics are in flight together)
                }
                __builtin_amdgcn_sched_barrier(0);
                u32 wmax = 0u;
#pragma unroll
                for (int h = 0; h < N; h++) {
                    sum[h] = ((sum[h] + cc[h]) >> fsh) & 0xFFFFu;   // (a lane past the half's end: its own contribution alone -- below any bound)
                    wmax = sum[h] > wmax ? sum[h] : wmax;
                }
                if (ballot(wmax >= thr_fx) != 0ull) {           // (rare once the bound stands)
                    // Some doc's integer sum reaches the image of the bound.  The accumulators hold all of the pass's
                    // contributions now: every posting whose doc's sum reaches the image goes on the rescoring list (all
                    // postings of such a doc, whatever order the additions took: its last term position is among them).
                    __builtin_amdgcn_wave_barrier();
                    u32 qrow = wrow0 + qi;
                    SA_OPAQUE_U32(qrow);                        // (nothing of this path is prepared outside it)
                    u64 fm[N];
                    u32 tot = 0;
#pragma unroll
                    for (int h = 0; h < N; h++) {
                        fm[h] = ballot(ok[h] && ((at(sl[h]) >> fsh) & 0xFFFFu) >= thr_fx);
                        tot += (u32)__popcll(fm[h]);
                    }
                    // The bound rises while the kernel runs: a flagged posting of the query's DESIGNATED term position counts its
                    // doc -- once: a position's docs are pairwise distinct -- in the query's histogram at a LOWER bound of the
                    // doc's exact score (the integer sum less what fp32 rounding can take: the contributions are truncated, so
                    // the sum is at most the real sum's image), and whenever a query's count crosses a multiple of
                    // SA_GRP_REFRESH_STEP its bound is re-derived: k counted docs score at least the bin edge, exactly.
                    {
                        u32 ndes = 0;
#pragma unroll
                        for (int h = 0; h < N; h++) {
                            const u32 hi = (u32)__builtin_amdgcn_readlane((int)dhi, (int)hb + h);
                            if (fm[h] && ((hi >> (SA_GFX_DES_SHIFT - 32)) & 1u)) {
                                if ((fm[h] >> lane) & 1ull) {
                                    const u32 f = (at(sl[h]) >> fsh) & 0xFFFFu;
                                    const u32 lo = f > gp.fx_slack_lo ? f - gp.fx_slack_lo : 0u;
                                    const u32 lb = __float_as_uint(__fdiv_rn((float)lo, scale));
                                    atomicAdd(&p.hist[(u64)qrow * SA_HBINS + sa_score_bin(lb)], 1u);
                                }
                                ndes += (u32)__popcll(fm[h]);
                            }
                        }
                        if (ndes) {
                            u32 old = 0;
                            if (lane == 0u) old = atomicAdd(&p.slots[(u64)qrow * 32u], ndes);
                            old = (u32)__builtin_amdgcn_readfirstlane((int)old);
                            if (old / (u32)SA_GRP_REFRESH_STEP != (old + ndes) / (u32)SA_GRP_REFRESH_STEP)
                                sa_hist_refresh(p.hist + (u64)qrow * SA_HBINS, &p.gthr[qrow], p.k, lane);
                        }
                    }
                    if (tot) {
                        // one reservation per pass, in the sub-list of this workgroup (SA_FXC_LISTS cursors: one address would serialise)
                        const u32 sub = blockIdx.x & (u32)(SA_FXC_LISTS - 1);
                        u32 fbase = 0;
                        if (lane == 0u) fbase = atomicAdd(gp.fxc_cnt + sub * SA_FXC_CNT_STRIDE, tot);
                        fbase = (u32)__builtin_amdgcn_readfirstlane((int)fbase);
#pragma unroll
                        for (int h = 0; h < N; h++) {
                            if (fm[h]) {
                                const u32 hi = (u32)__builtin_amdgcn_readlane((int)dhi, (int)hb + h);
                                const u32 tpos = ((hi >> (SA_GRPH_TERM_SHIFT - 32)) & 0x1Fu) + (loose ? 0u : 1u);
                                const u32 pos = fbase + (u32)__popcll(fm[h] & ((1ull << lane) - 1ull));
                                if (((fm[h] >> lane) & 1ull) && pos < gp.fxc_cap)
                                    gp.fxc[(u64)sub * gp.fxc_cap + pos] = (u64)((u32)tile_base + (sl[h] >> 2)) | ((u64)tpos << SA_FXC_TERM_SHIFT) |
                                                                         ((u64)qrow << SA_FXC_ROW_SHIFT);
                                fbase += (u32)__popcll(fm[h]);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (last_pass) {                                // the base comes back: subtract what was added
#pragma unroll
                    for (int h = 0; h < N; h++) if (ok[h]) atomicSub(&at(sl[h]), cc[h]);
                }
            };
            static_assert(NHP == 8, "pass dispatch");
            switch (nh) {
                case 1: pass(std::integral_constant<int, 1>{}); break;
                case 2: pass(std::integral_constant<int, 2>{}); break;
                case 3: pass(std::integral_constant<int, 3>{}); break;
                case 4: pass(std::integral_constant<int, 4>{}); break;
                case 5: pass(std::integral_constant<int, 5>{}); break;
                case 6: pass(std::integral_constant<int, 6>{}); break;
                case 7: pass(std::integral_constant<int, 7>{}); break;
                default: pass(std::integral_constant<int, 8>{}); break;
            }
        }
        if (nh_raw > (u32)NHP) {
            // the earlier passes' contributions, formed again from the postings
            const u32 early = ((nh_raw - 1u) / (u32)NHP) * (u32)NHP;
            for (u32 h = 0; h < early; h++) {
                const u32 hi = (u32)__builtin_amdgcn_readlane((int)dhi, (int)h);
                const u64 a = (u64)(u32)__builtin_amdgcn_readlane((int)dlo, (int)h) | ((u64)(hi & 0xFFFFu) << 32);
                const u32 lim = (hi >> (SA_GRPH_LIM_SHIFT - 32)) & 0x3Fu;
                const u64 v = ((gptr_u64)a)[lane < lim ? lane : lim];
                const float w = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ws), (int)h));
                const u32 c = sa_f2u_rz(__fmul_rn(__uint_as_float((u32)v), w));
                if (lane <= lim) atomicSub(&at((u32)(v >> 32) - tile_base_b), c << fsh);
            }
        }
    }
    // ---- general path: hand the (tile, query) pairs to the per-query kernel that follows (sa_k_bm25_tiles_wl)
    if (deferred) {
        const u32 c = (u32)__popcll(deferred) * tspan;
        u32 wbase = 0;
        if (lane == 0) wbase = atomicAdd(gp.wl_cnt, c);
        wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
        if ((deferred >> lane) & 1ull)
            for (u32 ts = 0; ts < tspan; ts++)
                gp.wl[wbase + (u32)__popcll(deferred & ((1ull << lane) - 1ull)) * tspan + ts] = ((u64)(tile + ts) << 32) | (u64)(wrow0 + lane);
    }
}

// The exact scores of the listed docs: one thread per record (doc, term position, row).  The doc's posting of every query term
// is searched in the term's slice of the doc's tile (the batch's slice table), `fl(factor * w)` is added in query-term order --
// the reference's `((s0 + s1) + s2) + s3` with +0.0 for absent terms, which changes nothing (similarity.py:19-38 through the
// impact stream, np.sum's row order: test/test_msmarco.py:353) -- and the record of the doc's LAST term position reports the doc
// if its score reaches the query's bound: it joins the query's candidate list (the bound's histogram was fed by the filter).  A list that ran
// over raises the run's redo flag (the batch is then redone without bounds).
__global__ void __launch_bounds__(256)
sa_k_bm25_fx_rescore(const Bm25Params p, const u64* __restrict__ lists, const u32* __restrict__ cnt_p, u32 cap, u32 tile_shift,
                     u32* __restrict__ overflow) {
    // block b walks sub-list b % SA_FXC_LISTS with the other blocks of that residue
    const u32 sub = blockIdx.x & (u32)(SA_FXC_LISTS - 1);
    const u32 cnt = cnt_p[sub * SA_FXC_CNT_STRIDE];
    if (cnt > cap && blockIdx.x < (u32)SA_FXC_LISTS && threadIdx.x == 0) {
        if (overflow) atomicMax(overflow, 1u);
        else atomicMax(&p.cand_cnt[0], 0xFFFFFFFFu);            // (a run whose lists are checked by the host: reads as an overflowing candidate list)
    }
    const u32 n = cnt < cap ? cnt : cap;
    const u64* const list = lists + (u64)sub * cap;
    const u32 nb = gridDim.x / (u32)SA_FXC_LISTS;
    for (u32 i = (blockIdx.x / (u32)SA_FXC_LISTS) * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
        const u64 rec = list[i];
        const u32 d = (u32)rec & 0x0FFFFFFFu, tflag = (u32)(rec >> SA_FXC_TERM_SHIFT) & 31u, q = (u32)(rec >> SA_FXC_ROW_SHIFT);
        const u32 tile = d >> tile_shift, d4 = d * 4u;
        float acc = 0.f;
        u32 last = 0xFFFFFFFFu;
        for (u32 t = 0; t < p.T; t++) {
            const u32 qt = q * p.T + t;
            const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
            const u64 b0 = ((const sa_u64x2*)p.qbase_imp)[qt].x;
            u64 lo = b0 + row[0], hi = b0 + row[1];
            const u64 end = hi;
            while (lo < hi) {
                const u64 mid = lo + ((hi - lo) >> 1);
                if ((u32)(p.imp[mid] >> 32) < d4) lo = mid + 1; else hi = mid;
            }
            if (lo < end) {
                const u64 cell = p.imp[lo];
                if ((u32)(cell >> 32) == d4) {
                    acc = __fadd_rn(acc, __fmul_rn(__uint_as_float((u32)cell), p.idf[qt]));
                    last = t;
                }
            }
        }
        if (last != tflag) continue;                            // (another record of this doc reports it, or none has to)
        const u32 fin = __float_as_uint(acc);
        u32 thr = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.seed) { const u32 sd = p.seed[q]; thr = sd > thr ? sd : thr; }
        if (fin < (thr > 1u ? thr : 1u)) continue;
        const u32 pos = atomicAdd(&p.cand_cnt[q], 1u);                // (the doc was counted for the query's bound by sa_k_bm25_group_fx)
        const u64 doc = p.doc_base + (u64)d;
        if (pos < p.cand_cap) p.cand[(u64)q * p.cand_cap + pos] = ((u64)fin << 32) | (u64)(u32)(~(u32)doc);
    }
}

// sa_bm25_hg.hip -- the head-group kernel: exhaustive BM25 + top-k for queries that share their first term,
// with NO per-query accumulators.
//
// The reference scores a query term by term into dense vectors and sums them (postings.py:652-680,
// bm25.pyx:11-25, test/test_msmarco.py:353-354), i.e. a doc's score is ((s0 + s1) + s2) + s3 in query-term order
// with +0.0 for the terms it lacks.  The grouped kernel of sa_bm25.hip forms those sums by a read-modify-write of
// per-(tile, group) LDS accumulators, query after query, and puts the base values back after each query.  Measured
// (rocprofv3 SQ counters, both kernels): the time of these kernels is their VALU + SALU instruction count x 4 cycles /
// 1024 SIMDs -- so this kernel is built to spend few instructions per (tile, query) pair, and to move what it can to
// the LDS pipe:
//
//   super-tile  a workgroup scores ST index tiles (4096 docs) at once: what a pair costs is mostly fixed (descriptors,
//               address selection), so the pairs are made bigger
//   base        the group's shared first term, scored ONCE per (super-tile, group) into LDS by the whole workgroup
//               (dense factor row: lane = doc, 16-byte loads; or a scatter of its postings); read-only afterwards,
//               so ALL waves of the workgroup share it and each takes its own queries of the group
//   candidates  the query's sparse further terms (all but its longest list): their postings are the docs that can
//               hold more than one further term.  Each is registered in the wave's CANDIDATE MAP in LDS -- one byte
//               per doc of the super-tile: 0, or the candidate's index + 1 -- by its first list; a later list that
//               finds its doc registered hands its value to that entry's mailbox instead (the doc has one owner)
//   stream      the query's longest further list S: every posting is loaded (coalesced 8-byte loads, the next chunk
//               of 256 requested before the current one is scored) and scored against the base -- base + s is the
//               complete score of a doc that holds nothing else of the query -- and looks its doc up in the candidate
//               map: a registered doc gets S's factor into its mailbox (one LDS write)
//   sum         every candidate adds up its doc IN QUERY-TERM ORDER -- base, own value, mailbox values, +0.0 for what
//               the doc lacks: bit for bit the reference's sum -- compares with the query's bound, and clears its
//               map byte and mailbox.
//
// Every posting of every query term is read and scored; nothing is skipped on a score bound.  A (super-tile, query)
// pair whose base values reach the query's bound (early tiles, before the bound stands) or whose candidate lists do
// not fit the map goes to the per-query kernel through the work list, like the grouped kernel's pairs.  Survivors
// (score >= bound) are appended to the query's candidate list and counted in its histogram exactly as
// sa_tile_topk_hist does, so the merge and the exactness argument are unchanged.
//
// Roofline: HBM-bound integer/bitwise + scalar fp32 work, no MFMA.  Compulsory bytes per launch = every distinct
// posting list once.
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_bm25_params.hpp"

#include <algorithm>
#include <stdlib.h>

#define SA_HG_NW 4          // waves per workgroup: they share the super-tile's base
#define SA_HG_QPR 16        // queries per wave and round (lane = 4 * query + slot)
#define SA_HG_CH 4          // 64-posting vectors per chunk of the stream pass
#define SA_HG_NCV 3         // candidate vectors of a (super-tile, query) pair: two of the first candidate list, one of the second
#define SA_HG_CAP (SA_HG_NCV * 64)

static int sa_hg_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// what the kernel needs of a batch (a compact argument block: everything in it is wave-uniform and most of it stays in
// scalar registers for the whole kernel)
struct HgArgs {
    const u64* imp; u64 imp_tail;
    const u32* bounds; const u64* qbase_imp; const float* idf; const u32* qrole; const u32* grp;
    const float* dense; u64 dense_stride;
    u32* gthr; const u32* seed; u32* hist; u64* cand; u32* cand_cnt; u64* wl; u32* wl_cnt;
    u64 doc_base;
    u32 n_tiles, T, k, cand_cap, group0, n_groups, tile0, n_tiles_run;
    u32 spx;                // super-tiles per XCD (an XCD takes a range of consecutive super-tiles), 0: dealt round-robin
};

// survivors of a vector -> the query's histogram and candidate list (as sa_tile_topk_hist); rare once the bounds stand, so
// it is a call, not inline code in every vector's path
__device__ __forceinline__ void sa_hg_emit(const HgArgs& a, u32 qrow, bool sv, u32 vbits, u64 doc, u32 lane) {
    const u64 mbits = (u64)__builtin_amdgcn_ballot_w64(sv);
    const u32 c = (u32)__popcll(mbits);
    u32 cbase = 0;
    if (lane == 0) cbase = atomicAdd(&a.cand_cnt[qrow], c);
    cbase = (u32)__builtin_amdgcn_readfirstlane((int)cbase);
    if (sv) {
        atomicAdd(&a.hist[(u64)qrow * SA_HBINS + sa_score_bin(vbits)], 1u);
        const u32 pos = cbase + (u32)__popcll(mbits & ((1ull << lane) - 1ull));
        if (pos < a.cand_cap) a.cand[(u64)qrow * a.cand_cap + pos] = ((u64)vbits << 32) | (u64)(u32)(~(u32)doc);
    }
    if (cbase / 32u != (cbase + c) / 32u) sa_hist_refresh(a.hist + (u64)qrow * SA_HBINS, &a.gthr[qrow], a.k, lane);
}

// the sum of a doc in query-term order: base, then the three further positions, which hold (own, s, oth) in the order `ord`
// (0 own,s,oth  1 own,oth,s  2 s,own,oth  3 s,oth,own  4 oth,own,s  5 oth,s,own); `ord` is wave-uniform, so the
// selections are v_cndmask on scalar conditions, no branches
__device__ __forceinline__ u32 sa_hg_fold(u32 ord, float bse, float own, float s, float oth) {
    const float x1 = ord < 2u ? own : (ord < 4u ? s : oth);
    const float x2 = (ord == 2u || ord == 4u) ? own : ((ord == 0u || ord == 5u) ? s : oth);
    const float x3 = (ord == 3u || ord == 5u) ? own : ((ord == 1u || ord == 4u) ? s : oth);
    return __float_as_uint(__fadd_rn(__fadd_rn(__fadd_rn(bse, x1), x2), x3));
}

__global__ void __launch_bounds__(SA_HG_NW * 64, 4) sa_k_bm25_headgroup(const HgArgs a) {
    constexpr int NW = SA_HG_NW, QPR = SA_HG_QPR, CH = SA_HG_CH, CAP = SA_HG_CAP, ST = SA_HG_ST;
    constexpr int THREADS = NW * 64;
    constexpr int TD = ST * 2048;                               // docs of a super-tile
    __shared__ alignas(16) float s_acc[TD];
    __shared__ u32 s_red[NW];
    __shared__ alignas(16) unsigned char s_cmap[NW][TD];        // per wave: 0, or 1 + index of the candidate that owns the doc; all zero between queries
    __shared__ alignas(16) u32 s_mb[NW][CAP + 1][2];            // per wave: mailbox of candidate j at [j + 1] -- [0] S's factor, [1] the second list's value; entry 0 takes what is addressed to nobody
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // (wave-uniform: everything derived from it stays scalar)
    // XCD-aware item order (as the grouped kernel): block b runs on XCD b % 8 -- the eight super-tiles of a chunk sit on
    // eight XCDs and all groups of a super-tile follow each other on the same XCD, so the lists that queries of different
    // groups share are fetched into one L2 only.
    const u32 per = 8u * a.n_groups;
    const u32 chunk = blockIdx.x / per, r = blockIdx.x % per;
    const u32 g = a.group0 + (r >> 3);
    const u32 srel = a.spx ? (r & 7u) * a.spx + chunk : chunk * 8u + (r & 7u);     // (an XCD walks a range of super-tiles: see the grouped kernel)
    if (srel * (u32)ST >= a.n_tiles_run) return;
    const u32 tile_lo = a.tile0 + srel * (u32)ST;
    const u32 tile_end = a.tile0 + a.n_tiles_run;
    const u32 tile_hi = tile_lo + (u32)ST < tile_end ? tile_lo + (u32)ST : tile_end;
    const u32 row0 = a.grp[3 * g], n = a.grp[3 * g + 1] & 0x7FFFFFFFu;
    const u32 T = a.T;
    const u64 tile_base = (u64)tile_lo * 2048ull;
    const u32 tb4 = (u32)tile_base * 4u;
    const u64* const stream = a.imp;
    const u64 sentinel = (u64)(stream + a.imp_tail);          // a cell that can always be loaded (doc field all ones)
    typedef const __attribute__((address_space(1))) char* gptr_c;
    auto rl = [](u32 v, u32 l) -> u32 { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); };
    auto rlf = [](float v, u32 l) -> float { return __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)l)); };
    auto ballot = [](bool c) -> u64 { return (u64)__builtin_amdgcn_ballot_w64(c); };
    // cell [off8 / 8] of a list: a uniform base and a 32-bit byte offset per lane (the load's scalar-base form)
    auto cell_at = [](u64 base, u32 off8) -> u64 { return *(const __attribute__((address_space(1))) u64*)((gptr_c)base + off8); };
    auto base_at = [&](u32 off4) -> float { return *(const float*)((const char*)s_acc + off4); };
    unsigned char* const cmap = &s_cmap[wave][0];
    u32* const mb = &s_mb[wave][0][0];

    // ---- per-lane tables of a round, ROLE-major: lane 4 * i + s of the wave's i-th query of the round holds, for
    //      s = 0 the stream list, s = 1 / 2 the first / second candidate list (first cell, postings in this super-tile, weight; a
    //      list the query does not have: 0 postings at a cell that can always be loaded), s = 3 the query's bound and the
    //      orders of its sums
    struct Round { u32 alo, ahi, ns; float w; u32 x; u64 todo, valid; };
    auto load_round = [&](u32 rbase) -> Round {
        Round R;
        const u32 qloc = rbase + wave + (u32)NW * (lane >> 2), t = lane & 3u;
        const bool qv = qloc < n && t < T;
        u32 r0 = 0, r1 = 0, role = 0, thr = 0;
        u64 cb = 0;
        float w = 0.f;
        if (qv) {
            const u32 qt = (row0 + qloc) * T + t;
            role = a.qrole[qt];
            const u32* row = a.bounds + (u64)qt * (a.n_tiles + 1);
            r0 = row[tile_lo]; r1 = row[tile_hi];
            cb = a.qbase_imp[2 * (u64)qt];
            w = a.idf[qt];
            if (t == 0u) {
                thr = __hip_atomic_load(&a.gthr[row0 + qloc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.seed) { const u32 sd = a.seed[row0 + qloc]; thr = sd > thr ? sd : thr; }
            }
        }
        const u32 kind = role & 0xFu;
        const u64 adr = (u64)(stream + cb + r0);                // the slice's first cell
        const u32 np = r1 - r0;
        // position-major -> role-major: slot s takes the values of the position that has role s + 2 (SA_HG_STREAM, _CAND0, _CAND1)
        const u32 l0 = lane & ~3u, sl = lane & 3u;
        const u64 mS = ballot(kind == SA_HG_STREAM), m0 = ballot(kind == SA_HG_CAND0), m1 = ballot(kind == SA_HG_CAND1);
        const u64 msel = sl == 0u ? mS : (sl == 1u ? m0 : m1);
        const u32 bits = sl == 3u ? 0u : (u32)(msel >> l0) & 0xFu;
        const u32 src = l0 + (bits ? (u32)__builtin_ctz(bits) : 0u);
        const u32 g_alo = (u32)__shfl((int)(u32)adr, (int)src), g_ahi = (u32)__shfl((int)(u32)(adr >> 32), (int)src);
        const u32 g_np = (u32)__shfl((int)np, (int)src);
        const float g_w = __shfl(w, (int)src);
        const u32 g_thr = (u32)__shfl((int)thr, (int)l0), g_role0 = (u32)__shfl((int)role, (int)l0);
        const bool have = bits != 0u && g_np != 0u;
        R.alo = have ? g_alo : (u32)sentinel; R.ahi = have ? g_ahi : (u32)(sentinel >> 32);
        R.ns = have ? g_np : 0u;
        R.w = have ? g_w : 0.f;
        R.x = sl == 3u ? (g_thr > 1u ? g_thr : 1u) : (g_role0 >> 4);     // slot 3: the bound; slots 0 .. 2: the head's role word >> 4 = the sum orders
        R.valid = ballot(qloc < n && sl == 0u);
        const u64 any = ballot(have);
        R.todo = (any | (any >> 1) | (any >> 2)) & 0x1111111111111111ull & R.valid;   // queries with postings here: bit at their slot-0 lane
        return R;
    };
    Round R = load_round(0);                                    // in flight while the base is built

    // ---- base: the shared first term of the group, scored once for all waves
    u32 base_max = 0;
    {
        const u32 dslot = a.grp[3 * g + 2];
        const u32 qt0 = row0 * T;
        const float hidf = a.idf[qt0];
        u32 lmax = 0;
        float4* a4 = (float4*)s_acc;
        if (dslot != 0xFFFFFFFFu) {
            // dense factor row: lane = doc, four docs per 16-byte load and LDS store (a doc without the term holds
            // 0.0 -> 0.0 * idf = +0.0); past the row's end (the last super-tile): zeros
            const float4* row4 = (const float4*)(a.dense + (u64)dslot * a.dense_stride + tile_base);
            const u32 lim4 = (u32)((a.dense_stride - tile_base) / 4ull < (u64)(TD / 4) ? (a.dense_stride - tile_base) / 4ull : (u64)(TD / 4));
            float4 v[TD / (4 * THREADS)];
#pragma unroll
            for (int j = 0; j < TD / (4 * THREADS); j++) {
                const u32 i4 = (u32)j * THREADS + tid;
                v[j] = i4 < lim4 ? row4[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < TD / (4 * THREADS); j++) {
                float4 w;
                w.x = __fmul_rn(v[j].x, hidf); w.y = __fmul_rn(v[j].y, hidf); w.z = __fmul_rn(v[j].z, hidf); w.w = __fmul_rn(v[j].w, hidf);
                a4[j * THREADS + (int)tid] = w;
                const u32 m0 = __float_as_uint(w.x) > __float_as_uint(w.y) ? __float_as_uint(w.x) : __float_as_uint(w.y);
                const u32 m1 = __float_as_uint(w.z) > __float_as_uint(w.w) ? __float_as_uint(w.z) : __float_as_uint(w.w);
                const u32 m = m0 > m1 ? m0 : m1;
                lmax = m > lmax ? m : lmax;
            }
        } else {
#pragma unroll
            for (int j = 0; j < TD / (4 * THREADS); j++) a4[j * THREADS + (int)tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            const u32* hrow = a.bounds + (u64)qt0 * (a.n_tiles + 1);
            const u32 h0 = hrow[tile_lo], h1 = hrow[tile_hi];
            const u64* cells = stream + a.qbase_imp[2 * (u64)qt0];
            for (u32 i = h0 + tid; i < h1; i += (u32)THREADS) {
                const u64 c = cells[i];
                const u32 wbits = __float_as_uint(__fmul_rn(__uint_as_float((u32)c), hidf));
                s_acc[((u32)(c >> 32) - tb4) >> 2] = __uint_as_float(wbits);
                lmax = wbits > lmax ? wbits : lmax;
            }
        }
        const u32 wm = sa_wave_max_u32(lmax);
        if (lane == 0) s_red[wave] = wm;
    }
    {
        // the wave's candidate map and mailboxes start out empty
        u32* cm4 = (u32*)cmap;
#pragma unroll
        for (int j = 0; j < TD / 4 / 64; j++) cm4[j * 64 + (int)lane] = 0u;
        for (u32 j = lane; j < 2u * (u32)(CAP + 1); j += 64u) mb[j] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++) base_max = s_red[w] > base_max ? s_red[w] : base_max;

    // ---- the scalars of one query (fixed lanes of the round's tables)
    struct QC {
        u64 sa, c0a, c1a; u32 sn, c0n, c1n; float ws, c0w, c1w; u32 thr, ord, l0;
    };
    auto qctx = [&](u32 l0) -> QC {
        QC q;
        q.l0 = l0;
        q.sa = (u64)rl(R.alo, l0) | ((u64)rl(R.ahi, l0) << 32);
        q.c0a = (u64)rl(R.alo, l0 + 1u) | ((u64)rl(R.ahi, l0 + 1u) << 32);
        q.c1a = (u64)rl(R.alo, l0 + 2u) | ((u64)rl(R.ahi, l0 + 2u) << 32);
        q.sn = rl(R.ns, l0); q.c0n = rl(R.ns, l0 + 1u); q.c1n = rl(R.ns, l0 + 2u);
        q.ws = rlf(R.w, l0); q.c0w = rlf(R.w, l0 + 1u); q.c1w = rlf(R.w, l0 + 2u);
        q.thr = rl(R.x, l0 + 3u);
        q.ord = rl(R.x, l0);
        return q;
    };
    // Loads: ALWAYS the same number per step (addresses are clamped / replaced by a cell that can always be loaded, never
    // branched around), so that every wait is an exact count.  Lanes past a list's end load copies of its last entry.
    struct SB { u64 v[CH]; };
    struct CB { u64 v0, v1, v2; };
    const u32 lane8 = lane << 3;
    auto issue_stream = [&](u64 sa, u32 sn, u32 off, SB& X) {
        const u32 last8 = sn ? (sn - 1u) << 3 : 0u;
        const u32 o8 = (off << 3) + lane8;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const u32 x = o8 + (u32)j * 512u;
            X.v[j] = cell_at(sa, x < last8 ? x : last8);
        }
    };
    auto issue_cands = [&](const QC& q, CB& X) {
        const u32 l0 = q.c0n ? (q.c0n - 1u) << 3 : 0u, l1 = q.c1n ? (q.c1n - 1u) << 3 : 0u;
        X.v0 = cell_at(q.c0a, lane8 < l0 ? lane8 : l0);
        X.v1 = cell_at(q.c0a, lane8 + 512u < l0 ? lane8 + 512u : l0);
        X.v2 = cell_at(q.c1a, lane8 < l1 ? lane8 : l1);
    };

    // ---- rounds of up to QPR queries per wave
    for (u32 rbase = 0; rbase < n; rbase += (u32)(NW * QPR)) {
        if (rbase) R = load_round(rbase);
        // queries left to the per-query kernel: bit 4 * i (the slot-0 lane of query i).  First those whose bound is not above
        // the base values yet.
        u64 deferred = (ballot((lane & 3u) == 3u && base_max >= R.x) >> 3) & R.valid;
        u64 todo = R.todo & ~deferred;
        if (todo) {
            // The wave's work is a flat sequence of STEPS: one chunk (CH vectors) of a query's stream list each; a query with
            // no stream postings here takes one step too.  A query's first step registers its candidates first (B1), its last
            // step sums them up afterwards (B2).  Every step requests the loads of the step after it -- the next chunk of the
            // same list, or the next query's candidate vectors and first chunk -- before it scores its own chunk, and always the
            // same number of them: 3 candidate vectors (copies of an always-loadable cell unless a new query starts) + CH.
            QC Q = qctx((u32)__builtin_ctzll(todo));
            todo &= todo - 1ull;
            SB SA_, SB_;
            CB CV;
            issue_cands(Q, CV);
            issue_stream(Q.sa, Q.sn, 0u, SA_);
            u32 off = 0u;
            // per-query state of the candidates, set by the first step of a query
            bool fits = true, ok0 = false, ok1 = false, ok2 = false, mine2 = false;
            u32 d0 = 0, d1 = 0, d2 = 0;
            float own0 = 0.f, own1 = 0.f, own2 = 0.f;
            auto step = [&](SB& cur, SB& nxt) -> bool {
                const u32 qrow = row0 + rbase + wave + (u32)NW * (Q.l0 >> 2);
                const bool first = off == 0u;
                const bool last = off + (u32)CH * 64u >= Q.sn;
                if (first) {
                    // ---- B1: the candidates register in the map.  The first list's docs are new; a doc of the second list that
                    //      is already there belongs to the first list's entry, which gets the value.
                    fits = Q.c0n <= 128u && Q.c1n <= 64u;
                    if (!fits) deferred |= 1ull << Q.l0;        // (more candidate postings than the map takes: the per-query kernel)
                    ok0 = lane < Q.c0n && fits; ok1 = lane + 64u < Q.c0n && fits; ok2 = lane < Q.c1n && fits;
                    d0 = ok0 ? ((u32)(CV.v0 >> 32) - tb4) >> 2 : 0u;
                    d1 = ok1 ? ((u32)(CV.v1 >> 32) - tb4) >> 2 : 0u;
                    d2 = ok2 ? ((u32)(CV.v2 >> 32) - tb4) >> 2 : 0u;
                    own0 = __fmul_rn(__uint_as_float((u32)CV.v0), Q.c0w);
                    own1 = __fmul_rn(__uint_as_float((u32)CV.v1), Q.c0w);
                    own2 = __fmul_rn(__uint_as_float((u32)CV.v2), Q.c1w);
                    if (ok0) cmap[d0] = (unsigned char)(lane + 1u);
                    if (ok1) cmap[d1] = (unsigned char)(lane + 65u);
                    __builtin_amdgcn_wave_barrier();
                    const u32 t = ok2 ? (u32)cmap[d2] : 0u;
                    if (t != 0u) mb[2u * t + 1u] = __float_as_uint(own2);
                    mine2 = ok2 && t == 0u;
                    if (mine2) cmap[d2] = (unsigned char)(lane + 129u);
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- the loads of the next step
                const bool have_next = todo != 0ull;
                const u32 l0n = (last && have_next) ? (u32)__builtin_ctzll(todo) : Q.l0;
                const QC N = qctx(l0n);                         // (not the last step: the same query again)
                {
                    QC C = N;                                   // candidate vectors: only a new query's are real
                    if (!(last && have_next)) { C.c0n = 0u; C.c1n = 0u; C.c0a = sentinel; C.c1a = sentinel; }
                    issue_cands(C, CV);
                    issue_stream(N.sa, N.sn, last ? 0u : off + (u32)CH * 64u, nxt);
                }
                // ---- A: this step's chunk against the base; a doc that is registered gets S's factor into its mailbox
                if (Q.sn) {
                    u32 d4[CH], vb[CH], t[CH];
                    float bs[CH];
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        d4[j] = (u32)(cur.v[j] >> 32) - tb4;
                        bs[j] = base_at(d4[j]);
                        t[j] = (u32)cmap[d4[j] >> 2];
                    }
                    u32 m = 0u;
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        vb[j] = __float_as_uint(__fadd_rn(bs[j], __fmul_rn(__uint_as_float((u32)cur.v[j]), Q.ws)));
                        mb[2u * t[j]] = (u32)cur.v[j];          // S's factor to the doc's owner (entry 0: nobody)
                        m = vb[j] > m ? vb[j] : m;
                    }
                    if (ballot(m >= Q.thr) != 0ull) {           // (rare once the bound stands)
                        // a doc of S alone -- not registered -- scores exactly base + s
#pragma unroll
                        for (int j = 0; j < CH; j++) {
                            const u32 idx = off + (u32)j * 64u + lane;          // (past the slice: copies of its last posting)
                            const bool sv = idx < Q.sn && t[j] == 0u && vb[j] >= Q.thr;
                            if (ballot(sv) != 0ull) sa_hg_emit(a, qrow, sv, vb[j], a.doc_base + tile_base + (u64)(d4[j] >> 2), lane);
                        }
                    }
                }
                off += (u32)CH * 64u;
                if (!last) return true;
                // ---- B2: every candidate sums up its doc in query-term order, then clears its map byte and its mailbox
                if ((Q.c0n | Q.c1n) != 0u && fits) {
                    __builtin_amdgcn_wave_barrier();
                    const u32 ordA = Q.ord & 7u, ordB = (Q.ord >> 3) & 7u;
                    {
                        const u32 m0 = mb[2u * (lane + 1u)], m1 = mb[2u * (lane + 1u) + 1u];
                        const u32 v = sa_hg_fold(ordA, base_at(d0 << 2), own0, __fmul_rn(__uint_as_float(m0), Q.ws), __uint_as_float(m1));
                        const bool sv = ok0 && v >= Q.thr;
                        if (ballot(sv) != 0ull) sa_hg_emit(a, qrow, sv, v, a.doc_base + tile_base + (u64)d0, lane);
                        if (ok0) cmap[d0] = 0;
                        mb[2u * (lane + 1u)] = 0u; mb[2u * (lane + 1u) + 1u] = 0u;
                    }
                    if (Q.c0n > 64u) {
                        const u32 m0 = mb[2u * (lane + 65u)], m1 = mb[2u * (lane + 65u) + 1u];
                        const u32 v = sa_hg_fold(ordA, base_at(d1 << 2), own1, __fmul_rn(__uint_as_float(m0), Q.ws), __uint_as_float(m1));
                        const bool sv = ok1 && v >= Q.thr;
                        if (ballot(sv) != 0ull) sa_hg_emit(a, qrow, sv, v, a.doc_base + tile_base + (u64)d1, lane);
                        if (ok1) cmap[d1] = 0;
                        mb[2u * (lane + 65u)] = 0u; mb[2u * (lane + 65u) + 1u] = 0u;
                    }
                    if (Q.c1n != 0u) {
                        const u32 m0 = mb[2u * (lane + 129u)];
                        const u32 v = sa_hg_fold(ordB, base_at(d2 << 2), own2, __fmul_rn(__uint_as_float(m0), Q.ws), 0.f);
                        const bool sv = mine2 && v >= Q.thr;
                        if (ballot(sv) != 0ull) sa_hg_emit(a, qrow, sv, v, a.doc_base + tile_base + (u64)d2, lane);
                        if (mine2) cmap[d2] = 0;
                        mb[2u * (lane + 129u)] = 0u;
                    }
                    if (lane == 0) { mb[0] = 0u; mb[1] = 0u; }
                    __builtin_amdgcn_wave_barrier();
                }
                if (!have_next) return false;
                todo &= todo - 1ull;
                Q = N; off = 0u;
                return true;
            };
            for (;;) {
                if (!step(SA_, SB_)) break;
                if (!step(SB_, SA_)) break;
            }
        }

        // general path: hand the (tile, query) pairs to the per-query kernel that follows (sa_k_bm25_tiles_wl), one item per
        // index tile of the super-tile: lane 4 * i + s = tile s of query i
        if (deferred) {
            const u32 ntl = tile_hi - tile_lo;
            const u64 dq4 = deferred | (deferred << 1) | (deferred << 2) | (deferred << 3);
            const u64 mine_ = ballot((lane & 3u) < ntl) & dq4;
            const u32 c = (u32)__popcll(mine_);
            u32 wbase = 0;
            if (lane == 0) wbase = atomicAdd(a.wl_cnt, c);
            wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
            if ((mine_ >> lane) & 1ull)
                a.wl[wbase + (u32)__popcll(mine_ & ((1ull << lane) - 1ull))] =
                    ((u64)(tile_lo + (lane & 3u)) << 32) | (u64)(row0 + rbase + wave + (u32)NW * (lane >> 2));
        }
    }
}

int sa_launch_bm25_headgroups(sa_index* ix, const sa_batch* bt, const Bm25Params& p, u32 tile0, hipStream_t st) {
    if (bt->n_hg_groups == 0 || ix->n_tiles <= tile0) return SA_OK;
    HgArgs a;
    a.imp = p.imp; a.imp_tail = p.imp_tail;
    a.bounds = p.bounds; a.qbase_imp = p.qbase_imp; a.idf = p.idf; a.qrole = bt->d_qrole; a.grp = bt->d_grp;
    a.dense = bt->impacts ? bt->impacts->d_dense : nullptr;
    a.dense_stride = bt->impacts ? bt->impacts->dense_stride : 0;
    a.gthr = p.gthr; a.seed = p.seed; a.hist = p.hist; a.cand = p.cand; a.cand_cnt = p.cand_cnt; a.wl = bt->d_wl; a.wl_cnt = bt->d_wl_cnt;
    a.doc_base = p.doc_base;
    a.n_tiles = p.n_tiles; a.T = p.T; a.k = p.k; a.cand_cap = p.cand_cap;
    a.group0 = 0; a.n_groups = bt->n_hg_groups; a.tile0 = tile0; a.n_tiles_run = ix->n_tiles - tile0;
    if (ix->tile_docs != 2048u) { sa_set_error("head-group kernel: tile_docs %u", ix->tile_docs); return SA_ERR_STATE; }
    const u32 n_st = (a.n_tiles_run + (u32)SA_HG_ST - 1u) / (u32)SA_HG_ST;
    const u64 blocks = (u64)((n_st + 7u) / 8u) * 8u * a.n_groups;
    const char* xr = getenv("SA_XCD_RANGE");
    a.spx = (xr && atoi(xr) == 0) ? 0u : (n_st + 7u) / 8u;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) { sa_set_error("head-group launch: bad grid"); return SA_ERR_STATE; }
    hipLaunchKernelGGL(sa_k_bm25_headgroup, dim3((u32)blocks), dim3(SA_HG_NW * 64), 0, st, a);
    return SA_OK;
}

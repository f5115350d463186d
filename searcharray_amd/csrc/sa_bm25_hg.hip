// sa_bm25_hg.hip -- the head-group kernel: exhaustive BM25 + top-k for queries that share their first term,
// with NO per-query accumulators.
//
// The reference scores a query term by term into dense vectors and sums them (postings.py:652-680,
// bm25.pyx:11-25, test/test_msmarco.py:353-354), i.e. a doc's score is ((s0 + s1) + s2) + s3 in query-term order
// with +0.0 for the terms it lacks.  The grouped kernel of sa_bm25.hip forms those sums by a read-modify-write of
// per-(tile, group) LDS accumulators, query after query, and puts the base values back after each query: three
// dependent LDS round trips per 64 postings, one wave per 10 KiB of LDS.  Here nothing is ever written per query:
//
//   base       the group's shared first term, scored ONCE per (tile, group) into LDS by the whole workgroup
//              (dense factor row: lane = doc, 16-byte loads; or a scatter of its postings); read-only afterwards,
//              so ALL waves of the workgroup share it and each takes its own queries of the group
//   stream     the query's one dense further term S (ranks ~11-340 of a Zipf corpus: tens to hundreds of postings
//              per tile): every posting is loaded (coalesced 8-byte loads, 8 vectors in flight), scored and added
//              to its doc's base -- base + s -- and only compared with the query's bound.  A doc that holds S and
//              nothing else of the query's sparse terms scores exactly that.
//   candidates the query's sparse terms (a handful of postings per tile each): their postings ARE the docs that can
//              score more than base + s.  They sit one per lane; a lane gathers every contribution of its doc --
//              base from LDS; S through the term's RANK BITMAP (sa_index::d_sbits: bit test + popcount of the
//              tile's bits below = index into S's tile slice, one 4-byte gather); the other sparse terms by
//              comparing doc ids across lanes (v_readlane broadcast of the smaller lists) -- and adds them IN
//              QUERY-TERM ORDER with +0.0 for what the doc lacks: bit for bit the reference's sum.
//
// Every posting of every query term is read and scored; nothing is skipped on a score bound.  A (tile, query)
// pair whose base or base + s values reach the query's bound (early tiles, before the bound stands), or whose
// sparse terms have more than 64 postings in the tile, goes to the per-query kernel through the work list, like
// the grouped kernel's pairs.  Survivors (score >= bound) are appended to the query's candidate list and counted
// in its histogram exactly as sa_tile_topk_hist does, so the merge and the exactness argument are unchanged.
//
// Roofline: HBM-bound integer/bitwise + scalar fp32 work, no MFMA.  Compulsory bytes per launch = every distinct
// posting list once (+ 256 B of rank bitmap per (tile, stream term)).
#include "sa_index.hpp"
#include "sa_topk.hpp"
#include "sa_batch.hpp"
#include "sa_bm25_params.hpp"

#include <algorithm>
#include <stdlib.h>

#define SA_HG_NW 4          // waves per workgroup: they share the tile's base
#define SA_HG_NVL 8         // 64-posting vectors of the stream term requested per query ahead of their use
#define SA_HG_QPR 16        // queries per wave and round (lane = 4 * query + term position)

__global__ void __launch_bounds__(256)
sa_k_build_sbits(const u64* __restrict__ tfp, u32 n, u32* __restrict__ bits) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u64 doc = tfp[i] >> SA_KEY_SHIFT;
        atomicOr(&bits[doc >> 5], 1u << (doc & 31u));          // (u32 view of the u64 words: little endian)
    }
}

static int sa_hg_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int sa_index_ensure_sbits(sa_index* ix) {
    if (ix->sbits_built) return SA_OK;
    ix->sbits_built = true;
    ix->h_sbits_slot.assign((size_t)ix->n_terms + 1, SA_DD_NONE);
    if (ix->n_docs == 0 || ix->n_tiles == 0) return SA_OK;
    // terms dense enough to be a query's stream term: at least ~16 postings per 2048 docs (the sparse ones are
    // candidates and need no bitmap); most frequent first, at most 4096 rows
    const int div = sa_hg_env_int("SA_SBITS_DIV", 128);
    std::vector<std::pair<u64, u32>> cand;
    if (div > 0)
        for (u32 t = 0; t < ix->n_terms; t++) {
            const u64 df = ix->h_tf_off[t + 1] - ix->h_tf_off[t];
            if (df >= 16 && df * (u64)div >= ix->n_docs && df < 0xFFFFFFFFull) cand.push_back({df, t});
        }
    std::sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    if (cand.size() > 4096) cand.resize(4096);
    if (cand.empty()) return SA_OK;
    // whole tiles of at least 4096 docs, so that a wave's 64 loads of a tile's words never leave the row
    const u64 docs_pad = ((u64)ix->n_tiles * ix->tile_docs + 4095ull) & ~4095ull;
    ix->sbits_stride = docs_pad / 64;
    const size_t bytes = cand.size() * ix->sbits_stride * sizeof(u64);
    if (hipMalloc(&ix->d_sbits, bytes) != hipSuccess) {
        (void)hipGetLastError();
        ix->d_sbits = nullptr;                                // (HBM short: no head groups)
        return SA_OK;
    }
    hipStream_t st = ix->stream;
    SA_HIP(hipMemsetAsync(ix->d_sbits, 0, bytes, st));
    for (size_t r = 0; r < cand.size(); r++) {
        const u32 t = cand[r].second;
        const u32 n = (u32)cand[r].first;
        const u32 grid = n / 256 + 1 < 16384 ? n / 256 + 1 : 16384;
        hipLaunchKernelGGL(sa_k_build_sbits, dim3(grid), dim3(256), 0, st, ix->d_tfp + ix->h_tf_off[t], n,
                           (u32*)(ix->d_sbits + r * ix->sbits_stride));
        ix->h_sbits_slot[t] = (u32)r;
    }
    ix->n_sbits_terms = (u32)cand.size();
    SA_HIP(hipStreamSynchronize(st));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// inclusive prefix sum over the 64 lanes of a wave (DPP: four row_shr steps inside the rows of 16, then the row totals)
__device__ __forceinline__ u32 sa_wave_incl_scan_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, SA_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}

// lane L reads `v` of lane `src` (any permutation; ds_bpermute_b32: the LDS crossbar, no LDS memory)
__device__ __forceinline__ u32 sa_bperm(u32 v, u32 src) {
    return (u32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}

template <int TILE>
__global__ void __launch_bounds__(SA_HG_NW * 64) sa_k_bm25_headgroup(const Bm25Params p, const HgParams hp) {
    constexpr int NW = SA_HG_NW, NVL = SA_HG_NVL, QPR = SA_HG_QPR;
    constexpr int THREADS = NW * 64;
    static_assert(TILE == 2048, "one 32-bit bitmap word per lane");
    __shared__ alignas(16) float s_acc[TILE];
    __shared__ u32 s_red[NW];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // XCD-aware item order (as the grouped kernel): block b runs on XCD b % 8 -- the eight tiles of a chunk sit on
    // eight XCDs and all groups of a tile follow each other on the same XCD, so the stream / candidate slices that
    // queries of different groups share are fetched into one L2 only.
    const u32 per = 8u * hp.n_groups;
    const u32 chunk = blockIdx.x / per, r = blockIdx.x % per;
    const u32 g = r >> 3;
    const u32 trel = chunk * 8u + (r & 7u);
    if (trel >= hp.n_tiles_run) return;
    const u32 tile = hp.tile0 + trel;
    const u32 row0 = hp.grp[3 * g], n = hp.grp[3 * g + 1] & 0x7FFFFFFFu, dslot = hp.grp[3 * g + 2];
    const bool has_head = dslot != SA_HG_NOHEAD;
    const u32 T = p.T;
    const u64 tile_base = (u64)tile * TILE;
    const u32 tb4 = (u32)tile_base * 4u;
    const u64* const stream = p.imp;
    const u64 sentinel = (u64)(stream + p.imp_tail);          // a cell that can always be loaded (doc field all ones)
    typedef const __attribute__((address_space(1))) u64* gptr_u64;
    typedef const __attribute__((address_space(1))) u32* gptr_u32;
    auto rl = [](u32 v, u32 l) -> u32 { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); };

    // ---- per-lane tables of a round: lane 4 * i + t = term position t of the wave's i-th query of the round
    struct Round { u32 alo, ahi, e, ns, role, thr; float w; u64 roleS, roleC; };
    auto load_round = [&](u32 rbase) -> Round {
        Round R;
        const u32 qloc = rbase + wave + (u32)NW * (lane >> 2), t = lane & 3u;
        const bool qv = qloc < n && t < T;
        u32 r0 = 0, r1 = 0, role = 0, thr = 0;
        u64 cb = 0;
        float w = 0.f;
        if (qv) {
            const u32 qt = (row0 + qloc) * T + t;
            role = hp.qrole[qt];
            const u32* row = p.bounds + (u64)qt * (p.n_tiles + 1) + tile;
            r0 = row[0]; r1 = row[1];
            cb = p.qbase_imp[2 * (u64)qt];
            w = p.idf[qt];
            if (t == 0u) thr = __hip_atomic_load(&p.gthr[row0 + qloc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u32 kind = role & 0xFu;
        const u32 np = r1 - r0;
        const u32 nc = kind == SA_HG_CAND ? np : 0u;
        // inclusive sum of the candidate counts over the query's four lanes
        u32 inc = nc;
        u32 up = (u32)__shfl_up((int)inc, 1u, 4);
        if (t >= 1u) inc += up;
        up = (u32)__shfl_up((int)inc, 2u, 4);
        if (t >= 2u) inc += up;
        // address of the slice's first cell; a candidate term's minus its first lane, so that lane L of the candidate
        // vector loads cell [L] of it
        u64 a = sentinel;
        if (kind == SA_HG_STREAM) a = (u64)(stream + cb + r0);
        if (kind == SA_HG_CAND) a = (u64)(stream + cb + r0) - 8ull * (u64)(inc - nc);
        R.alo = (u32)a; R.ahi = (u32)(a >> 32);
        R.e = inc; R.ns = np; R.role = role; R.thr = thr; R.w = w;
        R.roleS = (u64)__builtin_amdgcn_ballot_w64(kind == SA_HG_STREAM);
        R.roleC = (u64)__builtin_amdgcn_ballot_w64(kind == SA_HG_CAND && np != 0u);
        return R;
    };
    Round R = load_round(0);                                    // in flight while the base is built

    // ---- base: the shared first term of the group, scored once for all waves
    u32 base_max = 0;
    if (has_head) {
        const u32 qt0 = row0 * T;
        const float hidf = p.idf[qt0];
        u32 lmax = 0;
        if (dslot != 0xFFFFFFFFu) {
            // dense factor row: lane = doc, four docs per 16-byte load and LDS store (a doc without the term holds
            // 0.0 -> 0.0 * idf = +0.0)
            const float4* row4 = (const float4*)(hp.dense + (u64)dslot * hp.dense_stride + tile_base);
            float4* a4 = (float4*)s_acc;
            float4 v[TILE / (4 * THREADS)];
#pragma unroll
            for (int j = 0; j < TILE / (4 * THREADS); j++) v[j] = row4[j * THREADS + (int)tid];
#pragma unroll
            for (int j = 0; j < TILE / (4 * THREADS); j++) {
                float4 w;
                w.x = __fmul_rn(v[j].x, hidf); w.y = __fmul_rn(v[j].y, hidf); w.z = __fmul_rn(v[j].z, hidf); w.w = __fmul_rn(v[j].w, hidf);
                a4[j * THREADS + (int)tid] = w;
                const u32 m0 = __float_as_uint(w.x) > __float_as_uint(w.y) ? __float_as_uint(w.x) : __float_as_uint(w.y);
                const u32 m1 = __float_as_uint(w.z) > __float_as_uint(w.w) ? __float_as_uint(w.z) : __float_as_uint(w.w);
                const u32 m = m0 > m1 ? m0 : m1;
                lmax = m > lmax ? m : lmax;
            }
        } else {
            float4* a4 = (float4*)s_acc;
#pragma unroll
            for (int j = 0; j < TILE / (4 * THREADS); j++) a4[j * THREADS + (int)tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            const u32* hrow = p.bounds + (u64)qt0 * (p.n_tiles + 1) + tile;
            const u32 h0 = hrow[0], h1 = hrow[1];
            const u64* cells = stream + p.qbase_imp[2 * (u64)qt0];
            for (u32 i = h0 + tid; i < h1; i += (u32)THREADS) {
                const u64 c = cells[i];
                const u32 wbits = __float_as_uint(__fmul_rn(__uint_as_float((u32)c), hidf));
                s_acc[((u32)(c >> 32) - tb4) >> 2] = __uint_as_float(wbits);
                lmax = wbits > lmax ? wbits : lmax;
            }
        }
        const u32 wm = sa_wave_max_u32(lmax);
        if (lane == 0) s_red[wave] = wm;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; w++) base_max = s_red[w] > base_max ? s_red[w] : base_max;
    }

    // loads of one query, requested one query ahead of their use: NVL vectors of the stream term's slice (lanes past its
    // end: copies of its last posting), the tile's words of the term's rank bitmap, and the candidate vector
    struct QL { u64 sv[NVL]; u64 cv; u32 bw, pos; };
    auto issue = [&](u32 i, QL& X) {
        const u32 l0 = 4u * i;
        const u32 sb = (u32)(R.roleS >> l0) & 0xFu;
        const u32 ls = l0 + (sb ? (u32)__builtin_ctz(sb) : 0u);
        const u32 sn = sb ? rl(R.ns, ls) : 0u;
        const u64 sa = sn ? ((u64)rl(R.alo, ls) | ((u64)rl(R.ahi, ls) << 32)) : sentinel;
        const u32 last = sn ? sn - 1u : 0u;
#pragma unroll
        for (int j = 0; j < NVL; j++) {
            const u32 idx = (u32)j * 64u + lane;
            X.sv[j] = ((gptr_u64)sa)[idx < last ? idx : last];
        }
        if (sb) {
            const u32 slot = rl(R.role, ls) >> 4;
            X.bw = ((gptr_u32)(hp.sbits + (u64)slot * hp.sbits_stride + tile_base / 64u))[lane];
        } else {
            X.bw = 0u;
        }
        const u32 e0 = rl(R.e, l0), e1 = rl(R.e, l0 + 1u), e2 = rl(R.e, l0 + 2u), nt = rl(R.e, l0 + 3u);
        const u32 cl = lane < nt ? lane : (nt ? nt - 1u : 0u);
        const u32 pos = (cl >= e0 ? 1u : 0u) + (cl >= e1 ? 1u : 0u) + (cl >= e2 ? 1u : 0u);
        X.pos = pos;
        const u32 src = l0 + pos;
        const u64 ca = (u64)sa_bperm(R.alo, src) | ((u64)sa_bperm(R.ahi, src) << 32);
        const u64 cp = nt ? ca + 8ull * cl : sentinel;
        X.cv = *(gptr_u64)cp;
    };

    u32 deferred = 0u;                                          // queries of the round left to the per-query kernel (bit i)
    // ---- one query: stage 1 (before the next query's loads are requested): the candidates' lookups of the stream term
    struct G1 { u32 gv, cd4, thr, nt; float own, bse; bool present, skip; };
    auto stage1 = [&](u32 i, const QL& X) -> G1 {
        G1 G;
        const u32 l0 = 4u * i;
        const u32 thr_q = rl(R.thr, l0);
        G.thr = thr_q > 1u ? thr_q : 1u;
        G.skip = has_head && base_max >= G.thr;
        G.nt = rl(R.e, l0 + 3u);
        G.gv = 0u; G.present = false; G.own = 0.f; G.bse = 0.f;
        const u32 cd4 = G.nt ? (u32)(X.cv >> 32) - tb4 : 0u;
        G.cd4 = cd4;
        if (G.skip) { deferred |= 1u << i; return G; }
        if (G.nt == 0u) return G;
        G.own = __fmul_rn(__uint_as_float((u32)X.cv), __uint_as_float(sa_bperm(__float_as_uint(R.w), l0 + X.pos)));
        if (has_head) G.bse = *(const float*)((const char*)s_acc + cd4);
        const u32 sb = (u32)(R.roleS >> l0) & 0xFu;
        if (sb) {
            const u32 ls = l0 + (u32)__builtin_ctz(sb);
            const u32 pc = (u32)__popc(X.bw);
            const u32 excl = sa_wave_incl_scan_u32(pc) - pc;
            const u32 wi = cd4 >> 7;                            // word of the doc: (cd4 / 4) / 32
            const u32 ww = sa_bperm(X.bw, wi), pf = sa_bperm(excl, wi);
            const u32 bit = (cd4 >> 2) & 31u;
            G.present = ((ww >> bit) & 1u) != 0u && lane < G.nt;
            const u32 rank = pf + (u32)__popc(ww & ((1u << bit) - 1u));
            const u64 sa = (u64)rl(R.alo, ls) | ((u64)rl(R.ahi, ls) << 32);
            if (G.present) G.gv = *(gptr_u32)(sa + 8ull * rank);  // the factor: low half of the cell
        }
        return G;
    };
    // survivors of a candidate vector -> the query's histogram and candidate list (as sa_tile_topk_hist)
    auto emit = [&](u32 qrow, bool sv, u32 vbits, u32 cd4) {
        const u64 mb = (u64)__builtin_amdgcn_ballot_w64(sv);
        const u32 c = (u32)__popcll(mb);
        u32 cbase = 0;
        if (lane == 0) cbase = atomicAdd(&p.cand_cnt[qrow], c);
        cbase = (u32)__builtin_amdgcn_readfirstlane((int)cbase);
        if (sv) {
            atomicAdd(&p.hist[(u64)qrow * SA_HBINS + sa_score_bin(vbits)], 1u);
            const u32 pos = cbase + (u32)__popcll(mb & ((1ull << lane) - 1ull));
            const u64 doc = p.doc_base + tile_base + (u64)(cd4 >> 2);
            if (pos < p.cand_cap) p.cand[(u64)qrow * p.cand_cap + pos] = ((u64)vbits << 32) | (u64)(u32)(~(u32)doc);
        }
        if (cbase / 32u != (cbase + c) / 32u) sa_hist_refresh(p.hist + (u64)qrow * SA_HBINS, &p.gthr[qrow], p.k, lane);
    };
    // stage 2: the stream term against the base, then the candidate vector
    auto stage2 = [&](u32 i, u32 qrow, const QL& X, const G1& G) {
        if (G.skip) return;
        const u32 l0 = 4u * i;
        const u32 thr = G.thr;
        const u32 sb = (u32)(R.roleS >> l0) & 0xFu;
        u32 ps = 4u;                                            // position of the stream term (4: none)
        float ws = 0.f;
        if (sb) {
            ps = (u32)__builtin_ctz(sb);
            const u32 ls = l0 + ps;
            const u32 sn = rl(R.ns, ls);
            ws = __uint_as_float(rl(__float_as_uint(R.w), ls));
            u32 m = 0u;
#pragma unroll
            for (int j = 0; j < NVL; j++) {
                if ((u32)j * 64u < sn) {
                    const u32 d4 = (u32)(X.sv[j] >> 32) - tb4;
                    const float bse = has_head ? *(const float*)((const char*)s_acc + d4) : 0.f;
                    const u32 v = __float_as_uint(__fadd_rn(bse, __fmul_rn(__uint_as_float((u32)X.sv[j]), ws)));
                    m = v > m ? v : m;
                }
            }
            if (sn > (u32)NVL * 64u) {                          // (a longer slice: the rest one vector at a time)
                const u64 sa = (u64)rl(R.alo, ls) | ((u64)rl(R.ahi, ls) << 32);
                for (u32 j0 = (u32)NVL * 64u; j0 < sn; j0 += 64u) {
                    const u32 idx = j0 + lane;
                    const u64 c = ((gptr_u64)sa)[idx < sn - 1u ? idx : sn - 1u];
                    const u32 d4 = (u32)(c >> 32) - tb4;
                    const float bse = has_head ? *(const float*)((const char*)s_acc + d4) : 0.f;
                    const u32 v = __float_as_uint(__fadd_rn(bse, __fmul_rn(__uint_as_float((u32)c), ws)));
                    m = v > m ? v : m;
                }
            }
            // a doc of S alone that reaches the bound: the pair is scored again by the per-query kernel (which doc also
            // holds sparse terms is not known here)
            if (__builtin_amdgcn_ballot_w64(m >= thr) != 0ull) { deferred |= 1u << i; return; }
        }
        const u32 nt = G.nt;
        if (nt == 0u) return;
        if (nt > 64u) { deferred |= 1u << i; return; }
        // contributions of the lane's doc by term position: the head's base, the lane's own posting ...
        float x0, x1, x2, x3;
        x0 = X.pos == 0u ? G.own : 0.f; x1 = X.pos == 1u ? G.own : 0.f; x2 = X.pos == 2u ? G.own : 0.f; x3 = X.pos == 3u ? G.own : 0.f;
        if (has_head) x0 = G.bse;
        // ... the stream term's, looked up through the rank bitmap ...
        {
            const float cs = G.present ? __fmul_rn(__uint_as_float(G.gv), ws) : 0.f;
            if (ps == 0u) x0 = cs;
            if (ps == 1u) x1 = cs;
            if (ps == 2u) x2 = cs;
            if (ps == 3u) x3 = cs;
        }
        // ... and the other sparse terms': every list but the longest is broadcast posting by posting; the lanes that hold
        // the same doc take the value, and the broadcast lane gives its doc up to them
        bool alive = lane < nt;
        const u32 e0 = rl(R.e, l0), e1 = rl(R.e, l0 + 1u), e2 = rl(R.e, l0 + 2u);
        const u32 n0 = e0, n1 = e1 - e0, n2 = e2 - e1, n3 = nt - e2;
        if (n0 != nt && n1 != nt && n2 != nt && n3 != nt) {    // (more than one list)
            u32 big = 0u, nb = n0;
            if (n1 > nb) { big = 1u; nb = n1; }
            if (n2 > nb) { big = 2u; nb = n2; }
            if (n3 > nb) { big = 3u; nb = n3; }
            const u32 ownb = __float_as_uint(G.own);
            auto bcast = [&](u32 lo, u32 hi, float& xp) {
                for (u32 ii = lo; ii < hi; ii++) {
                    const u32 dd = rl(G.cd4, ii), vv = rl(ownb, ii);
                    const bool hit = G.cd4 == dd && alive && lane != ii;
                    xp = hit ? __uint_as_float(vv) : xp;
                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) alive = alive && lane != ii;
                }
            };
            if (big != 0u) bcast(0u, e0, x0);
            if (big != 1u) bcast(e0, e1, x1);
            if (big != 2u) bcast(e1, e2, x2);
            if (big != 3u) bcast(e2, nt, x3);
        }
        const u32 v = __float_as_uint(__fadd_rn(__fadd_rn(__fadd_rn(x0, x1), x2), x3));
        const bool sv = alive && v >= thr;
        if (__builtin_amdgcn_ballot_w64(sv) != 0ull) emit(qrow, sv, v, G.cd4);
    };

    // ---- rounds of up to QPR queries per wave
    for (u32 rbase = 0; rbase < n; rbase += (u32)(NW * QPR)) {
        if (rbase) R = load_round(rbase);
        // queries of this wave in the round: rbase + wave + NW * i < n
        const u32 left = n - rbase;
        u32 nq = left > wave ? (left - wave + (u32)NW - 1u) / (u32)NW : 0u;
        if (nq > (u32)QPR) nq = (u32)QPR;
        deferred = 0u;
        if (nq) {
            QL A, B;
            issue(0, A);
            for (u32 i = 0; i < nq; i += 2) {
                {
                    const G1 G = stage1(i, A);
                    if (i + 1u < nq) issue(i + 1u, B);
                    stage2(i, row0 + rbase + wave + (u32)NW * i, A, G);
                }
                if (i + 1u < nq) {
                    const G1 G = stage1(i + 1u, B);
                    if (i + 2u < nq) issue(i + 2u, A);
                    stage2(i + 1u, row0 + rbase + wave + (u32)NW * (i + 1u), B, G);
                }
            }
        }
        // general path: hand the (tile, query) pairs to the per-query kernel that follows (sa_k_bm25_tiles_wl)
        if (deferred) {
            const u32 c = (u32)__popc(deferred);
            u32 wbase = 0;
            if (lane == 0) wbase = atomicAdd(hp.wl_cnt, c);
            wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
            if (lane < (u32)QPR && ((deferred >> lane) & 1u))
                hp.wl[wbase + (u32)__popc(deferred & ((1u << lane) - 1u))] = ((u64)tile << 32) | (u64)(row0 + rbase + wave + (u32)NW * lane);
        }
    }
}

int sa_launch_bm25_headgroups(sa_index* ix, const sa_batch* bt, const Bm25Params& p, u32 tile0, hipStream_t st) {
    if (bt->n_hg_groups == 0 || ix->n_tiles <= tile0) return SA_OK;
    HgParams hp;
    hp.grp = bt->d_grp; hp.qrole = bt->d_qrole; hp.n_groups = bt->n_hg_groups;
    hp.dense = bt->impacts ? bt->impacts->d_dense : nullptr;
    hp.dense_stride = bt->impacts ? bt->impacts->dense_stride : 0;
    hp.sbits = ix->d_sbits; hp.sbits_stride = ix->sbits_stride;
    hp.tile0 = tile0; hp.n_tiles_run = ix->n_tiles - tile0;
    hp.wl = bt->d_wl; hp.wl_cnt = bt->d_wl_cnt;
    const u64 blocks = (u64)((hp.n_tiles_run + 7u) / 8u) * 8u * hp.n_groups;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) { sa_set_error("head-group launch: bad grid"); return SA_ERR_STATE; }
    if (ix->tile_docs != 2048u) { sa_set_error("head-group kernel: tile_docs %u", ix->tile_docs); return SA_ERR_STATE; }
    hipLaunchKernelGGL((sa_k_bm25_headgroup<2048>), dim3((u32)blocks), dim3(SA_HG_NW * 64), 0, st, p, hp);
    return SA_OK;
}

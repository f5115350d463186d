"""CPU model of the staged-tile route's pruning power: how many CANDIDATE postings a
(tile, query) pair has to evaluate when every query starts from its rank-table bound.

For a query set on zipf-N it computes, per query: the starting bound (weight x k-th largest factor of a
term, max over the terms), the terms' upper bounds (weight x largest factor), the essential terms (the
prefix in descending-bound order that the rest cannot replace), the candidate postings (sum of df over
the essential terms) and the true k-th best score.  Pure numpy; no GPU, no library.

    python scripts/stage_model.py [--docs 1000000] [--k 10] [--set baseline|distinct]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from searcharray_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--set", default="baseline")
    ap.add_argument("--tile", type=int, default=512)
    a = ap.parse_args()
    N, V, k = a.docs, 100_000, a.k
    lens, terms = synth.zipf_batch_tokens(0, N, V, 32, 1234, fast=True)
    doc = np.repeat(np.arange(N, dtype=np.int64), lens)
    key = terms.astype(np.int64) * N + doc
    uk, tf = np.unique(key, return_counts=True)
    pt, pd = uk // N, uk % N
    off = np.searchsorted(pt, np.arange(V + 1))
    dl = lens.astype(np.float32)
    avgdl = np.float32(dl.mean())
    k1, b = np.float32(1.2), np.float32(0.75)
    norm = k1 * ((np.float32(1) - b) + b * (dl[pd] / avgdl))
    fac = tf.astype(np.float32) / (tf.astype(np.float32) + norm)
    df = np.diff(off)
    idf = np.log(1 + (N - df + 0.5) / (df + 0.5)).astype(np.float32)
    qs = synth.bm25_queries(256, V) if a.set == "baseline" else synth.bm25_queries_distinct(256, 4, V)
    tot_cand = 0
    tot_post = 0
    ness = np.zeros(5, dtype=np.int64)
    gaps = []
    for q in qs:
        sl = [slice(off[t], off[t + 1]) for t in q]
        w = idf[q]
        kth = [np.partition(fac[s], -k)[-k] if df[t] >= k else 0.0 for s, t in zip(sl, q)]
        seed = max(wi * f for wi, f in zip(w, kth))
        ub = np.array([wi * (fac[s].max() if df[t] else 0.0) for wi, s, t in zip(w, sl, q)], dtype=np.float32)
        order = np.argsort(-ub)
        sfx = np.cumsum(ub[order][::-1])[::-1]            # sfx[i] = sum of ub over positions >= i
        n_ess = 4
        for i in range(4):
            if sfx[i] * 1.00001 < seed:
                n_ess = i
                break
        cand = sum(int(df[q[order[i]]]) for i in range(n_ess))
        # exact scores for the true bound
        sc = np.zeros(N, dtype=np.float32)
        for wi, s in zip(w, sl):
            sc[pd[s]] += fac[s] * wi
        true_kth = np.partition(sc, -k)[-k]
        gaps.append(seed / true_kth)
        tot_cand += cand
        tot_post += int(df[q].sum())
        ness[n_ess] += 1
    n_tiles = (N + a.tile - 1) // a.tile
    print(f"docs={N} k={k} set={a.set}: queries by number of essential terms {ness.tolist()}")
    print(f"  candidates/batch {tot_cand:,} of {tot_post:,} postings ({100.0 * tot_cand / tot_post:.2f} %);"
          f" per {a.tile}-doc tile: {tot_cand / n_tiles:.1f} candidates, {tot_post / n_tiles:.0f} postings (all queries)")
    print(f"  seed / true k-th score: min {min(gaps):.3f} median {np.median(gaps):.3f} max {max(gaps):.3f}")


def finalists(docs=1_000_000, k=10):
    """fraction of the candidates (postings of the essential terms) whose staged contributions + the probed terms' bounds reach the
    starting bound: what stage B hands to stage C"""
    N, V = docs, 100_000
    lens, terms = synth.zipf_batch_tokens(0, N, V, 32, 1234, fast=True)
    doc = np.repeat(np.arange(N, dtype=np.int64), lens)
    key = terms.astype(np.int64) * N + doc
    uk, tf = np.unique(key, return_counts=True)
    pt, pd = uk // N, uk % N
    off = np.searchsorted(pt, np.arange(V + 1))
    dl = lens.astype(np.float32)
    avgdl = np.float32(dl.mean())
    norm = np.float32(1.2) * ((np.float32(1) - np.float32(0.75)) + np.float32(0.75) * (dl[pd] / avgdl))
    fac = tf.astype(np.float32) / (tf.astype(np.float32) + norm)
    df = np.diff(off)
    idf = np.log(1 + (N - df + 0.5) / (df + 0.5)).astype(np.float32)
    qs = synth.bm25_queries(256, V)
    tot_c = tot_f = tot_surv = 0
    for q in qs:
        sl = [slice(off[t], off[t + 1]) for t in q]
        w = idf[q]
        kth = [np.partition(fac[s], -k)[-k] if df[t] >= k else 0.0 for s, t in zip(sl, q)]
        seed = max(wi * f for wi, f in zip(w, kth))
        ub = np.array([wi * fac[s].max() for wi, s in zip(w, sl)], dtype=np.float32)
        order = np.argsort(-ub)
        sfx = np.cumsum(ub[order][::-1])[::-1]
        n_ess = 4
        for i in range(4):
            if sfx[i] * 1.00001 < seed:
                n_ess = i
                break
        ess = order[:n_ess]
        ne = order[n_ess:]
        known = np.zeros(N, dtype=np.float32)
        for i in ess:
            known[pd[sl[i]]] += fac[sl[i]] * w[i]
        pend = float(ub[ne].sum())
        cand = known > 0
        full = known.copy()
        for i in ne:
            full[pd[sl[i]]] += fac[sl[i]] * w[i]
        tot_c += int(cand.sum())
        tot_f += int((known[cand] + pend >= seed).sum())
        tot_surv += int((full[cand] >= seed).sum())
    print(f"candidates {tot_c:,}  finalists (known + pend >= seed) {tot_f:,} ({100.0 * tot_f / tot_c:.1f} %)  at or above the seed {tot_surv:,}")


if __name__ == "__main__":
    if "--finalists" in sys.argv:
        sys.argv.remove("--finalists")
        finalists()
    else:
        main()

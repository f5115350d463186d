#!/usr/bin/env python
"""GPU stress (development tool): a few-million-doc corpus, random query batches of every shape;
dynamic pruning, the exhaustive kernel with the histogram bound, with the slot bound and with the
block-level selection must return identical top-k lists."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth
from searcharray_amd.device_index import DeviceIndex, QueryBatch

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=3_000_000)
ap.add_argument("--batches", type=int, default=24)
args = ap.parse_args()
V = 100_000
corpus = synth.zipf_corpus(args.docs, vocab=V, workers=8)
index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
rng = np.random.default_rng(2024)
fails = 0
for it in range(args.batches):
    B, T = int(rng.integers(1, 300)), int(rng.integers(1, 7))
    k = int(rng.choice([1, 10, 32, 33, 100, 1000]))
    ranks = np.exp(rng.uniform(0, np.log(V), size=(B, T))).astype(np.int64) - 1      # log-uniform term ranks
    ranks[rng.random((B, T)) < 0.05] = V + 7                                           # some unknown terms
    batch = QueryBatch(index, ranks, k=k)
    got = {}
    for name, env in (("pruned", {"SA_SPARSE": "1"}), ("hist", {"SA_SPARSE": "0"}),
                      ("slots", {"SA_SPARSE": "0", "SA_TOPK_HIST": "0"}), ("block", {"SA_SPARSE": "0", "SA_PRUNED_TOPK": "0"})):
        for key in ("SA_SPARSE", "SA_TOPK_HIST", "SA_PRUNED_TOPK"):
            os.environ.pop(key, None)
        os.environ.update(env)
        batch.run()
        got[name] = batch.fetch()
    ref = got["block"]
    for name in ("pruned", "hist", "slots"):
        if not (np.array_equal(got[name][0], ref[0]) and np.array_equal(got[name][1], ref[1])):
            fails += 1
            bad = np.flatnonzero((got[name][1] != ref[1]).any(axis=1))
            print("MISMATCH", it, name, dict(B=B, T=T, k=k), "queries", bad[:5].tolist(), flush=True)
    batch.close()
# phrase batches against the single-phrase dense path (both phrase kernels) + host-side top-k
pf = 0
for it in range(max(2, args.batches // 8)):
    n, L, k = int(rng.integers(1, 100)), int(rng.integers(2, 5)), int(rng.choice([1, 10, 50]))
    phrases = [[int(x) for x in rng.choice(300, L, replace=False)] for _ in range(n)]
    pb = index.phrase_batch(phrases, k=k)
    pb.run()
    s, d = pb.fetch()
    pb.close()
    for i in range(0, n, max(1, n // 6)):
        for mode in ("fused", "general"):
            os.environ["SA_PHRASE_MODE"] = mode
            dense = index.bm25_phrase_dense(phrases[i])
            order = np.lexsort((np.arange(len(dense)), -dense))[:k]
            m = int((dense[order] > 0).sum())
            if not (np.array_equal(d[i, :m], order[:m].astype(np.uint64)) and np.array_equal(s[i, :m], dense[order][:m])):
                pf += 1
                print("PHRASE MISMATCH", it, mode, phrases[i], flush=True)
    os.environ.pop("SA_PHRASE_MODE", None)
print(f"stress done: {args.batches} batches, {fails} mismatches; phrase mismatches {pf}")

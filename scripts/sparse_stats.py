#!/usr/bin/env python
"""Diagnostics of the dynamic pruning on the bench workload: sparse-routed queries, candidates scored."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth, _lib
from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=10)
args = ap.parse_args()
D, V, B = args.docs, 100_000, 256
corpus = synth.zipf_corpus(D, vocab=V, workers=8)
index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
df = index.docfreqs().astype(np.int64)
queries = synth.bm25_queries(B, vocab=V)
batch = QueryBatch(index, queries, k=args.k)
batch.stats(True)
batch.run()
cands, nq = batch.stats(True)
tot_post = int(sum(df[t] for q in queries for t in q))
print(json.dumps({"sparse_queries": nq, "of": B, "candidates": cands, "all_postings": tot_post,
                  "cand_per_sparse_query": cands / max(nq, 1)}))
batch.stats(False)
for _ in range(3):
    batch.run()
index.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    batch.run(sync=False)
index.synchronize()
print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
# which queries cannot take the sparse route statically (lead too frequent)?
idf_of = lambda t: float(compute_idf(D, np.asarray([df[t]])))
bad = []
for qi, q in enumerate(queries):
    lead = max(q, key=lambda t: idf_of(int(t)))
    if df[lead] > 24000:
        bad.append((qi, [int(df[t]) for t in q]))
print("lead too frequent:", bad[:10])

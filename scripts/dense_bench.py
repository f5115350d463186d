#!/usr/bin/env python
"""Latency of the dense drop-in calls (float32[N] results copied to the host): single-term BM25,
4-term sum and a phrase, at N docs."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth
from searcharray_amd.device_index import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
args = ap.parse_args()
D, V = args.docs, 100_000
corpus = synth.zipf_corpus(D, vocab=V, workers=8)
index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
out = {"docs": D}
for name, fn in (("bm25_1term_t5", lambda: index.bm25_dense([5])), ("bm25_4terms", lambda: index.bm25_dense([0, 9, 99, 999])),
                 ("termfreqs_t5", lambda: index.termfreqs_dense(5)), ("phrase_t0_t1", lambda: index.bm25_phrase_dense([0, 1]))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        r = fn()
    out[name + "_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
t0 = time.perf_counter()
for _ in range(10):
    a = np.empty(D, np.float32); a[:] = 0
out["numpy_alloc_fill_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(out))

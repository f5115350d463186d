#!/usr/bin/env python
"""The reference's other stock similarities: device kernel (score -> float32/float64[N] on the host) next to
the host route (tf from the device, then the numpy expression of reference similarity.py:41-89)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth                                          # noqa: E402
from searcharray_amd.device_index import DeviceIndex                      # noqa: E402
from searcharray_amd.similarity import bm25_impact, bm25_legacy_similarity, classic_similarity, compute_idf  # noqa: E402


def best(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def main():
    D = int(os.environ.get("DOCS", 10_000_000))
    corpus = synth.zipf_corpus(D, vocab=100_000, workers=8)
    dev = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
    term = 9
    df = np.asarray([dev.docfreq(term)])
    out = {"docs": D, "term_df": int(df[0])}
    for name, sim, kind in (("bm25_impact", bm25_impact(), "bm25_impact"), ("bm25_legacy", bm25_legacy_similarity(), "bm25_legacy"),
                            ("classic", classic_similarity(), "classic")):
        idf = (np.log((D + 1) / (np.sum(df, axis=0) + 1)) + 1) if kind == "classic" else compute_idf(D, df)
        ms_dev, a = best(lambda: dev.similarity_dense(kind, [term], idf=idf, k1=1.2, b=0.75))
        ms_host, b = best(lambda: sim(dev.termfreqs_dense(term), df, corpus.doc_lens, dev.avg_doc_len, D), n=3)
        out[name] = {"device_ms": round(ms_dev, 2), "host_numpy_ms": round(ms_host, 2), "dtype": str(a.dtype),
                     "identical": bool(np.array_equal(a, b, equal_nan=True))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

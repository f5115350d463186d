#!/usr/bin/env python
"""The fresh-batch loop of bench.py alone (10 M docs, rotating query sets, P batches in flight), `--steps` steps after the
warm-up: run under `rocprofv3 --hip-trace --stats` with two step counts, the difference of the two HIP-API summaries is
what the steady state calls -- no hipMalloc / hipFree / hipHostMalloc / blocking hipMemcpy among it."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth, _lib                          # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--corpus-cache", default="")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--pipeline", type=int, default=6)
    args = ap.parse_args()
    D, V, B = args.docs, args.vocab, 256
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    api = _lib.api()
    index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=api)
    df = index.docfreqs().astype(np.uint64)
    idf_table = np.log(1 + (D - df + 0.5) / (df + 0.5)).astype(np.float32)
    os.environ["SA_SPARSE"] = "0"
    sets = [synth.bm25_queries(B, vocab=V, seed=1000 + i) for i in range(8)]
    P = args.pipeline
    ring = [QueryBatch(index, sets[i % 8], k=10, idf=idf_table[sets[i % 8]]) for i in range(P)]
    pend = [False] * P

    def step(i):
        b = i % P
        if pend[b]:
            ring[b].fetch()
        q = sets[i % 8]
        ring[b].reset(q, idf=idf_table[q])
        ring[b].run(sync=False)
        pend[b] = True
    for i in range(2 * P):
        step(i)
    index.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(2 * P + i)
    for b in range(P):
        ring[b].fetch()
    index.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"docs": D, "steps": args.steps, "batches_in_flight": P, "ms_per_step": round(dt / args.steps * 1e3, 4),
                      "queries_per_s": round(B * args.steps / dt, 1)}))
    for b in ring:
        b.close()
    index.close()


if __name__ == "__main__":
    main()

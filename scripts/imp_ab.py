#!/usr/bin/env python
"""Same-box A/B of the exhaustive tile kernel's posting routes (development tool).

SA_IMPACT=0: TF postings + saturation table; 1: impact stream (the default).  One index per corpus size,
one batch, every route timed on the same box; results of all routes must be identical.  --slots / --empty
time subsets of the query terms (what a phase costs).  Prints one JSON line per (docs, k, route)."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth, _lib                              # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", default="10000000,1250000")
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--ks", default="10,1000")
    ap.add_argument("--routes", default="0,1,0,1")
    ap.add_argument("--corpus-cache", default="")
    ap.add_argument("--slots", default="", help="keep only these query-term columns, e.g. 0,1 (cost of a subset of the terms)")
    ap.add_argument("--empty", action="store_true",
                    help="queries of unknown terms only: the fixed cost of a launch (zero, slice lookup, barriers, selection)")
    args = ap.parse_args()
    api = _lib.api()
    V, B = args.vocab, args.queries
    for D in [int(x) for x in args.docs.split(",")]:
        cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
        if cpath and os.path.exists(cpath):
            z = np.load(cpath)
            corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
        else:
            corpus = synth.zipf_corpus(D, vocab=V, workers=min(8, os.cpu_count() or 8))
        queries = synth.bm25_queries(B, vocab=V)
        if args.slots:
            queries = np.asarray(queries)[:, [int(x) for x in args.slots.split(",")]]
        if args.empty:
            queries = np.full_like(np.asarray(queries), V + 5)
        index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=api)
        for k in [int(x) for x in args.ks.split(",")]:
            os.environ["SA_IMPACT"] = "1"
            t0 = time.perf_counter()
            batch = QueryBatch(index, queries, k=k)
            create_ms = (time.perf_counter() - t0) * 1e3
            ref = None
            for sparse in ("0", "1"):
                for route in args.routes.split(","):
                    os.environ["SA_SPARSE"] = sparse
                    os.environ["SA_IMPACT"] = route
                    for _ in range(3):
                        batch.run(sync=False)
                    index.synchronize()
                    batch.profile()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        batch.run(sync=False)
                    index.synchronize()
                    dt = time.perf_counter() - t0
                    ms, alg, post = batch.profile()
                    res = batch.fetch()
                    if ref is None:
                        ref = res
                    same = bool(np.array_equal(ref[0], res[0]) and np.array_equal(ref[1], res[1]))
                    print(json.dumps({"docs": D, "k": k, "sparse": sparse, "impact_route": route, "no_topk": os.environ.get("SA_NO_TOPK", "0"), "slots": args.slots,
                                      "qps": round(B * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4),
                                      "kernel_ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                                      "postings_GBps": round(post / ms / 1e6, 1), "same_results": same,
                                      "batch_create_ms": round(create_ms, 1),
                                      "hbm_GB": round(index.info().hbm_bytes / 1e9, 2)}), flush=True)
            batch.close()
        index.close()
        del corpus


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Same-box A/B of the exhaustive scoring paths on the bench workload: per-query kernel (SA_GROUP=0) vs grouped
kernel (shared first term) with different warm-up tile counts / minimum group sizes; BASELINE and
pairwise-distinct query sets; identical results required.  One JSON line per configuration."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth, _lib                                     # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--corpus-cache", default="")
    ap.add_argument("--ks", default="10,1000")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--tiles", default="2048")
    ap.add_argument("--only", default="", help="run only configuration number i (0 = per-query kernel, 1 = grouped, ...)")
    ap.add_argument("--qsets", default="baseline,distinct")
    args = ap.parse_args()
    api = _lib.api()
    D, V = args.docs, args.vocab
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    qsets = {"baseline": synth.bm25_queries(256, vocab=V), "distinct": synth.bm25_queries_distinct(256, vocab=V)}
    qsets = {k_: v for k_, v in qsets.items() if k_ in args.qsets.split(",")}
    for tile in [int(x) for x in args.tiles.split(",")]:
        index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, tile_docs=tile, api=api)
        df = index.docfreqs()
        for qname, queries in qsets.items():
            idf = np.asarray([[compute_idf(D, np.asarray([df[t]])) for t in q] for q in queries], dtype=np.float32)
            for k in [int(x) for x in args.ks.split(",")]:
                ref = None
                configs = [{"SA_GROUP": "0"}, {"SA_GROUP": "1"}, {"SA_GROUP": "1", "SA_GROUP_WARM": "64"},
                           {"SA_GROUP": "1", "SA_GROUP_WARM": "4"}, {"SA_GROUP": "1", "SA_GROUP_MIN": "1"}]
                if args.only:
                    configs = [configs[int(i)] for i in args.only.split(",")]
                for cfg in configs:
                    for key in ("SA_GROUP", "SA_GROUP_WARM", "SA_GROUP_MIN"):
                        os.environ.pop(key, None)
                    os.environ.update(cfg)
                    os.environ["SA_SPARSE"] = "0"
                    batch = QueryBatch(index, queries, k=k, idf=idf)
                    for _ in range(3):
                        batch.run(sync=False)
                    index.synchronize()
                    batch.profile()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        batch.run(sync=False)
                    index.synchronize()
                    dt = (time.perf_counter() - t0) / args.steps
                    kms, _, _ = batch.profile()
                    res = batch.fetch()
                    if ref is None:
                        ref = res
                    same = bool(np.array_equal(ref[0], res[0]) and np.array_equal(ref[1], res[1]))
                    print(json.dumps({"tile": tile, "queries": qname, "k": k, **cfg, "ms_per_step": round(dt * 1e3, 4),
                                      "kernel_ms": round(kms, 4), "same_results": same}), flush=True)
                    batch.close()
        index.close()


if __name__ == "__main__":
    main()

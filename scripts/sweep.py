#!/usr/bin/env python
"""GPU tuning sweep (development tool): one corpus, several tile sizes / kernel modes.

Prints one JSON line per configuration: queries/s, scoring-kernel ms, algorithmic GB/s."""
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
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--tiles", default="2048,4096,8192,16384")
    ap.add_argument("--ks", default="10")
    ap.add_argument("--dir-divs", default="8")
    ap.add_argument("--corpus-cache", default="")
    args = ap.parse_args()
    api = _lib.api()
    g = _lib.c_double(0)
    for mode in (0, 1):
        api.call("sa_stream_probe", 2 << 30, mode, 5, _lib.ctypes.byref(g))
        print(json.dumps({"probe": f"{8 * (mode + 1)}B/lane read", "GBps": round(g.value, 1)}), flush=True)
    D, V, B = args.docs, args.vocab, args.queries
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    t0 = time.time()
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=min(8, os.cpu_count() or 8))
        if cpath:
            os.makedirs(args.corpus_cache, exist_ok=True)
            np.savez(cpath, words=corpus.words, term_off=corpus.term_off, doc_lens=corpus.doc_lens)
    print(json.dumps({"corpus_s": round(time.time() - t0, 1), "words": int(len(corpus.words))}), flush=True)
    queries = synth.bm25_queries(B, vocab=V)
    for tile, ddiv in [(int(x), y) for x in args.tiles.split(",") for y in args.dir_divs.split(",")]:
        os.environ["SA_DIR_DIV"] = ddiv
        t0 = time.time()
        index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, tile_docs=tile, api=api)
        build_s = time.time() - t0
        for k in [int(x) for x in args.ks.split(",")]:
            batch = QueryBatch(index, queries, k=k)
            # (sparse, pruned selection, skip selection): dynamic pruning; exhaustive with the pruned
            # wave-level top-k; exhaustive with the block-level selection; exhaustive without any selection
            modes = [("1", "1", "0"), ("0", "1", "0"), ("0", "0", "0"), ("0", "1", "1")]
            for sparse, pruned, notopk in modes:
                os.environ["SA_SPARSE"] = sparse
                os.environ["SA_PRUNED_TOPK"] = pruned
                os.environ["SA_NO_TOPK"] = notopk
                for _ in range(2):
                    batch.run(sync=False)
                index.synchronize()
                batch.profile()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    batch.run(sync=False)
                index.synchronize()
                dt = time.perf_counter() - t0
                ms, alg, post = batch.profile()
                print(json.dumps({"tile": tile, "dir_div": ddiv, "dir_terms": int(index.info().n_dir_terms), "k": k,
                                  "sparse": sparse, "pruned_topk": pruned, "no_topk": notopk,
                                  "qps": round(B * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
                                  "kernel_ms": round(ms, 3), "alg_GBps": round(alg / ms / 1e6, 1),
                                  "postings_GBps": round(post / ms / 1e6, 1), "index_build_s": round(build_s, 1)}),
                      flush=True)
            batch.close()
        index.close()


if __name__ == "__main__":
    main()

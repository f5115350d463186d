#!/usr/bin/env python
"""Queries of 2 / 6 / 8 terms (256 per set, one term per frequency band like the BASELINE set, the bands repeated for the longer ones):
the staged-tile route (its 8-term instance for T > 4) against the exhaustive overlay and the library's own choice.  One JSON line per
(T, k, route)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth, _lib                                     # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--corpus-cache", default="")
    args = ap.parse_args()
    D, V = args.docs, 100_000
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=_lib.api())
    rng = np.random.default_rng(8)
    bands = [(0, 10), (10, 100), (100, 1000), (1000, 10000)]
    for T in (2, 6, 8):
        use = [bands[1], bands[3]] if T == 2 else [bands[i % 4] for i in range(T)]
        queries = np.stack([rng.integers(lo, hi, 256) for lo, hi in use], axis=1)
        for k in (10, 100):
            ref = None
            for name, opts in (("exhaustive", {"sparse": 0, "stage": 0}), ("staged", {"stage": 1}), ("default", {})):
                bt = QueryBatch(index, queries, k=k, opts=opts)
                for _ in range(3):
                    bt.run(sync=False)
                index.synchronize()
                bt.profile()
                for _ in range(10):
                    bt.run(sync=False)
                index.synchronize()
                kms, _, _ = bt.profile()
                res = bt.fetch()
                ref = ref or res
                print(json.dumps({"T": T, "k": k, "forced": name, "route": bt.last_route(), "kernel_ms": round(kms, 4),
                                  "same_results": bool(np.array_equal(ref[0], res[0]) and np.array_equal(ref[1], res[1]))}), flush=True)
                bt.close()
    index.close()


if __name__ == "__main__":
    main()

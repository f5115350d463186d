#!/usr/bin/env python
"""BASELINE config 2's literal call -- SearchArray.score of ONE term (and of 4 terms summed), the float32[n_docs] result left in a device
vector -- through the one-launch route (default) and the rounds 1-5 route (option dense_direct = 0: TF postings -> scratch -> scale / copy),
same index contents, results compared bit for bit.  One JSON line per (route, query)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth, _lib, options                              # noqa: E402
from searcharray_amd.device_index import DeviceIndex, DeviceVec              # noqa: E402
from searcharray_amd._lib import p_u32, p_f32                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--corpus-cache", default="")
    ap.add_argument("--calls", type=int, default=50)
    args = ap.parse_args()
    D, V = args.docs, args.vocab
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    api = _lib.api()
    ref = {}
    for route, direct in (("one_launch", None), ("tf_scratch_copy", 0)):
        index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=api, opts=None if direct is None else {"dense_direct": direct})
        vec = DeviceVec(api, D, False)
        df_all = index.docfreqs()
        for q in ([0], [9], [99], [999], [0, 9, 99, 999]):
            tarr = np.asarray(q, dtype=np.uint32)
            idf = index.idfs(q)
            call = lambda: index.into_vec(vec, None, "sa_index_bm25_dense", p_u32(tarr), p_f32(idf), len(q), np.float32(1.2), np.float32(0.75))   # noqa: E731
            for _ in range(3):
                call()
            index.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.calls):
                call()
            index.synchronize()
            dt = (time.perf_counter() - t0) / args.calls
            got = vec.fetch()
            r0 = ref.setdefault(tuple(q), got)
            dfs = [int(df_all[t]) for t in q]
            alg = 8 * sum(dfs) + 4 * D
            print(json.dumps({"route": route, "terms": q, "df": dfs, "docs": D, "ms_per_call": round(dt * 1e3, 4), "algorithmic_bytes": alg,
                              "GBps": round(alg / dt / 1e9, 1), "frac_of_8TBps": round(alg / dt / 8e12, 4), "same_results": bool(np.array_equal(r0, got)),
                              "nonzero_docs": int((got != 0).sum())}), flush=True)
        vec.close()
        index.close()


if __name__ == "__main__":
    main()

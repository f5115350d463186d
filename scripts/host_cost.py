#!/usr/bin/env python
"""Host-side cost of one fresh-batch step, by call: idf (numpy), QueryBatch.reset (Python packing + sa_batch_reset),
run (kernel launches), fetch.  The device is kept idle-free or starved depending on --docs: with a small shard the host
is the bottleneck and these are what a step costs."""
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
from searcharray_amd.device_index import DeviceIndex, QueryBatch           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_250_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--comm", action="store_true")
    ap.add_argument("--default-route", action="store_true")
    args = ap.parse_args()
    api = _lib.api()
    D, V, B = args.docs, args.vocab, args.queries
    corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=api)
    df = index.docfreqs().astype(np.uint64)
    if args.comm:
        index.comm_init(0, 1, DeviceIndex.comm_unique_id(api))
    if not args.default_route:
        os.environ["SA_SPARSE"] = "0"                          # (rounds 3-5: the exhaustive overlay route; --default-route: the library's own choice)
    sets = [synth.bm25_queries(B, vocab=V, seed=1000 + i) for i in range(8)]

    def idf_of(q):
        dfs = df[np.asarray(q, dtype=np.int64)]
        return np.log(1 + (D - dfs + 0.5) / (dfs + 0.5)).astype(np.float32)

    pair = [QueryBatch(index, sets[0], k=10, idf=idf_of(sets[0])), QueryBatch(index, sets[1], k=10, idf=idf_of(sets[1]))]
    t = {"idf": 0.0, "reset": 0.0, "reset_c": 0.0, "run": 0.0, "fetch": 0.0}
    pend = [False, False]
    import ctypes
    from searcharray_amd._lib import p_u32, p_f32, as_u32
    for phase in ("warm", "timed"):
        n = 20 if phase == "warm" else args.steps
        for k_ in t:
            t[k_] = 0.0
        index.synchronize()
        t00 = time.perf_counter()
        for i in range(n):
            b = i & 1
            t0 = time.perf_counter()
            if pend[b]:
                pair[b].fetch()
            t1 = time.perf_counter()
            w = idf_of(sets[i % 8])
            t2 = time.perf_counter()
            terms = as_u32(sets[i % 8])
            t3 = time.perf_counter()
            api.call("sa_batch_reset", pair[b]._h, p_u32(terms), p_f32(w))
            t4 = time.perf_counter()
            pair[b].run(sync=False)
            t5 = time.perf_counter()
            pend[b] = True
            t["fetch"] += t1 - t0; t["idf"] += t2 - t1; t["reset"] += t3 - t2; t["reset_c"] += t4 - t3; t["run"] += t5 - t4
        index.synchronize()
        total = time.perf_counter() - t00
    out = {k_: round(v / args.steps * 1e6, 1) for k_, v in t.items()}
    out.update({"docs": D, "queries": B, "comm": args.comm, "us_per_step_total": round(total / args.steps * 1e6, 1),
                "unit": "microseconds of host time per step (fetch includes waiting for the device)"})
    print(json.dumps(out))
    # the same stream through ONE library call per step (sa_batch_step: idf gathered from the index's table inside)
    dfs = df.astype(np.float64)
    index.set_idf_table(np.log(1 + (D - dfs + 0.5) / (dfs + 0.5)).astype(np.float32))
    sets_u32 = [np.ascontiguousarray(q, dtype=np.uint32) for q in sets]
    for b in (0, 1):
        pair[b].fetch()
    ts = {"step": 0.0, "fetch": 0.0}
    pend = [False, False]
    for phase in ("warm", "timed"):
        n = 20 if phase == "warm" else args.steps
        ts = {"step": 0.0, "fetch": 0.0}
        index.synchronize()
        h0 = [pair[b].host_times() for b in (0, 1)]
        t00 = time.perf_counter()
        for i in range(n):
            b = i & 1
            t0 = time.perf_counter()
            if pend[b]:
                pair[b].fetch()
            t1 = time.perf_counter()
            pair[b].step(sets_u32[i % 8])
            t2 = time.perf_counter()
            pend[b] = True
            ts["fetch"] += t1 - t0; ts["step"] += t2 - t1
        index.synchronize()
        total = time.perf_counter() - t00
    out = {k_: round(v / args.steps * 1e6, 1) for k_, v in ts.items()}
    h1 = [pair[b].host_times() for b in (0, 1)]
    fills = sum(h1[b]["fills"] - h0[b]["fills"] for b in (0, 1))
    inside = {k_: round(sum(h1[b][k_] - h0[b][k_] for b in (0, 1)) / max(fills, 1), 1) for k_ in ("fill_cpu_us", "fill_enqueue_us", "run_enqueue_us")}
    out.update({"docs": D, "queries": B, "comm": args.comm, "us_per_step_total": round(total / args.steps * 1e6, 1),
                "inside_the_library": inside,
                "call": "sa_batch_step (idf gather + reset + run in one call)"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

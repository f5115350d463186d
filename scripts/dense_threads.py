#!/usr/bin/env python
"""Dense drop-in calls of SEVERAL THREADS on one index handle (the reference's callers score from thread pools:
test/test_tmdb.py:285-312, test/test_msmarco.py:483-507; its native kernels release the GIL): single-term BM25 `score()`
calls -- float32[n_docs] back to the host -- from 1, 2, 4, 8 threads, calls per second and the speed-up over one thread.
Since round 5 such calls enqueue on lanes of the index (csrc/sa_index.hpp, DenseLane) and wait outside its lock.  Also: the wall
time of ONE single-phrase score() call against the device time of its kernels.

  python scripts/dense_threads.py [--docs 1000000] > profiles/dense_threads_r05.jsonl
"""
import _envopts  # noqa: F401
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth                                       # noqa: E402
from searcharray_amd.device_index import DeviceIndex                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--calls", type=int, default=400)
    args = ap.parse_args()
    corpus = synth.zipf_corpus(args.docs, vocab=args.vocab, workers=8)
    dev = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
    terms = [int(t) for t in synth.bm25_queries(64, vocab=args.vocab).reshape(-1)]
    idf = dev.idfs(terms)
    want = {t: dev.bm25_dense([t], idf=idf[i:i + 1]).copy() for i, t in list(enumerate(terms))[:8]}

    def call(i):
        j = i % len(terms)
        out = dev.bm25_dense([terms[j]], idf=idf[j:j + 1])
        return float(out[0])                                            # (touch the result; the buffer goes back to the pool)
    base = None
    for n_threads in (1, 2, 4, 8, 16):
        with ThreadPoolExecutor(n_threads) as ex:
            list(ex.map(call, range(n_threads * 4)))                    # warm (lanes, pinned buffers)
            t0 = time.perf_counter()
            list(ex.map(call, range(args.calls)))
            dt = time.perf_counter() - t0
        rate = args.calls / dt
        base = base or rate
        print(json.dumps({"docs": args.docs, "threads": n_threads, "calls": args.calls, "calls_per_s": round(rate, 1),
                          "ms_per_call": round(dt / args.calls * 1e3, 4), "speedup_vs_1_thread": round(rate / base, 2),
                          "host_GBps": round(rate * 4 * args.docs / 1e9, 2)}), flush=True)
    for t, w in want.items():                                           # results under threads equal the single-thread ones
        with ThreadPoolExecutor(8) as ex:
            outs = list(ex.map(lambda _: dev.bm25_dense([t], idf=idf[terms.index(t):terms.index(t) + 1]).copy(), range(16)))
        assert all(np.array_equal(o, w) for o in outs), f"term {t}: a threaded call returned another result"
    # one single-phrase call: wall vs device
    ph = [0, 1, 2]
    dev.bm25_phrase_dense(ph)
    walls, devs = [], []
    for _ in range(50):
        t0 = time.perf_counter()
        dev.bm25_phrase_dense(ph)
        walls.append(time.perf_counter() - t0)
        devs.append(dev.last_profile()[0])
    print(json.dumps({"docs": args.docs, "single_phrase_call": "t0 t1 t2", "wall_ms_median": round(float(np.median(walls)) * 1e3, 4),
                      "device_ms_median": round(float(np.median(devs)), 4)}), flush=True)
    dev.close()


if __name__ == "__main__":
    main()

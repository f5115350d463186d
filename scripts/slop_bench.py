#!/usr/bin/env python
"""Config 5 stand-in (BASELINE.json; MSMARCO is not available offline): synthetic Zipf docs, 2- and
3-token phrases with slop=2 on one MI355X.  Times SearchArray.termfreqs-style dense results
(float32[N] copied to the host) and the device-only part, next to the CPU oracle (the C restatement
of the reference's span search), and checks the counts bit-exact."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth, _lib                              # noqa: E402
from searcharray_amd.device_index import DeviceIndex                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--phrases", type=int, default=32)
    ap.add_argument("--cpu-phrases", type=int, default=4)
    ap.add_argument("--slop", type=int, default=2)
    args = ap.parse_args()
    api = _lib.api()
    D, V = args.docs, args.vocab
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    out_words, term_off = synth.concat_term_major([(words, counts)], V)
    doc_lens = lens.astype(np.float32)
    index = DeviceIndex(out_words, term_off, doc_lens, api=api)
    from oracle import refimpl as O
    orc = O.OracleIndex(out_words, np.arange(V), term_off, doc_lens, D)
    res = {}
    for length in (2, 3):
        phrases = [list(range(length))] + [[int(t) for t in p]
                                           for p in synth.phrase_queries_from_tokens(lens, terms, args.phrases, length, seed=5)]
        index.phrase_freqs_dense(phrases[0], slop=args.slop)
        index.phrase_freqs_dense([5000 + j for j in range(length)], slop=args.slop)      # (the general route's kernels loaded too: terms without a directory row)
        t0 = time.perf_counter()
        outs, kms, kbytes = [], 0.0, 0
        for i, p in enumerate(phrases):
            r = index.phrase_freqs_dense(p, slop=args.slop)
            outs.append(r if i <= args.cpu_phrases else None)     # (retaining every result would page-lock a buffer per call)
            ms, ab = index.last_profile()
            kms += ms
            kbytes += ab
        dt = time.perf_counter() - t0
        ms0, ab0 = 1e9, 0
        for _ in range(3):                                        # the heaviest phrase alone: best of three
            index.phrase_freqs_dense(phrases[0], slop=args.slop)
            m, ab0 = index.last_profile()
            ms0 = min(ms0, m)
        t0 = time.perf_counter()
        ok = True
        ncpu = min(args.cpu_phrases, len(phrases))
        for i in range(1, ncpu + 1):
            want = orc.phrase_freqs(phrases[i % len(phrases)], slop=args.slop)
            ok &= bool(np.array_equal(want, outs[i % len(phrases)]))
        cpu_dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        want0 = orc.phrase_freqs(phrases[0], slop=args.slop)
        cpu0 = time.perf_counter() - t0
        ok &= bool(np.array_equal(want0, outs[0]))
        # BM25 scores against the oracle (tolerance of BASELINE.json's north_star: 1e-5 relative)
        s_dev = index.bm25_phrase_dense(phrases[1], slop=args.slop)
        s_cpu = orc.score(phrases[1], slop=args.slop)
        # device-resident throughput: the same phrases as ONE slop batch -> BM25 -> top-10 (nothing but B x k results leaves HBM)
        pb = index.phrase_batch(phrases, k=10, slop=args.slop)
        pb.run()
        tb = time.perf_counter()
        reps = 5
        for _ in range(reps):
            pb.run(sync=False)
        index.synchronize()
        batch_dt = (time.perf_counter() - tb) / reps
        bs, bd = pb.fetch()
        ws, wd = O.topk(orc.score(phrases[1], slop=args.slop), 10)
        batch_ok = bool(np.array_equal(bs[1], ws) and np.array_equal(bd[1][ws > 0], wd[ws > 0]))
        pb.close()
        res[f"{length}_terms"] = {
            "slop_batch_phrases_per_s": round(len(phrases) / batch_dt, 1), "slop_batch_top10_matches_oracle": batch_ok,
            "phrases": len(phrases), "ms_per_phrase": round(dt / len(phrases) * 1e3, 3),
            "phrases_per_s": round(len(phrases) / dt, 1), "device_ms_per_phrase": round(kms / len(phrases), 4),
            "device_alg_GBps": round(kbytes / max(kms, 1e-9) / 1e6, 1),
            "heaviest_device_ms": round(ms0, 4), "heaviest_alg_GBps": round(ab0 / ms0 / 1e6, 1),
            "heaviest_matches": int(want0.sum()), "cpu_oracle_ms_per_phrase": round(cpu_dt / ncpu * 1e3, 2),
            "cpu_oracle_heaviest_ms": round(cpu0 * 1e3, 2), "counts_bit_exact": ok,
            "score_max_rel_err": float(np.max(np.abs(s_dev - s_cpu) / np.maximum(np.abs(s_cpu), 1e-30))) if s_cpu.any() else 0.0}
    print(json.dumps({"config": f"zipf-{D} slop={args.slop} phrases", **res}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Races between the index stream and the side stream / lanes would show as run-to-run differences: a mixed batch
(groups with a shared first term + loose groups + queries for the per-query kernel on the side stream) and a slop
batch (two lanes) are run many times back to back, asynchronously and interleaved, and every fetch must equal the
first one (which tests/test_config_scale.py checks against the oracle)."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth, _lib                              # noqa: E402
from searcharray_amd.device_index import DeviceIndex                 # noqa: E402


def main():
    D, V = 1_000_000, 100_000
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    words, term_off = synth.concat_term_major([(words, counts)], V)
    dev = DeviceIndex(words, term_off, lens.astype(np.float32), api=_lib.api())
    os.environ["SA_SPARSE"] = "0"
    mixed = np.concatenate([synth.bm25_queries(96, vocab=V), synth.bm25_queries_distinct(160, vocab=V)[:160]])
    b1 = dev.batch(mixed, k=10)
    b2 = dev.batch(mixed[::-1].copy(), k=100)
    phrases = [[0, 1]] + [[int(t) for t in p] for p in synth.phrase_queries_from_tokens(lens, terms, 24, 2, seed=5)]
    pb = dev.phrase_batch(phrases, k=10, slop=2)
    print("grouping:", b1.group_info())
    ref = None
    bad = 0
    for it in range(40):
        for _ in range(1 + it % 3):
            b1.run(sync=False)
            pb.run(sync=False)
            b2.run(sync=False)
        got = (b1.fetch(), b2.fetch(), pb.fetch())
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
                    bad += 1
    print(f"stream stress done: 40 rounds, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Device time of single slop-2 phrases through the doc-parallel route (SA_SPAN_DOC=2: whenever the phrase qualifies) and
the general route (SA_SPAN_DOC=0), zipf-1M: where the list-length rule of sa_span_counts_device should sit."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import json
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
    out_words, term_off = synth.concat_term_major([(words, counts)], V)
    index = DeviceIndex(out_words, term_off, lens.astype(np.float32), api=_lib.api())
    nwords = np.diff(term_off)
    phrases = [(0, 1), (0, 2), (1, 2), (2, 3), (0, 5), (3, 4), (0, 10), (5, 6), (1, 20), (0, 30), (10, 11), (8, 9), (20, 30), (15, 16), (1, 2, 3), (3, 4, 5), (5, 8, 9)]
    phrases += [tuple(int(t) for t in q) for q in synth.phrase_queries_from_tokens(lens, terms, 32, 2, seed=5)]      # (slop_bench.py's sample)
    for ph in phrases:
        row = {"phrase": list(ph), "words": [int(nwords[t]) for t in ph]}
        for mode in ("2", "0"):
            os.environ["SA_SPAN_DOC"] = mode
            ms = []
            for _ in range(5):
                r = index.phrase_freqs_dense(list(ph), slop=2)
                ms.append(index.last_profile()[0])
            row["doc_route_ms" if mode == "2" else "general_ms"] = round(min(ms[1:]), 4)
            row["first_ms_doc" if mode == "2" else "first_ms_general"] = round(ms[0], 4)
            row["matches"] = int(r.sum())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

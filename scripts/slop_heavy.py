#!/usr/bin/env python
"""The heaviest slop phrases of scripts/slop_bench.py ([t0 t1] and [t0 t1 t2], slop 2, zipf-1M) alone, a few times
each: run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown of one slop query."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth, _lib                              # noqa: E402
from searcharray_amd.device_index import DeviceIndex                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--terms", default="2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--slop", type=int, default=2)
    args = ap.parse_args()
    D, V = args.docs, args.vocab
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    out_words, term_off = synth.concat_term_major([(words, counts)], V)
    index = DeviceIndex(out_words, term_off, lens.astype(np.float32), api=_lib.api())
    for ce in (0,):
        for nt in [int(x) for x in args.terms.split(",")]:
            ph = list(range(nt))
            ms = []
            for _ in range(args.reps + 1):
                r = index.phrase_freqs_dense(ph, slop=args.slop)
                ms.append(round(index.last_profile()[0], 4))
            print(json.dumps({"phrase": ph, "slop": args.slop, "device_ms": ms, "matches": int(r.sum())}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Differential run at scale (development tool, GPU): random slop phrases and repeated-term phrases over the frequent
terms of zipf-1M, each through the one-launch route (doc-parallel slop / phrase chain per document) and through the
general route (SA_SPAN_DOC=0 / SA_PHRASE_DOCS=0), which tests pin to the oracle -- dense counts must be identical; a few
are checked against the CPU oracle as well."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
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
    ap.add_argument("--n", type=int, default=120)
    ap.add_argument("--oracle", type=int, default=6)
    args = ap.parse_args()
    D, V = args.docs, 100_000
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    out_words, term_off = synth.concat_term_major([(words, counts)], V)
    doc_lens = lens.astype(np.float32)
    index = DeviceIndex(out_words, term_off, doc_lens, api=_lib.api())
    from oracle import refimpl as O
    orc = O.OracleIndex(out_words, np.arange(V), term_off, doc_lens, D)
    rng = np.random.default_rng(77)
    bad = 0
    checked = {"slop": 0, "phrase": 0, "oracle": 0}
    for i in range(args.n):
        T = int(rng.integers(2, 5))
        ph = [int(x) for x in rng.integers(0, 70, T)]
        slop = int(rng.integers(1, 7))
        os.environ.pop("SA_SPAN_DOC", None)
        a = index.phrase_freqs_dense(ph, slop=slop)
        os.environ["SA_SPAN_DOC"] = "0"
        b = index.phrase_freqs_dense(ph, slop=slop)
        os.environ.pop("SA_SPAN_DOC", None)
        checked["slop"] += 1
        if not np.array_equal(a, b):
            bad += 1
            print(json.dumps({"slop_mismatch": ph, "slop": slop, "docs": np.flatnonzero(a != b)[:5].tolist()}), flush=True)
        if i < args.oracle and min(ph) > 3:
            checked["oracle"] += 1
            if not np.array_equal(a, orc.phrase_freqs(ph, slop=slop)):
                bad += 1
                print(json.dumps({"slop_vs_oracle_mismatch": ph, "slop": slop}), flush=True)
    for i in range(args.n):
        T = int(rng.integers(2, 9))
        ph = [int(x) for x in rng.integers(0, 12, T)]
        if len(set(ph)) == len(ph):
            ph[-1] = ph[0]
        os.environ.pop("SA_PHRASE_DOCS", None)
        a = index.phrase_freqs_dense(ph)
        os.environ["SA_PHRASE_DOCS"] = "0"
        b = index.phrase_freqs_dense(ph)
        os.environ.pop("SA_PHRASE_DOCS", None)
        checked["phrase"] += 1
        if not np.array_equal(a, b):
            bad += 1
            print(json.dumps({"phrase_mismatch": ph, "docs": np.flatnonzero(a != b)[:5].tolist()}), flush=True)
        if i < args.oracle:
            checked["oracle"] += 1
            if not np.array_equal(a, orc.phrase_freqs(ph)):
                bad += 1
                print(json.dumps({"phrase_vs_oracle_mismatch": ph}), flush=True)
    print(json.dumps({"checked": checked, "mismatches": bad}))


if __name__ == "__main__":
    main()

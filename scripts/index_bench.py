#!/usr/bin/env python
"""SearchArray.index end to end: Python tokenizer + term dictionary on the host, sort-by-term +
roaringish encode + tf/df derivation on the device (csrc/sa_build.hip), next to the host encoder
(the numpy restatement of the reference's indexer) on the same token stream."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import SearchArray, synth, roaringish as rz          # noqa: E402
from searcharray_amd.device_index import DeviceIndex                      # noqa: E402
from searcharray_amd.indexing import _triples                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=500_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    args = ap.parse_args()
    D, V = args.docs, args.vocab
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    starts = np.concatenate([[0], np.cumsum(lens)])
    names = np.array([f"t{i}" for i in range(V)])
    docs = [" ".join(names[terms[starts[i]:starts[i + 1]]]) for i in range(D)]
    t0 = time.perf_counter()
    arr = SearchArray.index(docs)                     # autowarm: device build included
    arr._core.device().synchronize()
    t_index = time.perf_counter() - t0
    h = arr._core.host
    t0 = time.perf_counter()
    dev = DeviceIndex.from_tokens(h.tokens, h.doc_ptr, len(arr.term_dict), doc_lens=h.doc_lens)
    dev.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    t, d, p = _triples(np.diff(h.doc_ptr.astype(np.int64)), h.tokens)
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, len(arr.term_dict))
    t_host = time.perf_counter() - t0
    got_words, got_off = dev.words()
    print(json.dumps({"docs": D, "tokens": int(len(h.tokens)), "words": int(len(words)),
                      "searcharray_index_s": round(t_index, 3), "device_build_s_incl_h2d": round(t_dev, 4),
                      "host_encode_s": round(t_host, 3), "byte_identical": bool(np.array_equal(got_words, words) and np.array_equal(got_off, off)),
                      "score_check": float(arr.score("t5").sum())}))


if __name__ == "__main__":
    main()

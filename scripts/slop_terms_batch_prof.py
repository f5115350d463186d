#!/usr/bin/env python
"""A batch of 256 slop-2 phrases of T terms (default 3; the bench's slop_batch recipe: half on ranks 50-5000, half on ranks 1-50) on
zipf-1M -> top-10: time per batch, and the batch's top-k against the one-launch-per-phrase route (option span_doc_multi = 0)."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from searcharray_amd import _lib, options               # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
api = _lib.api()
side = bench.PhraseSide(api, 1_000_000, 100_000)
rng = np.random.default_rng(9)
phrases = []
for i in range(256):
    lo, hi = (49, 5000) if i % 2 == 0 else (0, 50)
    ph = []
    while len(ph) < T:
        x = int(rng.integers(lo, hi))
        if x not in ph:
            ph.append(x)
    phrases.append(ph)
b = side.index.phrase_batch(phrases, k=K, slop=2)
dt, kms = side.timed(b, 3, 20)
got = b.fetch()
out = {"terms": T, "k": K, "ms_per_step": round(dt / 20 * 1e3, 4), "kernel_ms": round(kms, 4), "word_bytes": side.word_bytes(phrases)}
if os.environ.get("SA_CHECK", "1") != "0":
    with options.scoped(span_doc_multi=0):
        b1 = side.index.phrase_batch(phrases[:48], k=K, slop=2)
        b1.run()
        want = b1.fetch()
        b1.close()
    out["first_48_equal_one_launch_per_phrase"] = bool(np.array_equal(got[0][:48], want[0]) and np.array_equal(got[1][:48], want[1]))
print(json.dumps(out))
b.close()
side.close()

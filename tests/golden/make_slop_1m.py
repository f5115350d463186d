"""Generate tests/golden/slop_1m.npz: the REFERENCE's slop-2 match counts and BM25 top-10 at BASELINE config 5's scale.

The reference (oracle/_ref = /root/reference built by oracle/build_ref.sh) needs up to seconds per query here
(roaringish/spans.pyx:189-319), so this is an offline job of the build container (about five minutes):

    bash oracle/build_ref.sh && python tests/golden/make_slop_1m.py

Corpus: zipf-1M (V = 100k, Poisson(32) lengths, seed 1234 -- searcharray_amd.synth, the corpus of
tests/test_config_scale.py and of bench.py's phrase side).  Queries:

  set "t"  the 32 two-term queries of tests/test_config_scale.py::test_config5_slop2_at_1m_docs (ranks 50-5000)
  set "b"  of bench.py's 256 slop-2 phrases (PhraseSide.slop2): the first 32 mid-frequency ones, the first 8 on the
           most frequent terms and the four heaviest by posting words

Per query the file holds the reference's `termfreqs(tokens, slop=2)` as (nnz, sum, sha1 of the float32[1M] vector) --
a bit-exact pin whatever the number of matches -- the sparse (doc, count) pairs when there are at most 20 000 of them,
and the top-10 (scores, docs) of the reference's `score(tokens, slop=2)` (utils/sort.py:24 order).  Only outputs are stored.
"""
import hashlib
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from oracle import ref_loader                                     # noqa: E402
from oracle import refimpl as O                                   # noqa: E402
from searcharray_amd import synth                                 # noqa: E402

D, V = 1_000_000, 100_000
SPARSE_MAX = 20_000


def test_queries():
    rng = np.random.default_rng(5)
    out = []
    for _ in range(32):
        a, b = (int(x) for x in rng.integers(49, 5000, 2))
        out.append([a, b + 1 if a == b else b])
    return out


def bench_queries(term_off):
    rng = np.random.default_rng(5)
    allq = []
    for i in range(256):
        lo, hi = (49, min(5000, V - 1)) if i % 2 == 0 else (0, min(50, V - 1))
        a, b = (int(x) for x in rng.integers(lo, hi, 2))
        allq.append([a, b + 1 if a == b else b])
    weight = [int(sum(term_off[t + 1] - term_off[t] for t in q)) for q in allq]
    pick = [i for i in range(0, 64, 2)] + [i for i in range(1, 16, 2)] + [int(i) for i in np.argsort(weight)[-4:]]
    pick = sorted(set(pick))
    return pick, [allq[i] for i in pick]


def digest(v):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(v, dtype=np.float32).tobytes()).digest(), dtype=np.uint8)


def main():
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    words, term_off = synth.concat_term_major([(words, counts)], V)
    sa = ref_loader.reference_array(words, term_off, lens.astype(np.float32))
    tq = test_queries()
    bidx, bq = bench_queries(term_off)
    out = {"meta": np.asarray([D, V], dtype=np.int64), "t_queries": np.asarray(tq, dtype=np.int64),
           "b_queries": np.asarray(bq, dtype=np.int64), "b_index": np.asarray(bidx, dtype=np.int64)}
    for tag, qs in (("t", tq), ("b", bq)):
        for i, q in enumerate(qs):
            t0 = time.time()
            toks = [f"t{t}" for t in q]
            tf = np.asarray(sa.termfreqs(toks, slop=2), dtype=np.float32)
            sc = np.asarray(sa.score(toks, slop=2), dtype=np.float32)
            nz = np.flatnonzero(tf)
            out[f"{tag}{i}_stat"] = np.asarray([len(nz), float(tf.sum())], dtype=np.float64)
            out[f"{tag}{i}_sha1"] = digest(tf)
            if len(nz) <= SPARSE_MAX:
                out[f"{tag}{i}_idx"], out[f"{tag}{i}_val"] = nz.astype(np.uint32), tf[nz]
            ws, wd = O.topk(sc, 10)
            out[f"{tag}{i}_top_scores"], out[f"{tag}{i}_top_docs"] = ws, wd
            print(f"{tag}{i} {q}: {len(nz)} docs, {int(tf.sum())} matches, {time.time() - t0:.1f}s", flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slop_1m.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

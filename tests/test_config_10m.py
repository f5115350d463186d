"""BASELINE config 4 at its FULL size under pytest (GPU only): the zipf-10M corpus, the 256 x 4-term BASELINE query set
(seed 42), top-10.  Every exhaustive route of the library -- the grouped kernel (the default), the per-query tile
kernel (SA_GROUP=0) -- and dynamic pruning (SA_SPARSE=1) must return the same keys for
ALL 256 queries, and the first 16 queries plus the probe query must equal the CPU oracle's dense scores (the reference's
np.sum of per-term score vectors, test/test_msmarco.py:345-395) + deterministic top-k, bit for bit.  A fresh query set
through sa_batch_step (the bench's step) is checked the same way.

The corpus is generated once per session (about a minute on the GPU box's host; `SA_CORPUS_CACHE` -- default /tmp/corpus --
keeps the encoded shard for later runs)."""
import os

import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import synth
from searcharray_amd.device_index import DeviceIndex
from tests.helpers import set_opt, unset_opt

pytestmark = pytest.mark.gpu
D, V, K = 10_000_000, 100_000, 10


@pytest.fixture(scope="module")
def zipf10m():
    from searcharray_amd import _lib
    api = _lib.api()
    cache = os.environ.get("SA_CORPUS_CACHE", "/tmp/corpus")
    path = os.path.join(cache, f"zipf_{D}_{V}_0_{D}.npz")
    if os.path.exists(path):
        z = np.load(path)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=min(8, os.cpu_count() or 1))
        try:
            os.makedirs(cache, exist_ok=True)
            np.savez(path, words=corpus.words, term_off=corpus.term_off, doc_lens=corpus.doc_lens)
        except OSError:
            pass
    dev = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, api=api)
    orc = O.OracleIndex(corpus.words, np.arange(V), corpus.term_off, corpus.doc_lens, D)
    yield dev, orc
    dev.close()


def run(dev, queries, env, monkeypatch):
    for k_ in ("SA_SPARSE", "SA_GROUP", "SA_GROUP_ITEM"):
        unset_opt(k_)
    for k_, v in env.items():
        set_opt(k_, v)
    bt = dev.batch(queries, k=K)
    bt.run()
    res = bt.fetch()
    gi = bt.group_info()
    bt.close()
    return res, gi


def check_oracle(orc, queries, scores, docs, rows):
    for qi in rows:
        ws, wd = O.topk(orc.score_terms_sum([int(t) for t in queries[qi]]), K)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} scores vs oracle"
        assert np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} docs vs oracle"


def test_config4_all_routes_agree_and_equal_the_oracle(zipf10m, monkeypatch):
    dev, orc = zipf10m
    queries = synth.bm25_queries(256, vocab=V)
    (grouped, gi) = run(dev, queries, {"SA_SPARSE": "0"}, monkeypatch)
    assert gi["grouped_queries"] >= 200, gi
    (per_query, _) = run(dev, queries, {"SA_SPARSE": "0", "SA_GROUP": "0"}, monkeypatch)
    (pruned, _) = run(dev, queries, {"SA_SPARSE": "1"}, monkeypatch)
    # the default at this size: items of up to 32 queries in two table passes; one-pass items and four-pass items beside it
    (one_pass, g16) = run(dev, queries, {"SA_SPARSE": "0", "SA_GROUP_ITEM": "16"}, monkeypatch)
    (four_pass, g64) = run(dev, queries, {"SA_SPARSE": "0", "SA_GROUP_ITEM": "64"}, monkeypatch)
    assert g64["groups"] < gi["groups"] < g16["groups"], (g64, gi, g16)
    for name, got in (("per-query", per_query), ("pruned", pruned), ("one-pass items", one_pass), ("four-pass items", four_pass)):
        assert np.array_equal(grouped[0], got[0]), f"scores: grouped vs {name}"
        assert np.array_equal(grouped[1], got[1]), f"docs: grouped vs {name}"
    check_oracle(orc, queries, grouped[0], grouped[1], range(17))       # the probe query t0 t9 t99 t999 and 16 more


def test_config4_fresh_query_sets_through_one_call_per_step(zipf10m, monkeypatch):
    """the bench's step: sa_batch_step (idf gathered from the index's table + reset + run) on batch objects in flight"""
    dev, orc = zipf10m
    set_opt("SA_SPARSE", "0")
    df = dev.docfreqs().astype(np.float64)
    dev.set_idf_table(np.log(1 + (D - df + 0.5) / (df + 0.5)).astype(np.float32))
    sets = [synth.bm25_queries(256, vocab=V, seed=1000 + i) for i in range(3)]
    ring = [dev.batch(sets[0], k=K) for _ in range(2)]
    got = {}
    for i, qs in enumerate(sets):
        b = ring[i % 2]
        if i >= 2:
            got[i - 2] = b.fetch()
        b.step(np.ascontiguousarray(qs, dtype=np.uint32))
    for i in range(max(0, len(sets) - 2), len(sets)):
        got[i] = ring[i % 2].fetch()
    for i, qs in enumerate(sets):
        check_oracle(orc, qs, got[i][0], got[i][1], (0, 100, 255))
    for b in ring:
        b.close()

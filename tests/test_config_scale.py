"""Parity at the scale of BASELINE.json's configurations (GPU only; the CPU suite covers the same paths on small
corpora).  One seeded zipf-1M corpus (V = 100k, Poisson(32) lengths, seed 1234), then

  config 2  256 x 4-term disjunctive BM25, top-10 and top-1000: grouped exhaustive kernel, per-query exhaustive
            kernel and dynamic pruning agree on every query; ALL 256 queries equal the oracle's dense scores + top-k
            (reference shapes: test/test_msmarco.py:345-395, utils/sort.py:24)
            + the same on 256 queries of pairwise-distinct terms (loose groups, side stream)
  config 3  the 64 sampled consecutive trigrams + `t0 t1 t2` + the same-term set: match counts np.array_equal
            the oracle's (reference shapes: test/test_msmarco.py:227-295, test_phrase_matches.py:73-117)
  config 5  32 two-term slop-2 queries on mid-frequency terms (ranks 50-5000): counts exact, BM25 within 1e-5
            relative of the oracle (reference shapes: test/test_msmarco.py:247-254)
The oracle (oracle/refimpl.py + the C restatements) is pinned to the reference's outputs by tests/golden."""
import os

import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import synth
from searcharray_amd.device_index import DeviceIndex

pytestmark = pytest.mark.gpu

D, V = 1_000_000, 100_000


@pytest.fixture(scope="module")
def zipf1m():
    from searcharray_amd import _lib
    api = _lib.api()
    lens, terms = synth.zipf_batch_tokens(0, D, V, fast=True)
    words, counts = synth.encode_batch(lens, terms, V)
    words, term_off = synth.concat_term_major([(words, counts)], V)
    doc_lens = lens.astype(np.float32)
    dev = DeviceIndex(words, term_off, doc_lens, api=api)
    orc = O.OracleIndex(words, np.arange(V), term_off, doc_lens, D)
    yield dev, orc, lens, terms
    dev.close()


def run_batch(dev, queries, k, env):
    bt = dev.batch(queries, k=k, opts=env)                     # (the routes are options of the batch: "SA_SPARSE" -> sparse, ...)
    bt.run()
    res = bt.fetch()
    bt.close()
    return res


@pytest.mark.parametrize("k", [10, 1000])
def test_config2_bm25_topk_at_1m_docs(zipf1m, k):
    dev, orc, _, _ = zipf1m
    queries = synth.bm25_queries(256, vocab=V)
    grouped = run_batch(dev, queries, k, {"SA_SPARSE": "0", "SA_GROUP": "1"})
    per_query = run_batch(dev, queries, k, {"SA_SPARSE": "0", "SA_GROUP": "0"})
    pruned = run_batch(dev, queries, k, {"SA_SPARSE": "1"})
    for name, got in (("per-query", per_query), ("pruned", pruned)):
        assert np.array_equal(grouped[0], got[0]), f"scores: grouped vs {name}"
        assert np.array_equal(grouped[1], got[1]), f"docs: grouped vs {name}"
    for qi in range(len(queries)):                                   # ALL queries against the oracle, bit for bit
        ws, wd = O.topk(orc.score_terms_sum([int(t) for t in queries[qi]]), k)
        n = int((ws > 0).sum())
        assert np.array_equal(grouped[0][qi, :n], ws[:n]), f"q{qi} scores vs oracle"
        assert np.array_equal(grouped[1][qi, :n], wd[:n]), f"q{qi} docs vs oracle"


@pytest.mark.parametrize("k", [10, 100])
def test_config2_queries_without_shared_terms_at_1m_docs(zipf1m, k):
    """256 queries of pairwise-distinct terms (no posting list shared): the exhaustive path scores the sparse ones as
    LOOSE groups and the ones with a dense term as groups of ONE (round 6; option group_one = 0: the per-query kernel on the
    side stream, as rounds 3-5 did) -- equal to each other, to the per-query kernel alone, to dynamic pruning, and (all queries)
    to the oracle"""
    dev, orc, _, _ = zipf1m
    queries = synth.bm25_queries_distinct(256, vocab=V)
    bt = dev.batch(queries, k=k)
    gi = bt.group_info()
    bt.close()
    bt = dev.batch(queries, k=k, opts={"group_one": 0})
    gi0 = bt.group_info()
    bt.close()
    assert gi["shared_first_term"] == 0 and gi["grouped_queries"] == 256 and gi["per_query_kernel"] == 0, gi
    assert gi0["shared_first_term"] == 0 and gi0["grouped_queries"] >= 128 and gi0["per_query_kernel"] >= 1, gi0
    loose = run_batch(dev, queries, k, {"SA_SPARSE": "0", "SA_GROUP": "1"})
    side = run_batch(dev, queries, k, {"SA_SPARSE": "0", "SA_GROUP": "1", "group_one": 0})
    assert np.array_equal(loose[0], side[0]) and np.array_equal(loose[1], side[1]), "groups of one vs the per-query kernel on the side stream"
    per_query = run_batch(dev, queries, k, {"SA_SPARSE": "0", "SA_GROUP": "0"})
    pruned = run_batch(dev, queries, k, {"SA_SPARSE": "1"})
    for name, got in (("per-query", per_query), ("pruned", pruned)):
        assert np.array_equal(loose[0], got[0]), f"scores: loose groups vs {name}"
        assert np.array_equal(loose[1], got[1]), f"docs: loose groups vs {name}"
    for qi in range(len(queries)):                                   # ALL queries against the oracle, bit for bit
        ws, wd = O.topk(orc.score_terms_sum([int(t) for t in queries[qi]]), k)
        n = int((ws > 0).sum())
        assert np.array_equal(loose[0][qi, :n], ws[:n]), f"q{qi} scores vs oracle"
        assert np.array_equal(loose[1][qi, :n], wd[:n]), f"q{qi} docs vs oracle"


def test_config3_trigram_counts_at_1m_docs(zipf1m):
    dev, orc, lens, terms = zipf1m
    phrases = [[0, 1, 2]] + [[int(t) for t in p] for p in synth.phrase_queries_from_tokens(lens, terms, 64, 3, seed=7)]
    phrases += [[0, 0], [0, 0, 1], [1, 1, 1]]                       # the same-term rule (bigram_freqs.py:48-101)
    for ph in phrases:
        got = dev.phrase_freqs_dense(ph)
        assert np.array_equal(got, orc.phrase_freqs(ph)), f"phrase {ph}"
    assert dev.phrase_freqs_dense(phrases[1]).sum() >= 1            # sampled from a real doc: at least one match
    # ... and ranked on the device: the phrase batch's top-10 of the 64 sampled trigrams
    pb = dev.phrase_batch(phrases[:65], k=10)
    pb.run()
    ps, pd_ = pb.fetch()
    pb.close()
    for i in (0, 1, 17, 40, 64):
        ws, wd = O.topk(orc.score(phrases[i]), 10)
        n = int((ws > 0).sum())
        assert np.array_equal(ps[i, :n], ws[:n]) and np.array_equal(pd_[i, :n], wd[:n]), f"phrase batch {phrases[i]}"


def test_config5_slop2_at_1m_docs(zipf1m):
    dev, orc, _, _ = zipf1m
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(32):
        a, b = (int(x) for x in rng.integers(49, 5000, 2))
        if a == b:
            b += 1
        got = dev.phrase_freqs_dense([a, b], slop=2)
        want = orc.phrase_freqs([a, b], slop=2)
        assert np.array_equal(got, want), f"slop-2 counts [{a}, {b}]"
        s_dev = dev.bm25_phrase_dense([a, b], slop=2)
        s_cpu = orc.score([a, b], slop=2)
        nz = s_cpu != 0
        assert np.array_equal(s_dev != 0, nz)
        if nz.any():
            worst = max(worst, float(np.max(np.abs(s_dev[nz] - s_cpu[nz]) / np.abs(s_cpu[nz]))))
    assert worst <= 1e-5, worst


def test_config5_slop2_equals_the_reference_golden(zipf1m):
    """tests/golden/slop_1m.npz: the REFERENCE's own termfreqs(slop=2) and BM25 top-10 on this corpus (tests/golden/make_slop_1m.py,
    oracle/_ref) for the 32 queries above and 44 of the bench's, the four heaviest included -- the device's counts bit for bit (sha1 of
    the float32[1M] vector, plus the sparse pairs where the fixture holds them), its slop scores within 1e-5 relative"""
    import hashlib
    from searcharray_amd import options
    from tests.helpers import load_golden
    dev, _, _, _ = zipf1m
    g = load_golden("slop_1m")
    assert [int(x) for x in g["meta"]] == [D, V]
    for tag in "tb":
        qs = [[int(x) for x in q] for q in g[f"{tag}_queries"]]
        # (both instances of the batch launch: four and two of a block's waves with span tables -- the size of the launch decides otherwise)
        tops = []
        for tw in (4, 2):
            with options.scoped(span_tab_waves=tw):
                pb = dev.phrase_batch(qs, k=10, slop=2)
                pb.run()
                tops.append(pb.fetch())
                pb.close()
        assert np.array_equal(tops[0][0], tops[1][0]) and np.array_equal(tops[0][1], tops[1][1]), "two vs four table waves per block"
        ps = tops[1][0]
        for i, ph in enumerate(qs):
            got = dev.phrase_freqs_dense(ph, slop=2)
            sha = np.frombuffer(hashlib.sha1(np.ascontiguousarray(got, dtype=np.float32).tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(sha, g[f"{tag}{i}_sha1"]), f"slop-2 counts {ph} vs the reference"
            if f"{tag}{i}_idx" in g.files:
                nz = np.flatnonzero(got)
                assert np.array_equal(nz, g[f"{tag}{i}_idx"]) and np.array_equal(got[nz], g[f"{tag}{i}_val"])
            ws = g[f"{tag}{i}_top_scores"]
            n = int((ws > 0).sum())
            assert np.allclose(ps[i, :n], ws[:n], rtol=1e-5, atol=0) and not (ps[i, n:] > 0).any(), f"slop-2 top-10 {ph} vs the reference"

"""Head-group kernel (csrc/sa_bm25_hg.hip, sa_k_bm25_headgroup; the alternative to the grouped kernel, SA_HG=1): queries
that share their FIRST term are scored against a read-only base of that term, one workgroup per (super-tile of 4096 docs,
group).  A query's longest further list is streamed (base + s); the postings of its other terms are candidate docs,
registered in a per-wave candidate map in LDS, that collect every contribution of their doc through mailboxes -- the
streamed term's factor, the other sparse list's value -- and add them in query-term order.  Results must equal the oracle's
dense score (the reference's np.sum of per-term score vectors, test/test_msmarco.py:353-354) + deterministic top-k bit for
bit, and the grouped / per-query kernels' (SA_HG=0, SA_GROUP=0)."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import roaringish as rz, synth
from searcharray_amd.device_index import DeviceIndex

N_DOCS, VOCAB = 11000, 400          # 6 tiles of 2048 docs; df of rank r ~ 23 K / r


@pytest.fixture(autouse=True)
def head_groups_on(monkeypatch):
    monkeypatch.setenv("SA_HG", "1")


@pytest.fixture(scope="module")
def corpus():
    t, d, p, lens = synth.corpus_triples(N_DOCS, VOCAB, 14, seed=77)
    words, wt = rz.encode_sorted(t, d, p)
    return words, rz.term_offsets(wt, VOCAB), lens, O.OracleIndex.from_triples(t, d, p, N_DOCS, doc_lens=lens)


def run(api, corpus, queries, k, doc_base=0, idf=None, want_hg=None):
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=2048, doc_base=doc_base, api=api)
    bt = dev.batch(np.asarray(queries), k=k, idf=idf)
    gi = bt.group_info()
    if want_hg is not None:
        assert gi["head_group_queries"] >= want_hg, gi
    for _ in range(2):
        bt.run()
    scores, docs = bt.fetch()
    bt.close()
    dev.close()
    return scores, docs, gi


def check(api, corpus, queries, k, doc_base=0, want_hg=None):
    scores, docs, gi = run(api, corpus, queries, k, doc_base=doc_base, want_hg=want_hg)
    orc = corpus[3]
    for qi, q in enumerate(queries):
        ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q if 0 <= int(x) < VOCAB]), k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} {q} scores"
        assert np.array_equal(docs[qi, :n], wd[:n] + np.uint64(doc_base)), f"q{qi} {q} docs"
    return scores, docs, gi


def shaped(rng, n, T, heads, bands):
    q = np.empty((n, T), dtype=np.int64)
    q[:, 0] = rng.choice(heads, n)
    for t in range(1, T):
        lo, hi = bands[t - 1]
        q[:, t] = rng.integers(lo, hi, n)
    return q


@pytest.mark.parametrize("warm", ["0", "1"])
@pytest.mark.parametrize("T,k", [(2, 5), (3, 10), (4, 10), (4, 60)])
def test_headgroups_equal_oracle(api, corpus, monkeypatch, warm, T, k):
    """the BASELINE shape: a shared frequent first term, one dense term, sparse terms -- every query is in a head group;
    SA_GROUP_WARM=0: no tile establishes the bounds first, so the first pairs take the work list"""
    monkeypatch.setenv("SA_SPARSE", "0")
    monkeypatch.setenv("SA_GROUP_WARM", warm)
    rng = np.random.default_rng(40 + T + k)
    queries = shaped(rng, 36, T, heads=[0, 1, 4], bands=[(8, 60), (220, 400), (300, 400)])
    _, _, gi = check(api, corpus, queries, k, doc_base=70_000 if T == 3 else 0, want_hg=30)
    assert gi["head_groups"] >= 3


def test_headgroup_roles_and_odd_queries(api, corpus, monkeypatch):
    """the stream term in every position, repeated sparse terms (both lists hold the same docs: the lanes find each
    other), the shared term again among the further terms, unknown terms, sparse terms only, a dense term only, a group
    bigger than one round of a workgroup (4 waves x 16 queries)"""
    monkeypatch.setenv("SA_SPARSE", "0")
    monkeypatch.setenv("SA_GROUP_WARM", "1")
    queries = [[0, 300, 20, 350], [0, 300, 350, 20], [0, 20, 300, 350],          # stream term in positions 2, 3, 1
               [0, 330, 330, 12], [0, 12, 390, 390], [0, 399, 399, 399],         # repeated sparse terms
               [0, 0, 310, 15], [0, 9000, 25, 380], [0, 9000, 9001, 385],        # the head again; unknown terms
               [0, 350, 360, 370], [0, 30, 9000, 9000], [0, 9000, 9000, 9000],   # sparse only; dense only; nothing
               [2, 300, 310, 30], [2, 31, 320, 320]]
    queries += [[1, 10 + (i % 50), 250 + i, 399 - i] for i in range(75)]
    check(api, corpus, queries, 8, want_hg=80)
    # groups of one as head groups too
    monkeypatch.setenv("SA_HG_MIN", "1")
    check(api, corpus, queries[:14], 8, want_hg=13)


def test_candidate_lists_beyond_the_map_and_dense_candidate_terms(api, corpus, monkeypatch):
    """SA_HG_CAND_EXP raised: terms with hundreds of postings per super-tile count as candidates -- pairs whose candidate
    lists exceed the map (128 / 64 postings) go to the per-query kernel through the work list; the others register long
    lists with many docs in both"""
    monkeypatch.setenv("SA_SPARSE", "0")
    monkeypatch.setenv("SA_GROUP_WARM", "1")
    monkeypatch.setenv("SA_HG_CAND_EXP", "100000")
    rng = np.random.default_rng(5)
    queries = shaped(rng, 24, 4, heads=[0, 2], bands=[(5, 40), (30, 120), (60, 400)])
    check(api, corpus, queries, 10, want_hg=24)
    queries = shaped(rng, 24, 4, heads=[0, 2], bands=[(120, 400), (150, 400), (200, 400)])
    check(api, corpus, queries, 10, want_hg=24)


def test_headgroup_grouped_and_per_query_kernels_agree(api, corpus, monkeypatch):
    """explicit (non-reference) weights incl. zero weights and two weights for one first term (two groups): the head-group
    kernel, the grouped kernel (SA_HG=0) and the per-query kernel (SA_GROUP=0) return the same keys"""
    monkeypatch.setenv("SA_SPARSE", "0")
    monkeypatch.setenv("SA_GROUP_WARM", "1")
    rng = np.random.default_rng(11)
    queries = shaped(rng, 30, 4, heads=[0, 1], bands=[(8, 60), (220, 400), (300, 400)])
    idf = rng.uniform(0.1, 9.0, size=queries.shape).astype(np.float32)
    idf[:, 0] = np.where(rng.random(len(queries)) < 0.5, np.float32(0.25), np.float32(1.5))
    idf[3] = 0.0                                                     # a query that scores nothing
    idf[5, 1] = 0.0
    got = run(api, corpus, queries, 20, idf=idf, want_hg=20)
    monkeypatch.setenv("SA_HG", "0")
    grouped = run(api, corpus, queries, 20, idf=idf)
    assert grouped[2]["head_groups"] == 0
    monkeypatch.setenv("SA_GROUP", "0")
    per_query = run(api, corpus, queries, 20, idf=idf)
    for other in (grouped, per_query):
        assert np.array_equal(got[0], other[0]) and np.array_equal(got[1], other[1])


def test_reset_regroups_head_groups(api, corpus, monkeypatch):
    """a new query set in an existing batch (sa_batch_reset): roles and groups are rebuilt, results equal a fresh batch's"""
    monkeypatch.setenv("SA_SPARSE", "0")
    monkeypatch.setenv("SA_GROUP_WARM", "1")
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=2048, api=api)
    rng = np.random.default_rng(3)
    sets = [shaped(rng, 20, 4, heads=h, bands=[(8, 60), (220, 400), (300, 400)]) for h in ([0, 1], [3], [0, 5, 6])]
    bt = dev.batch(sets[0], k=10)
    for qs in sets[1:] + sets[:1]:
        bt.reset(qs)
        bt.run()
        scores, docs = bt.fetch()
        assert bt.group_info()["head_group_queries"] >= 15
        for qi, q in enumerate(qs):
            ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), 10)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[qi, :n], ws[:n]) and np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} {q}"
    bt.close()
    dev.close()

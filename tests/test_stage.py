"""The staged-tile route of BM25 top-k batches (csrc/sa_stage.hip): every distinct term of the batch is staged
in LDS once per tile of documents, the queries are answered from there -- candidates from the essential terms only, exact
scores in query-term order for the few documents that can reach a query's bound.  Results must equal the oracle's dense
score (reference postings.py:652-680 + bm25.pyx:11-25, summed as test/test_msmarco.py:353-354 does) + deterministic top-k bit
for bit, whatever the query shapes, and the route must actually have run (last_route() == 'staged')."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import roaringish as rz, synth
from searcharray_amd.device_index import DeviceIndex
from tests.helpers import set_opt, unset_opt

N_DOCS, VOCAB = 9000, 400


@pytest.fixture(scope="module")
def corpus():
    t, d, p, lens = synth.corpus_triples(N_DOCS, VOCAB, 14, seed=31)
    words, wt = rz.encode_sorted(t, d, p)
    return words, rz.term_offsets(wt, VOCAB), lens, O.OracleIndex.from_triples(t, d, p, N_DOCS, doc_lens=lens)


def check(api, corpus, queries, k, tile_docs=1024, doc_base=0, expect="staged", runs=2, force=True):
    """force: option stage = 1 (the kernel is under test, whatever the library's own rule would pick for the shape)"""
    words, off, lens, orc = corpus
    if force:
        set_opt("stage", 1)
    dev = DeviceIndex(words, off, lens, tile_docs=tile_docs, doc_base=doc_base, api=api)
    bt = dev.batch(np.asarray(queries), k=k)
    for _ in range(runs):
        bt.run()
    assert bt.last_route() == expect
    scores, docs = bt.fetch()
    for qi, q in enumerate(queries):
        dense = orc.score_terms_sum([int(x) for x in q if 0 <= int(x) < VOCAB])
        ws, wd = O.topk(dense, k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} {q} scores"
        assert np.array_equal(docs[qi, :n], wd[:n] + np.uint64(doc_base)), f"q{qi} {q} docs"
    bt.close()
    dev.close()
    return scores, docs


def band_queries(rng, n, T, heads):
    q = np.empty((n, T), dtype=np.int64)
    q[:, 0] = rng.choice(heads, n)
    for t in range(1, T):
        lo = [3, 20, 100, 250][min(t - 1, 3)]
        q[:, t] = rng.integers(lo, VOCAB, n)
    return q


@pytest.mark.parametrize("T", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("k", [3, 50])
def test_staged_equals_oracle(api, corpus, T, k):
    rng = np.random.default_rng(100 + T + k)
    queries = band_queries(rng, 40, T, heads=[0, 1, 2, 7, 350])
    check(api, corpus, queries, k)


@pytest.mark.parametrize("docs", [64, 128, 512, 1024])
def test_stage_tile_sizes_and_doc_base(api, corpus, docs):
    set_opt("stage_docs", docs)
    rng = np.random.default_rng(7 + docs)
    queries = band_queries(rng, 24, 4, heads=[0, 3])
    check(api, corpus, queries, 10, doc_base=50_000)


def test_staged_dense_terms_duplicates_unknowns(api, corpus):
    """all-frequent queries (every term essential: every posting a candidate), a term repeated inside a query, unknown
    terms, queries of unknown terms only"""
    queries = [[0, 1, 2, 3], [0, 0, 0, 5], [0, 2, 1, 1], [0, 390, 390, 9], [0, 4000, 17, 4001], [0, 4000, 4000, 4000],
               [5, 1, 0, 2], [5, 300, 301, 302], [4000, 0, 1, 2], [9, 8, 7, 6], [4000, 4001, 4002, 4003]]
    queries += [[0, 10 + i, 200 + i, 399 - i] for i in range(70)]
    check(api, corpus, queries, 7)
    check(api, corpus, queries, 1000)


def test_stage_overflow_splits_the_tile(api, corpus):
    """1024-doc stage tiles of 256 queries x 4 frequent terms hold more postings than the stage: the tile is taken in doc
    sub-ranges (halved until it fits)"""
    set_opt("stage_docs", 1024)
    rng = np.random.default_rng(5)
    queries = np.stack([rng.permutation(256) % 64, 64 + rng.permutation(256) % 120, rng.integers(0, VOCAB, 256), rng.integers(0, VOCAB, 256)], axis=1)
    check(api, corpus, queries, 10)


def test_route_rule_and_options(api, corpus):
    """the library's own rule (csrc/sa_stage.hip, sa_stage_plan): the staged route where few candidates per document are expected at
    small k; an explicit `sparse` option chooses between the two older routes; `stage` = 1 / 0 forces / forbids"""
    rng = np.random.default_rng(3)
    queries = band_queries(rng, 24, 4, heads=[0, 3])
    check(api, corpus, queries, 5, expect="staged", force=False)
    check(api, corpus, queries, 200, expect="exhaustive", force=False)
    check(api, corpus, queries[:3], 5, expect="exhaustive", force=False)     # (one to three queries: the per-query kernel, the route walks every tile whatever the set)
    set_opt("SA_SPARSE", "0")
    check(api, corpus, queries, 10, expect="exhaustive", force=False)
    set_opt("stage", 1)
    check(api, corpus, queries, 10, expect="staged", force=False)
    unset_opt("SA_SPARSE")
    set_opt("stage", 0)
    check(api, corpus, queries, 10, expect="exhaustive", force=False)
    check(api, corpus, queries[:3], 5, expect="staged")                      # (a set of three, forced: stage = 1)


def test_probed_terms_and_streaming_everything(api, corpus):
    """stage_probe = 0: every term of the batch is staged (no probe rows); default: terms that cannot be essential are probed"""
    rng = np.random.default_rng(11)
    queries = band_queries(rng, 48, 4, heads=[0, 1, 2, 5])
    a = check(api, corpus, queries, 10)
    set_opt("stage_probe", 0)
    b = check(api, corpus, queries, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_wide_query_sets_run_in_slices_of_256_rows(api, corpus):
    """more than 256 queries per set: the plan and the launches go slice by slice (256 device rows each, the last one partial), every
    query's top-k still equals the oracle's; a set with a slice the route does not take runs another route as a whole"""
    rng = np.random.default_rng(77)
    queries = band_queries(rng, 600, 4, heads=[0, 1, 2, 7, 350])
    check(api, corpus, queries, 10)
    check(api, corpus, queries[:257], 3, doc_base=1 << 20)
    check(api, corpus, band_queries(rng, 513, 3, heads=[0, 5]), 10)
    unset_opt("stage")
    check(api, corpus, queries, 10, expect="exhaustive", force=False)     # (the library's own rule takes slices only on shards of 8 M docs and more: measured)


def test_candidate_list_overflow_is_redone(api, corpus):
    """a candidate list of 64 keys per query (test hook cand_cap) runs over on the staged route like on the others: the cursor keeps
    counting, the merge flags the query, the run is redone without bounds at fetch -- results still equal the oracle"""
    set_opt("SA_CAND_CAP", "64")
    rng = np.random.default_rng(9)
    queries = band_queries(rng, 32, 4, heads=[0, 1])
    check(api, corpus, queries, 50)


@pytest.mark.parametrize("cw", [1, 2, 4])
def test_workgroups_co_walking_tile_ranges(api, corpus, cw):
    """groups of `stage_cw` workgroups of an XCD share a tile range and take every cw-th tile of it (csrc/sa_stage.hip, StageParams::cw): a
    term's first posting comes from the directory per tile instead of a running cursor.  stage_wgs = 8 gives the emulated 4-CU device 4
    workgroups per XCD, 64-doc tiles give every workgroup several tiles; whatever the order the tiles are taken in, the top-k equals the
    oracle's.  (A set with a term too rare for a directory row keeps private ranges: the walked cursor cannot skip tiles.)"""
    set_opt("stage_wgs", 8)
    set_opt("stage_cw", cw)
    set_opt("stage_docs", 64)
    rng = np.random.default_rng(40 + cw)
    orc = corpus[3]
    frequent = np.asarray([t for t in range(VOCAB) if orc.docfreq(t) >= 64])          # (every one of them has a directory row: df >= max(32, tiles / 8))
    assert len(frequent) >= 40
    queries = frequent[rng.integers(0, len(frequent), (40, 4))]
    check(api, corpus, queries, 10)
    check(api, corpus, queries[:24, :3], 3, doc_base=77)
    check(api, corpus, band_queries(rng, 24, 3, heads=[0, 350]), 3)                  # (with rare terms: private ranges)

"""Parity of the BM25 / tf / df / top-k path with the CPU oracle and the reference goldens.

Every test runs twice: backend "emu" (kernel sources compiled for the host, tests/hipemu --
checks kernel logic without a GPU) and backend "gpu" (the real gfx950 library through the C ABI;
marked gpu).  tf / df / top-k doc ids are compared bit-exact; BM25 scores are compared bit-exact
too (np.array_equal), which is stricter than the 1e-5 relative tolerance BASELINE.json allows."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import ops, synth
from searcharray_amd import roaringish as rz
from searcharray_amd.device_index import DeviceIndex, NO_DOC
from tests.helpers import golden_corpus, oracle_index, set_opt, unset_opt


def build_pair(name, tile_docs=1024, api=None):
    g, (t, d, p), lens, num_docs, vocab = golden_corpus(name)
    words, word_terms = rz.encode_sorted(t, d, p)
    term_off = rz.term_offsets(word_terms, vocab)
    dev = DeviceIndex(words, term_off, lens, tile_docs=tile_docs, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    return g, dev, orc, vocab


@pytest.fixture(scope="module")
def small(api):
    return build_pair("zipf_small", api=api)


def test_ops_mirrors_match_oracle(api):
    rng = np.random.default_rng(0)
    tf = rng.integers(0, 6, 5000).astype(np.float32)
    dl = rng.integers(1, 90, 5000).astype(np.float32)
    want = tf.copy()
    O.bm25_score(want, dl, 31.7, 2.345, 1.2, 0.75)
    got = tf.copy()
    ops.bm25_score(got, dl, 31.7, 2.345, 1.2, 0.75, api=api)
    assert np.array_equal(got, want)
    idx = np.sort(rng.choice(20000, 3000, replace=False)).astype(np.uint64)
    val = rng.random(3000).astype(np.float32)
    assert np.array_equal(ops.as_dense(idx, val, 20000, api=api), O.as_dense(idx, val, 20000))
    docs = np.sort(rng.integers(0, 900, 7000)).astype(np.uint64)
    arr = np.unique((docs << np.uint64(36)) | (rng.integers(0, 4, 7000).astype(np.uint64) << np.uint64(18))
                    | rng.integers(1, 2 ** 18, 7000).astype(np.uint64))
    k1, c1 = ops.popcount64_reduce(arr, 36, 0x3FFFF, api=api)
    k2, c2 = O.popcount64_reduce(arr, 36, 0x3FFFF)
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2)
    assert np.array_equal(ops.unique(arr, 36, api=api), O.unique(arr, 36))
    assert np.array_equal(ops.unique(arr >> np.uint64(36), 0, api=api), O.unique(arr >> np.uint64(36), 0))
    assert np.array_equal(ops.popcount64(arr, api=api), O.popcount64(arr))
    assert len(ops.unique(np.empty(0, np.uint64), 36, api=api)) == 0


def test_df_tf_bit_exact(small):
    g, dev, orc, vocab = small
    assert np.array_equal(dev.docfreqs(), g["df"])
    for t in g["tf_terms"]:
        assert np.array_equal(dev.termfreqs_dense(int(t)), orc.termfreqs(int(t))), f"tf t{t}"
        ids, tfs = dev.termfreqs_sparse(int(t))
        oi, ot = orc.termfreqs_sparse(int(t))
        assert np.array_equal(ids, oi) and np.array_equal(tfs, ot)
    assert dev.termfreqs_dense(vocab + 5).sum() == 0


def test_bm25_dense_bit_exact(small):
    g, dev, orc, vocab = small
    for t in g["score_terms"]:
        assert np.array_equal(dev.bm25_dense([int(t)]), g[f"score_{t}"]), f"score t{t}"
        assert np.array_equal(dev.bm25_dense([int(t)], k1=1.7, b=0.3), g[f"score_custom_{t}"])
    for q, want in zip(g["or_queries"], g["or_scores"]):
        assert np.array_equal(dev.bm25_dense([int(t) for t in q]), want)
    # unknown term contributes nothing
    assert np.array_equal(dev.bm25_dense([3, vocab + 9]), orc.score(3))


@pytest.mark.parametrize("k", [1, 10, 32, 40, 300])
def test_topk_batch_matches_oracle(small, k):
    g, dev, orc, vocab = small
    queries = g["or_queries"][:6]
    bt = dev.batch(queries, k=k)
    bt.run()
    scores, docs = bt.fetch()
    for qi, q in enumerate(queries):
        dense = orc.score_terms_sum([int(t) for t in q])
        ws, wd = O.topk(dense, k)
        nz = ws > 0
        n = int(nz.sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} scores"
        assert np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} docs"
        assert (scores[qi, n:] == 0).all() and (docs[qi, n:] == NO_DOC).all()
    ms, alg, post = bt.profile()
    want_post = sum(8 * int(orc.docfreq(int(t))) for q in queries for t in q)
    assert post == want_post and alg == want_post + 4 * orc.num_docs * len(queries)
    bt.close()


@pytest.mark.parametrize("n,cap", [(1500, ""), (7000, ""), (7000, "100")])
def test_topk_massive_ties_takes_fallback_path(api, n, cap, monkeypatch):
    """All docs identical -> every score ties -> the bound cuts nothing; with n = 7000 the survivors
    overflow the merge kernel's LDS list (in-place compaction + bisection over the survivors), and with
    a candidate list of 100 keys the list itself runs over: the batch is redone unpruned at fetch."""
    if cap:
        set_opt("SA_CAND_CAP", cap)
    t = np.repeat(np.arange(3), n).astype(np.uint32)
    d = np.tile(np.arange(n), 3).astype(np.uint64)
    p = np.repeat(np.arange(3), n).astype(np.uint64)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, 3), np.full(n, 3, np.float32), tile_docs=1024, api=api)
    for k in (5, 40, 700):
        bt = dev.batch(np.asarray([[0, 1, 2]]), k=k)
        bt.run()
        scores, docs = bt.fetch()
        assert np.array_equal(docs[0], np.arange(k, dtype=np.uint64))         # ties -> smallest doc ids
        assert (scores[0] == scores[0, 0]).all() and scores[0, 0] > 0
        bt.close()


@pytest.mark.parametrize("seg_words", ["1", "37", "700"])
def test_segmented_posting_derivation(api, seg_words, monkeypatch, on_emu):
    """Shards beyond 2^32 words derive their TF postings in segments of whole terms (< 2^31 words
    each); SA_SEG_WORDS forces tiny segments so the same code runs here -- down to one term per
    segment -- and must give the postings of the one-segment build."""
    if on_emu:                        # (one term per segment is thousands of emulated launches, 45 s: the GPU run does that;
        seg_words = {"1": "150", "37": "400"}.get(seg_words, seg_words)   #  here a few dozen segments of whole terms)
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_sparse")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    set_opt("SA_SEG_WORDS", seg_words)
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    assert np.array_equal(dev.docfreqs(), g["df"])
    assert dev.info().n_postings == int(np.sum(g["df"]))
    for row, want in zip(g["or_queries"][:6], g["or_scores"][:6]):
        assert np.array_equal(dev.bm25_dense([int(x) for x in row]), want)
    bt = dev.batch(np.asarray(g["or_queries"][:6]), k=10)
    bt.run()
    scores, docs = bt.fetch()
    for i, want in enumerate(g["or_scores"][:6]):
        order = np.lexsort((np.arange(num_docs), -want))[:10]
        keep = want[order] > 0
        assert np.array_equal(docs[i][keep], order[keep].astype(np.uint64))
    bt.close()


@pytest.mark.parametrize("tile_docs", [1024, 2048])
def test_sparse_corpus_and_doc_base(api, tile_docs):
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_sparse")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    dev = DeviceIndex(words, off, lens, tile_docs=tile_docs, api=api, doc_base=100000)
    assert np.array_equal(dev.docfreqs(), g["df"])
    q = g["or_queries"][:4]
    for row, want in zip(q, g["or_scores"][:4]):
        assert np.array_equal(dev.bm25_dense([int(x) for x in row]), want)
    bt = dev.batch(q, k=10)
    bt.run()
    scores, docs = bt.fetch()
    for qi, row in enumerate(q):
        ws, wd = O.topk(g["or_scores"][qi], 10)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]) and np.array_equal(docs[qi, :n], wd[:n] + 100000)
    bt.close()


def test_empty_and_degenerate_indexes(api):
    dev = DeviceIndex(np.empty(0, np.uint64), np.zeros(4, np.uint64), np.zeros(10, np.float32), tile_docs=1024, api=api)
    assert (dev.docfreqs() == 0).all()
    assert dev.bm25_dense([0, 1]).sum() == 0            # avg_doc_len == 0 -> zeros (similarity.py:31-32)
    assert dev.termfreqs_dense(1).sum() == 0
    bt = dev.batch(np.asarray([[0, 1]]), k=3)
    bt.run()
    s, d_ = bt.fetch()
    assert (s == 0).all() and (d_ == NO_DOC).all()


@pytest.mark.parametrize("tile_docs,k", [(1024, 10), (2048, 3), (1024, 32), (1024, 1000), (2048, 100), (4096, 10), (8192, 40)])
def test_topk_pruning_over_many_tiles(api, tile_docs, k):
    """Enough tiles that the global pruning slots fill up (> 32 waves per query): most waves are
    rejected by the bound, and the result must still be the exact top-k."""
    n_docs, vocab = 40000, 400
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 10, seed=11)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=tile_docs, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    queries = np.asarray([[0, 5, 50, 300], [1, 2, 3, 4], [7, 90, 200, 399], [399, 398, 397, 396]])
    bt = dev.batch(queries, k=k)
    for _ in range(2):                      # a second run must reset the slots / cursors
        bt.run()
        scores, docs = bt.fetch()
        for qi, q in enumerate(queries):
            ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), k)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} scores"
            assert np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} docs"
            assert (docs[qi, n:] == NO_DOC).all()
    bt.close()


def _long_doc_corpus():
    """Docs up to 700 tokens with term frequencies up to ~60: exercises postings outside the
    LDS saturation table (tf > 8, doc length >= 256)."""
    rng = np.random.default_rng(21)
    n_docs, vocab = 3000, 50
    lens = rng.integers(1, 700, n_docs)
    lens[::7] = rng.integers(1, 20, len(lens[::7]))
    terms = np.concatenate([rng.choice(vocab, L, p=np.r_[0.3, np.full(vocab - 1, 0.7 / (vocab - 1))]) for L in lens])
    t, d, p = synth.tokens_to_triples(lens.astype(np.int64), terms.astype(np.uint32))
    return t, d, p, lens.astype(np.float32), n_docs, vocab


def test_out_of_table_postings_and_unpacked_doc_lens(api):
    t, d, p, lens, n_docs, vocab = _long_doc_corpus()
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    for dl in (lens, lens + 0.5):                 # integer lengths ride in the postings; others are gathered
        orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=dl)
        dev = DeviceIndex(words, off, dl, tile_docs=1024, api=api)
        assert dev.info().dl_packed == (1 if dl is lens else 0)
        for q in ([0], [0, 1, 2, 3], [5, 0, 49]):
            assert np.array_equal(dev.bm25_dense(q), orc.score_terms_sum(q)), q
            assert np.array_equal(dev.bm25_dense(q, k1=0.9, b=0.4),
                                  np.sum([orc.score(x, k1=0.9, b=0.4) for x in q], axis=0))
        bt = dev.batch(np.asarray([[0, 1, 2, 3], [4, 0, 9, 30]]), k=10)
        bt.run()
        scores, docs = bt.fetch()
        for qi, q in enumerate([[0, 1, 2, 3], [4, 0, 9, 30]]):
            ws, wd = O.topk(orc.score_terms_sum(q), 10)
            assert np.array_equal(scores[qi], ws) and np.array_equal(docs[qi], wd)
        bt.close()


@pytest.mark.gpu
def test_rccl_exchange_single_rank():
    """The library-internal RCCL path (ncclCommInitRank + ncclAllGather on the index stream +
    regroup + merge) with a one-rank communicator: results must equal the no-communicator run."""
    import ctypes
    from searcharray_amd import _lib
    api = _lib.api()
    g, dev, orc, vocab = build_pair("zipf_small", api=api)
    queries = g["or_queries"][:8]
    plain = dev.batch(queries, k=10)
    plain.run()
    want = plain.fetch()
    buf = ctypes.create_string_buffer(128)
    api.call("sa_comm_unique_id", buf, 128)
    dev.comm_init(0, 1, buf.raw)
    try:
        bt = dev.batch(queries, k=10)
        for _ in range(3):
            bt.run()
        got = bt.fetch()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        bt.close()
    finally:
        dev.comm_destroy()
    plain.close()


@pytest.mark.parametrize("k", [10, 100])
def test_unpruned_block_selection_path(small, k, monkeypatch):
    """SA_PRUNED_TOPK=0 forces the block-level threshold selection (the overflow fallback of the
    pruned path): same exact results."""
    set_opt("SA_PRUNED_TOPK", "0")
    g, dev, orc, vocab = small
    queries = g["or_queries"][:4]
    bt = dev.batch(queries, k=k)
    bt.run()
    scores, docs = bt.fetch()
    for qi, q in enumerate(queries):
        ws, wd = O.topk(orc.score_terms_sum([int(t) for t in q]), k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]) and np.array_equal(docs[qi, :n], wd[:n])
    bt.close()


# ---------------------------------------------------------------------------------------------
# index build on the device (sa_index_create_from_tokens)
# ---------------------------------------------------------------------------------------------
def _tokens_of(t, d, p, n_docs):
    """(term, doc, pos) triples -> flat token stream in (doc, pos) order + doc offsets"""
    order = np.lexsort((p, d))
    lens = np.bincount(d.astype(np.int64), minlength=n_docs)
    ptr = np.zeros(n_docs + 1, dtype=np.uint64)
    np.cumsum(lens, out=ptr[1:])
    return t[order].astype(np.uint32), ptr


@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_device_index_build_is_byte_identical(api, name):
    """token stream -> sort by term + roaringish encode on the device == the host encoder (which is
    pinned to the reference's encoder on these corpora), and the derived statistics agree"""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus(name)
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    tokens, ptr = _tokens_of(t, d, p, num_docs)
    dev = DeviceIndex.from_tokens(tokens, ptr, vocab, doc_lens=lens, tile_docs=1024, api=api)
    got_words, got_off = dev.words()
    assert np.array_equal(got_off, off)
    assert np.array_equal(got_words, words)
    assert np.array_equal(dev.docfreqs(), g["df"])
    row = [int(x) for x in g["or_queries"][0]]
    assert np.array_equal(dev.bm25_dense(row), g["or_scores"][0])


def test_device_index_build_edge_cases(api):
    # empty docs, a term that never occurs, positions crossing the 18-bit block boundary
    lens = np.asarray([0, 40, 0, 3, 0], dtype=np.int64)
    ptr = np.zeros(6, dtype=np.uint64)
    np.cumsum(lens, out=ptr[1:])
    tokens = np.concatenate([np.tile([0, 2], 20), [2, 2, 4]]).astype(np.uint32)
    dev = DeviceIndex.from_tokens(tokens, ptr, 6, api=api, tile_docs=1024)
    words, off = dev.words()
    d = np.repeat(np.arange(5), lens).astype(np.uint64)
    p = np.concatenate([np.arange(n) for n in lens]).astype(np.uint64)
    order = np.lexsort((p, d, tokens))
    want_words, wt = rz.encode_sorted(tokens[order], d[order], p[order])
    assert np.array_equal(words, want_words) and np.array_equal(off, rz.term_offsets(wt, 6))
    assert list(dev.docfreqs()) == [1, 0, 2, 0, 1, 0]
    assert dev.termfreqs_dense(2)[1] == 20 and dev.termfreqs_dense(2)[3] == 2
    # no tokens at all
    dev0 = DeviceIndex.from_tokens(np.empty(0, np.uint32), np.zeros(4, np.uint64), 3, api=api, tile_docs=1024)
    assert dev0.words()[0].size == 0 and (dev0.docfreqs() == 0).all()
    with pytest.raises(Exception, match="term id"):
        DeviceIndex.from_tokens(np.asarray([7], np.uint32), np.asarray([0, 1], np.uint64), 3, api=api)


@pytest.mark.parametrize("k", [40, 300])
def test_large_k_slot_bound_path(small, k, monkeypatch):
    """k > 32 defaults to the histogram bound; SA_TOPK_HIST=0 keeps the slot bound (also what tiles of
    more than 4 waves and phrase batches use)"""
    set_opt("SA_TOPK_HIST", "0")
    test_topk_batch_matches_oracle(small, k)


# ---------------------------------------------------------------------------------------------
# dynamic pruning (sa_sparse.hip): only docs that can still reach the top-k are scored
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sparse,tf8_div", [("1", "128"), ("1", "0"), ("0", "128")])
@pytest.mark.parametrize("k", [5, 40])
def test_dynamic_pruning_is_exact(api, monkeypatch, sparse, tf8_div, k):
    """rare + frequent terms: queries with a rare term are answered by scoring candidates only (checked
    through the diagnostics counters), with and without dense tf rows in the index; all-frequent
    queries fall back to the tile scan; the top-k equals the exhaustive oracle bit for bit"""
    set_opt("SA_SPARSE", sparse)
    set_opt("SA_TF8_DIV", tf8_div)
    n_docs, vocab = 60000, 3000
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 12, seed=5)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    assert (dev.info().n_tf8_terms > 0) == (tf8_div != "0")
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    queries = np.asarray([[0, 40, 700, 2500], [2900, 1, 3, 1500], [5, 6, 7, 8], [2999, 2998, 0, 1], [0, 2000, 2000, 9],
                          [1, 0, 2, 3], [3100, 2, 1, 0], [2950, 2951, 2952, 2953]])
    bt = dev.batch(queries, k=k)
    bt.stats(True)
    for _ in range(2):
        bt.run()
        scores, docs = bt.fetch()
        for qi, q in enumerate(queries):
            ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), k)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} scores"
            assert np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} docs"
            assert (docs[qi, n:] == NO_DOC).all()
    cands, sparse_queries = bt.stats(False)
    if sparse == "1":
        assert 2 <= sparse_queries < len(queries), sparse_queries     # rare-term queries sparse, all-frequent ones scanned
        assert 0 < cands < 2 * len(queries) * n_docs // 8
    else:
        assert cands == 0
    bt.close()


def test_dense_calls_with_a_row_selection(api):
    """sa_index_select_rows: the next dense call returns only the selected rows (gathered on the device);
    a failed call must not leave the selection pending"""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    rows = np.asarray([5, 0, 1499, 700, 5, 3], dtype=np.uint64)
    full_tf, full_bm = dev.termfreqs_dense(3), dev.bm25_dense([3, 7, 1])
    assert np.array_equal(dev.termfreqs_dense(3, rows=rows), full_tf[rows.astype(np.int64)])
    assert np.array_equal(dev.bm25_dense([3, 7, 1], rows=rows), full_bm[rows.astype(np.int64)])
    assert np.array_equal(dev.phrase_freqs_dense([3, 7], rows=rows), dev.phrase_freqs_dense([3, 7])[rows.astype(np.int64)])
    assert np.array_equal(dev.phrase_freqs_dense([3, 7], slop=2, rows=rows), dev.phrase_freqs_dense([3, 7], slop=2)[rows.astype(np.int64)])
    assert np.array_equal(dev.bm25_phrase_dense([3, 7], rows=rows), dev.bm25_phrase_dense([3, 7])[rows.astype(np.int64)])
    assert np.array_equal(dev.termfreqs_dense(3, min_posn=0, max_posn=17, rows=rows),
                          dev.termfreqs_dense(3, min_posn=0, max_posn=17)[rows.astype(np.int64)])
    assert dev.termfreqs_dense(3, rows=np.empty(0, np.uint64)).size == 0
    assert dev.bm25_dense([3], rows=np.asarray([num_docs + 5], np.uint64))[0] == 0          # beyond the index: 0
    with pytest.raises(Exception):
        dev.phrase_freqs_dense(list(range(40)), slop=1, rows=rows)                          # a slop phrase of more than 32 terms is refused
    assert np.array_equal(dev.termfreqs_dense(3), full_tf)                                  # selection was cleared


@pytest.mark.parametrize("tile_docs", [1024, 2048, 4096, 8192])
def test_dense_bm25_one_launch_equals_the_tf_route_and_the_oracle(api, tile_docs):
    """BASELINE config 2 itself (reference postings.py:652-680 + bm25.pyx:11-25, summed over the terms as test/test_msmarco.py:353-354):
    sa_index_bm25_dense as ONE launch over the impact stream, written straight into the destination (host copy, row selection, a
    device vector times a boost) == the rounds 1-5 route (TF postings -> scratch -> copy; option dense_direct = 0) == the oracle, bit
    for bit; 1 .. 8 terms, a term twice, unknown terms, (k1, b) other than the stream's (the TF route serves those), a shard
    whose last tile is partial; tile sizes the launch has no instance for (8192) take the TF route"""
    from searcharray_amd.device_index import DeviceVec
    from searcharray_amd._lib import p_u32, p_f32
    t, d, p, lens = synth.corpus_triples(5003, 300, 12, seed=9)
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, 300)
    orc = O.OracleIndex.from_triples(t, d, p, 5003, doc_lens=lens)
    dev = DeviceIndex(words, off, lens, tile_docs=tile_docs, api=api)
    old = DeviceIndex(words, off, lens, tile_docs=tile_docs, api=api, opts={"dense_direct": 0})
    rng = np.random.default_rng(tile_docs)
    queries = [[0], [299], [5, 5], [0, 1, 2], [7, 4000, 3], [4000], [250, 0, 17, 3, 120, 9, 1, 77]] + [list(rng.integers(0, 300, int(n))) for n in rng.integers(1, 9, 12)]
    rows = np.asarray([5, 0, 5002, 700, 5, 3, 4095, 4096], dtype=np.uint64)
    vec = DeviceVec(api, 5003, False)
    for q in queries:
        q = [int(x) for x in q]
        known = [x for x in q if x < 300]
        want = orc.score_terms_sum(known) if known else np.zeros(5003, dtype=np.float32)
        got = dev.bm25_dense(q)
        assert np.array_equal(got, want), q
        assert np.array_equal(old.bm25_dense(q), want), q
        assert np.array_equal(dev.bm25_dense(q, rows=rows), want[rows.astype(np.int64)]), q
        tarr = np.asarray([x if x < 300 else 0xFFFFFFFF for x in q], dtype=np.uint32)
        idf = dev.idfs(q)
        for boost in (None, 2.5):
            dev.into_vec(vec, boost, "sa_index_bm25_dense", p_u32(tarr), p_f32(idf), len(tarr), np.float32(1.2), np.float32(0.75))
            assert np.array_equal(vec.fetch(), want if boost is None else want * np.float32(boost)), (q, boost)
        if known:
            assert np.array_equal(dev.bm25_dense(q, k1=0.9, b=0.4), orc.score_terms_sum(known, k1=0.9, b=0.4)), q
    vec.close()
    dev.close()
    old.close()


# ---------------------------------------------------------------------------------------------
# impact stream (sa_k_make_impacts): the per-posting factor evaluated once per (k1, b)
# ---------------------------------------------------------------------------------------------
def _check_batch(bt, orc, queries, k, k1=1.2, b=0.75):
    bt.run()
    scores, docs = bt.fetch()
    for qi, q in enumerate(queries):
        ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q], k1=k1, b=b), k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"q{qi} scores"
        assert np.array_equal(docs[qi, :n], wd[:n]), f"q{qi} docs"
        assert (docs[qi, n:] == NO_DOC).all()
    return scores, docs


@pytest.mark.parametrize("integer_lens", [True, False])
def test_impact_stream_exhaustive_scan(api, monkeypatch, integer_lens):
    """SA_SPARSE=0: every posting is scored by the tile kernel.  With the impact stream (default) and
    with the TF postings (SA_IMPACT=0) the top-k must equal the oracle bit for bit: long docs and high
    term frequencies (outside the saturation table), odd and even posting counts (padding), unknown and
    repeated terms, fractional doc lengths (doc_lens gathered at build time), two batches with different
    (k1, b) alive at once, and a batch that outlives the index's cached stream."""
    set_opt("SA_SPARSE", "0")
    n_docs, vocab = 9000, 400
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 150, seed=11)        # mean length 150: dl >= 128, tf > 8
    if not integer_lens:
        lens = (lens + np.float32(0.5)).astype(np.float32)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    assert bool(dev.info().dl_packed) == integer_lens
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    queries = np.asarray([[0, 1, 2, 3], [399, 398, 0, 200], [5, 5, 450, 7], [100, 101, 102, 103], [17, 390, 391, 2]])
    a = dev.batch(queries, k=20)
    c = dev.batch(queries, k=20, k1=1.7, b=0.3)                               # replaces the index's cached stream
    s_a, d_a = _check_batch(a, orc, queries, 20)
    _check_batch(c, orc, queries, 20, k1=1.7, b=0.3)
    s_a2, d_a2 = _check_batch(a, orc, queries, 20)                            # `a` still owns its stream
    assert np.array_equal(s_a, s_a2) and np.array_equal(d_a, d_a2)
    set_opt("SA_IMPACT", "0")                                      # same batch, TF-posting route
    s_b, d_b = _check_batch(a, orc, queries, 20)
    assert np.array_equal(s_a, s_b) and np.array_equal(d_a, d_b)
    e = dev.batch(queries, k=20, k1=0.9, b=0.0)                               # built without a stream at all
    _check_batch(e, orc, queries, 20, k1=0.9, b=0.0)
    for bt in (a, c, e):
        bt.close()
    # nine terms per query: three groups of term phases (4 + 4 + 1), long slices in every position of a group,
    # rare / unknown / repeated terms in between
    unset_opt("SA_IMPACT")
    wide = np.asarray([[0, 399, 1, 398, 2, 397, 3, 396, 4], [399, 398, 397, 396, 395, 0, 1, 2, 3],
                       [7, 7, 450, 8, 300, 0, 451, 9, 7], [390, 391, 392, 393, 394, 395, 396, 397, 398]])
    for env in ("1", "0"):
        set_opt("SA_IMPACT", env)
        w = dev.batch(wide, k=15)
        _check_batch(w, orc, wide, 15)
        w.close()


def test_dynamic_pruning_default_policy(api, monkeypatch):
    """Option `sparse` unset (round 5's measured rule, scripts/route_rule.py; round 6: the staged-tile route where the older rule scored
    exhaustively): a batch that can take the impact stream with histogram bounds takes the staged-tile route; one that cannot (impact = 0) is pruned while the shard holds at least 8192 docs
    per requested result.  Either way the top-k equals the oracle, and last_route() says which it was."""
    unset_opt("SA_SPARSE")
    n_docs, vocab = 60000, 3000
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 12, seed=5)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    queries = np.asarray([[0, 40, 700, 2500], [2900, 1, 3, 1500], [2999, 2998, 0, 1]] + [[i, 50 + i, 900 + i, 2000 + i] for i in range(6)])
    for impact, k, pruned, route in ((1, 1, False, "staged"), (0, 1, True, "pruned"), (0, 8, False, "exhaustive")):   # 60000 / 8192 = 7.3 results
        bt = dev.batch(queries, k=k, opts={"impact": impact})
        bt.stats(True)
        _check_batch(bt, orc, queries, k)
        cands, sparse_queries = bt.stats(False)
        assert bt.last_route() == route
        assert (sparse_queries > 0) == pruned and (cands > 0) == pruned
        bt.close()


def test_threaded_batches_share_impact_streams(api, monkeypatch):
    """Batches with different (k1, b) created, run and closed from several threads on one index (the C ABI
    serialises calls per index handle; an impact stream lives as long as a batch uses it, the index caches only
    the most recent one): every result equals the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    set_opt("SA_SPARSE", "0")
    n_docs, vocab = 6000, 300
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 40, seed=3)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    queries = np.asarray([[0, 1, 250, 299], [5, 120, 7, 2], [299, 298, 297, 0]])
    params = [(1.2, 0.75), (1.7, 0.3), (0.9, 0.0), (2.0, 1.0)]
    want = {kb: [O.topk(orc.score_terms_sum([int(x) for x in q], k1=kb[0], b=kb[1]), 10) for q in queries] for kb in params}

    def work(i):
        kb = params[i % len(params)]
        bt = dev.batch(queries, k=10, k1=kb[0], b=kb[1])
        try:
            for _ in range(2):
                bt.run()
                scores, docs = bt.fetch()
                for qi, (ws, wd) in enumerate(want[kb]):
                    n = int((ws > 0).sum())
                    if not (np.array_equal(scores[qi, :n], ws[:n]) and np.array_equal(docs[qi, :n], wd[:n])):
                        return False
            return True
        finally:
            bt.close()

    with ThreadPoolExecutor(4) as ex:
        assert all(ex.map(work, range(16)))

"""Exact-phrase match counts: device paths (general bigram chain and fused kernel) against the
CPU oracle and the reference goldens -- bit-exact counts.  Runs on the host emulator ("emu",
CPU suite) and on the real gfx950 library ("gpu")."""
import os

import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import synth
from searcharray_amd import roaringish as rz
from searcharray_amd.device_index import DeviceIndex, NO_DOC
from tests.helpers import golden_corpus, dense_from_sparse, set_opt, unset_opt
from tests.test_oracle_golden import PHRASE_SCENARIOS, _index_strings


def _device_from_strings(docs, api):
    vocab = {}
    t, d, p = [], [], []
    for di, doc in enumerate(docs):
        for pi, tok in enumerate(doc.split()):
            t.append(vocab.setdefault(tok, len(vocab))); d.append(di); p.append(pi)
    t, d, p = np.asarray(t, np.int64), np.asarray(d, np.int64), np.asarray(p, np.int64)
    order = np.argsort(t, kind="stable")
    lens = np.asarray([len(doc.split()) for doc in docs], np.float32)
    words, wt = rz.encode_sorted(t[order], d[order], p[order])
    dev = DeviceIndex(words, rz.term_offsets(wt, len(vocab)), lens, tile_docs=1024, api=api)
    return vocab, dev


@pytest.fixture(params=["general", "auto"])
def phrase_mode(request):
    if request.param == "general":
        set_opt("phrase_mode", "general")
    else:
        unset_opt("phrase_mode")
    yield request.param


@pytest.mark.parametrize("docs,phrase,expected", PHRASE_SCENARIOS[:23:2])
def test_reference_phrase_scenarios(api, phrase_mode, docs, phrase, expected):
    """known answers from the reference's test/test_phrase_matches.py:17-194"""
    vocab, dev = _device_from_strings(docs, api)
    got = dev.phrase_freqs_dense([vocab[t] for t in phrase.split()])
    assert np.array_equal(got, np.asarray(expected, np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["general", "auto"])
def test_all_reference_phrase_scenarios_and_offsets_on_the_device(mode, monkeypatch):
    """The reference runs every scenario of test/test_phrase_matches.py:17-194 and all 100 word-boundary offsets
    (:249-379); so does the device (the host-emulated runs above take every other scenario and six offsets to keep
    the CPU suite short)."""
    from searcharray_amd import _lib
    api = _lib.api()
    if mode == "general":
        set_opt("SA_PHRASE_MODE", "general")
    else:
        unset_opt("SA_PHRASE_MODE")
    for docs, phrase, expected in PHRASE_SCENARIOS:
        vocab, dev = _device_from_strings(docs, api)
        got = dev.phrase_freqs_dense([vocab[t] for t in phrase.split()])
        assert np.array_equal(got, np.asarray(expected, np.float32)), (docs, phrase)
        dev.close()
    for phrase in ["foo bar baz", "foo foo foo", "foo bar bar baz buz foo bar", "foo foo"]:
        for offset in range(100):
            vocab, dev = _device_from_strings([" ".join(["dummy"] * offset) + " " + phrase, "not match"], api)
            got = dev.phrase_freqs_dense([vocab[t] for t in phrase.split()])
            assert np.array_equal(got, [1, 0]), (phrase, offset)
            dev.close()


@pytest.mark.parametrize("phrase", ["foo bar baz", "foo foo foo", "foo bar bar baz buz foo bar", "foo foo"])
@pytest.mark.parametrize("offset", [0, 16, 17, 18, 35, 53])
def test_offsets_cross_word_boundary(api, phrase_mode, phrase, offset):
    vocab, dev = _device_from_strings([" ".join(["dummy"] * offset) + " " + phrase, "not match"], api)
    got = dev.phrase_freqs_dense([vocab[t] for t in phrase.split()])
    assert np.array_equal(got, [1, 0])


@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_corpus_phrases_match_reference(api, phrase_mode, name):
    """reference outputs (counts and BM25 scores) on the seeded corpora, incl. same-term phrases,
    2..7-term phrases and the T>=5 middle-out plan"""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus(name)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    for i in range(int(g["n_phrases"])):
        terms = [int(x) for x in g[f"phr_{i}_terms"]]
        want = dense_from_sparse(g[f"phr_{i}_idx"], g[f"phr_{i}_val"], num_docs)
        got = dev.phrase_freqs_dense(terms)
        assert np.array_equal(got, want), f"phrase {terms}: {np.flatnonzero(got != want)[:5]}"
    for i in range(0, int(g["n_phrases"]), 5):
        terms = [int(x) for x in g[f"phr_{i}_terms"]]
        wants = dense_from_sparse(g[f"phr_{i}_sidx"], g[f"phr_{i}_sval"], num_docs)
        assert np.array_equal(dev.bm25_phrase_dense(terms), wants), f"phrase score {terms}"


def test_fused_equals_general_on_random_distinct_phrases(api):
    n_docs, vocab = 3000, 60
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 45, seed=3)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    rng = np.random.default_rng(5)
    for length in (2, 3, 4, 5, 6, 8):
        for _ in range(4):
            terms = [int(x) for x in rng.choice(12, length, replace=False)]
            want = orc.phrase_freqs(terms)
            for mode in ("general", "fused"):
                set_opt("phrase_mode", mode)
                got = dev.phrase_freqs_dense(terms)
                assert np.array_equal(got, want), f"{mode} {terms}"
    unset_opt("phrase_mode")


@pytest.mark.parametrize("seed", range(4))
def test_repeated_term_phrases_chain_per_document(api, seed, monkeypatch, capfd):
    """Phrases with repeated terms run the bigram chain PER DOCUMENT in one launch (sa_k_phrase_docs: a thread per document
    of the rarest term, the general chain's steps on the document's few words, found through the doc directory or by
    search) instead of ~20 launches per bigram: counts equal to the oracle's (which is pinned to the reference, same-term rule and all) and to the general
    chain's, for every plan (left to right, right to left, middle out) and repeat pattern."""
    set_opt("trace", "1")
    if seed == 3:
        set_opt("SA_DOCDIR_DIV", "0")         # no doc directory: the documents' words are found by search
    rng = np.random.default_rng(40 + seed)
    n_docs, vocab = int(rng.integers(400, 2500)), 5
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, int(rng.integers(6, 40)), seed=70 + seed)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    phrases = [[0, 0], [1, 1], [0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 0], [0, 0, 0, 0], [2, 1, 1, 0], [0, 1, 1, 0, 2], [3, 3, 0, 1, 1],
               [4, 0, 0, 4], [1, 2, 3, 0, 0, 4, 1], [0, 1, 2, 3, 4, 0, 1, 2], [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1], [2, 2, 1, 0, 3, 4, 4, 1, 0, 2, 3, 1, 0, 0, 2, 4]]
    phrases += [[int(x) for x in rng.integers(0, vocab, int(rng.integers(2, 9)))] for _ in range(8)]
    taken = 0
    for ph in phrases:
        if len(set(ph)) == len(ph):
            continue
        want = orc.phrase_freqs(ph)
        capfd.readouterr()
        got = dev.phrase_freqs_dense(ph)
        taken += "chain per document taken" in capfd.readouterr().err
        assert np.array_equal(got, want), f"seed {seed} phrase {ph}: {np.flatnonzero(got != want)[:5]}"
        set_opt("SA_PHRASE_DOCS", "0")
        other = dev.phrase_freqs_dense(ph)
        unset_opt("SA_PHRASE_DOCS")
        assert np.array_equal(other, want), f"general chain: seed {seed} phrase {ph}"
    dev.close()
    assert taken >= 10, taken


def test_long_distinct_phrases_chain_per_document(api, monkeypatch, capfd):
    """Pairwise-distinct phrases of 19-32 terms (more than the fused kernel's window) take the chain per document too;
    every plan: the rarest term at the front, at the end, in the middle."""
    set_opt("trace", "1")
    rng = np.random.default_rng(8)
    toks = [f"w{i}" for i in range(34)]
    docs = []
    for i in range(300):
        a, b = sorted(int(x) for x in rng.integers(0, 34, 2))
        body = toks[a:b + 1] if i % 3 else toks                   # runs of the sequence, whole or partial
        noise = [toks[int(x)] for x in rng.integers(0, 34, int(rng.integers(0, 6)))]
        docs.append(" ".join(noise + body + noise))
    docs += ["w5 " * 3, "w17 w17", "w30"] * 20                     # makes some terms much more frequent than others
    vocab_o, dev = _device_from_strings(docs, api)
    vocab2, orc = _index_strings(docs)
    assert vocab2 == vocab_o
    taken = 0
    for a, b in ((0, 19), (3, 27), (0, 31), (10, 31), (2, 33), (14, 33)):
        ids = [vocab_o[t] for t in toks[a:b + 1]]
        capfd.readouterr()
        got = dev.phrase_freqs_dense(ids)
        taken += "chain per document taken" in capfd.readouterr().err
        want = orc.phrase_freqs(ids)
        assert np.array_equal(got, want), (a, b, np.flatnonzero(got != want)[:5])
        assert want.sum() > 0
    dev.close()
    assert taken == 6, taken


def test_words_in_a_documents_last_block_keep_the_general_routes(api, monkeypatch, capfd):
    """A word in a document's LAST 18-position block (positions >= 18 * (2^18 - 1)) has `header + 1` = the next document's
    block 0, so neither the bigram chain nor the slop candidate sets are local to a document then.  The index records
    whether any such word exists; with one, slop phrases of terms without a directory row take the general route, the
    phrase chain per document steps aside when it meets such a word -- and the answers stay the oracle's."""
    set_opt("trace", "1")
    set_opt("trace", "1")
    set_opt("SA_DOCDIR_DIV", "0")                 # no directory rows: only the index-wide flag can vouch for locality
    rng = np.random.default_rng(21)
    n_docs, vocab = 120, 4
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 12, seed=5)
    t = np.concatenate([t, [0, 1, 2]]); d = np.concatenate([d, [7, 8, 8]]); p = np.concatenate([p, [200, 0, 1]])
    order = np.lexsort((p, d, t))
    t, d, p = t[order], d[order], p[order]
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    # (the encoder takes positions < 2^18 only, like the reference's: the word is put into the last block by hand -- doc 7's
    #  last word of term 0, position bits 16 and 17)
    w0 = words[off[0]:off[1]]
    i = int(off[0]) + int(np.flatnonzero((w0 >> np.uint64(36)) == 7)[-1])
    words = words.copy()
    words[i] = (np.uint64(7) << np.uint64(36)) | (np.uint64(0x3FFFF) << np.uint64(18)) | np.uint64(0x30000)
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    orc = O.OracleIndex(words, np.arange(vocab), off, lens, n_docs)
    for ph, slop in (([0, 1], 2), ([1, 0, 2], 3), ([0, 0], 0), ([0, 0, 1], 0), ([1, 1, 2, 0], 0), ([0, 1], 0), ([0, 1, 2], 0)):
        capfd.readouterr()
        got = dev.phrase_freqs_dense(ph, slop=slop)
        err = capfd.readouterr().err
        assert np.array_equal(got, orc.phrase_freqs(ph, slop=slop)), (ph, slop)
        if slop:
            assert "slop route: general" in err, err
        elif len(set(ph)) < len(ph) and 0 in ph:
            assert "abandoned" in err, err
    dev.close()
    del rng


def test_chain_per_document_gives_way_when_its_checks_fail(api, monkeypatch, capfd):
    """The per-document chain predicts the reference's same-term test (`np.all(lhs_int == rhs_int)`, global over a step's
    matched pairs) and holds six words per list; it must step aside -- and the general chain give the reference's
    answer -- (a) when a step it calls "different" has matched pairs that are ALL equal: `a b b` over documents in which
    every `b` follows an `a` (the continuation of `a b` then equals b's words), (b) when a document has more than six
    words of a term."""
    set_opt("trace", "1")
    # (a) 200 documents "a b a b ... " (+ a few with other terms so that every term is frequent)
    docs = ["a b " * int(1 + i % 7) for i in range(200)]
    vocab_o, dev = _device_from_strings(docs, api)
    vocab2, orc = _index_strings(docs)
    assert vocab2 == vocab_o
    for ph in (["a", "b", "b"], ["a", "a", "b"], ["b", "a", "b", "b"]):
        ids = [vocab_o[x] for x in ph]
        capfd.readouterr()
        got = dev.phrase_freqs_dense(ids)
        err = capfd.readouterr().err
        assert np.array_equal(got, orc.phrase_freqs(ids)), ph
        if ph == ["a", "b", "b"]:
            assert "abandoned" in err, err
    dev.close()
    # (b) long documents: many words per term
    t, d, p, lens = synth.corpus_triples(60, 3, 500, seed=9)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, 3), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, 60, doc_lens=lens)
    for ph in ([0, 0], [1, 0, 0], [2, 2, 1, 2]):
        capfd.readouterr()
        got = dev.phrase_freqs_dense(ph)
        assert "abandoned" in capfd.readouterr().err
        assert np.array_equal(got, orc.phrase_freqs(ph)), ph
    dev.close()


def test_phrase_errors_and_unknown_terms(api):
    vocab, dev = _device_from_strings(["foo bar", "bar foo"], api)
    with pytest.raises(ValueError):
        dev.phrase_freqs_dense([0])
    assert dev.phrase_freqs_dense([0, 99]).sum() == 0
    assert dev.bm25_phrase_dense([0, 99]).sum() == 0


@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_slop_counts_match_reference(api, name):
    """slop > 0: candidate selection + per-document span state machine on the device vs the outputs
    of the reference itself (tests/golden) -- bit-exact counts, and BM25 scores within 1e-5."""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus(name)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    for i in range(int(g["n_slop"])):
        terms = [int(x) for x in g[f"slop_{i}_terms"]]
        slop = int(g[f"slop_{i}_slop"])
        want = dense_from_sparse(g[f"slop_{i}_idx"], g[f"slop_{i}_val"], num_docs)
        got = dev.phrase_freqs_dense(terms, slop=slop)
        assert np.array_equal(got, want), f"slop {terms} {slop}: {np.flatnonzero(got != want)[:5]}"
    terms, slop = [3, 7], 2
    assert np.allclose(dev.bm25_phrase_dense(terms, slop=slop), orc.score(terms, slop=slop), rtol=1e-5, atol=0)


def test_slop_more_doc_groups_than_resident_threads(api, monkeypatch):
    """the state-machine grid is resident (threads stride over the document groups): force a grid
    of 64 threads over ~1500 groups"""
    set_opt("SA_SPAN_THREADS", "64")
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    for i in range(0, int(g["n_slop"]), 3):
        terms = [int(x) for x in g[f"slop_{i}_terms"]]
        slop = int(g[f"slop_{i}_slop"])
        want = dense_from_sparse(g[f"slop_{i}_idx"], g[f"slop_{i}_val"], num_docs)
        assert np.array_equal(dev.phrase_freqs_dense(terms, slop=slop), want), f"slop {terms} {slop}"


@pytest.mark.parametrize("fast,docdir,sort", [("1", "1", "1"), ("0", "1", "1"), ("1", "0", "1"), ("1", "1", "0"), ("1", "1", "-1")])
def test_slop_table_placements_and_probe_routes_agree(api, monkeypatch, fast, docdir, sort):
    """the fast pass (span tables in LDS, abandoned documents redone with full tables) vs the full-table pass alone
    (SA_SPAN_FAST=0), and header probes through the doc directory vs binary searches (SA_SPAN_DOCDIR=0): docs
    with few positions and docs whose table outgrows the LDS column (> 12 spans), against the oracle (a table
    beyond the reference's 512 spans is undefined behaviour there -- see the fuzz test); document groups in work
    order vs index order (SA_SPAN_SORT=0)"""
    set_opt("SA_SPAN_FAST", fast)
    if sort != "-1":                                       # ("1" forces the work order on this small corpus, unset leaves it to the size rule)
        set_opt("SA_SPAN_SORT", sort)
    set_opt("SA_SPAN_DOCDIR", docdir)
    set_opt("SA_DOCDIR_DIV", "1000000")          # a doc directory for every term of >= 64 words
    rng = np.random.default_rng(3)
    docs = []
    for i in range(400):
        n = int(rng.integers(2, 12)) if i % 7 else int(rng.integers(60, 140))
        docs.append(" ".join(rng.choice(["a", "b", "c", "x", "y"], n, p=[0.3, 0.3, 0.1, 0.15, 0.15])))
    vocab, dev = _device_from_strings(docs, api)
    vocab_o, orc = _index_strings(docs)
    for q, slop in ((["a", "b"], 2), (["b", "a"], 1), (["a", "b", "c"], 3), (["a", "a"], 2)):
        got = dev.phrase_freqs_dense([vocab[x] for x in q], slop=slop)
        want = orc.phrase_freqs([vocab_o[x] for x in q], slop=slop)
        assert np.array_equal(got, want), (q, slop, np.flatnonzero(got != want)[:5])


def test_slop_scenarios_from_reference_tests(api):
    """match / no-match booleans of reference test/test_slop_matches.py:7-88"""
    docs = ["foo bar baz", "foo x bar", "foo x y bar", "bar foo", "foo foo bar", "nothing here"]
    vocab, dev = _device_from_strings(docs, api)
    vocab_o, orc = _index_strings(docs)
    q = [vocab["foo"], vocab["bar"]]
    for slop in (1, 2, 3):
        got = dev.phrase_freqs_dense(q, slop=slop)
        want = orc.phrase_freqs([vocab_o["foo"], vocab_o["bar"]], slop=slop)
        assert np.array_equal(got, want), slop
        assert got[0] > 0 and got[5] == 0


@pytest.mark.parametrize("seed", range(6))
def test_slop_random_differential(api, seed, monkeypatch):
    """random corpora / queries: device span search == oracle (which is pinned to the reference); document groups in
    work order (forced: the size rule would leave these small corpora in index order) on the odd seeds"""
    from oracle import spans as S
    set_opt("SA_SPAN_SORT", str(seed % 2))
    rng = np.random.default_rng(100 + seed)
    n_docs, vocab = int(rng.integers(200, 900)), int(rng.integers(8, 40))
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, int(rng.integers(10, 70)), seed=seed)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    for _ in range(6):
        T = int(rng.integers(2, 5))
        terms = [int(x) for x in rng.integers(0, vocab, T)]
        slop = int(rng.integers(1, 6))
        enc = [orc.enc(x) if orc.has_term(x) else np.empty(0, np.uint64) for x in terms]
        if any(len(e) == 0 for e in enc):
            continue
        ids, counts, overflow = S.span_search(enc, slop, return_overflow=True)
        want = np.zeros(n_docs, dtype=np.float32)
        want[ids.astype(np.int64)] = counts
        got = dev.phrase_freqs_dense(terms, slop=slop)
        assert np.array_equal(got, want), f"seed {seed} terms {terms} slop {slop}: {np.flatnonzero(got != want)[:5]}"


@pytest.mark.parametrize("seed", range(4))
def test_slop_doc_parallel_route(api, seed, monkeypatch, capfd):
    """The doc-parallel slop route (sa_spans.hip: frequent terms with directory rows -- sort blocks write per-document
    position records in work order, the machine takes them a lane each) against the oracle AND the general route, on
    corpora that reach each of its parts: 64-, 32- and 16-lane chunks, documents with more positions than a lane takes
    (heavy: a wave each), documents with more than four words of a term (the sort pass's slow path), lane tables that
    outgrow their column (redone by the lane's own wave), one and several sort blocks.  Two-term phrases take it whether or not header 0 is in L; three and four terms only when it is
    not -- doc 0 is kept free of the frequent terms on the even seeds so that both cases occur."""
    from oracle import spans as S
    set_opt("trace", "1")
    rng = np.random.default_rng(700 + seed)
    # (one / several blocks; seed 3: a Zipf vocabulary of 60 -- rare terms without a directory row, found by search, and
    #  blocks over the rarest term's documents instead of over all documents)
    n_docs, vocab = (int(rng.integers(300, 600)) if seed < 2 else int(rng.integers(4500, 7000))), (6 if seed < 3 else 60)
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, int(rng.integers(8, 30)), seed=60 + seed)
    # a few long documents: many positions, many words per term
    extra_t, extra_d, extra_p = [], [], []
    for doc in rng.choice(np.arange(1, n_docs), 12, replace=False):
        base = int(lens[doc])
        n = int(rng.integers(40, 260))
        extra_t.append(rng.integers(0, vocab, n)); extra_d.append(np.full(n, doc)); extra_p.append(base + np.arange(n))
        lens[doc] += n
    t = np.concatenate([t] + extra_t); d = np.concatenate([d] + extra_d); p = np.concatenate([p] + extra_p)
    if seed % 2 == 0:                                    # no frequent term at the very start of doc 0: header 0 not in L
        keep = ~((d == 0) & (p < 40))
        t, d, p = t[keep], d[keep], p[keep]
    order = np.lexsort((p, d, t))
    t, d, p = t[order], d[order], p[order]
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    routes = {"doc-parallel": 0, "general": 0, "doc route: over the rarest": 0, "doc route: over all": 0}
    for T in (2, 2, 3, 3, 4) if seed < 3 else (2, 2, 2, 3, 3, 3, 4, 4):
        terms = [int(x) for x in rng.integers(0, vocab, T)]
        if seed == 3:
            terms[0] = int(rng.integers(0, 4))           # a frequent term beside the rare ones: some documents do match
        slop = int(rng.integers(1, 7))
        enc = [orc.enc(x) if orc.has_term(x) else np.empty(0, np.uint64) for x in terms]
        if any(len(e) == 0 for e in enc):
            continue
        ids, counts, overflow = S.span_search(enc, slop, return_overflow=True)
        want = np.zeros(n_docs, dtype=np.float32)
        want[ids.astype(np.int64)] = counts
        capfd.readouterr()
        got = dev.phrase_freqs_dense(terms, slop=slop)
        err = capfd.readouterr().err
        for r in routes:
            routes[r] += (f"slop route: {r}" in err) or (f"slop {r}" in err)
        assert np.array_equal(got, want), f"seed {seed} terms {terms} slop {slop}: {np.flatnonzero(got != want)[:5]}"
        set_opt("SA_SPAN_DOC", "0")
        other = dev.phrase_freqs_dense(terms, slop=slop)
        unset_opt("SA_SPAN_DOC")
        assert np.array_equal(other, want), f"general route: seed {seed} terms {terms} slop {slop}"
    assert routes["doc-parallel"] >= 2, routes
    if seed % 2 == 0:
        assert routes["doc-parallel"] == 5, routes
    if seed == 3:
        assert routes["doc route: over the rarest"] >= 2, routes
    else:
        assert routes["doc route: over all"] >= 2, routes
    dev.close()


@pytest.mark.parametrize("seed", range(3))
def test_slop_five_to_eight_terms(api, seed, monkeypatch, on_emu):
    """phrases of more terms than the span kernels are specialised for (flags: 2-4 terms; the fast pass requests the
    first four terms' loads together): the generic paths, with and without the doc directory, vs the oracle"""
    from oracle import spans as S
    set_opt("SA_DOCDIR_DIV", "1000000" if seed % 2 else "0")
    set_opt("SA_SPAN_SORT", "1" if seed != 1 else "0")     # (work order forced / off)
    rng = np.random.default_rng(300 + seed)
    n_docs, vocab = int(rng.integers(300, 700)), int(rng.integers(6, 12))
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, int(rng.integers(20, 50)), seed=40 + seed)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    checked = 0
    for T in ((5, 8, 6) if on_emu else (5, 6, 7, 8, 5, 6)):
        terms = [int(x) for x in rng.integers(0, vocab, T)]
        slop = int(rng.integers(1, 4))
        enc = [orc.enc(x) if orc.has_term(x) else np.empty(0, np.uint64) for x in terms]
        if any(len(e) == 0 for e in enc):
            continue
        ids, counts, overflow = S.span_search(enc, slop, return_overflow=True)
        want = np.zeros(n_docs, dtype=np.float32)
        want[ids.astype(np.int64)] = counts
        got = dev.phrase_freqs_dense(terms, slop=slop)
        assert np.array_equal(got, want), f"seed {seed} terms {terms} slop {slop}: {np.flatnonzero(got != want)[:5]}"
        checked += 1
    dev.close()
    assert checked > 0


@pytest.mark.parametrize("seed", [5, 6])
def test_slop_span_table_overflow_matches_the_oracle(api, seed, monkeypatch):
    """Documents whose span table reaches the reference's 512 entries (long documents, a wide slop window).  The
    reference itself indexes one past its arrays there (spans.pyx:238-246, undefined behaviour), so the pinned behaviour
    is the oracle's guarded restatement: the "full" rule (min over terms of the summed popcounts, spans.pyx:306-311), the
    rest of the term's words of the document skipped -- except in the term's LAST document group, where the
    reference's search for the next document finds none and the remaining words still count.  Both table placements
    (LDS fast pass + wave-per-document heavy pass, and the slab machine) must reproduce it bit for bit."""
    from oracle import spans as S
    rng = np.random.default_rng(900 + seed)
    n_docs, vocab = 24, 3
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 700, seed=seed)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    seen_overflow = 0
    # (wide windows: the general route; T + slop <= 15: the doc-parallel route's wave-per-document machine and its
    #  lookup of the last document with candidates)
    small = 0
    for terms, slop in (([0, 1], 97), ([2, 0], 48), ([1, 2], 203), ([0, 1], 9), ([2, 0], 4), ([1, 2, 0], 11)):
        enc = [orc.enc(x) for x in terms]
        ids, counts, overflow = S.span_search(enc, slop, return_overflow=True)
        seen_overflow += overflow
        small += overflow if len(terms) + slop <= 15 else 0
        want = np.zeros(n_docs, dtype=np.float32)
        want[ids.astype(np.int64)] = counts
        for fast in ("1", "0"):
            set_opt("SA_SPAN_FAST", fast)
            got = dev.phrase_freqs_dense(terms, slop=slop)
            assert np.array_equal(got, want), f"seed {seed} terms {terms} slop {slop} fast {fast}: {np.flatnonzero(got != want)[:5]}"
    dev.close()
    assert seen_overflow > 0 and small > 0
    del rng


@pytest.mark.parametrize("min_posn,max_posn", [(0, 17), (18, None), (0, 35), (18, 53), (None, 17)])
def test_posn_range_matches_oracle(api, min_posn, max_posn):
    """min_posn / max_posn restriction (reference roaringish.py:266-282 incl. its unshifted-msb
    comparison) on exact phrases, same-term phrases and slop, device vs oracle"""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    for terms, slop in (([0, 1], 0), ([3, 0, 2], 0), ([0, 0], 0), ([1, 0, 0, 2], 0), ([5, 4, 0, 2, 1], 0), ([3, 7], 2)):
        want = orc.phrase_freqs(terms, slop=slop, min_posn=min_posn, max_posn=max_posn)
        got = dev.phrase_freqs_dense(terms, slop=slop, min_posn=min_posn, max_posn=max_posn)
        assert np.array_equal(got, want), (terms, slop, min_posn, max_posn)
    ids, tfs = O.popcount64_reduce(orc.posn_slice(orc.enc(0), min_posn, max_posn), 36, 0x3FFFF)
    assert np.array_equal(dev.termfreqs_dense(0, min_posn=min_posn, max_posn=max_posn), O.as_dense(ids, tfs, num_docs))


# ---------------------------------------------------------------------------------------------
# phrase batches: B phrases -> BM25 -> top-k on the device (sa_phrase_batch_create)
# ---------------------------------------------------------------------------------------------
def _check_phrase_batch(dev, orc, phrases, k, num_docs, doc_base=0):
    bt = dev.phrase_batch(phrases, k=k)
    for _ in range(2):                                   # a second run must reset slots / cursors
        bt.run()
        scores, docs = bt.fetch()
        for i, ph in enumerate(phrases):
            known = all(0 <= t < dev.n_terms for t in ph)
            want = orc.score(list(ph)) if known else np.zeros(num_docs, np.float32)
            ws, wd = O.topk(want, k)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[i, :n], ws[:n]), f"phrase {ph}: scores"
            assert np.array_equal(docs[i, :n], wd[:n] + doc_base), f"phrase {ph}: docs"
            assert (docs[i, n:] == NO_DOC).all() and (scores[i, n:] == 0).all()
    bt.close()


@pytest.mark.parametrize("k", [1, 10, 100])
def test_phrase_batch_matches_oracle(api, k):
    """2..8-term distinct phrases incl. the middle-out plan (shortest list strictly inside), ragged
    batch, an unknown term; scores bit-exact and ties broken by doc id"""
    n_docs, vocab = 9000, 60                             # 3 phrase tiles of 4096 docs
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 45, seed=3)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    rng = np.random.default_rng(8)
    phrases = [[0, 1], [1, 0], [3, 2, 1], [0, 5, 1, 2, 3], [4, 0, 30, 1, 2, 3], [7, 99], [2, 3]]
    for length in (2, 3, 4, 5, 6, 8):
        for _ in range(3):
            phrases.append([int(x) for x in rng.choice(14, length, replace=False)])
    # make sure at least one phrase takes the two-part plan
    lens_w = np.diff(rz.term_offsets(wt, vocab).astype(np.int64))
    assert any(1 < int(np.argmin(lens_w[ph])) < len(ph) - 2 for ph in phrases if max(ph) < vocab)
    _check_phrase_batch(dev, orc, phrases, k, n_docs)


@pytest.mark.parametrize("ptile,docdir,div", [("2048", "0", "32"), ("4096", "1", "32"), ("2048", "1", "1000000"),
                                              ("4096", "1", "0")])
def test_phrase_batch_tile_and_directory_variants(api, monkeypatch, ptile, docdir, div):
    """both tile sizes; probes by binary search only, by the doc directory of the frequent terms, with
    a directory for every term of >= 64 words, and with no directory in the index at all.  The
    single-phrase fused kernel takes the same probes."""
    set_opt("SA_PTILE", ptile)
    set_opt("SA_PHRASE_DOCDIR", docdir)
    set_opt("SA_DOCDIR_DIV", div)
    n_docs, vocab = 9000, 60
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 45, seed=3)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    info = dev.info()
    assert (info.n_docdir_terms > 0) == (div != "0")
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    phrases = [[0, 1], [3, 2, 1], [0, 5, 1, 2, 3], [4, 0, 30, 1, 2, 3], [2, 3], [1, 0, 2], [40, 41], [1, 45, 2]]
    _check_phrase_batch(dev, orc, phrases, 10, n_docs)
    for ph in phrases[:5]:
        assert np.array_equal(dev.phrase_freqs_dense(ph), orc.phrase_freqs(ph)), ph


def test_phrase_batch_golden_corpus_and_doc_base(api):
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api, doc_base=50000)
    phrases = []
    for i in range(int(g["n_phrases"])):
        terms = [int(x) for x in g[f"phr_{i}_terms"]]
        if len(set(terms)) == len(terms):
            phrases.append(terms)
    assert len(phrases) >= 5
    _check_phrase_batch(dev, orc, phrases, 10, num_docs, doc_base=50000)
    # the reference's own scores for the phrases it scored (golden fixture)
    bt = dev.phrase_batch(phrases, k=5)
    bt.run()
    scores, docs = bt.fetch()
    for i in range(0, int(g["n_phrases"]), 5):
        terms = [int(x) for x in g[f"phr_{i}_terms"]]
        if terms not in phrases:
            continue
        wants = dense_from_sparse(g[f"phr_{i}_sidx"], g[f"phr_{i}_sval"], num_docs)
        ws, wd = O.topk(wants, 5)
        n = int((ws > 0).sum())
        j = phrases.index(terms)
        assert np.array_equal(scores[j, :n], ws[:n]) and np.array_equal(docs[j, :n], wd[:n] + 50000)
    bt.close()


def test_phrase_batch_errors(api):
    vocab, dev = _device_from_strings(["foo bar baz", "bar baz foo"], api)
    with pytest.raises(Exception, match="at least two terms"):
        dev.phrase_batch([[0]], k=3)
    with pytest.raises(ValueError, match="slop"):
        dev.phrase_batch([[0, 1]], k=3, slop=-1)
    bt = dev.phrase_batch([[vocab["foo"], vocab["bar"]], [vocab["bar"], vocab["baz"]]], k=3)
    bt.run()
    scores, docs = bt.fetch()
    assert list(docs[0, :1]) == [0] and docs[0, 1] == NO_DOC
    assert list(docs[1]) == [0, 1, NO_DOC]
    bt.close()


def test_span_search_mirror_matches_oracle(api):
    """ops.span_search -- the kernel-level mirror of the reference's span_search(posns, lengths, Counter, slop,
    masks...) (roaringish/spans.pyx:322-330) -- fed with the candidate words _intersect_all selects, against the
    oracle's restatement of the same state machine (pinned to the reference by the slop goldens)."""
    from collections import Counter
    from oracle import spans as S
    from searcharray_amd import ops
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab).astype(np.int64)
    rng = np.random.default_rng(3)
    checked = 0
    for trial in range(25):
        T = int(rng.integers(2, 5))
        terms = [int(x) for x in rng.choice(30, size=T, replace=False)]
        slop = int(rng.integers(1, 5))
        enc = [words[off[x]:off[x + 1]] for x in terms]
        posns, lengths = S.intersect_all(enc)
        want_ids, want_counts = S.span_search(enc, slop)
        want = {int(k): int(v) for k, v in zip(want_ids, want_counts) if v != 0}
        got = Counter()
        ops.span_search(posns, lengths, got, np.uint64(slop), np.uint64(0xFFFFFFF000000000), np.uint64(0xFFFFFFFFFFFC0000),
                        np.uint64(28), np.uint64(18), api=api)
        assert {k: v for k, v in got.items() if v} == want, (terms, slop)
        checked += len(want)
    assert checked > 50
    # empty first term: nothing to walk; other layouts are refused
    got = Counter()
    ops.span_search(np.asarray([5 << 36 | 1], dtype=np.uint64), np.asarray([0, 0, 1], dtype=np.uint64), got, 2, api=api)
    assert not got
    with pytest.raises(NotImplementedError):
        ops.span_search(np.empty(0, np.uint64), np.zeros(3, np.uint64), got, 1, key_bits=32, api=api)


def test_phrase_batch_takes_every_phrase_score_takes(api):
    """Phrases the tile kernel does not take -- repeated terms (the reference's same-term rule,
    bigram_freqs.py:48-101), more than 18 terms, slop > 0 (span search, spans.py:71-187) -- mixed with ordinary
    ones in ONE batch: counted by the single-phrase kernels, ranked on the device, top-k equal to the oracle's
    score() + deterministic top-k.  Plus the reference's own outputs for its phrase and slop goldens."""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api, doc_base=7000)
    phrases, slops = [], []
    for i in range(int(g["n_phrases"])):                           # incl. the same-term phrases of the fixture
        phrases.append([int(x) for x in g[f"phr_{i}_terms"]]); slops.append(0)
    for i in range(int(g["n_slop"])):
        phrases.append([int(x) for x in g[f"slop_{i}_terms"]]); slops.append(int(g[f"slop_{i}_slop"]))
    phrases += [[0, 0], [0, 0, 1], [1, 0, 1, 0], [0, 1] * 10, list(range(19)), [0, vocab + 5, 0], [2, 1]]
    slops += [0, 0, 0, 0, 0, 0, 3]
    k = 6
    bt = dev.phrase_batch(phrases, k=k, slop=slops)
    for _ in range(2):
        bt.run()
        scores, docs = bt.fetch()
    bt.close()
    for i, (ph, sl) in enumerate(zip(phrases, slops)):
        known = all(0 <= x < vocab for x in ph)
        want = orc.score(list(ph), slop=sl) if known else np.zeros(num_docs, np.float32)
        ws, wd = O.topk(want, k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[i, :n], ws[:n]), f"phrase {ph} slop {sl}: scores"
        assert np.array_equal(docs[i, :n], wd[:n] + 7000), f"phrase {ph} slop {sl}: docs"
        assert (docs[i, n:] == NO_DOC).all()
    # the reference's own scores (golden fixture): exact phrases ...
    for i in range(int(g["n_phrases"])):
        ws, wd = O.topk(dense_from_sparse(g[f"phr_{i}_sidx"], g[f"phr_{i}_sval"], num_docs), k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[i, :n], ws[:n]) and np.array_equal(docs[i, :n], wd[:n] + 7000), f"golden phrase {i}"
    dev.close()


@pytest.mark.parametrize("lanes", ["1", "2", "4"])
def test_slop_batch_replays_and_scratch_moves(api, monkeypatch, lanes, on_emu):
    """A batch of slop phrases run again and again on 1, 2 and 4 lanes (streams with their own scratch areas, swapped
    in around the single-phrase kernels): every run must give the oracle's top-k, also after a larger dense query has
    grown and moved the first lane's scratch area."""
    if on_emu and lanes == "1":
        pytest.skip("one lane is what every other batch test of the CPU suite runs with SA_PHRASE_LANES unset on a single phrase")
    set_opt("SA_PHRASE_LANES", lanes)
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    rng = np.random.default_rng(17)
    phrases = [[int(x) for x in rng.choice(min(vocab, 30), 2, replace=False)] for _ in range(7)] + [[3, 1, 2]]
    slops = [int(x) for x in rng.integers(1, 4, len(phrases))]
    k = 5
    want = [O.topk(orc.score(list(ph), slop=sl), k) for ph, sl in zip(phrases, slops)]
    bt = dev.phrase_batch(phrases, k=k, slop=slops)

    def check(tag):
        bt.run()
        scores, docs = bt.fetch()
        for i, (ws, wd) in enumerate(want):
            n = int((ws > 0).sum())
            assert np.array_equal(scores[i, :n], ws[:n]) and np.array_equal(docs[i, :n], wd[:n]), f"{tag}: phrase {phrases[i]} slop {slops[i]}"
    for r in range(2 if on_emu else 4):
        check(f"run {r}")
    # a dense slop query over the most frequent terms needs more scratch than the batch did: the areas move
    dev.phrase_freqs_dense([0, 1, 2, 3, 4, 5], slop=3)
    for r in range(1 if on_emu else 3):
        check(f"run {r} after the scratch moved")
    bt.close()
    dev.close()


def test_search_phrases_with_repeated_tokens_and_slop(default_api):
    """SearchArray.search_phrases == top-k of SearchArray.score for phrases with repeated tokens and for slop"""
    from searcharray_amd import SearchArray
    docs = ["foo foo foo foo bar", "foo bar foo bar", "bar foo foo", "baz foo x y bar", "nothing here"] * 7
    arr = SearchArray.index(docs)
    phrases = [["foo", "foo"], ["foo", "bar", "foo", "bar"], ["foo", "bar"], ["foo", "missing"]]
    s, dd = arr.search_phrases(phrases, k=4)
    for i, ph in enumerate(phrases):
        ws, wd = O.topk(arr.score(ph), 4)
        n = int((ws > 0).sum())
        assert np.array_equal(s[i, :n], ws[:n]) and np.array_equal(dd[i, :n], wd[:n]), ph
    s, dd = arr.search_phrases([["foo", "bar"], ["baz", "bar"]], k=4, slop=3)
    for i, ph in enumerate([["foo", "bar"], ["baz", "bar"]]):
        ws, wd = O.topk(arr.score(ph, slop=3), 4)
        n = int((ws > 0).sum())
        assert np.array_equal(s[i, :n], ws[:n]) and np.array_equal(dd[i, :n], wd[:n]), ph


def test_phrases_of_any_length_and_slop_phrases_up_to_32_terms(api):
    """The reference has no limit on the terms of a phrase (compute_phrase_freqs chains bigrams, middle_out.py:73-168;
    PosnBitArray.phrase_freqs, :418-446) -- neither has the device's general chain (rounds 1-2 stopped at 128 terms);
    slop phrases go up to 32 terms (the span machine's term sets are 32-bit).  Device counts == oracle, dense calls
    and a phrase batch."""
    rng = np.random.default_rng(77)
    n_docs, vocab = 60, 6
    lens = rng.integers(150, 400, n_docs)
    toks = [rng.integers(0, vocab, int(n)) for n in lens]
    t = np.concatenate(toks).astype(np.uint32)
    d = np.repeat(np.arange(n_docs), lens).astype(np.uint64)
    p = np.concatenate([np.arange(int(n)) for n in lens]).astype(np.uint64)
    order = np.lexsort((p, d, t))
    t, d, p = t[order], d[order], p[order]
    words, wt = rz.encode_sorted(t, d, p)
    dl = lens.astype(np.float32)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), dl, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=dl)
    long_phrases = []
    for n, doc, start in ((130, 3, 5), (200, 17, 40), (19, 9, 0)):
        ph = [int(x) for x in toks[doc][start:start + n]]
        got = dev.phrase_freqs_dense(ph)
        want = orc.phrase_freqs(ph)
        assert np.array_equal(got, want), f"{n}-term phrase"
        assert got[doc] >= 1
        long_phrases.append(ph)
    for n, doc, start, slop in ((17, 4, 10, 1), (24, 30, 0, 2), (32, 11, 7, 1)):
        ph = [int(x) for x in toks[doc][start:start + n]]
        got = dev.phrase_freqs_dense(ph, slop=slop)
        want = orc.phrase_freqs(ph, slop=slop)
        assert np.array_equal(got, want), f"{n}-term slop-{slop} phrase"
    with pytest.raises(Exception):
        dev.phrase_freqs_dense([int(x) for x in toks[0][:33]], slop=1)
    pb = dev.phrase_batch(long_phrases + [[0, 1]], k=4)
    pb.run()
    ps, pd_ = pb.fetch()
    for i, ph in enumerate(long_phrases + [[0, 1]]):
        ws, wd = O.topk(orc.score(ph), 4)
        n = int((ws > 0).sum())
        assert np.array_equal(ps[i, :n], ws[:n]) and np.array_equal(pd_[i, :n], wd[:n]), f"batch phrase {i}"
    pb.close()
    dev.close()


def test_slop_phrases_of_a_batch_share_their_launches(api, monkeypatch):
    """Phrase batches score their slop phrases in SHARED launches (sa_span_counts_batch: blockIdx.y picks the phrase; one
    launch per stage and class of phrases -- 2, 3, 4, more terms -- and one ranking launch for all of them).  Mixed term
    counts and slops, an unknown term, an exact and a repeated-term phrase beside them: top-k equal to the oracle's and to
    the one-phrase-at-a-time route (SA_SPAN_MULTI=0).  Phrases of 2 - 4 terms that qualify take the doc-parallel kernel, ALL of a
    term count in one launch (sa_k_span_doc_fused_multi; SA_SPAN_DOC_MULTI=0: the five-stage shared launches for them too)."""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)
    rng = np.random.default_rng(3)
    phrases, slops = [], []
    for T in (2, 2, 3, 2, 4, 3, 5, 2, 6, 2, 3, 4):
        phrases.append([int(x) for x in rng.choice(min(vocab, 30), T, replace=False)])
        slops.append(int(rng.integers(1, 4)))
    phrases += [[0, 1], [2, 2, 3], [1, vocab + 7]]
    slops += [0, 0, 2]
    # the most frequent terms (their lists are longer than half the collection: the doc-parallel kernel walks ALL documents and,
    # in a batch, stages a block's counts in LDS and stores them together), incl. documents with many positions
    phrases += [[0, 1], [1, 0], [0, 2, 1], [3, 0], [1, 2, 0, 3]]
    slops += [2, 3, 2, 1, 3]
    k = 8
    results = {}
    for multi, docm in (("1", "1"), ("1", "0"), ("0", "1")):
        set_opt("SA_SPAN_MULTI", multi)
        set_opt("SA_SPAN_DOC_MULTI", docm)
        pb = dev.phrase_batch(phrases, k=k, slop=slops)
        for _ in range(2):
            pb.run()
        results[multi + docm] = pb.fetch()
        pb.close()
    for other in ("10", "01"):
        assert np.array_equal(results["11"][0], results[other][0]) and np.array_equal(results["11"][1], results[other][1]), other
    ps, pd_ = results["11"]
    for i, (ph, sl) in enumerate(zip(phrases, slops)):
        ws, wd = O.topk(orc.score(list(ph), slop=sl), k)
        n = int((ws > 0).sum())
        assert np.array_equal(ps[i, :n], ws[:n]) and np.array_equal(pd_[i, :n], wd[:n]), f"phrase {ph} slop {sl}"
    # a batch of two-term slop phrases only: every phrase ranks inside the doc-parallel kernel, no count vector, no ranking launch
    unset_opt("SA_SPAN_MULTI")
    unset_opt("SA_SPAN_DOC_MULTI")
    pairs = [[int(a), int(b)] for a, b in rng.choice(min(vocab, 40), (24, 2)) if a != b] + [[0, 1], [1, 0], [2, 0]]
    # ... with two or four of a block's waves holding span tables (SA_SPAN_TAB_WAVES; unset: by the size of the launch -- two when it
    # fills the device: six resident blocks per CU instead of four), then the mixed batch again on the two-wave instances of 2 - 4 terms
    for tw in (None, "2", "4"):
        if tw is None:
            unset_opt("SA_SPAN_TAB_WAVES")
        else:
            set_opt("SA_SPAN_TAB_WAVES", tw)
        pb = dev.phrase_batch(pairs, k=k, slop=2)
        for _ in range(2):
            pb.run()
        ps, pd_ = pb.fetch()
        pb.close()
        for i, ph in enumerate(pairs):
            ws, wd = O.topk(orc.score(list(ph), slop=2), k)
            n = int((ws > 0).sum())
            assert np.array_equal(ps[i, :n], ws[:n]) and np.array_equal(pd_[i, :n], wd[:n]), f"phrase {ph} slop 2 (pairs only, table waves {tw})"
    set_opt("SA_SPAN_TAB_WAVES", "2")
    pb = dev.phrase_batch(phrases, k=k, slop=slops)
    pb.run()
    got = pb.fetch()
    pb.close()
    assert np.array_equal(got[0], results["11"][0]) and np.array_equal(got[1], results["11"][1]), "mixed batch, two table waves per block"
    unset_opt("SA_SPAN_TAB_WAVES")
    dev.close()


@pytest.mark.parametrize("div", ["0", "32", "1000000"])
def test_fused_kernel_routes_dd_and_searched(api, monkeypatch, div):
    """sa_k_phrase_fused (csrc/sa_phrase.hip), one anchor word per lane: another term's words around it come through the
    term's doc directory row (SA_DOCDIR_DIV=1000000: every term of >= 64 words has one) or -- frequent terms without
    rows, SA_DOCDIR_DIV=0 -- by a search of the term's list.  All equal the
    oracle's counts (the reference's phrase_freqs, bigram_freqs.py:48-307 + middle_out.py:73-168), on lists long enough for
    several chunks per anchor, anchors in every phrase position, lists that end inside a wave's last round"""
    set_opt("SA_DOCDIR_DIV", div)
    set_opt("SA_PHRASE_MODE", "fused")
    n_docs, vocab = 6000, 300
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 40, seed=21)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    rng = np.random.default_rng(3)
    phrases = [[0, 1], [1, 0], [0, 1, 2], [2, 0, 1], [5, 0], [0, 40], [40, 0, 1], [1, 120, 0], [0, 1, 2, 3, 4], [7, 3, 0, 250],
               [299, 0], [0, 299], [150, 151], [30, 2, 31]]
    phrases += [[int(x) for x in rng.choice(60, 3, replace=False)] for _ in range(6)]
    for ph in phrases:
        got = dev.phrase_freqs_dense(ph)
        want = orc.phrase_freqs(ph)
        assert np.array_equal(got, want), f"phrase {ph} (SA_DOCDIR_DIV={div}): {int((got != want).sum())} docs differ"
    dev.close()

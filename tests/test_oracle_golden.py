"""Pin the CPU oracle (oracle/) to the reference: (a) known answers quoted from the
reference's own tests, (b) outputs of the reference itself captured in tests/golden/."""
import numpy as np
import pytest

from oracle import refimpl as O
from tests.helpers import load_golden, oracle_index, dense_from_sparse, golden_corpus

u64 = lambda x: np.asarray(x, dtype=np.uint64)  # noqa: E731


# ---- BM25 known answers: reference test/test_similarity.py:16-61 (Lucene explain values)
LUCENE = [
    (2, 14, 4, 2.7322686, 8516, 3.52482),
    (1, 5, 35, 50.580456, 8514, 3.8199246),
    (2, 7, 44, 50.580456, 8514, 4.5636616),
    (25, 7823, 152, 119.18542, 8516, 0.08028283),
]


@pytest.mark.parametrize("tf,df,dl,avgdl,n,expected", LUCENE)
def test_bm25_matches_lucene(tf, df, dl, avgdl, n, expected):
    got = O.bm25(np.asarray([tf], np.float32), np.asarray([df], np.float32),
                 np.asarray([dl], np.float32), avgdl, n)
    assert np.isclose(got, expected).all()


def _index_strings(docs):
    vocab = {}
    t, d, p = [], [], []
    for di, doc in enumerate(docs):
        for pi, tok in enumerate(doc.split()):
            t.append(vocab.setdefault(tok, len(vocab))); d.append(di); p.append(pi)
    t, d, p = np.asarray(t, np.int64), np.asarray(d, np.int64), np.asarray(p, np.int64)
    order = np.argsort(t, kind="stable")
    lens = np.asarray([len(doc.split()) for doc in docs], np.float32)
    return vocab, O.OracleIndex.from_triples(t[order], d[order], p[order], len(docs), doc_lens=lens)


FIXTURE_100 = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25


def test_search_fixture_known_answers():
    """reference test/test_search.py:86-102,121-124."""
    vocab, idx = _index_strings(FIXTURE_100)
    assert (idx.termfreqs(vocab["bar"]) == [2, 0, 1, 0] * 25).all()
    assert idx.docfreq(vocab["bar"]) == 50 and idx.docfreq(vocab["foo"]) == 25
    assert idx.avg_doc_length == 2.5
    assert np.isclose(idx.score(vocab["bar"]), [0.37066694, 0., 0.34314217, 0.] * 25).all()
    assert idx.score(12345).sum() == 0


# ---- phrase known answers: reference test/test_phrase_matches.py:17-194
PHRASE_SCENARIOS = [
    (["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [1, 0, 0, 0] * 25),
    (["foo bear bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [0, 0, 0, 0] * 25),
    (["foo foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [1, 0, 0, 0] * 25),
    (["foo bar bar bar foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [1, 0, 0, 0] * 25),
    (["foo bar baz baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar baz", [1, 0, 0, 0] * 25),
    (["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar baz", [0, 0, 0, 0] * 25),
    (["foo bar EEK foo URG bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar baz", [0, 0, 0, 0] * 25),
    (["foo foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo", [1, 0, 0, 0] * 25),
    (["foo foo bar", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo bar", [1, 0, 0, 0] * 25),
    (["foo bar bar", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar bar", [1, 0, 0, 0] * 25),
    (["foo bar bar foo bar bar", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar bar", [2, 0, 0, 0] * 25),
    (["foo foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo foo", [1, 0, 0, 0] * 25),
    (["foo foo foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo foo foo", [1, 0, 0, 0] * 25),
    (["foo foo foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo", [2, 0, 0, 0] * 25),
    (["foo foo foo foo baz foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo", [3, 0, 0, 0] * 25),
    (["foo foo bar bar", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo foo bar bar", [1, 0, 0, 0] * 25),
    (["foo bar foo bar", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [2, 0, 0, 0] * 25),
    (["foo bar baz foo bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar baz", [2, 0, 0, 0] * 25),
    (["foo bar baz foo bar buzz", "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar baz", [1, 0, 0, 0] * 25),
    (["foo " + " ".join(["bar"] * 50), "data2", "data3 bar", "bunny funny wunny"] * 25, "foo bar", [1, 0, 0, 0] * 25),
    (["data3 bar bar foo foo", "foo " + " ".join(["bar"] * 5), "foo " + " ".join(["bar"] * 50), "foo data2 bar",
      "bunny funny wunny"] * 25, "foo bar", [0, 1, 1, 0, 0] * 25),
    (["foo la ma bar bar baz", "data2 ma ta", "data3 bar ma", "bunny funny wunny",
      "la ma ta wa ga ao a b c d e f g a be ae i la ma ta wa ga ao a foo bar foo bar"] * 25,
     "la ma ta wa ga ao a", [0, 0, 0, 0, 2] * 25),
    (["foo bar bar baz " + " ".join([" dummy foo bar baz"] * 100), "data2", "data3 bar",
      "bunny funny wunny foo bar"] * 25, "foo bar", [101, 0, 0, 1] * 25),
]


@pytest.mark.parametrize("docs,phrase,expected", PHRASE_SCENARIOS)
def test_phrase_known_answers(docs, phrase, expected):
    vocab, idx = _index_strings(docs)
    got = idx.phrase_freqs([vocab[t] for t in phrase.split()])
    assert (got == np.asarray(expected, np.float32)).all()


@pytest.mark.parametrize("phrase", ["foo bar baz", "foo bar", "foo foo foo", "foo foo bar", "foo bar bar",
                                    "foo bar bar baz buz foo bar", "foo bar bar baz buz foo foo", "foo foo"])
@pytest.mark.parametrize("offset", [0, 1, 15, 16, 17, 18, 19, 34, 35, 36, 53, 54, 71, 99])
def test_phrase_offsets_cross_word_boundary(phrase, offset):
    """reference test/test_phrase_matches.py:249-265 (offsets crossing the 18-bit word)."""
    vocab, idx = _index_strings([" ".join(["dummy"] * offset) + " " + phrase, "not match"])
    got = idx.phrase_freqs([vocab[t] for t in phrase.split()])
    assert (got == [1, 0]).all()


# ---- set primitives: scenario dicts from reference test/test_snp_ops.py:96-154,457-548
def test_intersect_scenarios():
    lhs, rhs = u64([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]), u64([2, 4, 6, 8, 10])
    li, ri = O.intersect(lhs, rhs)
    assert (lhs[li.astype(int)] == rhs).all() and (rhs[ri.astype(int)] == rhs).all()
    lhs, rhs = u64([1, 1, 2, 2, 3, 3, 4, 4, 5, 5]), u64([1, 2, 2, 10])
    li, ri = O.intersect(lhs, rhs)
    assert (lhs[li.astype(int)] == [1, 2]).all()
    lhs = u64([0x1F, 0x2F, 0x3F, 0x4F, 0x5F, 0x6F, 0x7F, 0x8F, 0x9F, 0xAF])
    rhs = u64([0x2F, 0x4F, 0x6F, 0x8F, 0xAF])
    li, ri = O.intersect(lhs, rhs, mask=np.uint64(0xF0))
    assert ((lhs[li.astype(int)] & np.uint64(0xF0)) == [0x20, 0x40, 0x60, 0x80, 0xA0]).all()
    lhs, rhs = u64([9, 25, 28, 31, 31, 32, 38, 39, 42]), u64([0, 3, 11, 23, 32, 36, 41, 42])
    li, ri = O.intersect(lhs, rhs)
    assert (lhs[li.astype(int)] == [32, 42]).all()
    lhs, rhs = u64([0, 0, 1]), u64([0, 0, 0, 0, 1])
    li, ri = O.intersect(lhs, rhs)
    assert (lhs[li.astype(int)] == [0, 1]).all()
    with pytest.raises(ValueError):
        O.intersect(lhs, rhs, mask=np.uint64(0))


def test_merge_and_adjacent_scenarios():
    assert (O.merge(u64([1, 3, 5]), u64([2, 4, 6])) == [1, 2, 3, 4, 5, 6]).all()
    assert (O.merge(u64([1, 2, 5]), u64([2, 4])) == [1, 2, 2, 4, 5]).all()
    assert (O.merge(u64([1, 2, 5]), u64([2, 4]), drop_duplicates=True) == [1, 2, 4, 5]).all()
    lhs, rhs = u64([1, 5, 9]), u64([2, 5, 6, 7, 10])
    li, ri = O.adjacent(lhs, rhs)
    assert (lhs[li.astype(int)] == [1, 5, 9]).all() and (rhs[ri.astype(int)] == [2, 6, 10]).all()
    a, b, c, d = O.intersect_with_adjacents(lhs, rhs)
    assert (lhs[a.astype(int)] == [5]).all() and (lhs[c.astype(int)] == [1, 5, 9]).all()


def test_popcount_known_answers():
    """reference test/test_bitcount64.py:9-34."""
    assert (O.popcount64(u64([0, 1, 3, 0xFFFFFFFFFFFFFFFF, 0x8000000000000000])) == [0, 1, 2, 64, 1]).all()


# ---- reference outputs on its own captured arrays
@pytest.mark.parametrize("tag", ["128", "24179", "27685", "44358"])          # every complete triple the reference captured
def test_snp_fixtures_match_reference(tag):
    g = load_golden("snp_fixtures")
    lhs, rhs, mask = g[f"{tag}_lhs"], g[f"{tag}_rhs"], np.uint64(g[f"{tag}_mask"])
    li, ri = O.intersect(lhs, rhs, mask=mask)
    assert np.array_equal(li, g[f"{tag}_int_drop_l"]) and np.array_equal(ri, g[f"{tag}_int_drop_r"])
    lk, rk = O.intersect(lhs, rhs, mask=mask, drop_duplicates=False)
    assert np.array_equal(lk, g[f"{tag}_int_keep_l"]) and np.array_equal(rk, g[f"{tag}_int_keep_r"])
    a, b, c, d = O.intersect_with_adjacents(lhs, rhs, mask=mask)
    for got, key in ((a, "iwa_l"), (b, "iwa_r"), (c, "iwa_al"), (d, "iwa_ar")):
        assert np.array_equal(got, g[f"{tag}_{key}"]), key
    al, ar = O.adjacent(lhs, rhs, mask=mask)
    assert np.array_equal(al, g[f"{tag}_adj_l"]) and np.array_equal(ar, g[f"{tag}_adj_r"])
    assert np.array_equal(O.merge(lhs, rhs), g[f"{tag}_merge"])
    assert np.array_equal(O.merge(lhs, rhs, drop_duplicates=True), g[f"{tag}_merge_drop"])
    assert np.array_equal(O.unique(lhs, 36), g[f"{tag}_unique36"])
    k, c = O.popcount64_reduce(lhs, 36, 0x3FFFF)
    assert np.array_equal(k, g[f"{tag}_pcr_keys"]) and np.array_equal(c, g[f"{tag}_pcr_counts"])
    (ids, cnt), (_, rn) = O.bigram_freqs(lhs, rhs, O.CONT_RHS)
    assert np.array_equal(ids, g[f"{tag}_bg_rhs_ids"]) and np.array_equal(cnt, g[f"{tag}_bg_rhs_counts"])
    assert np.array_equal(rn, g[f"{tag}_bg_rhs_next"])
    (ids, cnt), (ln, _) = O.bigram_freqs(lhs, rhs, O.CONT_LHS)
    assert np.array_equal(ids, g[f"{tag}_bg_lhs_ids"]) and np.array_equal(cnt, g[f"{tag}_bg_lhs_counts"])
    assert np.array_equal(ln, g[f"{tag}_bg_lhs_next"])


@pytest.mark.parametrize("tag", ["185", "45907", "90596"])                       # captured without an rhs (fixtures/lhs_*.npy)
def test_snp_lhs_only_fixtures_match_reference(tag):
    g = load_golden("snp_fixtures_lhs")
    lhs = g[f"{tag}_lhs"]
    assert np.array_equal(O.unique(lhs, 36), g[f"{tag}_unique36"])
    assert np.array_equal(O.unique(lhs, 18), g[f"{tag}_unique18"])
    k, c = O.popcount64_reduce(lhs, 36, 0x3FFFF)
    assert np.array_equal(k, g[f"{tag}_pcr_keys"]) and np.array_equal(c, g[f"{tag}_pcr_counts"])


# ---- reference outputs on seeded synthetic corpora
@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_corpus_matches_reference(name):
    g, idx = oracle_index(name)
    n = idx.num_docs
    assert np.float32(idx.avg_doc_length) == g["avg_doc_length"]
    assert np.array_equal(idx.doc_lens, g["doc_lens"])
    vocab = int(g["meta"][1])
    dfs = np.asarray([idx.docfreq(t) for t in range(vocab)], dtype=np.uint64)
    assert np.array_equal(dfs, g["df"])                                   # bit-exact df
    for t in g["tf_terms"]:
        want = dense_from_sparse(g[f"tf_{t}_idx"], g[f"tf_{t}_val"], n)
        assert np.array_equal(idx.termfreqs(int(t)), want), f"tf t{t}"    # bit-exact tf
    for t in g["score_terms"]:
        assert np.array_equal(idx.score(int(t)), g[f"score_{t}"]), f"score t{t}"      # bit-exact fp32
        assert np.array_equal(idx.score(int(t), k1=1.7, b=0.3), g[f"score_custom_{t}"])
    for q, want in zip(g["or_queries"], g["or_scores"]):
        assert np.array_equal(idx.score_terms_sum([int(t) for t in q]), want)
    for i in range(int(g["n_phrases"])):
        terms = [int(t) for t in g[f"phr_{i}_terms"]]
        want = dense_from_sparse(g[f"phr_{i}_idx"], g[f"phr_{i}_val"], n)
        assert np.array_equal(idx.phrase_freqs(terms), want), f"phrase {terms}"       # bit-exact counts
        wants = dense_from_sparse(g[f"phr_{i}_sidx"], g[f"phr_{i}_sval"], n)
        assert np.array_equal(idx.score(terms), wants), f"phrase score {terms}"


@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_slop_matches_reference(name):
    """slop > 0 counts (reference phrase/spans.py + roaringish/spans.pyx) on the seeded corpora:
    only match/no-match booleans are pinned by the reference's own tests (test_slop_matches.py), so
    the counts are pinned against outputs of the reference itself."""
    from oracle import spans as S
    g, idx = oracle_index(name)
    n = idx.num_docs
    for i in range(int(g["n_slop"])):
        terms = [int(t) for t in g[f"slop_{i}_terms"]]
        slop = int(g[f"slop_{i}_slop"])
        want = dense_from_sparse(g[f"slop_{i}_idx"], g[f"slop_{i}_val"], n)
        got = idx.phrase_freqs(terms, slop=slop)
        assert np.array_equal(got, want), f"slop {terms} {slop}"
        _, _, overflow = S.span_search([idx.enc(t) for t in terms], slop, return_overflow=True)
        assert overflow == 0            # no doc fills the 512-span table (reference UB territory)


# ---- the goldens against the REFERENCE ITSELF (oracle/_ref: /root/reference compiled by oracle/build_ref.sh) ----------------
# The fixtures were written by tests/golden/make_golden.py from the reference's outputs; this re-scores them through the built
# tree whenever it is present (this container; not the GPU box), so a drift between the generator, the fixtures and the
# reference shows up under pytest, not only in a reviewer's shell.
def _ref_or_skip():
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("oracle/_ref is not built (bash oracle/build_ref.sh where /root/reference exists)")
    return ref_loader


@pytest.mark.parametrize("name", ["zipf_small", "zipf_sparse"])
def test_corpus_goldens_equal_the_built_reference(name):
    """df of every term, sparse tf, single-term BM25 (default and custom k1 / b), 4-term disjunctions (np.sum of score vectors,
    test/test_msmarco.py:353), phrase counts + scores and slop counts of tests/golden/<name>.npz, recomputed by the reference's
    own code (SearchArray.score / termfreqs / docfreq over the same words) -- bit for bit"""
    _ref_or_skip().reference()
    from searcharray.postings import SearchArray                    # (oracle/_ref is first on sys.path once loaded)
    from searcharray.similarity import bm25_similarity
    g, _, lens, num_docs, vocab = golden_corpus(name)
    # indexed as tests/golden/make_golden.py indexed it: the token stream as whitespace documents through SearchArray.index
    starts = np.zeros(num_docs + 1, dtype=np.int64)
    np.cumsum(g["lens"], out=starts[1:])
    docs = [" ".join(f"t{t}" for t in g["terms"][starts[i]:starts[i + 1]]) for i in range(num_docs)]
    sa = SearchArray.index(docs, autowarm=False)
    assert np.float32(sa.avg_doc_length) == g["avg_doc_length"]
    assert np.array_equal(np.asarray([sa.docfreq(f"t{x}") for x in range(vocab)], dtype=np.uint64), g["df"])

    def dense(idx, val):
        out = np.zeros(num_docs, dtype=np.float32)
        out[idx] = val
        return out
    for x in g["tf_terms"]:
        assert np.array_equal(sa.termfreqs(f"t{x}"), dense(g[f"tf_{x}_idx"], g[f"tf_{x}_val"])), f"tf t{x}"
    custom = bm25_similarity(k1=1.7, b=0.3)
    for x in g["score_terms"]:
        assert np.array_equal(sa.score(f"t{x}"), g[f"score_{x}"]), f"score t{x}"
        assert np.array_equal(sa.score(f"t{x}", similarity=custom), g[f"score_custom_{x}"]), f"custom score t{x}"
    for q, want in zip(g["or_queries"], g["or_scores"]):
        assert np.array_equal(np.sum([sa.score(f"t{x}") for x in q], axis=0), want), f"disjunction {q}"
    for i in range(int(g["n_phrases"])):
        toks = [f"t{x}" for x in g[f"phr_{i}_terms"]]
        assert np.array_equal(sa.termfreqs(toks), dense(g[f"phr_{i}_idx"], g[f"phr_{i}_val"])), f"phrase counts {toks}"
        assert np.array_equal(sa.score(toks), dense(g[f"phr_{i}_sidx"], g[f"phr_{i}_sval"])), f"phrase scores {toks}"
    for i in range(int(g["n_slop"])):
        toks = [f"t{x}" for x in g[f"slop_{i}_terms"]]
        slop = int(g[f"slop_{i}_slop"])
        assert np.array_equal(sa.termfreqs(toks, slop=slop), dense(g[f"slop_{i}_idx"], g[f"slop_{i}_val"])), f"slop {slop} {toks}"


def test_snp_fixture_goldens_equal_the_built_reference():
    """tests/golden/snp_fixtures.npz (intersect / adjacent / intersect_with_adjacents / merge / unique / popcount64_reduce on
    arrays captured from a reference run) recomputed by the reference's Cython kernels, called as make_golden.py calls them"""
    _ref_or_skip().reference()
    from searcharray.roaringish.intersect import intersect, adjacent, intersect_with_adjacents
    from searcharray.roaringish.merge import merge
    from searcharray.roaringish.unique import unique
    from searcharray.roaringish.popcount import popcount64_reduce
    g = load_golden("snp_fixtures")
    tags = sorted({k.split("_")[0] for k in g.files})
    assert tags
    lsb_mask = np.uint64((1 << 18) - 1)
    for tag in tags:
        lhs, rhs, mask = g[f"{tag}_lhs"], g[f"{tag}_rhs"], np.uint64(g[f"{tag}_mask"])
        li, ri = intersect(lhs, rhs, mask=mask)
        assert np.array_equal(li, g[f"{tag}_int_drop_l"]) and np.array_equal(ri, g[f"{tag}_int_drop_r"]), tag
        lk, rk = intersect(lhs, rhs, mask=mask, drop_duplicates=False)
        assert np.array_equal(lk, g[f"{tag}_int_keep_l"]) and np.array_equal(rk, g[f"{tag}_int_keep_r"]), tag
        a, b, c, d = intersect_with_adjacents(lhs, rhs, mask=mask)
        for got, key in ((a, "iwa_l"), (b, "iwa_r"), (c, "iwa_al"), (d, "iwa_ar")):
            assert np.array_equal(got, g[f"{tag}_{key}"]), f"{tag} {key}"
        al, ar = adjacent(lhs, rhs, mask=mask)
        assert np.array_equal(al, g[f"{tag}_adj_l"]) and np.array_equal(ar, g[f"{tag}_adj_r"]), tag
        assert np.array_equal(merge(lhs, rhs), g[f"{tag}_merge"]), tag
        assert np.array_equal(merge(lhs, rhs, drop_duplicates=True), g[f"{tag}_merge_drop"]), tag
        assert np.array_equal(unique(lhs, 36), g[f"{tag}_unique36"]), tag
        keys, counts = popcount64_reduce(lhs, np.uint64(36), lsb_mask)
        assert np.array_equal(keys, g[f"{tag}_pcr_keys"]) and np.array_equal(counts, g[f"{tag}_pcr_counts"]), tag


# ---- slop counts at BASELINE config 5's scale: tests/golden/slop_1m.npz = the reference's outputs on zipf-1M ----------------
def _slop_1m_corpus():
    from searcharray_amd import synth
    lens, terms = synth.zipf_batch_tokens(0, 1_000_000, 100_000, fast=True)
    words, counts = synth.encode_batch(lens, terms, 100_000)
    words, term_off = synth.concat_term_major([(words, counts)], 100_000)
    return words, term_off, lens.astype(np.float32)


def slop_1m_digest(tf):
    import hashlib
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(tf, dtype=np.float32).tobytes()).digest(), dtype=np.uint8)


def test_slop_1m_golden_oracle_port_and_built_reference():
    """oracle/spans.c (the port the GPU tests and the bench compare with at 1M docs) equals the REFERENCE's slop-2 counts and BM25
    top-10 on every query of the fixture -- the 32 of test_config_scale.py, 44 of the bench's 256, the four heaviest among them
    (up to 1.6 M matches); and, where oracle/_ref is built, the reference recomputes the light ones (the heavy ones take it minutes:
    tests/golden/make_slop_1m.py is the offline job)"""
    g = load_golden("slop_1m")
    words, term_off, doc_lens = _slop_1m_corpus()
    orc = O.OracleIndex(words, np.arange(100_000), term_off, doc_lens, 1_000_000)
    n = 0
    for tag in "tb":
        for i, q in enumerate(g[f"{tag}_queries"]):
            ph = [int(x) for x in q]
            tf = orc.phrase_freqs(ph, slop=2)
            nz = np.flatnonzero(tf)
            assert len(nz) == int(g[f"{tag}{i}_stat"][0]) and float(tf.sum()) == float(g[f"{tag}{i}_stat"][1]), (tag, i, ph)
            assert np.array_equal(slop_1m_digest(tf), g[f"{tag}{i}_sha1"]), (tag, i, ph)
            if f"{tag}{i}_idx" in g.files:
                assert np.array_equal(nz, g[f"{tag}{i}_idx"]) and np.array_equal(tf[nz], g[f"{tag}{i}_val"])
            ws, wd = O.topk(orc.score(ph, slop=2), 10)
            m = int((g[f"{tag}{i}_top_scores"] > 0).sum())
            assert int((ws > 0).sum()) == m
            assert np.allclose(ws[:m], g[f"{tag}{i}_top_scores"][:m], rtol=1e-5, atol=0), (tag, i, ph)   # (north_star: slop scores within 1e-5)
            n += 1
    assert n >= 72
    from oracle import ref_loader
    if ref_loader.available():
        sa = ref_loader.reference_array(words, term_off, doc_lens)
        done = 0
        for tag in "tb":
            for i, q in enumerate(g[f"{tag}_queries"]):
                if int(g[f"{tag}{i}_stat"][0]) > 200:
                    continue
                tf = np.asarray(sa.termfreqs([f"t{int(x)}" for x in q], slop=2), dtype=np.float32)
                assert np.array_equal(slop_1m_digest(tf), g[f"{tag}{i}_sha1"]), (tag, i)
                done += 1
        assert done >= 40

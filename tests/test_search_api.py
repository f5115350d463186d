"""The drop-in surface: SearchArray.index / termfreqs / docfreq / score / phrases / pandas
behaviour, written after the reference's own tests (test/test_search.py, test/test_phrase_matches.py,
test/test_similarity.py) with their known answers.  Runs on "emu" (CPU suite) and "gpu"."""
import pickle
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pandas as pd
import pytest

from searcharray_amd import SearchArray, Terms, bm25_similarity, bm25_impact, classic_similarity
from searcharray_amd.similarity import compute_idf
from tests.test_oracle_golden import PHRASE_SCENARIOS, LUCENE
from tests.helpers import load_golden

DOCS = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25


@pytest.fixture(scope="module")
def data(default_api):
    return SearchArray.index(DOCS)


def test_term_freqs_and_doc_freq(data):
    """reference test_search.py:75-95"""
    assert ((data.termfreqs("foo") > 0) == [True, False, False, False] * 25).all()
    assert (data.termfreqs("not_present") == 0).all()
    assert (data.termfreqs("bar") == [2, 0, 1, 0] * 25).all()
    assert data.docfreq("bar") == 50 and data.docfreq("foo") == 25 and data.docfreq("nope") == 0
    with pytest.raises(TypeError):
        data.docfreq(["bar"])
    with pytest.raises(TypeError):
        data.termfreqs(5)


def test_doc_lengths(data):
    """reference test_search.py:98-102"""
    assert data.doclengths().shape == (100,)
    assert (data.doclengths() == [4, 1, 2, 3] * 25).all()
    assert data.avg_doc_length == 2.5
    assert len(data) == 100 and data.corpus_size == 100


def test_default_score_matches_lucene(data):
    """reference test_search.py:121-124"""
    bm25 = data.score("bar")
    assert bm25.shape == (100,) and bm25.dtype == np.float32
    assert np.isclose(bm25, [0.37066694, 0., 0.34314217, 0.] * 25).all()
    assert np.array_equal(data.score(["bar"]), bm25)                  # one-element list == the term


def test_custom_similarities(data):
    """reference test_search.py:105-118, test_similarity.py:64-86"""
    bm25 = data.score("bar")
    custom = bm25_similarity(k1=10, b=0.01)
    c1, c2 = data.score("bar", similarity=custom), data.score("bar", similarity=custom)
    assert np.array_equal(c1, c2)
    assert not np.isclose(bm25[bm25 > 0], c1[c1 > 0]).any()
    impact = data.score("bar", similarity=bm25_impact())
    idf = compute_idf(100, np.asarray([50]))
    assert np.isclose(impact * idf, bm25).all()
    classic = data.score("bar", similarity=classic_similarity())
    assert classic.shape == (100,) and (classic[1::4] == 0).all() and (classic[0::4] > 0).all()


@pytest.mark.parametrize("tf,df,dl,avgdl,n,expected", LUCENE)
def test_bm25_closure_matches_lucene(default_api, tf, df, dl, avgdl, n, expected):
    """reference test_similarity.py:16-61: the Similarity protocol called directly"""
    got = bm25_similarity(k1=1.2, b=0.75)(np.asarray([tf], np.float32), np.asarray([df], np.float32),
                                          np.asarray([dl], np.float32), avgdl, n)
    assert np.isclose(got, expected).all()


def test_and_or_queries(data):
    """reference test_search.py:127-226: boolean combinations are plain numpy on score()"""
    foo, bar = data.score("foo") > 0, data.score("bar") > 0
    assert ((foo & bar) == [True, False, False, False] * 25).all()
    assert ((foo | bar) == [True, False, True, False] * 25).all()
    summed = np.sum([data.score(t) for t in ("foo", "bar")], axis=0)
    assert summed.argmax() % 4 == 0


def test_empty_docs(default_api):
    """reference test_search.py:42-59"""
    arr = SearchArray.index(pd.DataFrame({"data": [""] * 100})["data"])
    assert arr.score("foo").sum() == 0
    assert arr.score(["foo", "bar"]).sum() == 0
    assert arr.isna().all()


def test_slices_copies_and_views(data):
    sliced = data[1::2]
    assert len(sliced) == 50
    assert np.array_equal(sliced.termfreqs("bar"), data.termfreqs("bar")[1::2])
    assert np.array_equal(sliced.score("bar"), data.score("bar")[1::2])
    mask = np.asarray([True, False, True, False] * 25)
    assert np.array_equal(data[mask].termfreqs("bar"), np.asarray([2, 1] * 25, dtype=np.float32))
    cp = data.copy()
    assert (cp == data).all() and np.array_equal(cp.score("foo"), data.score("foo"))
    assert np.array_equal(data.take([0, 2, 2]).termfreqs("bar"), [2, 1, 1])
    first = data[0]
    assert isinstance(first, Terms) and first.postings == {"foo": 1, "bar": 2, "baz": 1} and first.doc_len == 4
    assert (first.positions("bar") == [1, 2]).all()


def test_positions(data):
    """reference test_phrase_matches.py:400-425"""
    posns = data.positions("bar")
    assert len(posns) == 100 and (posns[0] == [1, 2]).all() and (posns[2] == [1]).all() and len(posns[1]) == 0
    sub = data.positions("bar", np.asarray([True, False, False, False] * 25))
    assert len(sub) == 25 and all((p == [1, 2]).all() for p in sub)


@pytest.mark.parametrize("docs,phrase,expected", PHRASE_SCENARIOS[1::3])
def test_phrase_api(default_api, docs, phrase, expected):
    """reference test_phrase_matches.py:224-246 (full array and odd-doc slice)"""
    arr = SearchArray.index(docs)
    before = arr.copy()
    tfs = arr.termfreqs(phrase.split())
    assert (tfs == expected).all()
    assert (arr == before).all()
    assert (arr[1::2].termfreqs(phrase.split()) == np.asarray(expected)[1::2]).all()
    scores = arr.score(phrase.split())
    assert ((scores > 0) == (np.asarray(expected) > 0)).all()


def test_phrase_too_many_posns(default_api):
    """reference test_phrase_matches.py:382-397"""
    big = "foo bar baz " + " ".join(["dummy"] * (2 ** 18 - 1)) + " blah blah blah"
    with pytest.raises(ValueError):
        SearchArray.index([big, "not match"])
    arr = SearchArray.index([big, "not match"], truncate=True)
    assert (arr.termfreqs(["foo", "bar", "baz"]) == [1, 0]).all()


MINMAX_DOCS = ["foo bar bar baz" + " ".join(["boz"] * 25) + " foo bar", "data2", "data3 bar", "bunny funny wunny"] * 25


@pytest.mark.parametrize("phrase,min_posn,max_posn,expected", [
    (["foo", "bar"], 0, 17, [1, 0, 0, 0] * 25),
    (["foo", "bar"], 0, None, [2, 0, 0, 0] * 25),
    (["foo", "bar"], 18, None, [1, 0, 0, 0] * 25),
])
def test_min_max_posn(default_api, phrase, min_posn, max_posn, expected):
    """reference test/test_minmax_posns.py:5-52 (known answers)"""
    arr = SearchArray.index(MINMAX_DOCS)
    before = arr.copy()
    assert (arr.termfreqs(phrase, min_posn=min_posn, max_posn=max_posn) == expected).all()
    assert (arr == before).all()
    scores = arr.score(phrase, min_posn=min_posn, max_posn=max_posn)
    assert ((scores > 0) == (np.asarray(expected) > 0)).all()


def test_min_max_posn_same_term_and_single_term(default_api):
    docs = ["foo foo baz baz" + " ".join(["boz"] * 25) + " foo foo", "data2", "data3 bar", "bunny funny wunny"] * 25
    arr = SearchArray.index(docs)
    assert (arr.termfreqs(["foo", "foo"], min_posn=0, max_posn=17) == [1, 0, 0, 0] * 25).all()     # test_minmax_posns.py:35-43
    assert (arr.termfreqs("foo", min_posn=0, max_posn=17) == [2, 0, 0, 0] * 25).all()
    assert (arr.termfreqs("foo", min_posn=18) == [2, 0, 0, 0] * 25).all()
    assert (arr.termfreqs("foo") == [4, 0, 0, 0] * 25).all()
    assert ((arr.score("foo", min_posn=0, max_posn=17) > 0) == [True, False, False, False] * 25).all()
    with pytest.raises(ValueError):
        arr.termfreqs(["foo", "foo"], min_posn=5)
    with pytest.raises(ValueError):
        arr.termfreqs("foo", max_posn=18)


def test_threaded_scoring_is_deterministic(data):
    """reference test_tmdb.py:285-312: concurrent score() calls must agree with serial ones"""
    want = {t: data.score(t) for t in ("foo", "bar", "baz", "data2")}
    phrase = data.score(["foo", "bar"])
    with ThreadPoolExecutor(4) as ex:
        futs = [(t, ex.submit(data.score, t)) for t in list(want) * 8]
        pf = [ex.submit(data.score, ["foo", "bar"]) for _ in range(8)]
        for t, f in futs:
            assert np.array_equal(f.result(), want[t])
        for f in pf:
            assert np.array_equal(f.result(), phrase)


def test_pickle_round_trip(data):
    """reference test_search.py:62-73 (pickle of the array; the HBM copy is rebuilt on demand)"""
    again = pickle.loads(pickle.dumps(data))
    assert np.array_equal(again.score("bar"), data.score("bar"))
    assert (again == data).all()


def test_pandas_integration(data):
    df = pd.DataFrame({"text": DOCS, "idx": np.arange(100)})
    df["tokens"] = SearchArray.index(df["text"])
    assert str(df["tokens"].dtype) == "tokenized_text"
    top = df[df["tokens"].array.score("bar") > 0]
    assert len(top) == 50
    sub = df.iloc[::4]
    assert (sub["tokens"].array.termfreqs("foo") == 1).all()
    both = pd.concat([df.iloc[:4], df.iloc[4:8]])
    assert (both["tokens"].array.termfreqs("bar") == [2, 0, 1, 0, 2, 0, 1, 0]).all()


def test_setitem_and_from_dicts(default_api):
    arr = SearchArray.index(["foo bar", "baz", "foo foo"])
    arr[1] = arr[2]
    assert (arr.termfreqs("foo") == [1, 2, 2]).all()
    from_dicts = SearchArray([{"foo": 1, "bar": 2}, {}, {"baz": 1}])
    assert len(from_dicts) == 3 and from_dicts.isna().tolist() == [False, True, False]
    assert from_dicts[0] == Terms({"foo": 1, "bar": 2}, doc_len=2)


def test_slices_use_subset_local_docfreq(default_api):
    """Known answers produced by the reference itself (v0.0.73, `arr[mask].score(...)`): a slice keeps
    the global corpus_size / avg_doc_length but its docfreq -- hence the idf of every score on it --
    counts only the slice's docs (FilteredPosns, reference middle_out.py:291-317, postings.py:345-358)."""
    docs = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny", "bar foo", "bar bar"] * 3
    arr = SearchArray.index(docs)
    mask = np.zeros(len(docs), dtype=bool)
    mask[[0, 2, 4, 5, 9]] = True
    sl = arr[mask]
    assert arr.docfreq("bar") == 12 and sl.docfreq("bar") == 4
    assert sl.corpus_size == 18 and np.isclose(sl.avg_doc_length, 2.3333333)
    assert np.allclose(arr.score("bar")[mask], [0.21791613, 0.202136, 0.202136, 0.27264857, 0.0], rtol=1e-6)
    assert np.allclose(sl.score("bar"), [0.7496305, 0.69534695, 0.69534695, 0.93790984, 0.0], rtol=1e-6)
    assert np.allclose(sl.score(["foo", "bar"]), [1.2200788, 0, 0, 0, 0], rtol=1e-6)
    assert np.array_equal(sl.termfreqs("bar"), [2, 1, 1, 2, 0]) and np.array_equal(sl.termfreqs(["bar", "bar"]), [1, 0, 0, 1, 0])
    sl2 = arr[2:8]
    assert sl2.docfreq("bar") == 4
    assert np.allclose(sl2.score("bar"), [0.69534695, 0.0, 0.69534695, 0.93790984, 0.7496305, 0.0], rtol=1e-6)


def test_permuted_and_repeated_views_score_row_by_row(default_api):
    """A view in arbitrary row order or with repeated rows (a sorted DataFrame, `take`): every row gets the
    value of its own document, with the docfreq of the view's distinct docs.  (Deliberate difference: the
    reference's subset path assumes sorted unique rows and returns misaligned values here -- DESIGN 4.)"""
    docs = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25
    arr = SearchArray.index(docs)
    sorted_rows = np.asarray([0, 2, 4, 6])
    base = arr[sorted_rows]
    want = dict(zip(sorted_rows.tolist(), base.score("bar").tolist()))
    want_ph = dict(zip(sorted_rows.tolist(), base.score(["bar", "baz"]).tolist()))
    for rows in ([6, 4, 2, 0], [2, 0, 2, 6, 4, 4]):
        view = arr.take(rows)
        assert view.docfreq("bar") == base.docfreq("bar")
        assert view.score("bar").tolist() == [want[r] for r in rows]
        assert view.score(["bar", "baz"]).tolist() == [want_ph[r] for r in rows]
        assert view.termfreqs("bar").tolist() == [2.0 if r % 4 == 0 else 1.0 for r in rows]


def test_batched_search_matches_score_loops(default_api):
    """SearchArray.search / search_phrases == top-k of the dense score() the reference's callers build"""
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(30)]
    p = 1.0 / np.arange(1, 31)
    p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=max(1, rng.poisson(12)), p=p)) for _ in range(700)]
    arr = SearchArray.index(docs)
    queries = [["w0", "w7", "w20"], "w3 w29", ["w1"], ["nope", "w2"], ["w5", "w5"]]
    scores, ids = arr.search(queries, k=7)
    for i, q in enumerate(queries):
        toks = q.split() if isinstance(q, str) else q
        dense = np.sum([arr.score(t) for t in toks], axis=0)
        order = np.lexsort((np.arange(len(dense)), -dense))[:7]
        n = int((dense[order] > 0).sum())
        assert np.array_equal(ids[i, :n], order[:n].astype(np.uint64)) and np.allclose(scores[i, :n], dense[order][:n], rtol=1e-6)
    phrases = [["w0", "w1"], ["w2", "w0", "w1"], "w1 w0"]
    pscores, pids = arr.search_phrases(phrases, k=5)
    for i, ph in enumerate(phrases):
        toks = ph.split() if isinstance(ph, str) else ph
        dense = arr.score(toks)
        order = np.lexsort((np.arange(len(dense)), -dense))[:5]
        n = int((dense[order] > 0).sum())
        assert np.array_equal(pids[i, :n], order[:n].astype(np.uint64)) and np.allclose(pscores[i, :n], dense[order][:n], rtol=1e-6)
    with pytest.raises(ValueError, match="slice"):
        arr[:10].search(queries)
    with pytest.raises(ValueError, match="BM25"):
        arr.search(queries, similarity=lambda *a: a[0])


def test_degenerate_bm25_parameters_keep_the_reference_nan(default_api):
    """k1 = 0 makes the BM25 denominator 0 for docs without the term; the reference returns 0/0 = NaN
    there (known answers produced by the reference itself)"""
    from searcharray_amd.similarity import bm25_similarity
    arr = SearchArray.index(["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"])
    got = arr.score("bar", similarity=bm25_similarity(k1=0.0))
    assert np.allclose(got[[0, 2]], 0.6931472) and np.isnan(got[[1, 3]]).all()
    got = arr.score(["foo", "bar"], similarity=bm25_similarity(k1=0.0))
    assert np.isclose(got[0], 1.89712) and np.isnan(got[1:]).all()


def test_terms_helpers_memory_report_and_set_of_results(default_api):
    import pandas as pd
    from searcharray_amd import SetOfResults
    arr = SearchArray.index(["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"])
    doc = arr[0]
    dense = doc.tf_to_dense(arr.term_dict)
    assert dense[arr.term_dict.get_term_id("bar")] == 2 and dense.sum() == 4
    raw = dict(doc.raw_positions(arr.term_dict))
    assert list(raw[arr.term_dict.get_term_id("bar")]) == [1, 2]
    assert list(dict(doc.raw_positions(arr.term_dict, "foo"))[arr.term_dict.get_term_id("foo")]) == [0]
    report = arr.memory_report(N=3)
    assert "Number of Terms: 8" in report and "Term 0: bar" in report and "Cumulative" in report
    df = pd.DataFrame({"title": arr, "id": [10, 11, 12, 13]})
    res = SetOfResults(df)
    res.ins_top_n(arr.score("bar"), N=2, query="bar", metadata={"run": "a"})
    res.ins_top_n(arr.score("bunny"), N=1, query="bunny", metadata={"run": ["b"]})
    out = res.get_all()
    assert list(out.columns) == ["id", "score", "query", "run", "rank"]
    assert list(out["id"]) == [10, 12, 13] and list(out["rank"]) == [1, 2, 1]
    with pytest.raises(ValueError):
        res.ins_top_n(arr.score("bar"), N=2, query="x", metadata={"run": ["only one"]})


# ---- the reference's other stock similarities as device kernels ----------------------
def _sim(name):
    from searcharray_amd.similarity import bm25_impact, bm25_legacy_similarity, classic_similarity
    return {"impact": bm25_impact(), "impact_b": bm25_impact(k1=0.9, b=0.3), "legacy": bm25_legacy_similarity(),
            "legacy_b": bm25_legacy_similarity(k1=1.7, b=0.3), "classic": classic_similarity()}[name]


SIM_CASES = [("w0", {}), ("w7", {}), ("w49", {}), ("nope", {}), (["w0", "w1"], {}), (["w1", "w0", "w2"], {}),
             (["w0", "w0"], {}), (["w0", "nope"], {}), (["w3", "w1"], {"slop": 2}),
             ("w0", {"min_posn": 0, "max_posn": 17}), (["w0", "w1"], {"min_posn": 0, "max_posn": 35})]


@pytest.mark.parametrize("name", ["impact", "impact_b", "legacy", "legacy_b", "classic"])
def test_other_similarities_bit_identical_to_reference(default_api, name):
    """bm25_impact / bm25_legacy_similarity / classic_similarity (reference similarity.py:41-89) run on the
    device with numpy's rounding: same dtype and the same bits as the reference's host expressions
    (goldens: tests/golden/make_golden.py ONLY=similarity)."""
    g = load_golden("similarities")
    assert int(g["n_cases"]) == len(SIM_CASES)
    arr = SearchArray.index([str(d) for d in g["docs"]])
    sim = _sim(name)
    calls = []
    dev = arr._core.device()
    orig = dev.similarity_dense
    dev.similarity_dense = lambda *a, **k: (calls.append(a[0]), orig(*a, **k))[1]
    try:
        for i, (tok, kw) in enumerate(SIM_CASES):
            got, want = arr.score(tok, similarity=sim, **kw), g[f"{name}_{i}"]
            assert got.dtype == want.dtype, (tok, kw)
            assert np.array_equal(got, want, equal_nan=True), (tok, kw, np.abs(got - want).max())
        rows = g["rows"]
        for key, tok in (("slice", "w0"), ("slice_phrase", ["w0", "w1"])):
            got, want = arr[rows].score(tok, similarity=sim), g[f"{name}_{key}"]
            assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True)
    finally:
        dev.similarity_dense = orig
    assert len(calls) == len(SIM_CASES) + 2            # every score came from the device kernel
    # the closures keep working as plain Similarity callables (protocol use) and agree with the kernel
    tf = arr.termfreqs("w7")
    direct = sim(tf.copy(), np.asarray([arr.docfreq("w7")]), arr.doc_lens, arr.avg_doc_length, arr.corpus_size)
    assert np.array_equal(direct, g[f"{name}_1"], equal_nan=True)


def test_eq_compares_content_when_rows_differ(default_api):
    """Duplicate docs at different rows of one index are equal (reference postings.py:463-464 compares
    term_mat rows and doc_lens, not row numbers)."""
    from searcharray_amd import SearchArray
    arr = SearchArray.index(["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 3)
    assert (arr[0:4] == arr[4:8]).all()
    assert np.array_equal(arr.take([0, 4]) == arr.take([4, 0]), [True, True])
    assert np.array_equal(arr[0:4] == arr[1:5], [False, False, False, False])
    assert np.array_equal(arr[[0, 1, 2]] == arr[[4, 2, 6]], [True, False, True])


def test_eq_ignores_the_key_order_of_dict_postings(default_api):
    """an index built from Terms / dict postings keeps a doc's term ids in insertion order: equal content with permuted
    keys is equal (the reference compares term_mat rows, postings.py:463-464: order-insensitive)"""
    a = SearchArray([Terms({"a": 1, "b": 2}, doc_len=3), Terms({"b": 2, "a": 1}, doc_len=3), Terms({"a": 1, "c": 2}, doc_len=3)])
    assert np.array_equal(a == a.take([1, 0, 2]), [True, True, True])
    assert np.array_equal(a == a.take([2, 2, 0]), [False, False, False])
    assert bool(a[0] == a[1])


def test_nbytes_does_not_materialise_device_built_words(default_api):
    from searcharray_amd import SearchArray
    arr = SearchArray.index(["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25)
    arr.score("bar")                                        # index built on the device from the token stream
    host = arr._core.host
    had = host.has_words
    n = arr.nbytes
    assert n > 0 and arr.posns.nbytes > 0
    assert host.has_words == had                            # asking for sizes downloaded nothing
    assert arr.posns.nbytes == host.words.nbytes            # and the size was right


def test_score_device_keeps_the_result_in_hbm(data):
    """``score_device``: the same scores as ``score()`` (reference postings.py:652-680), left in a device vector --
    single terms, phrases, slop, unknown terms; slices and other similarities are refused"""
    for token, slop in (("bar", 0), (["foo", "bar"], 0), (["foo", "baz"], 2), ("not_present", 0), (["bunny", "nope"], 0)):
        vec = data.score_device(token, slop=slop)
        try:
            assert np.array_equal(vec.fetch(), data.score(token, slop=slop)), (token, slop)
        finally:
            vec.close()
    with pytest.raises(ValueError):
        data[:10].score_device("bar")
    with pytest.raises(ValueError):
        data.score_device("bar", similarity=classic_similarity())


@pytest.mark.parametrize("batch_size,workers", [(1, 1), (3, 1), (3, 4), (7, 2), (100000, 4)])
def test_index_in_batches_with_workers_is_the_same_index(default_api, batch_size, workers):
    """SearchArray.index(batch_size=, workers=) (reference indexing.py:235-296: batches tokenised on a thread pool): whatever
    the batch size and the number of workers, the index is the one a single pass builds -- same term ids (assigned in document
    order, not in thread order), same words, same scores -- and over-long docs still raise / truncate per batch"""
    rng = np.random.default_rng(5)
    vocab = [f"w{i}" for i in range(40)]
    docs = [" ".join(rng.choice(vocab, int(rng.integers(0, 12)))) for _ in range(50)]
    one = SearchArray.index(docs, batch_size=100000, workers=1)
    arr = SearchArray.index(docs, batch_size=batch_size, workers=workers)
    assert len(arr) == len(one) == 50
    assert [arr.term_dict.get_term(i) for i in range(len(arr.term_dict))] == [one.term_dict.get_term(i) for i in range(len(one.term_dict))]
    assert np.array_equal(arr.doclengths(), one.doclengths())
    for term in ("w0", "w7", "w39"):
        assert np.array_equal(arr.score(term), one.score(term))
        assert np.array_equal(arr.termfreqs(term), one.termfreqs(term))
    assert np.array_equal(arr.score(["w1", "w2"]), one.score(["w1", "w2"]))
    assert np.all(arr == one)
    with pytest.raises(ValueError):
        SearchArray.index(docs, batch_size=0)
    # the length check happens batch by batch, whatever thread tokenised the doc
    from searcharray_amd import roaringish as rz
    long_doc = " ".join(["w1"] * (rz.MAX_POSN + 5))
    with pytest.raises(ValueError):
        SearchArray.index(docs[:5] + [long_doc] + docs[5:9], batch_size=2, workers=3)
    cut = SearchArray.index(docs[:5] + [long_doc], batch_size=2, workers=3, truncate=True)
    assert cut.doclengths()[5] == rz.MAX_POSN


def test_threaded_dense_calls_on_one_handle(api):
    """the reference's callers score from thread pools (test/test_tmdb.py:285-312; its native kernels release the GIL): dense
    calls of several threads on ONE index handle enqueue on separate lanes (csrc/sa_index.hpp, DenseLane) and wait outside the
    index lock -- every thread must still get exactly its own result, single terms, disjunctions and row subsets mixed with
    phrase calls (which keep the exclusive path)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import refimpl as O
    from searcharray_amd import synth, roaringish as rz
    from searcharray_amd.device_index import DeviceIndex
    n_docs, vocab = 5000, 300
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 14, seed=41)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
    rng = np.random.default_rng(8)
    rows = np.sort(rng.choice(n_docs, 200, replace=False)).astype(np.uint64)
    jobs = []
    for i in range(96):
        kind = i % 4
        if kind == 0:
            term = int(rng.integers(0, vocab))
            jobs.append(("tf", term, orc.termfreqs(term)))
        elif kind == 1:
            q = [int(x) for x in rng.integers(0, vocab, 3)]
            jobs.append(("bm25", q, orc.score_terms_sum(q)))
        elif kind == 2:
            q = [int(x) for x in rng.integers(0, 40, 2)]
            jobs.append(("bm25_rows", q, orc.score_terms_sum(q)[rows.astype(np.int64)]))
        else:
            ph = [int(x) for x in rng.choice(12, 2, replace=False)]
            jobs.append(("phrase", ph, orc.phrase_freqs(ph)))

    def run(job):
        kind, arg, want = job
        if kind == "tf":
            got = dev.termfreqs_dense(arg)
        elif kind == "bm25":
            got = dev.bm25_dense(arg)
        elif kind == "bm25_rows":
            got = dev.bm25_dense(arg, rows=rows)
        else:
            got = dev.phrase_freqs_dense(arg)
        return bool(np.array_equal(np.array(got), want))
    with ThreadPoolExecutor(8) as ex:
        ok = list(ex.map(run, jobs * 2))
    assert all(ok), f"{ok.count(False)} of {len(ok)} threaded calls returned another call's result"
    dev.close()

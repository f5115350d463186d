"""On-disk index format <-> HBM: the reference's raw uint64 ``.dat`` + ArrayDict
metadata (phrase/memmap_arrays.py) streamed to the device, and written back from it.
The golden file was produced by the reference's own ``SearchArray.index(..., data_dir=...)``
(tests/golden/make_golden.py, ONLY=memmap)."""
import os
import pickle

import numpy as np
import pytest

from searcharray_amd import SearchArray, roaringish as rz
from searcharray_amd._lib import SearchArrayHipError
from searcharray_amd.device_index import DeviceIndex
from tests.helpers import golden_corpus, load_golden, set_opt, unset_opt


def _golden_file(tmp_path):
    g = load_golden("memmap")
    path = str(tmp_path / "0.dat")
    g["dat"].tofile(path)
    md = {int(i): {"offset": int(o), "length": int(n)} for i, o, n in zip(g["ids"], g["offsets"], g["lengths"])}
    return g, path, md


def test_reference_file_scores_like_the_reference(default_api, tmp_path):
    g, path, md = _golden_file(tmp_path)
    arr = SearchArray.from_memmap(path, md, [str(t) for t in g["terms"]], g["doc_lens"])
    assert len(arr) == 400 and arr.corpus_size == 400
    for i, q in enumerate(g["queries"]):
        np.testing.assert_allclose(arr.score(str(q)), g[f"score_{i}"], rtol=1e-5, atol=0)
    for i, q in enumerate(g["phrases"]):
        np.testing.assert_allclose(arr.score(str(q).split("|")), g[f"phrase_{i}"], rtol=1e-5, atol=0)
    # the host view of the same file: docs reconstruct token for token
    docs = [str(d) for d in g["docs"]]
    for d in (0, 7, 399):
        t = arr[d]
        want = docs[d].split()
        assert t.doc_len == len(want)
        for term in set(want):
            assert list(t.positions(term)) == [i for i, w in enumerate(want) if w == term]


def test_file_in_any_term_order_and_with_gaps(api, tmp_path):
    """ArrayDict metadata may place terms anywhere in the file (after concat / __setitem__ the offsets
    are not in id order, memmap_arrays.py:56-110); the device layout is id order regardless."""
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_sparse")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab).astype(np.int64)
    rng = np.random.default_rng(5)
    order = rng.permutation(vocab)
    chunks, md, cur = [], {}, 0
    for k in order:
        w = words[off[k]:off[k + 1]]
        if len(w) == 0 and rng.random() < 0.5:
            continue                                         # absent key == empty term
        pad = int(rng.integers(0, 4))                        # garbage between the arrays
        chunks.append(np.full(pad, 0xDEADBEEF, dtype=np.uint64))
        cur += pad
        md[int(k)] = {"offset": cur, "length": len(w)}
        chunks.append(w)
        cur += len(w)
    path = str(tmp_path / "shuffled.dat")
    np.concatenate(chunks).tofile(path)
    dev = DeviceIndex.from_file(path, md, lens, n_terms=vocab, tile_docs=1024, api=api)
    back, back_off = dev.words()
    assert np.array_equal(back, words) and np.array_equal(back_off.astype(np.int64), off)
    assert np.array_equal(dev.docfreqs(), g["df"])
    for row, want in zip(g["or_queries"][:3], g["or_scores"][:3]):
        assert np.array_equal(dev.bm25_dense([int(x) for x in row]), want)
    dev.close()


@pytest.mark.parametrize("piece", ["64", "4096", "100000"])
def test_ring_wraps_with_many_pieces(api, tmp_path, monkeypatch, piece):
    """Tiny staging pieces (SA_IO_PIECE_BYTES) push hundreds of pieces through the 12-slot ring and its
    4 file threads, both directions."""
    set_opt("SA_IO_PIECE_BYTES", piece)
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    if piece == "64":                                            # keep the piece count reasonable
        words, off = words[:int(off[3])], np.concatenate([off[:4], np.full(vocab - 3, off[3], np.uint64)])
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    path = str(tmp_path / "ring.dat")
    dev.save(path)
    assert np.array_equal(np.fromfile(path, dtype=np.uint64), words)
    dev2 = DeviceIndex.from_file(path, off, lens, tile_docs=1024, api=api)
    back, back_off = dev2.words()
    assert np.array_equal(back, words) and np.array_equal(back_off, off)
    # terms in reverse file order: every term is its own run of pieces
    rev = np.concatenate([words[int(off[k]):int(off[k + 1])] for k in range(vocab - 1, -1, -1)])
    rpath = str(tmp_path / "rev.dat")
    rev.tofile(rpath)
    lens_t = np.diff(off.astype(np.int64))
    starts = np.concatenate([[0], np.cumsum(lens_t[::-1])])[:-1][::-1]
    md = {k: {"offset": int(starts[k]), "length": int(lens_t[k])} for k in range(vocab)}
    dev3 = DeviceIndex.from_file(rpath, md, lens, n_terms=vocab, tile_docs=1024, api=api)
    assert np.array_equal(dev3.words()[0], words)


def test_save_round_trip_and_empty_index(api, tmp_path):
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    off = rz.term_offsets(wt, vocab)
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    path = str(tmp_path / "saved.dat")
    got_off = dev.save(path)
    assert np.array_equal(got_off, off)
    assert np.array_equal(np.fromfile(path, dtype=np.uint64), words)     # == ArrayDict.data.tofile
    dev2 = DeviceIndex.from_file(path, got_off, lens, tile_docs=1024, api=api)
    assert np.array_equal(dev2.docfreqs(), dev.docfreqs())
    q = [int(x) for x in g["or_queries"][0]]
    assert np.array_equal(dev2.bm25_dense(q), dev.bm25_dense(q))
    # an index without words
    empty = DeviceIndex(np.empty(0, np.uint64), np.zeros(4, np.uint64), np.ones(5, np.float32), api=api)
    epath = str(tmp_path / "empty.dat")
    empty.save(epath)
    assert os.path.getsize(epath) == 0
    e2 = DeviceIndex.from_file(epath, np.zeros(4, np.uint64), np.ones(5, np.float32), api=api)
    assert e2.docfreqs().tolist() == [0, 0, 0]


def test_file_errors(api, tmp_path):
    lens = np.ones(4, np.float32)
    with pytest.raises(SearchArrayHipError, match="cannot open"):
        DeviceIndex.from_file(str(tmp_path / "missing.dat"), {0: {"offset": 0, "length": 1}}, lens, api=api)
    path = str(tmp_path / "short.dat")
    np.arange(10, dtype=np.uint64).tofile(path)
    with pytest.raises(SearchArrayHipError, match="lies outside"):
        DeviceIndex.from_file(path, {0: {"offset": 4, "length": 7}}, lens, api=api)
    with pytest.raises(SearchArrayHipError, match="cannot create"):
        DeviceIndex(np.empty(0, np.uint64), np.zeros(2, np.uint64), lens, api=api).save(str(tmp_path / "no" / "x.dat"))


def test_index_with_data_dir_pickles_the_filename_not_the_words(default_api, tmp_path):
    docs = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25
    plain = SearchArray.index(docs)
    ddir = tmp_path / "idx"
    ddir.mkdir()
    arr = SearchArray.index(docs, data_dir=str(ddir))
    assert sorted(os.listdir(ddir)) == ["0.dat"]                          # reference naming: <n files>.dat
    assert np.array_equal(np.fromfile(ddir / "0.dat", dtype=np.uint64), plain._core.host.words)
    blob = pickle.dumps(arr)
    assert len(blob) < len(pickle.dumps(plain))
    assert plain._core.host.words.tobytes()[:64] not in blob
    back = pickle.loads(blob)
    assert not back._core.host.has_words
    np.testing.assert_array_equal(back.score("bar"), plain.score("bar"))
    np.testing.assert_array_equal(back.score(["bar", "baz"]), plain.score(["bar", "baz"]))
    assert not back._core.host.has_words                                  # scoring never pulled the words to the host
    assert back[0] == plain[0] and back[3] == plain[3]                    # ... reconstructing a doc maps the file
    # a second index in the same directory takes the next name
    SearchArray.index(["x y", "y z"], data_dir=str(ddir))
    assert sorted(os.listdir(ddir)) == ["0.dat", "1.dat"]

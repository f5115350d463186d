"""Randomised differential tests: small random corpora and queries, every result against the oracle.
A few iterations per seed keep the suite fast; scripts in the development history ran hundreds."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import roaringish as rz
from searcharray_amd import synth
from searcharray_amd.device_index import DeviceIndex, NO_DOC
from tests.helpers import set_opt, unset_opt


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_bm25_batches_pruned_and_exhaustive(api, seed, monkeypatch):
    rng = np.random.default_rng(seed)
    for _ in range(4):
        n_docs, vocab, mean = int(rng.integers(50, 6000)), int(rng.integers(5, 600)), int(rng.integers(2, 30))
        t, d, p, lens = synth.corpus_triples(n_docs, vocab, mean, seed=int(rng.integers(1 << 30)))
        words, wt = rz.encode_sorted(t, d, p)
        set_opt("SA_TF8_DIV", str(rng.choice([0, 8, 128, 100000])))
        dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=int(rng.choice([1024, 2048, 4096])), api=api)
        orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
        T, B, k = int(rng.integers(1, 7)), int(rng.integers(1, 7)), int(rng.choice([1, 3, 10, 33, 100]))
        queries = rng.integers(0, vocab + 3, size=(B, T))                       # includes unknown ids
        queries[rng.random((B, T)) < 0.3] = vocab - 1 - rng.integers(0, max(1, vocab // 4))
        k1, b = float(rng.choice([1.2, 0.4, 2.0])), float(rng.choice([0.75, 0.0, 1.0]))
        got = {}
        for sparse in ("1", "0"):
            set_opt("SA_SPARSE", sparse)
            bt = dev.batch(queries, k=k, k1=k1, b=b)
            bt.run()
            got[sparse] = bt.fetch()
            bt.close()
        assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])
        for qi, q in enumerate(queries):
            known = [int(x) for x in q if x < vocab]
            dense = orc.score_terms_sum(known, k1=k1, b=b) if known else np.zeros(n_docs, np.float32)
            ws, wd = O.topk(dense, k)
            n = int((ws > 0).sum())
            assert np.array_equal(got["1"][0][qi, :n], ws[:n]) and np.array_equal(got["1"][1][qi, :n], wd[:n])
            assert (got["1"][1][qi, n:] == NO_DOC).all()
        dev.close()


STAGE_FUZZ_DOCS = (200, 9000)            # (scripts/fuzz_stage.py --big widens it on the GPU: enough tiles for the co-walking groups of a full device)


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_random_bm25_batches_on_the_staged_route(api, seed):
    """the staged-tile route (csrc/sa_stage.hip) forced on random corpora and query sets: any B (slices above 256), T = 1 .. 8, k = 1 .. 300,
    stage tiles of 64 .. 1024 docs, 1 / 2 / 8 workgroups per CU with co-walking groups of 1 / 2 / 4, with and without probe rows, BM25
    parameters at the edges of the range the bounds admit (b = 0, b = 1) -- top-k equal to the oracle's bit for bit, and the route
    taken is the staged one whenever the set has a known term"""
    rng = np.random.default_rng(seed)
    for _ in range(5):
        n_docs, vocab, mean = int(rng.integers(*STAGE_FUZZ_DOCS)), int(rng.integers(5, 500)), int(rng.integers(2, 24))
        t, d, p, lens = synth.corpus_triples(n_docs, vocab, mean, seed=int(rng.integers(1 << 30)))
        words, wt = rz.encode_sorted(t, d, p)
        doc_base = int(rng.choice([0, 12345]))
        dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=int(rng.choice([1024, 2048, 4096])), doc_base=doc_base, api=api)
        orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
        T, B, k = int(rng.integers(1, 9)), int(rng.choice([1, 2, 7, 40, 257, 300])), int(rng.choice([1, 3, 10, 33, 100, 300]))
        queries = rng.integers(0, vocab + 2, size=(B, T))                       # includes unknown ids
        queries[rng.random((B, T)) < 0.4] = rng.integers(0, max(1, vocab // 8))  # frequent terms, shared between queries
        k1, b = float(rng.choice([1.2, 0.4, 2.0])), float(rng.choice([0.75, 0.0, 1.0]))     # (k1 = 0 is 0 / 0 = NaN for every doc without the term in the reference's dense formula)
        opts = {"stage": 1, "stage_docs": int(rng.choice([64, 128, 256, 512, 1024])), "stage_wgs": int(rng.choice([1, 2, 8])),
                "stage_cw": int(rng.choice([1, 2, 4])), "stage_probe": int(rng.choice([0, 1])), "probe_div": int(rng.choice([4, 128]))}
        bt = dev.batch(queries, k=k, k1=k1, b=b, opts=opts)
        for _ in range(2):
            bt.run()
        scores, docs = bt.fetch()
        any_known = bool((queries < vocab).any())
        if any_known and k <= 1024:
            assert bt.last_route() in ("staged", "exhaustive"), bt.last_route()     # (exhaustive: a plan the route declines -- e.g. more than 1024 distinct terms)
        for qi, q in enumerate(queries):
            known = [int(x) for x in q if x < vocab]
            dense = orc.score_terms_sum(known, k1=k1, b=b) if known else np.zeros(n_docs, np.float32)
            ws, wd = O.topk(dense, k)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[qi, :n], ws[:n]), (seed, opts, qi)
            assert np.array_equal(docs[qi, :n], wd[:n] + np.uint64(doc_base)), (seed, opts, qi)
            assert (docs[qi, n:] == NO_DOC).all()
        bt.close()
        dev.close()


@pytest.mark.parametrize("seed", [21, 22])
def test_random_phrases_batches_and_slop(api, seed, monkeypatch):
    rng = np.random.default_rng(seed)
    for _ in range(3):
        n_docs, vocab, mean = int(rng.integers(50, 5000)), int(rng.integers(4, 60)), int(rng.integers(3, 60))
        t, d, p, lens = synth.corpus_triples(n_docs, vocab, mean, seed=int(rng.integers(1 << 30)))
        words, wt = rz.encode_sorted(t, d, p)
        set_opt("SA_DOCDIR_DIV", str(rng.choice([0, 4, 32, 100000])))
        set_opt("SA_PTILE", str(rng.choice([2048, 4096])))
        dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
        orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
        phrases = [[int(x) for x in rng.choice(vocab, int(rng.integers(2, min(8, vocab))), replace=False)]
                   for _ in range(int(rng.integers(1, 6)))]
        k = int(rng.choice([1, 5, 40]))
        bt = dev.phrase_batch(phrases, k=k)
        bt.run()
        s, dd = bt.fetch()
        bt.close()
        for i, ph in enumerate(phrases):
            ws, wd = O.topk(orc.score(list(ph)), k)
            n = int((ws > 0).sum())
            assert np.array_equal(s[i, :n], ws[:n]) and np.array_equal(dd[i, :n], wd[:n]) and (dd[i, n:] == NO_DOC).all()
            assert np.array_equal(dev.phrase_freqs_dense(ph), orc.phrase_freqs(ph))
            if len(ph) <= 4:
                slop = int(rng.integers(1, 4))
                assert np.array_equal(dev.phrase_freqs_dense(ph, slop=slop), orc.phrase_freqs(ph, slop=slop))
        dev.close()


@pytest.mark.parametrize("seed", [31, 32])
def test_random_slop_batches(api, seed, monkeypatch):
    """Batches of slop phrases of 2 - 4 terms (the doc-parallel batch launch, csrc/sa_spans.hip sa_k_span_doc_fused_multi: ranked inside
    the kernel) with random launch shapes -- phrases per bundle, waves with span tables, one launch or one per phrase -- against the
    oracle's score(): the slot list, both instances of the kernel, heavy documents, phrases without matches"""
    from tests import emu
    small = emu._api is api                                   # (the host stand-in runs a 256-thread block on fibers: smaller corpora, fewer phrases there)
    rng = np.random.default_rng(seed)
    for _ in range(1 if small else 3):
        n_docs, vocab, mean = int(rng.integers(200, 1200 if small else 6000)), int(rng.integers(4, 40)), int(rng.integers(4, 20 if small else 50))
        t, d, p, lens = synth.corpus_triples(n_docs, vocab, mean, seed=int(rng.integers(1 << 30)))
        words, wt = rz.encode_sorted(t, d, p)
        set_opt("SA_SPAN_BUNDLE", str(rng.choice([1, 2, 3, 32])))
        tw = rng.choice([0, 2, 4])
        if tw:
            set_opt("SA_SPAN_TAB_WAVES", str(tw))
        else:
            unset_opt("SA_SPAN_TAB_WAVES")
        set_opt("SA_SPAN_DOC_MULTI", str(rng.choice([1, 1, 1, 0])))
        dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
        orc = O.OracleIndex.from_triples(t, d, p, n_docs, doc_lens=lens)
        nb = int(rng.integers(1, 8 if small else 24))
        phrases = [[int(x) for x in rng.choice(vocab, int(rng.integers(2, min(5, vocab))), replace=False)] for _ in range(nb)]
        slops = [int(rng.integers(1, 4)) for _ in range(nb)]
        k = int(rng.choice([1, 5, 40]))
        bt = dev.phrase_batch(phrases, k=k, slop=slops)
        for _ in range(2):
            bt.run()
        s, dd = bt.fetch()
        bt.close()
        for i, (ph, sl) in enumerate(zip(phrases, slops)):
            ws, wd = O.topk(orc.score(list(ph), slop=sl), k)
            n = int((ws > 0).sum())
            assert np.array_equal(s[i, :n], ws[:n]) and np.array_equal(dd[i, :n], wd[:n]) and (dd[i, n:] == NO_DOC).all(), f"seed {seed}: {ph} slop {sl}"
        dev.close()

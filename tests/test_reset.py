"""Fresh query sets in an existing batch (sa_batch_reset / sa_phrase_batch_reset): what the reference does per call --
SearchArray.score on queries it has not seen (postings.py:652-680; timed as test/test_msmarco.py:345-395) -- for a
STREAM of batches.  A reset must leave the batch exactly as a newly created one would be: results equal the oracle's
dense score + deterministic top-k bit for bit on every path (grouped / per-query exhaustive, dynamic pruning), for query
sets whose grouping, lead terms and Bloom sizes all differ, with resets issued while earlier runs are still in flight
and two batches used alternately (the pipeline bench.py times)."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import roaringish as rz, synth
from searcharray_amd.device_index import DeviceIndex
from tests.helpers import set_opt, unset_opt

N_DOCS, VOCAB = 9000, 400


@pytest.fixture(scope="module")
def corpus():
    t, d, p, lens = synth.corpus_triples(N_DOCS, VOCAB, 14, seed=77)
    words, wt = rz.encode_sorted(t, d, p)
    return words, rz.term_offsets(wt, VOCAB), lens, O.OracleIndex.from_triples(t, d, p, N_DOCS, doc_lens=lens)


def query_sets(T=4, B=24):
    rng = np.random.default_rng(5)
    sets = []
    # shared frequent heads (groups), all rare (loose groups), mixed with unknown terms, all the same query, all distinct heads
    q = np.empty((B, T), dtype=np.int64)
    q[:, 0] = rng.choice([0, 1, 2], B)
    for t in range(1, T):
        q[:, t] = rng.integers(3 + 20 * t, VOCAB, B)
    sets.append(q)
    sets.append(rng.integers(200, VOCAB, (B, T)))
    q = rng.integers(0, VOCAB + 40, (B, T))           # ids >= VOCAB: unknown terms
    sets.append(q)
    sets.append(np.tile(np.asarray([[0, 5, 50, 300][:T]]), (B, 1)))
    q = rng.integers(0, VOCAB, (B, T))
    q[:, 0] = rng.permutation(VOCAB)[:B]
    sets.append(q)
    return sets


def expect(orc, q, k):
    dense = orc.score_terms_sum([int(x) for x in q if 0 <= int(x) < VOCAB])
    return O.topk(dense, k)


def assert_batch(orc, queries, scores, docs, k, what):
    for qi, q in enumerate(queries):
        ws, wd = expect(orc, q, k)
        n = int((ws > 0).sum())
        assert np.array_equal(scores[qi, :n], ws[:n]), f"{what}: q{qi} {q} scores"
        assert np.array_equal(docs[qi, :n], wd[:n]), f"{what}: q{qi} {q} docs"
        assert (scores[qi, n:] == 0).all()


@pytest.mark.parametrize("mode", [{"SA_SPARSE": "0"}, {"SA_SPARSE": "0", "SA_GROUP": "0"}, {"SA_SPARSE": "1"}])
@pytest.mark.parametrize("k", [5, 40])
def test_reset_equals_fresh_batch_and_oracle(api, corpus, monkeypatch, mode, k):
    for name, v in mode.items():
        set_opt(name, v)
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    sets = query_sets()
    bt = dev.batch(sets[0], k=k)
    for i, qs in enumerate(sets + sets[:2]):
        if i:
            bt.reset(qs)
        bt.run(sync=False)
        scores, docs = bt.fetch()
        assert_batch(orc, qs, scores, docs, k, f"{mode} reset {i}")
        fresh = dev.batch(qs, k=k)
        fresh.run()
        fs, fd = fresh.fetch()
        assert np.array_equal(fs, scores) and np.array_equal(fd, docs), f"{mode} reset {i}: differs from a new batch"
        assert fresh.group_info() == bt.group_info()
        fresh.close()
    bt.close()
    dev.close()


def test_bloom_buffer_grows_with_the_query_sets(api, corpus, monkeypatch):
    """dynamic pruning: the lead terms' Bloom filters live in a buffer sized from the query sets seen so far
    (sa_batch_ensure_bloom) -- sets whose lead terms get longer and longer make it grow between runs, with a run in flight"""
    set_opt("SA_SPARSE", "1")
    set_opt("SA_BLOOM_FLOOR", "1024")
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    rng = np.random.default_rng(9)
    B, k = 24, 7
    sets = [rng.integers(lo, hi, (B, 3)) for lo, hi in [(380, 400), (200, 260), (60, 90), (12, 30), (0, 6), (300, 400)]]
    bt = dev.batch(sets[0], k=k)
    for i, qs in enumerate(sets):
        if i:
            bt.reset(qs)
        bt.run(sync=False)
        bt.run(sync=False)
        scores, docs = bt.fetch()
        assert_batch(orc, qs, scores, docs, k, f"bloom growth, set {i}")
    bt.close()
    dev.close()


def test_two_batches_alternating_without_waiting(api, corpus, monkeypatch):
    """the pipeline of bench.py's fresh_batches leg: reset + run of batch i+1 are enqueued before batch i's results are
    fetched; every fetch waits for its own batch only"""
    set_opt("SA_SPARSE", "0")
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    sets = query_sets()
    k = 10
    pair = [dev.batch(sets[0], k=k), dev.batch(sets[1], k=k)]
    in_flight = [None, None]
    n_steps = 3 * len(sets)
    for step in range(n_steps + 2):
        b = step & 1
        if in_flight[b] is not None:
            scores, docs = pair[b].fetch()
            assert_batch(orc, sets[in_flight[b]], scores, docs, k, f"step {step - 2}")
            in_flight[b] = None
        if step < n_steps:
            si = (step * 3 + 1) % len(sets)
            pair[b].reset(sets[si])
            pair[b].run(sync=False)
            in_flight[b] = si
    for b in pair:
        b.close()
    dev.close()


def test_reset_with_explicit_weights_and_bad_shapes(api, corpus, monkeypatch):
    set_opt("SA_SPARSE", "0")
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    sets = query_sets()
    bt = dev.batch(sets[0], k=7)
    with pytest.raises(ValueError):
        bt.reset(sets[0][:, :3])
    # explicit idf weights (a caller with its own statistics, e.g. global df of a sharded corpus)
    idf = (np.abs(np.sin(np.arange(sets[1].size, dtype=np.float64))) + 0.25).astype(np.float32).reshape(sets[1].shape)
    bt.reset(sets[1], idf=idf)
    bt.run()
    s1, d1 = bt.fetch()
    fresh = dev.batch(sets[1], k=7, idf=idf)
    fresh.run()
    s2, d2 = fresh.fetch()
    assert np.array_equal(s1, s2) and np.array_equal(d1, d2)
    fresh.close()
    bt.close()
    dev.close()


def test_phrase_batch_reset(api, corpus):
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, api=api)
    rng = np.random.default_rng(11)
    k = 6

    def phrases_of(seed, with_dense):
        r = np.random.default_rng(seed)
        ph = [list(map(int, r.integers(0, 12, r.integers(2, 4)))) for _ in range(10)]
        slops = [0] * 10
        if with_dense:
            ph[3] = [0, 0]                      # repeated term: dense route
            ph[7] = [1, 2]
            slops[7] = 2                        # slop: dense route
        else:
            ph = [p if len(set(p)) == len(p) else [0, 1, 2][:len(p)] for p in ph]
        ph[9] = [2, VOCAB + 5]                  # unknown term: matches nothing
        return ph, slops

    first, sl0 = phrases_of(1, False)
    pb = dev.phrase_batch(first, k=k, slop=sl0)
    for i, (seed, dense) in enumerate([(1, False), (2, True), (3, False), (4, True), (1, False)]):
        ph, slops = phrases_of(seed, dense)
        if i:
            pb.reset(ph, slop=slops)
        pb.run(sync=False)
        scores, docs = pb.fetch()
        for j, (p, s) in enumerate(zip(ph, slops)):
            ws, wd = O.topk(orc.score(list(p), slop=s), k)
            n = int((ws > 0).sum())
            assert np.array_equal(scores[j, :n], ws[:n]) and np.array_equal(docs[j, :n], wd[:n]), f"set {i} phrase {p} slop {s}"
    with pytest.raises(ValueError):
        pb.reset(first[:5])
    pb.close()
    dev.close()
    del rng


def test_pruning_tables_on_demand(api, corpus, monkeypatch):
    """a batch whose queries are grouped is reset WITHOUT the pruning tables (the run scores every posting); SA_SPARSE switched
    on between reset and run -- what bench.py's pruning leg does with a resident batch -- derives them at that run and uploads
    the image again; back to the exhaustive path the starting bounds are still there (the slice-table kernel forms them again)"""
    words, off, lens, orc = corpus
    dev = DeviceIndex(words, off, lens, tile_docs=1024, api=api)
    sets = query_sets()
    unset_opt("SA_SPARSE")
    bt = dev.batch(sets[0], k=9)                         # shared heads: grouped by default, no tables
    for i, qs in enumerate([sets[0], sets[3], sets[0]]):
        if i:
            unset_opt("SA_SPARSE")
            bt.reset(qs)
        for mode in (None, "1", "0", None):
            if mode is None:
                unset_opt("SA_SPARSE")
            else:
                set_opt("SA_SPARSE", mode)
            bt.run(sync=False)
            scores, docs = bt.fetch()
            assert_batch(orc, qs, scores, docs, 9, f"set {i} SA_SPARSE={mode}")
        if i == 0:
            assert bt.seeds().max() > 0
    bt.close()
    dev.close()


def test_query_set_queue_with_a_worker_thread(api):
    """sa_queue_* (csrc/sa_queue.hip): a stream of query sets through a ring of batches stepped by a worker thread of the library -- every
    ticket's results equal a batch created for that set (and through it the oracle: tests above), in order, with more sets than the ring
    is deep, with the caller fetching late (submit blocks until a slot is free) and from a second caller thread; a ticket cannot be
    fetched twice; a failing step (no idf table) surfaces at fetch with the worker's error text."""
    import threading
    n_docs, vocab = 9000, 400
    t, d, p, lens = synth.corpus_triples(n_docs, vocab, 14, seed=31)
    words, wt = rz.encode_sorted(t, d, p)
    dev = DeviceIndex(words, rz.term_offsets(wt, vocab), lens, tile_docs=1024, api=api)
    rng = np.random.default_rng(77)
    sets = [np.stack([rng.integers(0, 6, 24), rng.integers(3, vocab, 24), rng.integers(100, vocab + 2, 24)], axis=1) for _ in range(11)]
    want = []
    for s_ in sets:
        bt = dev.batch(s_, k=7)
        bt.run()
        want.append(bt.fetch())
        bt.close()
    qq = dev.queue(24, 3, k=7, depth=3)
    tk = qq.submit(sets[0])
    with pytest.raises(Exception, match="set_idf_table"):
        qq.fetch(tk)                                             # (sa_batch_step needs sa_index_set_idf_table)
    df = dev.docfreqs().astype(np.float64)
    dev.set_idf_table(np.log(1 + (n_docs - df + 0.5) / (df + 0.5)).astype(np.float32))
    tickets = []
    got = {}
    for i, s_ in enumerate(sets):                                # 11 sets through a ring of 3: fetch when the ring is full
        if len(tickets) == 3:
            t0 = tickets.pop(0)
            got[t0[1]] = qq.fetch(t0[0])
        tickets.append((qq.submit(s_), i))
    for tk, i in tickets:
        got[i] = qq.fetch(tk)
    for i in range(len(sets)):
        assert np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1]), i
    with pytest.raises(Exception, match="fetched already|never handed out"):
        qq.fetch(tickets[-1][0])
    # a producer thread submits, the main thread fetches
    order = []

    def produce():
        for i in (4, 2, 9, 0, 7, 5):
            order.append((qq.submit(sets[i]), i))
    th = threading.Thread(target=produce)
    th.start()
    done = 0
    while done < 6:
        if len(order) > done:
            tk, i = order[done]
            r = qq.fetch(tk)
            assert np.array_equal(r[0], want[i][0]) and np.array_equal(r[1], want[i][1]), i
            done += 1
    th.join()
    assert qq.last_route(0) in ("staged", "exhaustive", "pruned")
    qq.close()
    dev.close()

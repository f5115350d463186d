"""N devices behind one handle (searcharray_amd/sharded.py): one process, one host thread per shard,
global BM25 statistics, all-gather of the per-shard top-k + merge on every shard.  On the CPU suite the
"devices" are shards of the host-emulated library and the collective is the test build's in-process
communicator (tests/hipemu/sa_comm_stub.cpp); with backend "gpu" the same code runs over RCCL on however
many GPUs the box has (one on the round-end box: the N = 1 path, no communicator).

Sharded results must be BIT-identical to the single-index oracle: avgdl is np.mean over the float32
lengths of the whole corpus (reference indexing.py:282-284), df is summed over the shards."""
import os

import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import roaringish as rz, synth
from searcharray_amd.sharded import ShardedIndex, split_by_doc_range
from tests.helpers import set_opt, unset_opt

N_DOCS, VOCAB, K = 5000, 250, 10
QUERIES = np.asarray([[0, 5, 50, 200], [1, 2, 3, 4], [7, 90, 150, 249], [10, 11, 12, 13], [249, 248, 400, 3]])
PHRASES = [[0, 1], [2, 1, 0], [5, 3], [1, 0, 4, 2]]


@pytest.fixture(scope="module")
def corpus():
    t, d, p, lens = synth.corpus_triples(N_DOCS, VOCAB, 12, seed=23)
    words, wt = rz.encode_sorted(t, d, p)
    return words, rz.term_offsets(wt, VOCAB), lens, O.OracleIndex.from_triples(t, d, p, N_DOCS, doc_lens=lens)


def n_devices(api, want):
    if api.path.endswith("libsearcharray_emu.so"):
        return want
    import ctypes
    n = ctypes.c_int(0)
    api.call("sa_device_count", ctypes.byref(n))
    return max(1, min(want, n.value))


def test_split_by_doc_range_is_a_partition(corpus):
    words, off, lens, _ = corpus
    bounds = [0, 1234, 1234, 4000, N_DOCS]                       # incl. an empty shard
    parts = split_by_doc_range(words, off, bounds)
    assert sum(len(w) for w, _ in parts) == len(words)
    for t in (0, 3, 77, VOCAB - 1):
        back = np.concatenate([w[int(o[t]):int(o[t + 1])] + (np.uint64(bounds[g]) << np.uint64(36))
                               for g, (w, o) in enumerate(parts)])
        assert np.array_equal(back, words[int(off[t]):int(off[t + 1])])
    assert len(parts[1][0]) == 0


@pytest.mark.parametrize("shards", [1, 2, 3])
def test_sharded_topk_is_bit_identical_to_the_single_index_oracle(api, corpus, shards):
    words, off, lens, orc = corpus
    G = n_devices(api, shards)
    ix = ShardedIndex(words, off, lens, devices=list(range(G)), tile_docs=1024, api=api)
    try:
        assert ix.avg_doc_len == np.float32(np.mean(lens))
        assert np.array_equal(ix._df, np.asarray([orc.docfreq(t) for t in range(VOCAB)], dtype=np.uint64))
        bt = ix.batch(QUERIES, k=K)
        for _ in range(2):                                       # double-buffered exchange: run twice
            bt.run()
        scores, docs = bt.fetch()
        for qi, q in enumerate(QUERIES):
            ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), K)
            assert np.array_equal(scores[qi], ws), f"q{qi}"
            assert np.array_equal(docs[qi][ws > 0], wd[ws > 0]), f"q{qi}"
        # a fresh query set in the same batch (sa_sharded_batch_reset on every shard), still bit-identical
        fresh = QUERIES[::-1].copy()
        bt.reset(fresh)
        bt.run(sync=False)
        scores, docs = bt.fetch()
        for qi, q in enumerate(fresh):
            ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q if 0 <= int(x) < VOCAB]), K)
            assert np.array_equal(scores[qi], ws) and np.array_equal(docs[qi][ws > 0], wd[ws > 0]), f"reset q{qi}"
        bt.close()
        pb = ix.phrase_batch(PHRASES, k=K)
        pb.run()
        ps, pd_ = pb.fetch()
        for pi, ph in enumerate(PHRASES):
            ws, wd = O.topk(orc.score(list(ph)), K)
            assert np.array_equal(ps[pi], ws) and np.array_equal(pd_[pi][ws > 0], wd[ws > 0]), f"phrase {ph}"
        pb.close()
        # dense drop-in results: per-shard vectors concatenated, no collective
        assert np.array_equal(ix.bm25_dense([0, 5, 50]), orc.score_terms_sum([0, 5, 50]))
        assert np.array_equal(ix.phrase_freqs_dense([2, 1, 0]), orc.phrase_freqs([2, 1, 0]))
        assert np.array_equal(ix.bm25_phrase_dense([0, 1]), orc.score([0, 1]))
    finally:
        ix.close()


def test_sharded_batch_follows_options_set_after_its_creation(api, corpus):
    """sa_sharded_batch_set_options fans out to the shards' batches: a switch set on an existing sharded batch -- set_options, or the
    thread's scoped options -- applies to its next run (it used to be ignored silently).  cand_cap is read when a batch is
    CREATED, so the observable switch here is the route: every route returns the same top-k, and the batch reports what it was given."""
    words, off, lens, orc = corpus
    ix = ShardedIndex(words, off, lens, devices=list(range(n_devices(api, 2))), tile_docs=1024, api=api)
    try:
        bt = ix.batch(QUERIES, k=K)
        want = None
        for kw in ({}, {"stage": 0}, {"stage": 1}, {"stage": 0, "sparse": 1}, {"stage": None, "sparse": None}):
            if kw:
                bt.set_options(**kw)
                for name, v in kw.items():
                    assert bt.options().get(name) == v
            bt.run()
            got = bt.fetch()
            if want is None:
                want = got
                for qi, q in enumerate(QUERIES):
                    ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), K)
                    assert np.array_equal(got[0][qi], ws) and np.array_equal(got[1][qi][ws > 0], wd[ws > 0])
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), kw
        set_opt("group", 0)                                      # (the thread's scope reaches the existing batch too)
        bt.run()
        got = bt.fetch()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert bt.options().get("group") == 0
        bt.close()
    finally:
        ix.close()


def test_search_array_search_over_devices(default_api):
    from searcharray_amd import SearchArray
    docs = ["foo bar bar baz", "data2", "data3 bar", "bunny funny wunny"] * 25
    arr = SearchArray.index(docs)
    G = n_devices(default_api, 2)
    one = arr.search([["bar"], ["foo", "baz"], ["nope"]], k=5)
    many = arr.search([["bar"], ["foo", "baz"], ["nope"]], k=5, devices=list(range(G)))
    assert np.array_equal(one[0], many[0]) and np.array_equal(one[1], many[1])
    assert np.isclose(one[0][0][0], 0.37066694)                  # reference test_search.py:121-124
    p1 = arr.search_phrases([["bar", "bar"], ["foo", "bar"]], k=3)
    p2 = arr.search_phrases([["bar", "bar"], ["foo", "bar"]], k=3, devices=list(range(G)))
    assert np.array_equal(p1[0], p2[0]) and np.array_equal(p1[1], p2[1])


@pytest.mark.parametrize("shards,k", [(2, 5), (3, 700)])
def test_sharded_candidate_overflow_and_wide_merge(api, shards, k, monkeypatch):
    """Every doc identical -> every score ties -> the bound cuts nothing and a 100-key candidate list runs over on every
    shard: the flag travels with the all-gather (one extra cell per rank), every shard sees it at fetch and all of them
    redo the batch unpruned.  k = 700 on 3 shards: 2100 gathered keys per query do not fit the merge's LDS list, so the
    exchange takes the regroup route instead of the fused one."""
    set_opt("SA_CAND_CAP", "100")
    n = 4000
    t = np.repeat(np.arange(3), n).astype(np.uint32)
    d = np.tile(np.arange(n), 3).astype(np.uint64)
    p = np.repeat(np.arange(3), n).astype(np.uint64)
    words, wt = rz.encode_sorted(t, d, p)
    G = n_devices(api, shards)
    ix = ShardedIndex(words, rz.term_offsets(wt, 3), np.full(n, 3, np.float32), devices=list(range(G)), tile_docs=1024, api=api)
    try:
        bt = ix.batch(np.asarray([[0, 1, 2], [2, 1, 0]]), k=k)
        for _ in range(2):
            bt.run()
        scores, docs = bt.fetch()
        for qi in range(2):
            assert np.array_equal(docs[qi], np.arange(k, dtype=np.uint64))       # ties -> smallest doc ids, over all shards
            assert (scores[qi] == scores[qi, 0]).all() and scores[qi, 0] > 0
        bt.run()                                                                 # (and the state is clean again)
        s2, d2 = bt.fetch()
        assert np.array_equal(s2, scores) and np.array_equal(d2, docs)
        bt.close()
    finally:
        ix.close()


def test_a_batch_that_outlives_its_sharded_handle_is_refused_not_freed_twice(api, corpus):
    """sa_sharded_destroy takes the per-shard parts of the batches still alive down with the shards and orphans them: later
    calls on such a batch fail with an error, its close only frees the shell (it used to touch joined worker threads and
    destroyed indexes)"""
    from searcharray_amd._lib import SearchArrayHipError
    words, off, lens, _ = corpus
    ix = ShardedIndex(words, off, lens, devices=list(range(n_devices(api, 2))), tile_docs=1024, api=api)
    bt = ix.batch(QUERIES, k=K)
    bt.run()
    want = bt.fetch()
    bt2 = ix.batch(QUERIES, k=K)
    bt2.close()                                                      # (a batch closed before the index leaves the registry)
    ix.close()
    with pytest.raises(SearchArrayHipError):
        bt.run()
    with pytest.raises(SearchArrayHipError):
        bt.fetch()
    bt.close()
    bt.close()
    assert want[0].shape == (len(QUERIES), K)


@pytest.mark.gpu
def test_collectives_come_from_the_rccl_the_library_was_linked_against():
    """libsearcharray_hip.so links /opt/rocm's RCCL (DT_RPATH); a process that has imported torch first resolves the same
    SONAME to the copy inside the torch wheel instead (round 4's GPU log showed exactly that).  Nothing the GPU suite collects
    imports torch any more (tests/test_dist_gloo.py imports it inside its test): the collectives of this process must be
    /opt/rocm's."""
    import sys
    from searcharray_amd import _lib
    from searcharray_amd.device_index import DeviceIndex
    assert "torch" not in sys.modules, "a collected test module imported torch at import time"
    version, path = DeviceIndex.comm_library_info(_lib.api())
    assert version > 0
    assert os.path.realpath(path).startswith("/opt/rocm"), f"collectives resolved to {path}"


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_over_rccl(tmp_path):
    """The N > 1 path on real hardware with the ONE GPU a test box has: two processes, both on device 0, each with one doc-range
    shard, exchange their per-shard top-k over libsearcharray_hip.so's RCCL communicator (ncclAllGather + cross-rank merge,
    csrc/sa_comm.hip, sa_k_topk_merge over the gathered keys) -- the first time a second rank's keys travel through it.  Both
    ranks must end with the single-index oracle's top-k.  If RCCL refuses two ranks on one device the test reports RCCL's
    message as an expected failure (that is evidence too); any other failure fails."""
    import subprocess
    import sys
    from tests import two_rank_worker as W
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_worker.py")
    id_file = str(tmp_path / "nccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", id_file, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank run timed out")
        outs.append(out.decode(errors="replace"))
    errs = [open(tmp_path / f"rank{r}.err").read() for r in range(2) if os.path.exists(tmp_path / f"rank{r}.err")]
    if errs:
        pytest.xfail("RCCL refused two ranks on one device: " + errs[0][:300])
    assert all(pr.returncode == 0 for pr in procs), "\n".join(o[-2000:] for o in outs)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["scores"], r1["scores"]) and np.array_equal(r0["docs"], r1["docs"])       # every rank merges
    assert np.array_equal(r0["scores"], r0["scores0"]) and np.array_equal(r0["docs"], r0["docs0"])     # every step the same
    assert int(r0["groups"]) >= 1
    t, d, p, lens = W.corpus()
    orc = O.OracleIndex.from_triples(t, d, p, W.N_DOCS, doc_lens=lens)
    for qi, q in enumerate(W.queries()):
        ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), W.K)
        n = int((ws > 0).sum())
        assert np.array_equal(r0["scores"][qi, :n], ws[:n]), f"q{qi}"
        assert np.array_equal(r0["docs"][qi, :n], wd[:n]), f"q{qi}"
    assert os.path.realpath(str(r0["lib"])).startswith("/opt/rocm")


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_cross_rank_merge_on_the_device(tmp_path):
    """What CAN run with one GPU: two processes on device 0, one doc-range shard each, through the external-collective route of
    the ABI (sa_batch_run_local -> exchange -> sa_batch_merge_gathered; the exchange itself goes through files, RCCL having
    refused -- test above).  Global df summed over the ranks, shard-local scoring with global statistics, and the cross-rank
    merge kernels fed BOTH ranks' keys run on the real device; both ranks end with the single-index oracle's top-k."""
    import subprocess
    import sys
    from tests import two_rank_worker as W
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path / "unused"), str(tmp_path), "files"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank run timed out")
        outs.append(out.decode(errors="replace"))
    assert all(pr.returncode == 0 for pr in procs), "\n".join(o[-2000:] for o in outs)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["scores"], r1["scores"]) and np.array_equal(r0["docs"], r1["docs"])       # every rank merges
    assert (r0["keys_per_rank"] > 0).all(), "the merge must have seen keys of both ranks"
    t, d, p, lens = W.corpus()
    orc = O.OracleIndex.from_triples(t, d, p, W.N_DOCS, doc_lens=lens)
    from_second = 0
    for qi, q in enumerate(W.queries()):
        ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), W.K)
        n = int((ws > 0).sum())
        assert np.array_equal(r0["scores"][qi, :n], ws[:n]), f"q{qi}"
        assert np.array_equal(r0["docs"][qi, :n], wd[:n]), f"q{qi}"
        from_second += int((wd[:n] >= W.N_DOCS // 2).sum())
    assert from_second > 0, "no result doc from the second rank's shard: the test would prove nothing"

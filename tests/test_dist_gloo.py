"""Doc-range sharding across ranks (N > 1 path) on CPU: world_size-2 `gloo` processes, each
holding one shard of a small corpus in the host-emulated kernel library, exchange their per-shard
top-k keys with all_gather and merge -- the same entry points (sa_batch_run_local /
sa_batch_merge_gathered) bench.py drives over RCCL on the GPUs.  The merged result must equal the
single-index oracle top-k, and BM25 must use the GLOBAL statistics."""
import os
import socket

import numpy as np
import pytest

# torch is imported INSIDE the test and its workers, never at collection: its wheel bundles a librccl.so whose SONAME matches
# /opt/rocm's, and a process that has mapped torch's copy first resolves libsearcharray_hip.so's collectives against it -- the
# GPU suite (which deselects this module) must load the RCCL the library was linked against (tests/test_sharded.py checks).

N_DOCS, VOCAB, K = 6000, 300, 10
QUERIES = np.asarray([[0, 5, 50, 200], [1, 2, 3, 4], [7, 90, 150, 299], [10, 11, 12, 13]])
PHRASES = [[0, 1], [2, 1, 0], [5, 3], [1, 0, 4, 2]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _corpus():
    from searcharray_amd import synth
    return synth.corpus_triples(N_DOCS, VOCAB, 14, seed=17)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from searcharray_amd import roaringish as rz
    from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf
    from tests.emu import emu_api
    t, d, p, lens = _corpus()
    lo, hi = N_DOCS * rank // world, N_DOCS * (rank + 1) // world
    sel = (d >= lo) & (d < hi)
    words, wt = rz.encode_sorted(t[sel], d[sel] - np.uint64(lo), p[sel])
    # global statistics.  avgdl exactly as the reference forms it -- np.mean over the float32 lengths of the
    # WHOLE corpus (indexing.py:282-284; a sum of per-shard sums divided by N is not the same float32) --
    # which every rank can do: doc lengths are index-time metadata (bench.py: synth.zipf_doc_lens).
    # Per-term df is summed over the shards.
    avgdl = np.float32(np.mean(lens))
    index = DeviceIndex(words, rz.term_offsets(wt, VOCAB), lens[lo:hi], avg_doc_len=avgdl, corpus_size=N_DOCS,
                        doc_base=lo, tile_docs=1024, api=emu_api())
    df = torch.from_numpy(index.docfreqs().astype(np.int64))
    dist.all_reduce(df)
    index.set_global_docfreqs(df.numpy().astype(np.uint64))
    idf = np.asarray([[compute_idf(N_DOCS, np.asarray([df[t_].item()])) for t_ in q] for q in QUERIES], dtype=np.float32)
    batch = QueryBatch(index, QUERIES, k=K, idf=idf)
    local = torch.zeros(len(QUERIES) * K, dtype=torch.int64)
    gathered = torch.zeros(world * len(QUERIES) * K, dtype=torch.int64)
    batch.run_local(local.data_ptr(), sync=True)
    dist.all_gather_into_tensor(gathered, local)
    batch.merge_gathered(gathered.data_ptr(), world, sync=True)
    scores, docs = batch.fetch()
    # exact phrases through the same exchange: phrase idf from the GLOBAL document frequencies
    pidf = np.asarray([compute_idf(N_DOCS, np.asarray([df[t_].item() for t_ in ph])) for ph in PHRASES], dtype=np.float32)
    pbatch = index.phrase_batch(PHRASES, k=K, idf=pidf)
    plocal = torch.zeros(len(PHRASES) * K, dtype=torch.int64)
    pgathered = torch.zeros(world * len(PHRASES) * K, dtype=torch.int64)
    pbatch.run_local(plocal.data_ptr(), sync=True)
    dist.all_gather_into_tensor(pgathered, plocal)
    pbatch.merge_gathered(pgathered.data_ptr(), world, sync=True)
    pscores, pdocs = pbatch.fetch()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores, docs=docs, avgdl=avgdl, pscores=pscores, pdocs=pdocs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_topk_matches_single_index_oracle(tmp_path):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import refimpl as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    t, d, p, lens = _corpus()
    orc = O.OracleIndex.from_triples(t, d, p, N_DOCS, doc_lens=lens)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["scores"], r1["scores"]) and np.array_equal(r0["docs"], r1["docs"])   # every rank merges
    assert r0["avgdl"] == np.float32(orc.avg_doc_length)
    for qi, q in enumerate(QUERIES):
        ws, wd = O.topk(orc.score_terms_sum([int(x) for x in q]), K)
        n = int((ws > 0).sum())
        assert np.array_equal(r0["scores"][qi, :n], ws[:n]), f"q{qi}"             # bit-exact, like the unsharded tests
        assert np.array_equal(r0["docs"][qi, :n], wd[:n]), f"q{qi}"
    assert np.array_equal(r0["pscores"], r1["pscores"]) and np.array_equal(r0["pdocs"], r1["pdocs"])
    for pi, ph in enumerate(PHRASES):
        ws, wd = O.topk(orc.score(list(ph)), K)
        n = int((ws > 0).sum())
        assert np.array_equal(r0["pscores"][pi, :n], ws[:n]), f"phrase {ph}"
        assert np.array_equal(r0["pdocs"][pi, :n], wd[:n]), f"phrase {ph}"

"""TEST INFRASTRUCTURE: one rank of tests/test_sharded.py::test_two_ranks_on_one_gpu_over_rccl -- two PROCESSES, both on GPU 0,
each holding one doc-range shard of a seeded corpus, over libsearcharray_hip.so's own RCCL communicator (the path
bench.py --gpus N takes with one process per GPU; here the second GPU is missing, not the second rank).

  python tests/two_rank_worker.py RANK WORLD ID_FILE OUT_DIR [rccl|files]

`files`: the external-collective route of the ABI (sa_batch_run_local / sa_batch_merge_gathered, what tests/test_dist_gloo.py drives
over gloo on the CPU) with the exchange done through files in OUT_DIR -- RCCL refuses two ranks on one device, the device-side
half of the N > 1 path (shard-local top-k with global statistics, the cross-rank merge kernels over BOTH ranks' keys) does not care.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_DOCS, VOCAB, K = 40_000, 2000, 10


def corpus():
    from searcharray_amd import synth
    return synth.corpus_triples(N_DOCS, VOCAB, 24, seed=23)


def queries():
    from searcharray_amd import synth
    q = synth.bm25_queries(48, vocab=VOCAB, seed=11)
    q[:, 0] = np.arange(48) % 4                                  # shared first terms: the grouped kernel has groups
    return q


def main():
    rank, world, id_file, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "rccl"
    from searcharray_amd import _lib, roaringish as rz
    from searcharray_amd.device_index import DeviceIndex, compute_idf
    api = _lib.api()
    t, d, p, lens = corpus()
    lo, hi = N_DOCS * rank // world, N_DOCS * (rank + 1) // world
    sel = (d >= lo) & (d < hi)
    words, wt = rz.encode_sorted(t[sel], d[sel] - np.uint64(lo), p[sel])
    index = DeviceIndex(words, rz.term_offsets(wt, VOCAB), lens[lo:hi], avg_doc_len=np.float32(np.mean(lens)), corpus_size=N_DOCS,
                        doc_base=lo, device=0, api=api)        # EVERY rank on device 0
    if mode == "files":
        return files_mode(api, index, rank, world, out_dir)
    if rank == 0:
        uid = DeviceIndex.comm_unique_id(api)
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise RuntimeError("no communicator id")
            time.sleep(0.02)
        uid = open(id_file, "rb").read()
    try:
        index.comm_init(rank, world, uid)
    except Exception as e:                                       # RCCL's answer to two ranks on one device is the test's result
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            f.write(str(e))
        return 3
    df = index.comm_allreduce(np.ascontiguousarray(index.docfreqs(), dtype=np.uint64).copy(), "sum")
    index.set_global_docfreqs(df)
    q = queries()
    idf = np.asarray([[compute_idf(N_DOCS, np.asarray([df[x]])) for x in row] for row in q], dtype=np.float32)
    bt = index.batch(q, k=K, idf=idf, opts={"sparse": 0})
    res = []
    for _ in range(3):                                           # (double-buffered exchange: several steps in flight)
        bt.run(sync=False)
        res.append(bt.fetch())
    ver, path = DeviceIndex.comm_library_info(api)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=res[-1][0], docs=res[-1][1], scores0=res[0][0], docs0=res[0][1],
             nccl_version=ver, lib=np.asarray(path), groups=np.asarray(bt.group_info()["groups"]))
    index.comm_barrier()
    bt.close()
    index.comm_destroy()
    index.close()
    return 0


def _publish(path, arr):
    np.save(path + ".tmp.npy", arr)
    os.replace(path + ".tmp.npy", path)


def _await(path, timeout=120):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError(f"peer never wrote {path}")
        time.sleep(0.02)
    return np.load(path)


def files_mode(api, index, rank, world, out_dir):
    import ctypes
    from searcharray_amd.device_index import compute_idf
    # global df: every rank publishes its shard's, sums all (what comm_allreduce does over RCCL)
    _publish(os.path.join(out_dir, f"df{rank}.npy"), index.docfreqs().astype(np.uint64))
    df = sum(_await(os.path.join(out_dir, f"df{r}.npy")) for r in range(world)).astype(np.uint64)
    index.set_global_docfreqs(df)
    q = queries()
    idf = np.asarray([[compute_idf(N_DOCS, np.asarray([df[x]])) for x in row] for row in q], dtype=np.float32)
    bt = index.batch(q, k=K, idf=idf, opts={"sparse": 0})
    n = len(q) * K
    # page-locked host buffers: the kernels of the library read and write them like device memory
    local_p, gath_p = ctypes.c_void_p(), ctypes.c_void_p()
    api.call("sa_host_alloc", n * 8, ctypes.byref(local_p))
    api.call("sa_host_alloc", world * n * 8, ctypes.byref(gath_p))
    local = np.ctypeslib.as_array(ctypes.cast(local_p, ctypes.POINTER(ctypes.c_uint64)), (n,))
    gathered = np.ctypeslib.as_array(ctypes.cast(gath_p, ctypes.POINTER(ctypes.c_uint64)), (world * n,))
    bt.run_local(local_p.value, sync=True)
    _publish(os.path.join(out_dir, f"keys{rank}.npy"), local.copy())
    for r in range(world):
        gathered[r * n:(r + 1) * n] = _await(os.path.join(out_dir, f"keys{r}.npy"))
    bt.merge_gathered(gath_p.value, world, sync=True)
    scores, docs = bt.fetch()
    other = [int((gathered[r * n:(r + 1) * n] != 0).sum()) for r in range(world)]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores, docs=docs, keys_per_rank=np.asarray(other),
             groups=np.asarray(bt.group_info()["groups"]))
    bt.close()
    api.call("sa_host_free", local_p)
    api.call("sa_host_free", gath_p)
    index.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

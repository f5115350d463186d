#!/usr/bin/env python
"""On-disk index <-> HBM (csrc/sa_io.hip): save a resident shard as the reference's raw uint64 .dat,
stream it back file -> page-locked ring -> HBM, next to the route through numpy (np.fromfile +
upload of a pageable array) and to what the reference does with the file (np.memmap + first touch
of every page)."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import synth                                          # noqa: E402
from searcharray_amd.device_index import DeviceIndex                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--dir", default="/tmp")
    args = ap.parse_args()
    corpus = synth.zipf_corpus(args.docs, vocab=args.vocab, workers=8)
    dev = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
    dev.synchronize()
    path = os.path.join(args.dir, "io_bench.dat")
    gb = corpus.words.nbytes / 1e9

    t0 = time.perf_counter()
    off = dev.save(path)
    t_save = time.perf_counter() - t0
    same_file = bool(np.array_equal(np.fromfile(path, dtype=np.uint64), corpus.words))

    def timed_load():
        t0 = time.perf_counter()
        d = DeviceIndex.from_file(path, off, corpus.doc_lens)
        d.synchronize()
        return d, time.perf_counter() - t0

    d1, t_first = timed_load()                    # page cache warm (just written); first call page-locks the ring
    d1.close()
    t_load = t_load2 = 1e9
    for _ in range(5):
        d1, t = timed_load()
        t_load = min(t_load, t)
        d1.close()
    d1, _t = timed_load()
    sweep = {}
    if os.environ.get("IO_SWEEP"):
        for threads in (1, 2, 4, 6):
            for piece in (1 << 20, 4 << 20, 16 << 20):
                os.environ["SA_IO_THREADS"], os.environ["SA_IO_PIECE_BYTES"] = str(threads), str(piece)
                ts = []
                for _ in range(3):
                    dd, t = timed_load()
                    dd.close()
                    ts.append(t)
                t0 = time.perf_counter()
                dev.save(path + ".2")
                sweep[f"t{threads}_p{piece >> 20}M"] = [round(min(ts), 3), round(time.perf_counter() - t0, 3)]
        os.unlink(path + ".2")
        del os.environ["SA_IO_THREADS"], os.environ["SA_IO_PIECE_BYTES"]
    q = [0, 9, 99, 999]
    same_scores = bool(np.array_equal(d1.bm25_dense(q), dev.bm25_dense(q)))
    d1.close()

    t0 = time.perf_counter()
    w = np.fromfile(path, dtype=np.uint64)
    d2 = DeviceIndex(w, off, corpus.doc_lens)
    d2.synchronize()
    t_numpy = time.perf_counter() - t0
    d2.close()

    t0 = time.perf_counter()
    mm = np.memmap(path, dtype=np.uint64, mode="r")
    touched = int(mm[::512].sum() & 1)            # the reference faults pages in as queries touch terms
    t_memmap_touch = time.perf_counter() - t0

    t_mem = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        d3 = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
        d3.synchronize()
        t_mem = min(t_mem, time.perf_counter() - t0)
        d3.close()
    os.unlink(path)
    print(json.dumps({"docs": args.docs, "file_GB": round(gb, 3),
                      "save_s": round(t_save, 3), "save_GBps": round(gb / t_save, 2),
                      "first_load_s_incl_ring_alloc": round(t_first, 3),
                      "load_s_incl_derive": round(min(t_load, t_load2), 3),
                      "load_GBps_incl_derive": round(gb / min(t_load, t_load2), 2),
                      "fromfile_then_upload_s": round(t_numpy, 3),
                      "upload_from_host_array_s": round(t_mem, 3),
                      "memmap_touch_every_page_s": round(t_memmap_touch, 3),
                      "sweep_load_save_s": sweep, "file_identical": same_file, "scores_identical": same_scores, "_": touched}))


if __name__ == "__main__":
    main()

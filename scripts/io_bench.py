#!/usr/bin/env python
"""On-disk index <-> HBM (csrc/sa_io.hip): save a resident shard as the reference's raw uint64 .dat,
stream it back file -> page-locked ring -> HBM, next to the route through numpy (np.fromfile +
upload of a pageable array) and to what the reference does with the file (np.memmap + first touch
of every page)."""
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

    d1, t_load = timed_load()                     # page cache warm (just written)
    d1.close()
    d1, t_load2 = timed_load()
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

    t0 = time.perf_counter()
    d3 = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
    d3.synchronize()
    t_mem = time.perf_counter() - t0
    os.unlink(path)
    print(json.dumps({"docs": args.docs, "file_GB": round(gb, 3),
                      "save_s": round(t_save, 3), "save_GBps": round(gb / t_save, 2),
                      "load_s_incl_derive": round(min(t_load, t_load2), 3),
                      "load_GBps_incl_derive": round(gb / min(t_load, t_load2), 2),
                      "fromfile_then_upload_s": round(t_numpy, 3),
                      "upload_from_host_array_s": round(t_mem, 3),
                      "memmap_touch_every_page_s": round(t_memmap_touch, 3),
                      "file_identical": same_file, "scores_identical": same_scores, "_": touched}))


if __name__ == "__main__":
    main()

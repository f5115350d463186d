#!/usr/bin/env python
"""Same-box A/B of builds and switches of the exhaustive scoring path on the bench workload.

  python scripts/ab.py --libs searcharray_amd/libsearcharray_hip_base.so,searcharray_amd/libsearcharray_hip.so \
                       --envs "SA_GROUP=1;SA_GROUP=1,SA_GROUP_WARM=8" --ks 10,1000 --qsets baseline,distinct

Every (library, environment) pair scores the same resident batch; results must be identical to the first
configuration's.  One JSON line per configuration (ms per step by the host clock over `--steps` asynchronous runs,
kernel ms by HIP events).  Libraries are loaded side by side in one process (ctypes, RTLD_LOCAL), each builds its own
index from the same corpus."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth, _lib                                     # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf   # noqa: E402

KEYS = ("SA_GROUP",  "SA_GROUP_WARM", "SA_GROUP_MIN", "SA_GROUP_LOOSE", "SA_GROUP_SIDE", "SA_SPARSE", "SA_GRP_VARIANT",
        "SA_SEED", "SA_SEED_WARM_DIV", "SA_SEED_J", "SA_XCD_RANGE", "SA_TERM_SEED", "SA_MERGE_SMALL", "SA_GROUP_MAXQ", "SA_LOOSE_POSTINGS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--corpus-cache", default="")
    ap.add_argument("--ks", default="10")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--libs", default="searcharray_amd/libsearcharray_hip.so")
    ap.add_argument("--envs", default="SA_SPARSE=0", help="';'-separated configurations of ','-separated NAME=VALUE")
    ap.add_argument("--qsets", default="baseline")
    ap.add_argument("--queries", type=int, default=256)
    args = ap.parse_args()
    D, V = args.docs, args.vocab
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz") if args.corpus_cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
        if cpath:
            os.makedirs(args.corpus_cache, exist_ok=True)
            np.savez(cpath, words=corpus.words, term_off=corpus.term_off, doc_lens=corpus.doc_lens)
    B = args.queries
    qsets = {"baseline": synth.bm25_queries(B, vocab=V), "distinct": synth.bm25_queries_distinct(B, vocab=V) if 4 * B <= V else None}
    # "hot": every query's further terms drawn from FOUR terms per band, so their posting slices stay in L2 -- what the
    # kernel costs when memory latency is out of the picture
    hot = qsets["baseline"].copy()
    for c in (1, 2, 3):
        hot[:, c] = hot[:4, c][np.arange(B) % 4]
    qsets["hot"] = hot
    qsets = {k_: v for k_, v in qsets.items() if k_ in args.qsets.split(",") and v is not None}
    envs = [dict(kv.split("=") for kv in cfg.split(",") if kv) for cfg in args.envs.split(";")]
    ref = {}
    for lib in args.libs.split(","):
        path = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
        api = _lib.bind(ctypes.CDLL(path), path, allow_missing=True)         # (an older build lacks the newer entry points)
        index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, tile_docs=args.tile, api=api)
        df = index.docfreqs()
        for qname, queries in qsets.items():
            idf = np.asarray([[compute_idf(D, np.asarray([df[t]])) for t in q] for q in queries], dtype=np.float32)
            for k in [int(x) for x in args.ks.split(",")]:
                for cfg in envs:
                    # the switches of a configuration: options of the batch (this round's library) -- and, for a build from
                    # before the options existed (--libs), the environment variables it read
                    for key in KEYS:
                        os.environ.pop(key, None)
                    os.environ["SA_SPARSE"] = "0"
                    os.environ.update(cfg)
                    # (configuration "default=1": no route option at all -- the library's own rule picks the route)
                    bopts = {kk: vv for kk, vv in cfg.items() if kk != "default"} if "default" in cfg else dict({"sparse": 0}, **cfg)
                    batch = QueryBatch(index, queries, k=k, idf=idf, opts=bopts)
                    for _ in range(3):
                        batch.run(sync=False)
                    index.synchronize()
                    batch.profile()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        batch.run(sync=False)
                    index.synchronize()
                    dt = (time.perf_counter() - t0) / args.steps
                    kms, _, _ = batch.profile()
                    res = batch.fetch()
                    probe = None
                    if hasattr(api._cdll, "sa_debug_probe_read"):          # a -DSA_PROBE build (scripts/build_probe.sh)
                        buf = (ctypes.c_ulonglong * 16)()
                        api._cdll.sa_debug_probe_read(buf, 1)              # (clear: counts of the runs so far)
                        batch.run(sync=True)
                        api._cdll.sa_debug_probe_read(buf, 1)
                        v = list(buf)
                        items, pairs = max(v[6], 1), max(v[7], 1)
                        names = ["tables", "request_q0_and_base", "wait_for_postings", "request_next_query", "overlay_query", "flush_and_defer"]
                        probe = {"items": v[6], "pairs": v[7], "deferred_pairs": v[8],
                                 "cycles_per_item": {nm: round(v[i] / items, 1) for i, nm in enumerate(names)},
                                 "cycles_per_pair_in_the_query_loop": {nm: round(v[i] / pairs, 1) for i, nm in enumerate(names) if 2 <= i <= 4},
                                 "cycles_per_item_total": round((sum(v[:6]) + sum(v[9:12])) / items, 1),
                                 "fine": {nm: round(v[i] / items, 1) for i, nm in ((9, "kernel_start_to_group_entry_loaded"), (10, "slice_bounds_loaded"), (11, "q0_postings_loaded_after_tables"))} if any(v[9:12]) else None,
                                 "note": "s_memtime ticks at 100 MHz x ... see DESIGN 3.1a; one launch; items that reach the overlay"}
                    if hasattr(api._cdll, "sa_debug_stage_probe_read") and batch.last_route() == "staged":   # (-DSA_PROBE: the staged-tile kernel's phases)
                        buf = (ctypes.c_ulonglong * 32)()
                        api._cdll.sa_debug_stage_probe_read(buf, 1)
                        batch.run(sync=True)
                        api._cdll.sa_debug_stage_probe_read(buf, 1)
                        v = list(buf)
                        passes = max(v[24], 1)
                        names = ["slice_ends_and_scan", "offsets_and_chunk_list", "barrier_a", "next_tile_reads_and_stage_loads_issued", "stage_loads_landed_and_written", "barrier_b",
                                 "query_phase_and_scan", "barrier_c", "stage_a_tail", "stage_b_rounds_tail", "pass_tail", "tile_top",
                                 "a_search", "a_check", "a_compact", "a_barrier", "b_round_head", "b_lookups", "b_compact", "b_barrier", "before_flush", "flush_atomics_append", "flush_refresh_and_barriers", "flush_probes_and_score"]
                        probe = {"tile_passes": v[24], "workgroups": v[26], "candidates_per_pass": round(v[25] / passes, 1), "finalists_per_pass": round(v[27] / passes, 1), "flushes_per_pass": round(v[28] / passes, 2),
                                 "cycles_per_pass": {nm: round(v[i] / passes, 1) for i, nm in enumerate(names)},
                                 "cycles_per_pass_total": round(sum(v[:24]) / passes, 1),
                                 "note": "s_memtime of wave 0 at the phase boundaries (shader cycles); one launch"}
                    r0 = ref.setdefault((qname, k), res)
                    same = bool(np.array_equal(r0[0], res[0]) and np.array_equal(r0[1], res[1]))
                    print(json.dumps({"lib": os.path.basename(lib), "queries": qname, "k": k, "docs": D, **cfg,
                                      "route": batch.last_route(), "ms_per_step": round(dt * 1e3, 4), "kernel_ms": round(kms, 4), "same_results": same, **({"probe": probe} if probe else {})}), flush=True)
                    batch.close()
        index.close()


if __name__ == "__main__":
    main()

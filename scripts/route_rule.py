#!/usr/bin/env python
"""Does the library's AUTOMATIC route (option `sparse` unset: csrc/sa_bm25.hip, sa_batch_run_shard) pick the faster of
exhaustive scoring and dynamic pruning?  Sweep: the share of a 256-query batch that the grouped exhaustive kernel can take
(BASELINE-shaped queries: first term from ranks 1-10) against queries it cannot (a rare first term nobody shares + three
dense terms: per-query kernel), x k.  One JSON line per cell: ms per step of both forced routes and of the default, which
route the default took, and whether it is within 5 % of the better one.

  python scripts/route_rule.py [--docs 10000000] > profiles/route_rule_r05.jsonl
"""
import _envopts  # noqa: F401
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from searcharray_amd import synth                                       # noqa: E402
from searcharray_amd.device_index import DeviceIndex                    # noqa: E402


def mixed(B, share, vocab, seed):
    rng = np.random.default_rng(seed)
    base = synth.bm25_queries(B, vocab=vocab, seed=seed)
    n_g = int(round(B * share))
    q = base.copy()
    for i in range(n_g, B):                                             # a rare, unshared first term + three dense terms
        q[i] = [5000 + i, *(rng.choice(np.arange(1, 40), 3, replace=False))]
    return q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--corpus-cache", default="/tmp/corpus")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    D, V = args.docs, args.vocab
    cpath = os.path.join(args.corpus_cache, f"zipf_{D}_{V}_0_{D}.npz")
    if os.path.exists(cpath):
        z = np.load(cpath)
        corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
    else:
        corpus = synth.zipf_corpus(D, vocab=V, workers=8)
    dev = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens)
    for k in (10, 100, 1000):
        for share in (0.0, 0.25, 0.5, 0.75, 1.0):
            q = mixed(256, share, V, 42)
            row = {"docs": D, "k": k, "groupable_share": share}
            res = {}
            for name, opts in (("exhaustive", {"sparse": 0}), ("pruned", {"sparse": 1}), ("default", {})):
                bt = dev.batch(q, k=k, opts=opts)
                for _ in range(3):
                    bt.run(sync=False)
                dev.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    bt.run(sync=False)
                dev.synchronize()
                row[f"{name}_ms"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
                if name == "default":
                    row["default_route"] = bt.last_route()
                    row["grouping"] = bt.group_info()
                res[name] = bt.fetch()
                bt.close()
            best = min(row["exhaustive_ms"], row["pruned_ms"])
            row["default_over_best"] = round(row["default_ms"] / best, 3)
            row["default_within_5pct"] = bool(row["default_ms"] <= 1.05 * best)
            row["same_results"] = bool(all(np.array_equal(res["exhaustive"][i], res[n][i]) for n in ("pruned", "default") for i in (0, 1)))
            print(json.dumps(row), flush=True)
    dev.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""BASELINE configs 1 / 5 on REAL data, when it is there: MSMARCO docs (the reference downloads
msmarco-docs.tsv.gz, /root/reference/test/msmarco_utils.py:52-62 -- there is no network here, so the file has to be mounted
at data/msmarco-docs.tsv[.gz]).  Loader as the reference's fixture builds its index (test/test_msmarco.py:71-91:
csv column 3 = body, the whitespace tokenizer of test/tokenizers.py:8-11, the first --docs documents), then config 5's
query shape: two-token phrases with slop = 2 through SearchArray.score, and the 4-term disjunction caller idiom
(test/test_msmarco.py:353-354) through a device batch.  Without the file the script says so and exits 0; the synthetic
stand-ins of SURVEY.md 8d (bench.py, scripts/slop_bench.py) stay the measured configuration.

  python scripts/msmarco.py [--docs 1000000] [--path data/msmarco-docs.tsv]
"""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import csv
import gzip
import json
import os
import string
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_FOLD = {ord(x): ord(y) for x, y in zip("‘’´“”–-", "'''\"\"--")}
_TRANS = {**_FOLD, **str.maketrans({c: " " for c in string.punctuation})}


def tokenize(text):                      # what test/tokenizers.py:8-11 does: fold quotes / dashes, punctuation -> space, lower, split
    return text.translate(_TRANS).lower().split()


def column(path, col, num_docs):
    opener = gzip.open if path.endswith(".gz") else open
    csv.field_size_limit(sys.maxsize)
    with opener(path, "rt", encoding="utf-8", newline="") as f:
        for i, row in enumerate(csv.reader(f, delimiter="\t")):
            if i >= num_docs:
                return
            yield row[col] if len(row) > col else ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", default="")
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=32)
    args = ap.parse_args()
    path = args.path or next((p for p in (os.path.join(ROOT, "data", "msmarco-docs.tsv"), os.path.join(ROOT, "data", "msmarco-docs.tsv.gz"))
                              if os.path.exists(p)), "")
    if not path or not os.path.exists(path):
        print(json.dumps({"msmarco": "not mounted", "expected": "data/msmarco-docs.tsv[.gz]",
                          "note": "BASELINE configs 1 / 5 are measured on the synthetic stand-ins of SURVEY.md 8d"}))
        return
    from searcharray_amd import SearchArray
    t0 = time.perf_counter()
    arr = SearchArray.index(column(path, 3, args.docs), tokenizer=tokenize, truncate=True)
    t_index = time.perf_counter() - t0
    # query terms: frequent body tokens (config 5: two-token phrases, slop 2), as SURVEY 8d draws them on the synthetic corpus
    dfs = arr._core.device().docfreqs()
    order = np.argsort(dfs)[::-1]
    rng = np.random.default_rng(5)
    out = {"msmarco": path, "docs": len(arr), "index_s": round(t_index, 2), "vocab": int(len(dfs))}
    if len(order) >= 2:
        top = min(200, len(order))
        name = arr.term_dict.get_term
        pairs = [(name(int(order[a])), name(int(order[b]))) for a, b in rng.integers(0, top, (args.queries, 2)) if a != b]
        t0 = time.perf_counter()
        hits = [int((arr.score(list(p), slop=2) > 0).sum()) for p in pairs]
        out["slop2_phrases"] = {"n": len(pairs), "ms_per_query": round((time.perf_counter() - t0) / max(1, len(pairs)) * 1e3, 3),
                                "matching_docs_mean": float(np.mean(hits)) if hits else 0.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

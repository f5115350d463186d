#!/usr/bin/env python
"""Per-query HBM traffic of the slop pipeline from the two --pmc passes of scripts/gpu_slop_pmc.sh.
slop_heavy.py runs the 2-term phrase reps+1 times, then the 3-term phrase reps+1 times; a query's kernels are the span
kernels from one sa_k_span_flags<T> (or sa_k_span_doc_fused<T>: the doc-parallel route's single launch) to the next, and
T labels the query.  Counter units as in bench.py: FETCH_SIZE and WRITE_SIZE in KiB;
FETCH_SIZE under-reports wide coalesced reads on gfx950 (exactly 1/2 for 16 bytes per lane, MI355X_MICROARCH.md), and
these kernels mix 1-, 4-, 8- and 16-byte accesses, so the read traffic is given as the raw figure (a lower bound) and
as twice it (the upper bound)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def queries(path):
    """[(label, {kernel: {counter: value}})] in dispatch order; label = terms of the phrase, from the query's first kernel"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "span" not in name:
            continue
        first = "sa_k_span_flags" in name or "sa_k_span_doc_fused" in name   # a query's first launch (the doc-parallel route: its only one)
        if first and (cur is None or r["Dispatch_Id"] != cur[2]):
            terms = name.split("<")[1].split(">")[0].split(",")[0].strip() if "<" in name else "n"
            cur = (f"{terms}_terms", collections.defaultdict(lambda: collections.defaultdict(float)), r["Dispatch_Id"])
            out.append(cur)
        if cur is not None:
            cur[1][name.split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    return [(lab, k) for lab, k, _ in out]


def main():
    base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    f = sorted(glob.glob(f"{base}/pmc_slop_f/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    w = sorted(glob.glob(f"{base}/pmc_slop_w/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    res = {}
    for label in ("2_terms", "3_terms"):
        kern = collections.defaultdict(lambda: collections.defaultdict(float))
        n = 0
        for path in (f, w):
            qs = [k for lab, k in queries(path) if lab == label][1:]          # (the first run of a phrase loads its kernels)
            n = max(n, len(qs))
            for q in qs:
                for k, v in q.items():
                    for c, x in v.items():
                        kern[k][c] += x
        if n == 0:
            continue
        per = {}
        tot_r = tot_w = 0.0
        for k, v in kern.items():
            rd = v.get("FETCH_SIZE", 0.0) / n * 1024.0
            wr = v.get("WRITE_SIZE", 0.0) / n * 1024.0
            hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
            per[k] = {"read_MB_raw": round(rd / 1e6, 2), "write_MB": round(wr / 1e6, 2),
                      "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None}
            tot_r += rd
            tot_w += wr
        res[label] = {"queries": n, "hbm_read_MB_raw": round(tot_r / 1e6, 2), "hbm_read_MB_x2": round(2 * tot_r / 1e6, 2),
                      "hbm_write_MB": round(tot_w / 1e6, 2), "kernels": per}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

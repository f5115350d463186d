#!/usr/bin/env python
"""Per-kernel means of the counters in rocprofv3 --pmc output directories (counter_collection.csv), one JSON object:
{kernel: {counter: mean per dispatch, "dispatches": n}} -- the scoring kernels only (sa_k_bm25*, sa_k_topk_merge), or those whose names start with one of SQ_PREFIXES (comma-separated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PREFIXES = os.environ.get("SQ_PREFIXES", "sa_k_bm25,sa_k_topk,sa_k_run_reset").split(",")
out = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getsize)
    if not files:
        continue
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(k.startswith(x) for x in PREFIXES):
            continue
        did = int(r["Dispatch_Id"])
        names[did] = k
        per[did][r["Counter_Name"]] += float(r["Counter_Value"])
    for did, cs in per.items():
        for c, v in cs.items():
            out[names[did]][c].append(v)
res = {}
for k, cs in out.items():
    res[k] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    res[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(res, indent=1))

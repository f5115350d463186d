#!/usr/bin/env python
"""Per-kernel means of the counters in rocprofv3 --pmc output directories (counter_collection.csv), one JSON object:
{kernel: {counter: mean per dispatch, "dispatches": n}} -- the scoring kernels only (sa_k_bm25*, sa_k_topk_merge)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getsize)
    if not files:
        continue
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not (k.startswith("sa_k_bm25") or k.startswith("sa_k_topk") or k.startswith("sa_k_run_reset")):
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

#!/usr/bin/env python
"""edismax over a two-field frame: combination on the device (Part 4 of the C ABI) vs with numpy on the
host (the reference's way, over the same GPU score() vectors)."""
import argparse, json, os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searcharray_amd import SearchArray, synth
from searcharray_amd.solr import edismax

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=500_000)
args = ap.parse_args()
D, V = args.docs, 50_000
names = np.array([f"t{i}" for i in range(V)])
frames = {}
for field, seed_off in (("title", 0), ("body", 1)):
    lens, terms = synth.zipf_batch_tokens(seed_off, D, V, fast=True)
    if field == "title":
        lens = np.maximum(1, lens // 4)
        starts = np.concatenate([[0], np.cumsum(lens)])
        terms = terms[:starts[-1]]
    starts = np.concatenate([[0], np.cumsum(lens)])
    frames[field] = SearchArray.index([" ".join(names[terms[starts[i]:starts[i + 1]]]) for i in range(D)])
frame = pd.DataFrame(frames)
params = dict(q="t3 t40 t7", qf=["title^3", "body"], pf=["body"], pf2=["title", "body"], mm="2<75%", tie=0.2)
out = {"docs": D}
res = {}
for route in (True, False):
    edismax(frame, use_device=route, **params)
    t0 = time.perf_counter()
    for _ in range(5):
        res[route], _ = edismax(frame, use_device=route, **params)
    out["device_ms" if route else "host_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
out["identical"] = bool(np.array_equal(res[True], res[False]))
out["matches"] = int((res[True] > 0).sum())
print(json.dumps(out))

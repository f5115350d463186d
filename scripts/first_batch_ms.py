"""How long the FIRST batch with a new (k1, b) takes at 10 M docs -- the impact stream, dense rows and rank tables are built then, and (round 6,
default route) the staged-tile route's stage directory, probe rows and presence bitmaps -- against a second batch with the same parameters.
One JSON line per (route, parameters)."""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np                                                    # noqa: E402
from searcharray_amd import synth                                     # noqa: E402
from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf   # noqa: E402

D, V = 10_000_000, 100_000
cache = sys.argv[1] if len(sys.argv) > 1 else ""
cpath = os.path.join(cache, f"zipf_{D}_{V}_0_{D}.npz") if cache else ""
if cpath and os.path.exists(cpath):
    z = np.load(cpath)
    c = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], D, V, 0)
else:
    c = synth.zipf_corpus(D, vocab=V, workers=8)
index = DeviceIndex(c.words, c.term_off, c.doc_lens)
df = index.docfreqs()
q = synth.bm25_queries(256, vocab=V)
idf = np.asarray([[compute_idf(D, np.asarray([df[t]])) for t in r] for r in q], dtype=np.float32)
for route, opts in (("staged (default)", {}), ("exhaustive overlay", {"sparse": 0, "stage": 0})):
    for i in range(2):
        for rep in ("first batch of these parameters", "second batch"):
            index.synchronize()
            t0 = time.perf_counter()
            b = QueryBatch(index, q, k=10, idf=idf, k1=1.2 + 0.1 * i + (0.05 if opts else 0.0), opts=opts)
            b.run()
            b.fetch()
            ms = (time.perf_counter() - t0) * 1e3
            print(json.dumps({"route": route, "k1": round(1.2 + 0.1 * i + (0.05 if opts else 0.0), 2), "what": rep, "create_run_fetch_ms": round(ms, 2), "last_route": b.last_route()}), flush=True)
            b.close()

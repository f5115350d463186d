"""How long the first batch with a new (k1, b) takes at 10 M docs: impact stream + dense rows + rank tables are built then."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from searcharray_amd import synth, _lib
from searcharray_amd.device_index import DeviceIndex, QueryBatch, compute_idf
D, V = 10_000_000, 100_000
c = synth.zipf_corpus(D, vocab=V, workers=8)
index = DeviceIndex(c.words, c.term_off, c.doc_lens)
df = index.docfreqs()
q = synth.bm25_queries(256, vocab=V)
idf = np.asarray([[compute_idf(D, np.asarray([df[t]])) for t in r] for r in q], dtype=np.float32)
os.environ["SA_SPARSE"] = "0"
for i in range(2):
    index.synchronize()
    t0 = time.perf_counter()
    b = QueryBatch(index, q, k=10, idf=idf, k1=1.2 + 0.1 * i)
    index.synchronize()
    print("batch create incl. impact stream + rank tables for a new (k1, b): ms", (time.perf_counter() - t0) * 1e3)
    b.close()

#!/usr/bin/env python
"""The bench's slop_batch / phrase_batch legs alone (zipf-1M, 32 two-token slop-2 phrases of ranks 50-5000; 256 sampled
trigrams), for rocprofv3 --kernel-trace --stats."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from searcharray_amd import _lib                         # noqa: E402

api = _lib.api()
side = bench.PhraseSide(api, 1_000_000, 100_000)
out = {}
for name, b in side.legs():
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    dt, kms = side.timed(b, 3, 20)
    out[name] = {"ms_per_step": round(dt / 20 * 1e3, 4), "kernel_ms": round(kms, 4)}
if hasattr(api._cdll, "sa_debug_span_probe_read"):      # a -DSA_PROBE build (SA_PROBE_LIB=build/libsearcharray_hip_probe.so: see below)
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    api._cdll.sa_debug_span_probe_read(buf, 1)
    side.sb.run(sync=True)
    api._cdll.sa_debug_span_probe_read(buf, 1)
    v = list(buf)
    blocks = max(v[7], 1)
    names = ["gather", "gather_barrier_wait", "order", "machines", "machines_barrier_wait", "heavy_documents", "score_and_rank"]
    out["slop_batch_probe"] = {"blocks": v[7], "cycles_per_block_wave0": {n: round(v[i] / blocks, 1) for i, n in enumerate(names)},
                               "cycles_per_block_total": round(sum(v[:7]) / blocks, 1), "note": "s_memtime of wave 0 at the phase boundaries (shader cycles); one launch"}
print(json.dumps(out))
side.close()

"""CPU ORACLE (test infrastructure): slop > 0 phrase search, restating
searcharray/phrase/spans.py:71-187 (candidate-word selection, numpy) on top of the C restatement
of the span state machine (oracle/spans.c <- searcharray/roaringish/spans.pyx)."""
from __future__ import annotations

import ctypes
from collections import Counter
from typing import List, Tuple

import numpy as np

from . import refimpl as O

_1 = np.uint64(1)
_HDR_UNIT = np.uint64(1 << 18)            # 1 << (64 - header_bits), spans.py:107-108


def header(arr: np.ndarray) -> np.ndarray:
    return arr & O.HEADER_MASK


def slice_by_header(encoded: np.ndarray, headers: np.ndarray) -> np.ndarray:
    """RoaringishEncoder.slice(header=...), roaringish.py:253-259."""
    _, idx_enc = O.intersect(headers, header(encoded), drop_duplicates=False)
    return encoded[idx_enc.astype(np.int64)]


def intersect_all(posns_encoded: List[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """spans.py:71-123."""
    if len(posns_encoded) < 2:
        raise ValueError("Need at least two positions to intersect")
    last_lhs = last_rhs = None
    curr = posns_encoded[0]
    for nxt in posns_encoded[1:]:
        lhs_int_idx, _ = O.intersect(curr, nxt, mask=O.HEADER_MASK)
        int_headers = header(curr[lhs_int_idx.astype(np.int64)])
        curr_to_right, next_to_left = O.adjacent(curr, nxt, mask=O.HEADER_MASK)
        lhs_headers = O.merge(int_headers, nxt[next_to_left.astype(np.int64)])
        rhs_headers = O.merge(int_headers, curr[curr_to_right.astype(np.int64)])
        next_to_right, curr_to_left = O.adjacent(nxt, curr, mask=O.HEADER_MASK)
        lhs_headers = O.merge(lhs_headers, curr[curr_to_left.astype(np.int64)])
        rhs_headers = O.merge(rhs_headers, nxt[next_to_right.astype(np.int64)])
        if last_lhs is not None:
            l, _ = O.intersect(last_lhs, lhs_headers, mask=O.HEADER_MASK)
            r, _ = O.intersect(last_rhs, rhs_headers, mask=O.HEADER_MASK)
            last_lhs = last_lhs[l.astype(np.int64)]
            last_rhs = last_rhs[r.astype(np.int64)]
        else:
            last_lhs, last_rhs = lhs_headers, rhs_headers
        # NOTE: `curr` is never advanced in the reference loop (spans.py:77-105): every pair is
        # (term 0, term i)
    to_rhs = last_rhs + _HDR_UNIT
    to_lhs = last_lhs - _HDR_UNIT
    all_headers = O.merge(to_rhs, to_lhs, drop_duplicates=True)
    all_headers = O.merge(last_lhs, all_headers, drop_duplicates=True)
    all_headers = O.merge(last_rhs, all_headers, drop_duplicates=True)
    all_headers = all_headers & O.HEADER_MASK
    sliced = [slice_by_header(p, all_headers) for p in posns_encoded]
    lengths = np.cumsum([0] + [len(p) for p in sliced], dtype=np.uint64)
    return np.concatenate(sliced).astype(np.uint64), lengths


def span_search(posns_encoded: List[np.ndarray], slop: int, return_overflow: bool = False):
    """spans.py:171-187 + spans.pyx:322-330: (ids, counts) in Counter order."""
    posns, lengths = intersect_all(posns_encoded)
    lib = O.lib()
    cap = max(1, int(lengths[1] - lengths[0]) + 1)
    keys = np.empty(cap, dtype=np.uint64)
    incr = np.empty(cap, dtype=np.float32)
    overflow = ctypes.c_long(0)
    lib.oracle_span_freqs.argtypes = [O.u64p, O.c_long, O.u64p, O.c_long, O.c_u64, O.u64p, O.f32p, O.c_long,
                                      ctypes.POINTER(ctypes.c_long)]
    lib.oracle_span_freqs.restype = O.c_long
    posns = np.ascontiguousarray(posns, dtype=np.uint64)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
    n = lib.oracle_span_freqs(posns.ctypes.data_as(O.u64p), len(posns), lengths.ctypes.data_as(O.u64p), len(lengths),
                              int(slop), keys.ctypes.data_as(O.u64p), incr.ctypes.data_as(O.f32p), cap,
                              ctypes.byref(overflow))
    freqs: Counter = Counter()
    for k, v in zip(keys[:n].tolist(), incr[:n].tolist()):
        freqs[k] += v
    ids = np.array(list(freqs.keys()), dtype=np.uint64)
    counts = np.array(list(freqs.values()), dtype=np.float32)
    if return_overflow:
        return ids, counts, overflow.value
    return ids, counts

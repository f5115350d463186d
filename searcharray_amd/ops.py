"""Kernel-level entry points with the reference's names and numpy signatures.

Each function has the same arguments, result types and error behaviour as the Cython
function it replaces (cited per function) but runs on the GPU through the C ABI (Part 1 of
include/searcharray_hip.h).  They exist for drop-in rebinding of the reference's call sites
 and for kernel-level parity tests; the resident-index classes in
``device_index`` are the fast path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import as_f32, as_u64, p_f32, p_u64

ALL_BITS = np.uint64(0xFFFFFFFFFFFFFFFF)


def _api(api):
    return api if api is not None else _lib.api()


def bm25_score(term_freqs: np.ndarray, doc_lens: np.ndarray, avg_doc_lens: float, idf: float,
               k1: float, b: float, api=None) -> None:
    """In-place BM25 (reference searcharray/bm25/bm25.pyx:28-41; argument order of the .pyx)."""
    if term_freqs.dtype != np.float32 or not term_freqs.flags.c_contiguous:
        raise ValueError("term_freqs must be a contiguous float32 array")
    dl = as_f32(doc_lens)
    if len(dl) != len(term_freqs):
        raise ValueError("term_freqs and doc_lens must have the same length")
    _api(api).call("sa_bm25_score", p_f32(term_freqs), p_f32(dl), np.float32(avg_doc_lens),
                   np.float32(idf), np.float32(k1), np.float32(b), len(term_freqs))


def as_dense(indices, values, size: int, api=None) -> np.ndarray:
    """reference searcharray/roaringish/roaringish_ops.pyx:84-98."""
    indices, values = as_u64(indices), as_f32(values)
    if len(indices) != len(values):
        raise ValueError("indices and values must have the same length")
    out = np.empty(int(size), dtype=np.float32)
    _api(api).call("sa_as_dense", p_u64(indices), p_f32(values), len(indices), p_f32(out), int(size))
    return out


def popcount64_reduce(arr, key_shift, value_mask, api=None) -> Tuple[np.ndarray, np.ndarray]:
    """reference searcharray/roaringish/popcount.pyx:271-278."""
    arr = as_u64(arr)
    keys = np.empty(len(arr), dtype=np.uint64)
    counts = np.empty(len(arr), dtype=np.float32)
    n = _lib.c_int64(0)
    _api(api).call("sa_popcount64_reduce", p_u64(arr), len(arr), int(key_shift), int(value_mask),
                   p_u64(keys), p_f32(counts), n)
    return keys[:n.value].copy(), counts[:n.value].copy()


def unique(arr, rshift=0, api=None) -> np.ndarray:
    """reference searcharray/roaringish/unique.pyx:139-145."""
    arr = as_u64(arr)
    out = np.empty(len(arr), dtype=np.uint64)
    n = _lib.c_int64(0)
    _api(api).call("sa_unique", p_u64(arr), len(arr), int(rshift), p_u64(out), n)
    return out[:n.value].copy()


def popcount64(arr, api=None) -> np.ndarray:
    """reference searcharray/roaringish/popcount.pyx:119-121."""
    arr = as_u64(arr)
    out = np.empty(len(arr), dtype=np.uint64)
    _api(api).call("sa_popcount64", p_u64(arr), len(arr), p_u64(out))
    return out


def _check_mask(mask):
    if mask is None:
        return int(ALL_BITS)
    if int(mask) == 0:
        raise ValueError("Mask cannot be zero")                     # reference intersect.pyx:290-291
    return int(mask)


def intersect(lhs, rhs, mask=ALL_BITS, drop_duplicates=True, api=None) -> Tuple[np.ndarray, np.ndarray]:
    """reference searcharray/roaringish/intersect.pyx:278-320."""
    lhs, rhs = as_u64(lhs), as_u64(rhs)
    mask = _check_mask(mask)
    n = min(len(lhs), len(rhs)) if drop_duplicates else max(len(lhs), len(rhs))
    lo, ro = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
    nl, nr = _lib.c_int64(0), _lib.c_int64(0)
    _api(api).call("sa_intersect", p_u64(lhs), len(lhs), p_u64(rhs), len(rhs), mask, 1 if drop_duplicates else 0,
                   p_u64(lo), p_u64(ro), nl, nr)
    return lo[:nl.value].copy(), ro[:nr.value].copy()


def adjacent(lhs, rhs, mask=ALL_BITS, api=None) -> Tuple[np.ndarray, np.ndarray]:
    """reference searcharray/roaringish/intersect.pyx:323-343."""
    lhs, rhs = as_u64(lhs), as_u64(rhs)
    mask = _check_mask(mask)
    n = min(len(lhs), len(rhs))
    lo, ro = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
    k = _lib.c_int64(0)
    _api(api).call("sa_adjacent", p_u64(lhs), len(lhs), p_u64(rhs), len(rhs), mask, p_u64(lo), p_u64(ro), k)
    return lo[:k.value].copy(), ro[:k.value].copy()


def intersect_with_adjacents(lhs, rhs, mask=ALL_BITS, api=None):
    """reference searcharray/roaringish/intersect.pyx:346-390."""
    lhs, rhs = as_u64(lhs), as_u64(rhs)
    mask = _check_mask(mask)
    n = min(len(lhs), len(rhs))
    lo, ro, la, ra = (np.empty(n, dtype=np.uint64) for _ in range(4))
    k, ka = _lib.c_int64(0), _lib.c_int64(0)
    _api(api).call("sa_intersect_with_adjacents", p_u64(lhs), len(lhs), p_u64(rhs), len(rhs), mask,
                   p_u64(lo), p_u64(ro), k, p_u64(la), p_u64(ra), ka)
    return lo[:k.value].copy(), ro[:k.value].copy(), la[:ka.value].copy(), ra[:ka.value].copy()


def merge(lhs, rhs, drop_duplicates=False, api=None) -> np.ndarray:
    """reference searcharray/roaringish/merge.pyx:135-158."""
    lhs, rhs = as_u64(lhs), as_u64(rhs)
    out = np.empty(len(lhs) + len(rhs), dtype=np.uint64)
    k = _lib.c_int64(0)
    _api(api).call("sa_merge", p_u64(lhs), len(lhs), p_u64(rhs), len(rhs), 1 if drop_duplicates else 0, p_u64(out), k)
    return out[:k.value].copy()


def sort_merge_counts(lhs_ids, lhs_counts, rhs_ids, rhs_counts, api=None):
    """reference searcharray/roaringish/merge.pyx:221-232."""
    lhs_ids, rhs_ids = as_u64(lhs_ids), as_u64(rhs_ids)
    lhs_counts, rhs_counts = as_f32(lhs_counts), as_f32(rhs_counts)
    n = len(lhs_ids) + len(rhs_ids)
    oi, oc = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.float32)
    k = _lib.c_int64(0)
    _api(api).call("sa_sort_merge_counts", p_u64(lhs_ids), p_f32(lhs_counts), len(lhs_ids), p_u64(rhs_ids),
                   p_f32(rhs_counts), len(rhs_ids), p_u64(oi), p_f32(oc), k)
    return oi[:k.value].copy(), oc[:k.value].copy()


def popcount_reduce_at(ids, payload, api=None):
    """reference searcharray/roaringish/popcount.pyx:151-165."""
    ids, payload = as_u64(ids), as_u64(payload)
    if len(ids) != len(payload):
        raise ValueError("ids and payload must have the same length")
    oi, oc = np.empty(len(ids), dtype=np.uint64), np.empty(len(ids), dtype=np.float32)
    k = _lib.c_int64(0)
    _api(api).call("sa_popcount_reduce_at", p_u64(ids), p_u64(payload), len(ids), p_u64(oi), p_f32(oc), k)
    return oi[:k.value].copy(), oc[:k.value].copy()


def key_sum_over(ids, count, api=None):
    """reference searcharray/roaringish/popcount.pyx:194-204."""
    ids, count = as_u64(ids), as_u64(count)
    if len(ids) != len(count):
        raise ValueError("ids and count must have the same length")
    oi, oc = np.empty(len(ids), dtype=np.uint64), np.empty(len(ids), dtype=np.float32)
    k = _lib.c_int64(0)
    _api(api).call("sa_key_sum_over", p_u64(ids), p_u64(count), len(ids), p_u64(oi), p_f32(oc), k)
    return oi[:k.value].copy(), oc[:k.value].copy()


def payload_slice(arr, payload_msb_mask, min_payload=0, max_payload=0xFFFFFFFFFFFFFFFF, api=None) -> np.ndarray:
    """reference searcharray/roaringish/roaringish_ops.pyx:63-68."""
    arr = as_u64(arr)
    out = np.empty(len(arr), dtype=np.uint64)
    k = _lib.c_int64(0)
    _api(api).call("sa_payload_slice", p_u64(arr), len(arr), int(payload_msb_mask), int(min_payload),
                   int(max_payload) & 0xFFFFFFFFFFFFFFFF, p_u64(out), k)
    return out[:k.value].copy()


def span_search(posns, lengths, phrase_freqs, slop, key_mask=None, header_mask=None, key_bits=None, lsb_bits=None,
                api=None) -> None:
    """``span_search`` of the reference (roaringish/spans.pyx:322-330): walk the terms' candidate words --
    term t = ``posns[lengths[t]:lengths[t + 1]]`` -- with the slop span machine and add each document's
    count into ``phrase_freqs`` (a Counter / dict-like).  Only the default 28 / 18 / 18 layout is
    implemented on the device."""
    if key_bits is not None and int(key_bits) != 28 or lsb_bits is not None and int(lsb_bits) != 18:
        raise NotImplementedError("span_search on the device supports the default 28/18/18 layout only")
    if key_mask is not None and int(key_mask) != 0xFFFFFFF000000000 or \
            header_mask is not None and int(header_mask) != 0xFFFFFFFFFFFC0000:
        raise NotImplementedError("span_search on the device supports the default 28/18/18 layout only")
    posns, lengths = as_u64(posns), as_u64(lengths)
    n_max = max(1, int(lengths[-1] - lengths[0]))
    docs = np.empty(n_max, dtype=np.uint64)
    counts = np.empty(n_max, dtype=np.uint64)
    n = ctypes.c_int64(0)
    _api(api).call("sa_span_search", p_u64(posns), p_u64(lengths), len(lengths) - 1, int(slop), p_u64(docs),
                   p_u64(counts), ctypes.byref(n))
    for d, c in zip(docs[:n.value].tolist(), counts[:n.value].tolist()):
        phrase_freqs[d] += c

"""Kernel-level entry points with the reference's names and numpy signatures.

Each function has the same arguments, result types and error behaviour as the Cython
function it replaces (cited per function) but runs on the GPU through the C ABI (Part 1 of
include/searcharray_hip.h).  They exist for drop-in rebinding of the reference's call sites
 and for kernel-level parity tests; the resident-index classes in
``device_index`` are the fast path.
"""
from __future__ import annotations

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

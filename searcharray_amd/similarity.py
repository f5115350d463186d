"""Similarity functions with the reference's protocol
``similarity(term_freqs f32[N], doc_freqs, doc_lens f32[N], avg_doc_lens, num_docs) -> ndarray``
(reference searcharray/similarity.py:8-16).

``bm25_similarity`` closures are tagged (``.kind == "bm25"``, ``.k1``, ``.b``) so
``SearchArray.score`` can run them entirely on the GPU.  Called directly (protocol use, e.g. by a
caller that already holds term frequencies) they apply the BM25 kernel to the given arrays through
the C ABI.  The other stock similarities are tagged the same way (``bm25_impact``, ``bm25_legacy``,
``classic``) and run on the device from ``SearchArray.score``; called directly they are the
reference's numpy expressions, which is also what the device kernels reproduce bit for bit.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from . import ops


# the reference's Similarity protocol (similarity.py:8-16): any
# callable(term_freqs, doc_freqs, doc_lens, avg_doc_lens, num_docs) -> ndarray
Similarity = Callable[..., np.ndarray]


def compute_idf(num_docs, dfs):
    """reference similarity.py:19-21 (float64 numpy math)."""
    dfs = np.asarray(dfs)
    return np.sum(np.log(1 + (num_docs - dfs + 0.5) / (dfs + 0.5)))


def bm25_similarity(k1: float = 1.2, b: float = 0.75):
    """BM25 as in Lucene 9 (reference similarity.py:24-38); mutates and returns term_freqs."""
    def bm25(term_freqs, doc_freqs, doc_lens, avg_doc_lens, num_docs):
        if avg_doc_lens == 0:
            return np.zeros_like(term_freqs)
        idf = compute_idf(num_docs, doc_freqs)
        ops.bm25_score(term_freqs, doc_lens, avg_doc_lens, idf, k1, b)
        return term_freqs
    bm25.kind = "bm25"
    bm25.k1 = float(k1)
    bm25.b = float(b)
    return bm25


def bm25_impact(k1: float = 1.2, b: float = 0.75):
    """BM25 without the idf factor (reference similarity.py:41-53)."""
    def bm25(term_freqs, doc_freqs, doc_lens, avg_doc_lens, num_docs):
        if avg_doc_lens == 0:
            return np.zeros_like(term_freqs)
        return term_freqs / (term_freqs + k1 * (1 - b + b * doc_lens / avg_doc_lens))
    bm25.kind = "bm25_impact"
    bm25.k1 = float(k1)
    bm25.b = float(b)
    return bm25


def bm25_legacy_similarity(k1: float = 1.2, b: float = 0.75):
    """BM25 with (k1 + 1) in the numerator (reference similarity.py:56-71)."""
    def bm25(term_freqs, doc_freqs, doc_lens, avg_doc_lens, num_docs):
        if avg_doc_lens == 0:
            return np.zeros_like(term_freqs)
        idf = compute_idf(num_docs, doc_freqs)
        tf = (term_freqs * (k1 + 1)) / (term_freqs + k1 * (1 - b + b * doc_lens / avg_doc_lens))
        return idf * tf
    bm25.kind = "bm25_legacy"
    bm25.k1 = float(k1)
    bm25.b = float(b)
    return bm25


def classic_similarity():
    """Classic Lucene TF-IDF (reference similarity.py:74-89)."""
    def classic(term_freqs, doc_freqs, doc_lens, avg_doc_lens, num_docs):
        sum_dfs = np.sum(doc_freqs, axis=0)
        idf = np.log((num_docs + 1) / (sum_dfs + 1)) + 1
        length_norm = 1.0 / np.sqrt(doc_lens)
        return idf * np.sqrt(term_freqs) * length_norm
    classic.kind = "classic"
    return classic


default_bm25 = bm25_similarity()

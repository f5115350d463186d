"""Roaringish wire format (host side).

The index is, per term, a sorted ``uint64[]`` of words

    [63:36] doc id (28 b) | [35:18] position // 18 (18 b) | [17:0] bitmap of position % 18

exactly the layout of the reference (searcharray/roaringish/roaringish.py:30-35,54-86),
so arrays produced here can be handed to the reference and vice versa.  This module
only holds the layout constants and the host-side encoder used by index build; every
query-time operation on these words runs on the GPU (searcharray_amd/csrc/).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

KEY_BITS = 28
KEY_SHIFT = 36
LSB_BITS = 18
MSB_BITS = 18
KEY_MASK = np.uint64(0xFFFFFFF000000000)
PAYLOAD_MSB_MASK = np.uint64(0x0000000FFFFC0000)
PAYLOAD_LSB_MASK = np.uint64(0x000000000003FFFF)
HEADER_MASK = np.uint64(0xFFFFFFFFFFFC0000)
MAX_POSN = (1 << LSB_BITS) - 1          # reference: roaringish.py:86, middle_out.py:41
MAX_DOC_ID = (1 << KEY_BITS) - 1


def encode_sorted(term_ids: np.ndarray, doc_ids: np.ndarray, posns: np.ndarray
                  ) -> Tuple[np.ndarray, np.ndarray]:
    """Pack (term, doc, posn) triples -- already sorted by term, doc, posn -- into words.

    Returns ``(words, word_terms)``: one word per distinct (term, doc, posn // 18) with the
    position bits OR-ed together, and the term id each word belongs to.  Same result as
    RoaringishEncoder.encode with term boundaries (reference roaringish.py:93-142).
    """
    n = len(term_ids)
    if n == 0:
        return np.empty(0, np.uint64), np.empty(0, np.uint32)
    posns = np.asarray(posns)
    if int(posns.max()) > MAX_POSN:
        raise ValueError(f"Positions must be less than {1 << LSB_BITS}")
    doc_ids = np.asarray(doc_ids, dtype=np.uint64)
    if int(doc_ids.max()) > MAX_DOC_ID:
        raise ValueError(f"Doc ids must be less than {1 << KEY_BITS}")
    p32 = posns.astype(np.uint32)
    blk = p32 // np.uint32(LSB_BITS)                      # 32-bit division is ~3x the 64-bit rate
    bit = (p32 - blk * np.uint32(LSB_BITS)).astype(np.uint64)
    word = (doc_ids << np.uint64(KEY_SHIFT)) | (blk.astype(np.uint64) << np.uint64(MSB_BITS))
    first = np.empty(n, dtype=bool)
    first[0] = True
    np.not_equal(word[1:], word[:-1], out=first[1:])
    term_ids = np.asarray(term_ids)
    first[1:] |= term_ids[1:] != term_ids[:-1]
    word |= np.uint64(1) << bit
    starts = np.flatnonzero(first)
    words = np.bitwise_or.reduceat(word.view(np.int64), starts).view(np.uint64)
    return words, term_ids[starts].astype(np.uint32)


def term_offsets(word_terms: np.ndarray, num_terms: int) -> np.ndarray:
    """CSR offsets ``uint64[num_terms + 1]`` over a term-sorted word array (the layout of the
    reference's ArrayDict, phrase/memmap_arrays.py:15-56)."""
    counts = np.bincount(word_terms, minlength=num_terms)
    off = np.zeros(num_terms + 1, dtype=np.uint64)
    np.cumsum(counts, out=off[1:])
    return off


def decode_positions(words: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Expand words into (doc_ids, posns), sorted by doc then posn (host utility for
    ``SearchArray.positions`` / ``__getitem__``; reference roaringish.py:144-166)."""
    words = np.asarray(words, dtype=np.uint64)
    if len(words) == 0:
        return np.empty(0, np.uint64), np.empty(0, np.uint64)
    docs = words >> np.uint64(KEY_SHIFT)
    blk = (words & PAYLOAD_MSB_MASK) >> np.uint64(MSB_BITS)
    payload = (words & PAYLOAD_LSB_MASK).astype(np.uint32)
    bits = ((payload[:, None] >> np.arange(LSB_BITS, dtype=np.uint32)[None, :]) & 1).astype(bool)
    rows, cols = np.nonzero(bits)
    return docs[rows], blk[rows] * np.uint64(LSB_BITS) + cols.astype(np.uint64)

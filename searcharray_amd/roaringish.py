"""Roaringish wire format (host side).

The index is, per term, a sorted ``uint64[]`` of words

    [63:36] doc id (28 b) | [35:18] position // 18 (18 b) | [17:0] bitmap of position % 18

exactly the layout of the reference (searcharray/roaringish/roaringish.py:30-35,54-86),
so arrays produced here can be handed to the reference and vice versa.  This module
only holds the layout constants and the host-side encoder used by index build; every
query-time operation on these words runs on the GPU (searcharray_amd/csrc/).
"""
from __future__ import annotations

from typing import Optional, Tuple

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


# ---------------------------------------------------------------------------------------------
# The reference's encoder object (searcharray/roaringish/roaringish.py:54-282) for arbitrary key
# widths.  Bit arithmetic is numpy; every set operation (intersect / adjacent / merge / unique /
# popcount-reduce / payload slice) runs on the GPU through ``ops`` (C ABI Part 1).
# ---------------------------------------------------------------------------------------------
DEFAULT_KEY_BITS = np.uint64(KEY_BITS)
DEFAULT_KEY_MASK = KEY_MASK
DEFAULT_PAYLOAD_MSB_MASK = PAYLOAD_MSB_MASK
DEFAULT_PAYLOAD_MSB_BITS = np.uint64(MSB_BITS)
DEFAULT_PAYLOAD_LSB_MASK = PAYLOAD_LSB_MASK
DEFAULT_PAYLOAD_LSB_BITS = np.uint64(LSB_BITS)


def n_msb_mask(n) -> np.uint64:
    """A mask of the ``n`` most significant of 64 bits (reference roaringish.py:45-47)."""
    n = int(n)
    return np.uint64(((1 << n) - 1) << (64 - n)) if n else np.uint64(0)


def sorted_unique(arr: np.ndarray) -> np.ndarray:
    from . import ops
    return ops.unique(arr)


def convert_keys(keys) -> np.ndarray:
    """A number, list, array or range of keys as uint64 (reference roaringish.py:285-299 -- including
    its treatment of a non-empty ``range``: ``arange(first, last + 1) + first``)."""
    import numbers
    if isinstance(keys, numbers.Number):
        return np.asarray([keys], dtype=np.uint64)
    if isinstance(keys, list):
        return np.asarray(keys, dtype=np.uint64)
    if isinstance(keys, np.ndarray):
        return keys.astype(np.uint64)
    if isinstance(keys, range):
        if len(keys) == 0:
            return np.asarray([], dtype=np.uint64)
        return np.arange(keys[0], keys[-1] + 1, dtype=np.uint64) + keys[0]
    raise ValueError(f"Unknown type for keys: {type(keys)}")


class RoaringishEncoder:
    """key -> sorted integer set packed in uint64 words ``| key | payload // L | 1 << payload % L |`` with
    ``key_bits`` key bits and the remaining bits split evenly into the block number (msb) and the
    L-bit bitmap (lsb).  Same attributes and methods as the reference's class."""

    def __init__(self, key_bits: np.uint64 = DEFAULT_KEY_BITS):
        key_bits = np.uint64(key_bits)
        payload_bits = np.uint64(64) - key_bits
        self.key_bits = key_bits
        self.payload_msb_bits = payload_bits // np.uint64(2)
        self.payload_lsb_bits = np.uint64(payload_bits - self.payload_msb_bits)
        self.key_mask = n_msb_mask(key_bits)
        self.header_bits = key_bits + self.payload_msb_bits
        self.payload_msb_mask = n_msb_mask(self.header_bits) & ~self.key_mask
        self.payload_lsb_mask = (np.uint64(1) << self.payload_lsb_bits) - np.uint64(1)
        self.header_mask = self.key_mask | self.payload_msb_mask
        self.max_payload = np.uint64(2 ** int(self.payload_lsb_bits) - 1)
        self._key_shift = np.uint64(64) - key_bits

    def validate_payload(self, payload: np.ndarray):
        if np.any(payload > self.max_payload):
            raise ValueError(f"Positions must be less than {2 ** int(self.payload_lsb_bits)}")

    def encode(self, payload: np.ndarray, keys: Optional[np.ndarray] = None,
               boundaries: Optional[np.ndarray] = None):
        """Pack sorted ``payload`` values (per key, keys ascending) into words; a new word starts when
        (key, payload // L) changes or at a ``boundaries`` index (several sets -- terms -- encoded in
        one call).  Returns ``(words, word index of every boundary + [n_words])`` or ``(words, None)``
        (reference roaringish.py:93-142)."""
        payload = np.asarray(payload)
        L = self.payload_lsb_bits
        head = (payload // L).astype(np.uint64) << self.payload_msb_bits
        if keys is not None:
            head |= np.asarray(keys).astype(np.uint64) << self._key_shift
        n = len(head)
        starts = np.zeros(n, dtype=bool)
        if n:
            starts[0] = True
            np.not_equal(head[1:], head[:-1], out=starts[1:])
        new_boundaries = None
        if boundaries is not None:
            b = np.unique(np.asarray(boundaries).astype(np.int64))
            starts[b[b < n]] = True
        first = np.flatnonzero(starts)
        if boundaries is not None:
            new_boundaries = np.concatenate([np.searchsorted(first, b[b < n]), [len(first)]]).astype(np.uint64)
        if n == 0:
            return head, new_boundaries
        words = head | (np.uint64(1) << (payload % L).astype(np.uint64))
        return np.bitwise_or.reduceat(words.view(np.int64), first).view(np.uint64), new_boundaries

    def decode(self, encoded: np.ndarray, get_keys: bool = True):
        """``[(key, payload values ascending), ...]`` by ascending key, or just the value arrays
        (reference roaringish.py:144-166)."""
        encoded = np.asarray(encoded, dtype=np.uint64)
        if len(encoded) == 0:
            return [] if get_keys else [np.empty(0, dtype=np.uint64)]
        L = int(self.payload_lsb_bits)
        bits = ((encoded[:, None] >> np.arange(L, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        word, bit = np.nonzero(bits)                          # row-major: by word, then by bit
        keys = (encoded >> self._key_shift)[word]
        values = self.payload_msb(encoded)[word] * self.payload_lsb_bits + bit.astype(np.uint64)
        order = np.lexsort((values, keys))
        keys, values = keys[order], values[order]
        uniq, first = np.unique(keys, return_index=True)
        groups = np.split(values, first[1:])
        return list(zip(uniq, groups)) if get_keys else groups

    def num_values_per_key(self, encoded: np.ndarray):
        from . import ops
        return ops.popcount64_reduce(encoded, self._key_shift, self.payload_lsb_mask)

    def keys(self, encoded: np.ndarray) -> np.ndarray:
        return encoded >> self._key_shift

    def keys_unique(self, encoded: np.ndarray) -> np.ndarray:
        from . import ops
        return ops.unique(encoded, self._key_shift)

    def payload_msb(self, encoded: np.ndarray) -> np.ndarray:
        return (encoded & self.payload_msb_mask) >> self.payload_msb_bits

    def payload_lsb(self, encoded: np.ndarray) -> np.ndarray:
        return encoded & self.payload_lsb_mask

    def header(self, encoded: np.ndarray) -> np.ndarray:
        return encoded & ~self.payload_lsb_mask

    def intersect_candidates(self, lhs: np.ndarray, rhs: np.ndarray):
        """Words with equal headers and words whose headers differ by one block, in one pass
        (reference roaringish.py:193-198)."""
        from . import ops
        li, ri, la, ra = ops.intersect_with_adjacents(lhs, rhs, mask=self.header_mask)
        return lhs[li], rhs[ri], lhs[la], rhs[ra]

    def intersect_rshift(self, lhs: np.ndarray, rhs: np.ndarray, rshift=np.int64(-1)):
        """Words of ``lhs`` whose header + 1 block is a header of ``rhs`` (reference roaringish.py:200-213;
        ``rshift`` is unused there too)."""
        from . import ops
        li, ri = ops.adjacent(lhs, rhs, mask=self.header_mask)
        return lhs[li], rhs[ri]

    def intersect(self, lhs: np.ndarray, rhs: np.ndarray):
        from . import ops
        li, ri = ops.intersect(lhs, rhs, mask=self.header_mask)
        return lhs[li], rhs[ri]

    def key_partition(self, encoded: np.ndarray, max_key, num_partitions=2) -> np.ndarray:
        """Indices that cut ``encoded`` at keys ``max_key * i // num_partitions`` -- the doc-range shard
        boundaries (reference roaringish.py:227-243)."""
        keys = np.asarray(encoded, dtype=np.uint64) & self.key_mask
        cuts = [np.uint64(0)]
        for i in range(num_partitions - 1):
            target = np.uint64(int(max_key) * (i + 1) // num_partitions) << self._key_shift
            cuts.append(np.uint64(np.searchsorted(keys, target, side="left")))
        cuts.append(np.uint64(len(encoded)))
        return np.asarray(cuts, dtype=np.uint64)

    def slice(self, encoded: np.ndarray, keys: Optional[np.ndarray] = None, header: Optional[np.ndarray] = None,
              max_payload: Optional[int] = None, min_payload: Optional[int] = None) -> np.ndarray:
        """The words of ``encoded`` with a key in ``keys`` (or a header in ``header``), optionally
        restricted to payload blocks ``[min_payload, max_payload]`` (reference roaringish.py:245-282)."""
        from . import ops
        if header is not None:
            if keys is not None:
                raise ValueError("Can't specify both keys and header")
            _, idx = ops.intersect(header.view(np.uint64), self.header(encoded).view(np.uint64), drop_duplicates=False)
            encoded = encoded[idx]
        if keys is not None:
            _, idx = ops.intersect(keys.view(np.uint64), self.keys(encoded).view(np.uint64), drop_duplicates=False)
            encoded = encoded[idx]
        if max_payload is None and min_payload is None:
            return encoded
        L = int(self.payload_lsb_bits)
        if min_payload is not None and min_payload % L != 0:
            raise ValueError(f"min_payload must be a multiple of {L}")
        if max_payload is not None and max_payload % L != L - 1:
            raise ValueError(f"max_payload must be a multiple of {L} - 1")
        lo = 0 if min_payload is None else min_payload
        hi = 0xFFFFFFFFFFFFFFFF if max_payload is None else max_payload
        return ops.payload_slice(encoded, self.payload_msb_mask, lo // L, hi // L)


# the rest of the reference package's surface (searcharray/roaringish/__init__.py:1-7), GPU-backed
from .ops import (adjacent, intersect, key_sum_over, merge, popcount64, popcount_reduce_at,   # noqa: E402,F401
                  sort_merge_counts, span_search, unique)

"""Host-side index build: tokenise -> (term, doc, posn) triples -> roaringish words.

The tokenizer is arbitrary Python and stays on the host together with the term dictionary; the
token stream it produces is sorted by term and roaringish-encoded ON THE DEVICE
(``DeviceIndex.from_tokens`` -> csrc/sa_build.hip).  The encoded index -- term-major roaringish
words, CSR offsets, doc lengths -- is byte-compatible with what the reference's indexer produces
(reference searcharray/indexing.py:118-160,235-296); ``HostIndex`` downloads it lazily when the host
needs the words.  Already-tokenised input (``Terms`` / dicts with explicit positions) is encoded on
the host and uploaded.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional

import numpy as np

from . import roaringish as rz
from .term_dict import TermDict


class HostIndex:
    """Host side of an index.  Always present: term dictionary and doc lengths.  The roaringish words
    and the doc -> distinct-terms CSR are materialised on first use: an index built from a token
    stream is encoded ON THE DEVICE (DeviceIndex.from_tokens) and only downloaded if the host asks for
    the words (positions(), scalar __getitem__, pickling)."""

    def __init__(self, term_dict: TermDict, doc_lens: np.ndarray, words: Optional[np.ndarray] = None,
                 term_off: Optional[np.ndarray] = None, doc_term_ptr: Optional[np.ndarray] = None,
                 doc_term_ids: Optional[np.ndarray] = None, doc_term_tfs: Optional[np.ndarray] = None,
                 tokens: Optional[np.ndarray] = None, doc_ptr: Optional[np.ndarray] = None):
        self.term_dict = term_dict
        self.doc_lens = doc_lens            # float32[N]
        self._words = words                 # uint64[W] term-major
        self._term_off = term_off           # uint64[V+1]
        self._doc_term_ptr = doc_term_ptr   # int64[N+1]  CSR over docs -> distinct term ids (for __getitem__)
        self._doc_term_ids = doc_term_ids   # uint32[...]
        self.doc_term_tfs = doc_term_tfs    # only for docs given as {term: tf} without positions
        self.tokens = tokens                # uint32[n_tokens] term id per token, docs back to back (or None)
        self.doc_ptr = doc_ptr              # uint64[N+1] token offsets
        self.words_source: Optional[Callable] = None     # () -> (words, term_off), set by the device owner
        # (path, term_src_off uint64[V], term_len uint64[V]) when the words live in a raw uint64 file --
        # the reference's MemoryMappedArrays .dat (phrase/memmap_arrays.py:145-165)
        self.words_file: Optional[tuple] = None

    @property
    def num_docs(self) -> int:
        return len(self.doc_lens)

    @property
    def has_words(self) -> bool:
        return self._words is not None

    def _materialise_words(self):
        if self._words is not None:
            return
        if self.words_source is not None:
            self._words, self._term_off = self.words_source()
            return
        if self.words_file is not None:
            path, src, length = self.words_file
            off = np.zeros(len(length) + 1, dtype=np.uint64)
            np.cumsum(length, out=off[1:])
            mm = np.memmap(path, dtype=np.uint64, mode="r") if os.path.getsize(path) else np.empty(0, np.uint64)
            if np.array_equal(src, off[:-1]):
                words = mm                                  # terms back to back in id order: map, do not copy
            else:
                words = np.empty(int(off[-1]), dtype=np.uint64)
                for t in np.flatnonzero(length):
                    words[int(off[t]):int(off[t + 1])] = mm[int(src[t]):int(src[t]) + int(length[t])]
            self._words, self._term_off = words, off
            return
        # no device copy yet: encode on the host (same bytes)
        lens = np.diff(self.doc_ptr.astype(np.int64))
        if len(self.tokens):
            t, d, p = _triples(lens, self.tokens)
            words, word_terms = rz.encode_sorted(t, d, p)
        else:
            words, word_terms = np.empty(0, np.uint64), np.empty(0, np.uint32)
        self._words, self._term_off = words, rz.term_offsets(word_terms, len(self.term_dict))

    @property
    def words(self) -> np.ndarray:
        self._materialise_words()
        return self._words

    @property
    def term_off(self) -> np.ndarray:
        self._materialise_words()
        return self._term_off

    def _materialise_doc_terms(self):
        if self._doc_term_ptr is not None:
            return
        if self.tokens is None:                             # only the words exist (index read from a file)
            words, off = self.words, self.term_off
            terms = np.repeat(np.arange(len(off) - 1, dtype=np.uint64), np.diff(off.astype(np.int64)))
            docs = np.asarray(words) >> np.uint64(rz.KEY_SHIFT)
            self._doc_term_ptr, self._doc_term_ids = _csr_doc_terms(terms, docs, len(self.doc_lens))
            return
        lens = np.diff(self.doc_ptr.astype(np.int64))
        docs = np.repeat(np.arange(len(lens), dtype=np.uint64), lens)
        self._doc_term_ptr, self._doc_term_ids = _csr_doc_terms(self.tokens, docs, len(lens))

    @property
    def doc_term_ptr(self) -> np.ndarray:
        self._materialise_doc_terms()
        return self._doc_term_ptr

    @property
    def doc_term_ids(self) -> np.ndarray:
        self._materialise_doc_terms()
        return self._doc_term_ids

    def __getstate__(self):
        state = dict(self.__dict__)
        state["words_source"] = None
        if self.words_file is not None:
            # like the reference's MemoryMappedArrays (memmap_arrays.py:196-208): the pickle carries the
            # filename and the per-term metadata, never the words
            state["_words"] = None
            state["_term_off"] = None
        else:
            self._materialise_words()               # pickles are self-contained
            state["_words"], state["_term_off"] = self._words, self._term_off
        return state


def _csr_doc_terms(terms: np.ndarray, docs: np.ndarray, n_docs: int):
    """distinct (doc, term) pairs as a CSR over docs"""
    if len(terms) == 0:
        return np.zeros(n_docs + 1, dtype=np.int64), np.empty(0, np.uint32)
    key = (docs.astype(np.uint64) << np.uint64(32)) | terms.astype(np.uint64)
    key = np.unique(key)
    d = (key >> np.uint64(32)).astype(np.int64)
    ptr = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(np.bincount(d, minlength=n_docs), out=ptr[1:])
    return ptr, (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def build_from_token_ids(term_dict: TermDict, doc_tokens: List[np.ndarray]) -> HostIndex:
    """doc_tokens[d] = term ids of doc d in position order.  Only the token stream is assembled here;
    sorting by term and the roaringish encoding run on the device when the index is first used."""
    n_docs = len(doc_tokens)
    lens = np.fromiter((len(t) for t in doc_tokens), dtype=np.int64, count=n_docs)
    total = int(lens.sum())
    tokens = np.concatenate(doc_tokens).astype(np.uint32) if total else np.empty(0, np.uint32)
    doc_ptr = np.zeros(n_docs + 1, dtype=np.uint64)
    np.cumsum(lens, out=doc_ptr[1:])
    return HostIndex(term_dict, lens.astype(np.float32), tokens=tokens, doc_ptr=doc_ptr)


def _triples(lens: np.ndarray, terms: np.ndarray):
    from .synth import tokens_to_triples
    return tokens_to_triples(lens, terms)


def _batches(array: Iterable, batch_size: int):
    """lists of at most batch_size docs, in order (the reference's batch_iterator, indexing.py:176-188)"""
    batch: list = []
    for doc in array:
        batch.append(doc)
        if len(batch) >= batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


def build_index_from_tokenizer(array: Iterable, tokenizer: Callable, truncate: bool = False,
                               batch_size: int = 100000, workers: int = 4) -> HostIndex:
    """reference indexing.py:235-296: the docs are taken ``batch_size`` at a time; a batch's token ids end as ONE uint32
    array (4 bytes per token) and its Python objects are dropped before the next batch is touched, so the transient host
    memory is a batch's, not the collection's (the reference's design size of 1 M docs is ~32 M tokens: > 1 GB as Python
    ints in one list, 128 MB as these arrays).  ``workers`` > 1 runs the TOKENIZER on a thread pool, as the reference does:
    a batch is cut into ``workers`` pieces that are tokenized side by side, and at most TWO batches exist as Python lists at
    any time (the one being taken and the one being tokenized) -- not ``workers`` of them.  Term ids are still assigned by
    this thread in document order -- unlike the
    reference's (its workers race for the shared dictionary, indexing.py:190-209), so an index does not depend on the
    thread timing.  Host work is the tokenizer and the term dictionary only; the result carries the token stream, which
    is sorted and roaringish-encoded on the device (csrc/sa_build.hip)."""
    if batch_size < 1:
        raise ValueError("batch_size must be positive")
    term_dict = TermDict()
    add_terms = term_dict.add_terms
    max_posn = rz.MAX_POSN
    chunks: List[np.ndarray] = []
    len_chunks: List[np.ndarray] = []

    def take(tokenized: list):
        """one batch of tokenized docs -> its ids and lengths as arrays"""
        flat: List[int] = []
        lens = np.empty(len(tokenized), dtype=np.int64)
        for i, toks in enumerate(tokenized):
            ids = add_terms(toks)
            if len(ids) > max_posn:
                if truncate:
                    ids = ids[:max_posn]                    # reference indexing.py:120-122,76-78
                else:
                    raise ValueError(f"Document length exceeds maximum of {max_posn}")    # indexing.py:141-142
            flat.extend(ids)
            lens[i] = len(ids)
        chunks.append(np.asarray(flat, dtype=np.uint32) if flat else np.empty(0, np.uint32))
        len_chunks.append(lens)

    def tokenize(batch: list) -> list:
        return [tokenizer(doc) for doc in batch]

    if workers is None or workers <= 1:
        for batch in _batches(array, batch_size):
            take(tokenize(batch))
    else:
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        pending: deque = deque()

        def submit(batch: list) -> list:
            step = (len(batch) + workers - 1) // workers
            return [pool.submit(tokenize, batch[i:i + step]) for i in range(0, len(batch), step)]

        def collect(parts: list) -> list:
            out: list = []
            for f in parts:
                out.extend(f.result())
            return out

        with ThreadPoolExecutor(max_workers=workers) as pool:
            for batch in _batches(array, batch_size):
                pending.append(submit(batch))
                if len(pending) >= 2:                       # (the next batch is tokenized while this one is taken)
                    take(collect(pending.popleft()))
            while pending:
                take(collect(pending.popleft()))
    lens_a = np.concatenate(len_chunks) if len_chunks else np.empty(0, np.int64)
    tokens = np.concatenate(chunks) if chunks else np.empty(0, np.uint32)
    chunks.clear()
    n_docs = len(lens_a)
    doc_ptr = np.zeros(n_docs + 1, dtype=np.uint64)
    np.cumsum(lens_a, out=doc_ptr[1:])
    return HostIndex(term_dict, lens_a.astype(np.float32), tokens=tokens, doc_ptr=doc_ptr)


def build_index_from_terms_list(postings, Terms) -> HostIndex:
    """Index already-tokenised docs: ``Terms`` objects or ``{term: tf}`` dicts
    (reference indexing.py:298-342).  Positions are indexed when the doc carries them; a doc
    without positions contributes to equality / __getitem__ only (its term frequencies cannot be
    searched), exactly as in the reference."""
    term_dict = TermDict()
    t_l, d_l, p_l = [], [], []
    doc_lens = []
    ptr = [0]
    ids: List[int] = []
    tfs: List[int] = []
    for doc_id, tokenized in enumerate(list(postings)):
        if isinstance(tokenized, dict):
            tokenized = Terms(tokenized, doc_len=len(tokenized))
        elif not isinstance(tokenized, Terms):
            raise TypeError("Expected a Terms or a dict")
        doc_lens.append(tokenized.doc_len)
        for token, tf in tokenized.terms():
            tid = term_dict.add_term(token)
            ids.append(tid)
            tfs.append(tf)
            positions = tokenized.positions(token) if tokenized.posns is not None else None
            if positions is not None and len(positions) > 0:
                pos = np.asarray(positions, dtype=np.int64)
                t_l.append(np.full(len(pos), tid, dtype=np.int64))
                d_l.append(np.full(len(pos), doc_id, dtype=np.int64))
                p_l.append(pos)
        ptr.append(len(ids))
    n_docs = len(doc_lens)
    V = len(term_dict)
    if t_l:
        t = np.concatenate(t_l); d = np.concatenate(d_l); p = np.concatenate(p_l)
        order = np.lexsort((p, d, t))
        words, word_terms = rz.encode_sorted(t[order].astype(np.uint32), d[order].astype(np.uint64), p[order].astype(np.uint64))
    else:
        words, word_terms = np.empty(0, np.uint64), np.empty(0, np.uint32)
    return HostIndex(term_dict, np.asarray(doc_lens, dtype=np.float32), words=words,
                     term_off=rz.term_offsets(word_terms, V), doc_term_ptr=np.asarray(ptr, dtype=np.int64),
                     doc_term_ids=np.asarray(ids, dtype=np.uint32), doc_term_tfs=np.asarray(tfs, dtype=np.float64))

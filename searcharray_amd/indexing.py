"""Host-side index build: tokenise -> (term, doc, posn) triples -> roaringish words.

Index build stays on the host in this version (the tokenizer is arbitrary Python; SURVEY.md
8f ranks device-side encode as the next step).  The output -- term-major roaringish words, CSR
offsets, doc lengths -- is byte-compatible with what the reference's indexer produces
(reference searcharray/indexing.py:118-160,235-296) and is what ``DeviceIndex`` uploads.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional

import numpy as np

from . import roaringish as rz
from .term_dict import TermDict


@dataclass
class HostIndex:
    term_dict: TermDict
    words: np.ndarray           # uint64[W] term-major
    term_off: np.ndarray        # uint64[V+1]
    doc_lens: np.ndarray        # float32[N]
    doc_term_ptr: np.ndarray    # int64[N+1]   CSR over docs -> distinct term ids (for __getitem__)
    doc_term_ids: np.ndarray    # uint32[...]
    doc_term_tfs: Optional[np.ndarray] = None     # only for docs given as {term: tf} without positions

    @property
    def num_docs(self) -> int:
        return len(self.doc_lens)


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
    """doc_tokens[d] = term ids of doc d in position order"""
    n_docs = len(doc_tokens)
    lens = np.fromiter((len(t) for t in doc_tokens), dtype=np.int64, count=n_docs)
    total = int(lens.sum())
    if total:
        terms = np.concatenate(doc_tokens).astype(np.uint32)
        t, d, p = _triples(lens, terms)
        words, word_terms = rz.encode_sorted(t, d, p)
    else:
        terms = np.empty(0, np.uint32)
        t = np.empty(0, np.uint32); d = np.empty(0, np.uint64)
        words, word_terms = np.empty(0, np.uint64), np.empty(0, np.uint32)
    V = len(term_dict)
    term_off = rz.term_offsets(word_terms, V)
    ptr, ids = _csr_doc_terms(t, d, n_docs)
    return HostIndex(term_dict, words, term_off, lens.astype(np.float32), ptr, ids)


def _triples(lens: np.ndarray, terms: np.ndarray):
    from .synth import tokens_to_triples
    return tokens_to_triples(lens, terms)


def build_index_from_tokenizer(array: Iterable, tokenizer: Callable, truncate: bool = False,
                               batch_size: int = 100000) -> HostIndex:
    """reference indexing.py:235-296 (single pass; batch_size only bounds the token buffers)."""
    term_dict = TermDict()
    doc_tokens: List[np.ndarray] = []
    max_posn = rz.MAX_POSN
    for doc in array:
        toks = [term_dict.add_term(tok) for tok in tokenizer(doc)]
        if len(toks) > max_posn:
            if truncate:
                toks = toks[:max_posn]                      # reference indexing.py:120-122,76-78
            else:
                raise ValueError(f"Document length exceeds maximum of {max_posn}")    # indexing.py:141-142
        doc_tokens.append(np.asarray(toks, dtype=np.uint32))
    return build_from_token_ids(term_dict, doc_tokens)


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
    return HostIndex(term_dict, words, rz.term_offsets(word_terms, V), np.asarray(doc_lens, dtype=np.float32),
                     np.asarray(ptr, dtype=np.int64), np.asarray(ids, dtype=np.uint32),
                     np.asarray(tfs, dtype=np.float64))

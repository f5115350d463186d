"""N GPUs behind ONE handle: the doc-range-sharded index of a single process.

Documents are independent units of this path -- tf, doc length and phrase matches are per doc, and the
roaringish key of a word is its doc id, so a doc-id range is a contiguous slice of every term's word
list (what the reference's ``key_partition`` computes, searcharray/roaringish/roaringish.py:227-243).
Device g holds docs ``[g*N/G, (g+1)*N/G)`` with shard-local ids; BM25 uses the GLOBAL corpus size, average
doc length and document frequencies on every shard (the reference computes them over the whole corpus:
indexing.py:282-284, middle_out.py:521-528).  A query batch is scored on all shards concurrently (one
host thread per device INSIDE the library, csrc/sa_sharded.hip -- Part 3b of the C ABI --, each shard on its own
HIP streams); the only data-path exchange is the RCCL all-gather of the per-shard top-k keys followed by a merge on
every device (csrc/sa_comm.hip), so ``fetch`` reads the final result from shard 0.  Dense drop-in results
(``score`` / ``termfreqs``) are per-shard vectors concatenated in doc order -- no collective.

``bench.py --gpus N`` uses the same library entry points with one PROCESS per GPU.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import options as _options
from . import roaringish as rz
from .device_index import NO_TERM, DeviceIndex, compute_idf


def split_by_doc_range(words: np.ndarray, term_off: np.ndarray, bounds: Sequence[int]):
    """Cut a term-major roaringish index at the doc ids ``bounds`` (``len(bounds) - 1`` shards): yields
    ``(words_g, term_off_g)`` with doc ids rebased to the shard.  Every term's words are doc-sorted, so
    a shard's words of a term are one contiguous run of the term's list."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    term_off = np.asarray(term_off, dtype=np.int64)
    vocab = len(term_off) - 1
    doc = (words >> np.uint64(rz.KEY_SHIFT)).astype(np.int64)
    shard_of = np.searchsorted(np.asarray(bounds[1:-1], dtype=np.int64), doc, side="right")
    word_term = np.repeat(np.arange(vocab, dtype=np.int64), np.diff(term_off))
    out = []
    for g in range(len(bounds) - 1):
        sel = shard_of == g
        w = words[sel] - (np.uint64(bounds[g]) << np.uint64(rz.KEY_SHIFT))
        off = np.zeros(vocab + 1, dtype=np.uint64)
        np.cumsum(np.bincount(word_term[sel], minlength=vocab), out=off[1:])
        out.append((w, off))
    return out


class _ShardView(DeviceIndex):
    """A shard of a sharded handle as a ``DeviceIndex`` (dense drop-in calls, statistics): the handle is BORROWED from
    ``sa_sharded_shard`` -- the sharded handle owns it and destroys it."""

    def __init__(self, api, handle, n_docs, n_terms, doc_base, avg_doc_len, corpus_size, global_df, opts=None):     # noqa: D107 (no create)
        self.api = api
        self._h = handle
        self._init_opts(opts)
        self.n_docs, self.n_terms, self.doc_base = int(n_docs), int(n_terms), int(doc_base)
        self.avg_doc_len = np.float32(avg_doc_len)
        self.corpus_size = int(corpus_size)
        self._local_df = None
        self._global_df = global_df

    def close(self):
        self._h = _lib.ctypes.c_void_p()


class ShardedIndex:
    """``len(devices)`` doc-range shards of one corpus, one per GPU, behind ONE handle of the C ABI
    (``sa_sharded_create``, csrc/sa_sharded.hip: shard split, one host thread per device, RCCL communicator, global
    document frequencies all happen inside the library).  This class only adds the numpy-side statistics (idf in the
    reference's float64 arithmetic) and fans the dense drop-in calls out to the shards."""

    def __init__(self, words: np.ndarray, term_off: np.ndarray, doc_lens: np.ndarray, devices: Sequence[int],
                 avg_doc_len: Optional[float] = None, tile_docs: int = 0, api=None, opts=None):
        self.api = api if api is not None else _lib.api()
        self._opts = _options.Options(opts)
        self.devices = [int(d) for d in devices]
        G = len(self.devices)
        if G < 1:
            raise ValueError("need at least one device")
        ct = _lib.ctypes
        words = _lib.as_u64(words)
        term_off = _lib.as_u64(term_off)
        doc_lens = np.ascontiguousarray(doc_lens, dtype=np.float32)
        self.n_docs = len(doc_lens)
        self.n_terms = len(term_off) - 1
        self.corpus_size = self.n_docs
        # reference indexing.py:282-284: np.mean over the float32 lengths of the WHOLE corpus
        self.avg_doc_len = np.float32(np.mean(doc_lens) if avg_doc_len is None and self.n_docs else (avg_doc_len or 0.0))
        self._h = ct.c_void_p()
        dev_arr = (ct.c_int * G)(*self.devices)
        with _options.creating(self.api, self._opts):          # (the shard threads of the library start from the caller's options)
            self.api.call("sa_sharded_create", dev_arr, G, self.n_docs, self.n_terms, _lib.p_u64(words), _lib.p_u64(term_off),
                          _lib.p_f32(doc_lens), self.avg_doc_len, int(tile_docs), ct.byref(self._h))
        bounds = np.zeros(G + 1, dtype=np.uint64)
        n = ct.c_int(0)
        self.api.call("sa_sharded_info", self._h, ct.byref(n), _lib.p_u64(bounds))
        self.bounds = [int(x) for x in bounds]
        self._df = np.zeros(self.n_terms, dtype=np.uint64)
        self.api.call("sa_sharded_docfreqs", self._h, _lib.p_u64(self._df))
        self._pool = ThreadPoolExecutor(max_workers=G)
        self.shards: List[DeviceIndex] = []
        for g in range(G):
            h = ct.c_void_p()
            self.api.call("sa_sharded_shard", self._h, g, ct.byref(h))
            self.shards.append(_ShardView(self.api, h, self.bounds[g + 1] - self.bounds[g], self.n_terms, self.bounds[g],
                                          self.avg_doc_len, self.n_docs, self._df, opts=self._opts))

    def map(self, fn):
        """fn(shard_index, DeviceIndex) on every shard concurrently (dense drop-in calls)."""
        amb = _options.ambient()                                # (the pool threads work under the CALLER's scoped options)

        def run(g):
            _options._set_ambient(amb)
            try:
                return fn(g, self.shards[g])
            finally:
                _options._set_ambient(_options.Options())
        return list(self._pool.map(run, range(len(self.shards))))

    def docfreq(self, term: int) -> np.uint64:
        return self._df[term] if 0 <= term < self.n_terms else np.uint64(0)

    def idfs(self, terms) -> np.ndarray:
        return np.asarray([compute_idf(self.corpus_size, np.asarray([self.docfreq(int(t))]))
                           if 0 <= int(t) < self.n_terms else np.float32(0) for t in terms], dtype=np.float32)

    # -- dense drop-in results: shard vectors concatenated in doc order
    def bm25_dense(self, terms, k1: float = 1.2, b: float = 0.75) -> np.ndarray:
        idf = self.idfs(terms)
        return np.concatenate(self.map(lambda g, s: s.bm25_dense(terms, k1=k1, b=b, idf=idf)))

    def termfreqs_dense(self, term: int) -> np.ndarray:
        return np.concatenate(self.map(lambda g, s: s.termfreqs_dense(term)))

    def phrase_freqs_dense(self, terms, slop: int = 0) -> np.ndarray:
        return np.concatenate(self.map(lambda g, s: s.phrase_freqs_dense(terms, slop=slop)))

    def bm25_phrase_dense(self, terms, k1: float = 1.2, b: float = 0.75, slop: int = 0) -> np.ndarray:
        dfs = np.asarray([self.docfreq(int(t)) if 0 <= int(t) < self.n_terms else 0 for t in terms])
        idf = compute_idf(self.corpus_size, dfs)
        return np.concatenate(self.map(lambda g, s: s.bm25_phrase_dense(terms, k1=k1, b=b, slop=slop, idf=idf)))

    # -- resident top-k batches
    def batch(self, queries: np.ndarray, k: int = 10, k1: float = 1.2, b: float = 0.75, opts=None) -> "ShardedBatch":
        q = np.asarray(queries, dtype=np.int64)
        if q.ndim != 2:
            raise ValueError("queries must be [B][T] term ids")
        idf = _lib.as_f32(self.idfs(q.reshape(-1)).reshape(q.shape))
        terms = _lib.as_u32(np.where((q >= 0) & (q < self.n_terms), q, NO_TERM).astype(np.uint32))
        h = _lib.ctypes.c_void_p()
        with _options.creating(self.api, _options.Options(self._opts, opts)):
            self.api.call("sa_sharded_batch_create", self._h, _lib.p_u32(terms), _lib.p_f32(idf), q.shape[0], q.shape[1], int(k),
                          np.float32(k1), np.float32(b), _lib.ctypes.byref(h))
        return ShardedBatch(self, h, q.shape[0], int(k), n_terms=q.shape[1], opts=_options.Options(self._opts, opts))

    def phrase_batch(self, phrases, k: int = 10, k1: float = 1.2, b: float = 0.75, slop=0, opts=None) -> "ShardedBatch":
        ct = _lib.ctypes
        B = len(phrases)
        if B == 0:
            raise ValueError("empty phrase batch")
        n_terms = np.asarray([len(p) for p in phrases], dtype=np.int32)
        T = int(max(2, n_terms.max()))
        terms = np.full((B, T), NO_TERM, dtype=np.uint32)
        for i, ph in enumerate(phrases):
            row = np.asarray(ph, dtype=np.int64)
            terms[i, :len(row)] = np.where((row >= 0) & (row < self.n_terms), row, NO_TERM)
        idf = np.asarray([compute_idf(self.corpus_size, np.asarray([self.docfreq(int(t)) if 0 <= int(t) < self.n_terms else 0
                                                                    for t in ph])) for ph in phrases], dtype=np.float32)
        slops = np.ascontiguousarray(np.broadcast_to(np.asarray(slop, dtype=np.int32), (B,)))
        if (slops < 0).any():
            raise ValueError("slop must be >= 0")
        h = ct.c_void_p()
        with _options.creating(self.api, _options.Options(self._opts, opts)):
            self.api.call("sa_sharded_phrase_batch_create", self._h, _lib.p_u32(_lib.as_u32(terms)),
                          n_terms.ctypes.data_as(ct.POINTER(ct.c_int32)), slops.ctypes.data_as(ct.POINTER(ct.c_int32)),
                          _lib.p_f32(_lib.as_f32(idf)), B, T, int(k), np.float32(k1), np.float32(b), ct.byref(h))
        return ShardedBatch(self, h, B, int(k), opts=_options.Options(self._opts, opts))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_sharded_destroy(self._h)
            self._h = _lib.ctypes.c_void_p()
        for s in self.shards:
            s.close()
        self.shards = []
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedBatch(_options.OptionsMixin):
    """One resident query batch per shard behind one handle (``sa_sharded_batch_*``); ``run`` = score every shard +
    all-gather + merge, ``fetch`` = the merged top-k.  Follows the thread's scoped options and ``set_options`` like a
    ``QueryBatch`` does (``sa_sharded_batch_set_options`` fans out to the shards' batches)."""
    _opt_setter = "sa_sharded_batch_set_options"

    def __init__(self, index: ShardedIndex, handle, B: int, k: int, n_terms: Optional[int] = None, opts=None):
        self.index, self.api, self._h = index, index.api, handle
        self.B, self.k, self.T = B, k, n_terms
        self._init_opts(opts)

    def reset(self, queries: np.ndarray):
        q = np.asarray(queries, dtype=np.int64)
        if self.T is None or q.shape != (self.B, self.T):
            raise ValueError("reset takes a BM25 batch and queries of its shape")
        idf = _lib.as_f32(self.index.idfs(q.reshape(-1)).reshape(q.shape))
        terms = _lib.as_u32(np.where((q >= 0) & (q < self.index.n_terms), q, NO_TERM).astype(np.uint32))
        self._call("sa_sharded_batch_reset", self._h, _lib.p_u32(terms), _lib.p_f32(idf))

    def run(self, sync: bool = True):
        self._call("sa_sharded_batch_run", self._h, 1 if sync else 0)

    def synchronize(self):
        self.index.map(lambda g, s: s.synchronize())

    def fetch(self) -> Tuple[np.ndarray, np.ndarray]:
        scores = np.empty((self.B, self.k), dtype=np.float32)
        docs = np.empty((self.B, self.k), dtype=np.uint64)
        self._call("sa_sharded_batch_fetch", self._h, _lib.p_f32(scores), _lib.p_u64(docs))
        return scores, docs

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_sharded_batch_destroy(self._h)
            self._h = _lib.ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

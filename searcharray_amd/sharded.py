"""N GPUs behind ONE handle: the doc-range-sharded index of a single process.

Documents are independent units of this path -- tf, doc length and phrase matches are per doc, and the
roaringish key of a word is its doc id, so a doc-id range is a contiguous slice of every term's word
list (what the reference's ``key_partition`` computes, searcharray/roaringish/roaringish.py:227-243).
Device g holds docs ``[g*N/G, (g+1)*N/G)`` with shard-local ids; BM25 uses the GLOBAL corpus size, average
doc length and document frequencies on every shard (the reference computes them over the whole corpus:
indexing.py:282-284, middle_out.py:521-528).  A query batch is scored on all shards concurrently (one
host thread per device -- ctypes releases the GIL -- each on its own HIP streams); the only data-path
exchange is the RCCL all-gather of the per-shard top-k keys followed by a merge on every device
(csrc/sa_comm.hip), so ``fetch`` reads the final result from shard 0.  Dense drop-in results
(``score`` / ``termfreqs``) are per-shard vectors concatenated in doc order -- no collective.

``bench.py --gpus N`` uses the same library entry points with one PROCESS per GPU.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import roaringish as rz
from .device_index import DeviceIndex, PhraseBatch, QueryBatch, compute_idf


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


class ShardedIndex:
    """``len(devices)`` doc-range shards of one corpus, one per GPU, driven from this process."""

    def __init__(self, words: np.ndarray, term_off: np.ndarray, doc_lens: np.ndarray, devices: Sequence[int],
                 avg_doc_len: Optional[float] = None, tile_docs: int = 0, api=None):
        self.api = api if api is not None else _lib.api()
        self.devices = [int(d) for d in devices]
        G = len(self.devices)
        if G < 1:
            raise ValueError("need at least one device")
        doc_lens = np.ascontiguousarray(doc_lens, dtype=np.float32)
        self.n_docs = len(doc_lens)
        self.n_terms = len(term_off) - 1
        self.corpus_size = self.n_docs
        # reference indexing.py:282-284: np.mean over the float32 lengths of the WHOLE corpus
        self.avg_doc_len = np.float32(np.mean(doc_lens) if avg_doc_len is None and self.n_docs else (avg_doc_len or 0.0))
        self.bounds = [self.n_docs * g // G for g in range(G + 1)]
        self._pool = ThreadPoolExecutor(max_workers=G)
        parts = split_by_doc_range(words, term_off, self.bounds)

        def make(g):
            w, off = parts[g]
            lo, hi = self.bounds[g], self.bounds[g + 1]
            return DeviceIndex(w, off, doc_lens[lo:hi], avg_doc_len=self.avg_doc_len, corpus_size=self.n_docs,
                               doc_base=lo, device=self.devices[g], tile_docs=tile_docs, api=self.api)
        self.shards: List[DeviceIndex] = list(self._pool.map(make, range(G)))
        self._comm = False
        if G > 1:
            uid = DeviceIndex.comm_unique_id(self.api)
            list(self._pool.map(lambda g: self.shards[g].comm_init(g, G, uid), range(G)))     # ncclCommInitRank blocks until all joined
            self._comm = True
        # global document frequencies: summed over the shards by the library's own all-reduce
        def gdf(g):
            df = self.shards[g].docfreqs().astype(np.uint64)
            if self._comm:
                df = self.shards[g].comm_allreduce(np.ascontiguousarray(df), "sum")
            self.shards[g].set_global_docfreqs(df)
            return df
        self._df = list(self._pool.map(gdf, range(G)))[0]

    def map(self, fn):
        """fn(shard_index, DeviceIndex) on every shard concurrently."""
        return list(self._pool.map(lambda g: fn(g, self.shards[g]), range(len(self.shards))))

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
    def batch(self, queries: np.ndarray, k: int = 10, k1: float = 1.2, b: float = 0.75) -> "ShardedBatch":
        q = np.asarray(queries, dtype=np.int64)
        idf = self.idfs(q.reshape(-1)).reshape(q.shape)
        return ShardedBatch(self, lambda g, s: QueryBatch(s, q, k=k, k1=k1, b=b, idf=idf))

    def phrase_batch(self, phrases, k: int = 10, k1: float = 1.2, b: float = 0.75, slop=0) -> "ShardedBatch":
        idf = np.asarray([compute_idf(self.corpus_size, np.asarray([self.docfreq(int(t)) if 0 <= int(t) < self.n_terms else 0
                                                                    for t in ph])) for ph in phrases], dtype=np.float32)
        return ShardedBatch(self, lambda g, s: PhraseBatch(s, phrases, k=k, k1=k1, b=b, idf=idf, slop=slop))

    def close(self):
        if self._comm:
            self.map(lambda g, s: s.comm_destroy())
            self._comm = False
        for s in self.shards:
            s.close()
        self.shards = []
        self._pool.shutdown(wait=True)

    def __del__(self):
        try:
            if self.shards:
                self.close()
        except Exception:
            pass


class ShardedBatch:
    """One resident query batch per shard; ``run`` = score every shard + all-gather + merge."""

    def __init__(self, index: ShardedIndex, make):
        self.index = index
        self.parts = index.map(make)

    def run(self, sync: bool = True):
        # every shard's run() enqueues its scoring kernels and then the collective: one host thread per
        # device, so the all-gathers of the ranks meet (a single thread would deadlock in the first one)
        self.index.map(lambda g, s: self.parts[g].run(sync=sync))

    def synchronize(self):
        self.index.map(lambda g, s: s.synchronize())

    def fetch(self) -> Tuple[np.ndarray, np.ndarray]:
        # fetch is collective when sharded (the ranks agree on a redo after a candidate-list overflow)
        return self.index.map(lambda g, s: self.parts[g].fetch())[0]

    def close(self):
        for p in self.parts:
            p.close()
        self.parts = []

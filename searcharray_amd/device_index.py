"""HBM-resident index and query batches (Part 2/3 of the C ABI) as Python objects.

``DeviceIndex`` is the device-side counterpart of the reference's ``PosnBitArray``
(searcharray/phrase/middle_out.py:320): it owns the roaringish words of one doc-range shard
in HBM plus the derived TF postings, and answers docfreq / termfreqs / BM25 / phrase queries
with GPU kernels.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import options as _options
from ._lib import IndexInfo, as_f32, as_u32, as_u64, p_f32, p_u32, p_u64

NO_TERM = 0xFFFFFFFF
NO_DOC = 0xFFFFFFFFFFFFFFFF


def compute_idf(num_docs, dfs) -> np.float32:
    """idf exactly as the reference computes it on the host: float64 numpy math, summed over
    the tokens of the phrase, truncated to float32 at the native call
    (searcharray/similarity.py:19-21, bm25.pyx:31)."""
    dfs = np.asarray(dfs)
    return np.float32(np.sum(np.log(1 + (num_docs - dfs + 0.5) / (dfs + 0.5))))


class _PinnedPool:
    """float32 result arrays backed by page-locked host memory, recycled when the array dies.

    The dense drop-in calls return a fresh ``float32[n_docs]`` every time.  Allocating that with numpy
    costs first-touch page faults (≈3 ms for 40 MB -- more than the device work plus the copy), and a
    pageable destination halves the PCIe rate.  Buffers here come from ``sa_host_alloc``; when the numpy
    array (and every view of it) has been garbage-collected its buffer goes back to the pool, so a caller
    that consumes results as they come keeps reusing the same few buffers.  A caller that keeps every
    result simply keeps every buffer: the arrays stay valid and caller-owned."""

    MAX_IDLE_BYTES = 1 << 30

    def __init__(self, api):
        self.api = api
        self.idle = {}                # nbytes -> [address, ...]
        self.idle_bytes = 0

    def _release(self, addr: int, nbytes: int):
        try:
            if self.idle_bytes + nbytes <= self.MAX_IDLE_BYTES:
                self.idle.setdefault(nbytes, []).append(addr)
                self.idle_bytes += nbytes
            else:
                self.api.sa_host_free(ctypes.c_void_p(addr))
        except Exception:             # interpreter shutdown
            pass

    def empty_f32(self, n: int) -> np.ndarray:
        return self.empty(n, np.float32)

    def empty(self, n: int, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        if n == 0:
            return np.empty(0, dtype=dtype)
        nbytes = dtype.itemsize * n
        stack = self.idle.get(nbytes)
        addr = None
        if stack:
            try:
                addr = stack.pop()                  # (threads share the pool: another one may have taken the last buffer)
                self.idle_bytes -= nbytes
            except IndexError:
                addr = None
        if addr is None:
            ptr = ctypes.c_void_p()
            self.api.call("sa_host_alloc", nbytes, ctypes.byref(ptr))
            addr = ptr.value
        buf = (ctypes.c_char * nbytes).from_address(addr)
        weakref.finalize(buf, self._release, addr, nbytes)     # runs when the last array / view is gone
        return np.frombuffer(buf, dtype=dtype)


_pools = {}


def _pool(api) -> _PinnedPool:
    pool = _pools.get(id(api))
    if pool is None:
        pool = _pools[id(api)] = _PinnedPool(api)
    return pool


class DeviceIndex(_options.OptionsMixin):
    _opt_setter = "sa_index_set_options"

    def __init__(self, words: np.ndarray, term_off: np.ndarray, doc_lens: np.ndarray,
                 avg_doc_len: Optional[float] = None, corpus_size: Optional[int] = None,
                 doc_base: int = 0, device: int = 0, tile_docs: int = 0, api=None,
                 global_df: Optional[np.ndarray] = None, opts=None):
        self.api = api if api is not None else _lib.api()
        words = as_u64(words)
        term_off = as_u64(term_off)
        doc_lens = as_f32(doc_lens)
        self.n_docs = len(doc_lens)
        self.n_terms = len(term_off) - 1
        self.doc_base = int(doc_base)
        self.avg_doc_len = np.float32(np.mean(doc_lens) if avg_doc_len is None and self.n_docs
                                      else (avg_doc_len or 0.0))
        self.corpus_size = int(self.n_docs if corpus_size is None else corpus_size)
        self._h = ctypes.c_void_p()
        with _options.creating(self.api, opts):
            self.api.call("sa_index_create", int(device), self.n_docs, self.doc_base, self.n_terms,
                          p_u64(words), p_u64(term_off), p_f32(doc_lens), self.avg_doc_len,
                          self.corpus_size, int(tile_docs), ctypes.byref(self._h))
        self._init_opts(opts)
        self._local_df = None
        self._global_df = None if global_df is None else np.asarray(global_df, dtype=np.uint64)

    @classmethod
    def from_tokens(cls, tokens: np.ndarray, doc_ptr: np.ndarray, n_terms: int,
                    doc_lens: Optional[np.ndarray] = None, avg_doc_len: Optional[float] = None,
                    corpus_size: Optional[int] = None, doc_base: int = 0, device: int = 0, tile_docs: int = 0,
                    api=None, global_df: Optional[np.ndarray] = None, opts=None) -> "DeviceIndex":
        """Build the index on the device from the token stream: ``tokens[doc_ptr[d]:doc_ptr[d+1]]`` are
        the term ids of doc ``d`` in position order (what a tokenizer + term dictionary produce).
        Sort by term and roaringish encoding happen on the GPU (reference indexing.py:102-115,
        roaringish.py:93-142); ``words()`` downloads the encoded index when the host needs it."""
        self = cls.__new__(cls)
        self.api = api if api is not None else _lib.api()
        tokens = as_u32(tokens)
        doc_ptr = as_u64(doc_ptr)
        self.n_docs = len(doc_ptr) - 1
        if doc_lens is None:
            doc_lens = np.diff(doc_ptr.astype(np.int64)).astype(np.float32)
        doc_lens = as_f32(doc_lens)
        self.n_terms = int(n_terms)
        self.doc_base = int(doc_base)
        self.avg_doc_len = np.float32(np.mean(doc_lens) if avg_doc_len is None and self.n_docs
                                      else (avg_doc_len or 0.0))
        self.corpus_size = int(self.n_docs if corpus_size is None else corpus_size)
        self._h = ctypes.c_void_p()
        with _options.creating(self.api, opts):
            self.api.call("sa_index_create_from_tokens", int(device), self.n_docs, self.doc_base, self.n_terms,
                          p_u32(tokens), p_u64(doc_ptr), p_f32(doc_lens), self.avg_doc_len,
                          self.corpus_size, int(tile_docs), ctypes.byref(self._h))
        self._init_opts(opts)
        self._local_df = None
        self._global_df = None if global_df is None else np.asarray(global_df, dtype=np.uint64)
        return self

    @classmethod
    def from_file(cls, path: str, metadata, doc_lens: np.ndarray, n_terms: Optional[int] = None,
                  avg_doc_len: Optional[float] = None, corpus_size: Optional[int] = None, doc_base: int = 0,
                  device: int = 0, tile_docs: int = 0, api=None,
                  global_df: Optional[np.ndarray] = None, opts=None) -> "DeviceIndex":
        """Stream an index from the reference's on-disk format straight into HBM: ``path`` is the raw
        uint64 ``.dat`` file MemoryMappedArrays writes (reference phrase/memmap_arrays.py:158-161) and
        ``metadata`` its ArrayDict metadata ``{term_id: {'offset': o, 'length': n}}`` (element units,
        memmap_arrays.py:28-54) -- or, equivalently, the CSR offsets ``uint64[V+1]`` of a file that
        holds the terms back to back in id order (what ``save`` writes), or the pair of arrays
        ``(term_src_off uint64[V], term_len uint64[V])``."""
        self = cls.__new__(cls)
        self.api = api if api is not None else _lib.api()
        if isinstance(metadata, tuple) and len(metadata) == 2:
            # (term_src_off uint64[V], term_len uint64[V]) as the C ABI takes them: no per-term Python work
            src, length = as_u64(metadata[0]), as_u64(metadata[1])
            V = len(length) if n_terms is None else int(n_terms)
            if len(src) != V or len(length) != V:
                raise ValueError("term_src_off / term_len must hold n_terms entries")
        elif isinstance(metadata, dict):
            V = int(n_terms) if n_terms is not None else (max(int(k) for k in metadata) + 1 if metadata else 0)
            src = np.zeros(V, dtype=np.uint64)
            length = np.zeros(V, dtype=np.uint64)
            for k, v in metadata.items():
                if int(k) >= V:
                    raise ValueError(f"metadata names term {k} but the index has {V} terms")
                src[int(k)] = int(v["offset"])
                length[int(k)] = int(v["length"])
        else:
            off = as_u64(np.asarray(metadata))
            V = len(off) - 1 if n_terms is None else int(n_terms)
            if len(off) != V + 1:
                raise ValueError("term offsets must hold n_terms + 1 entries")
            src = np.ascontiguousarray(off[:-1])
            length = np.ascontiguousarray(np.diff(off))
        doc_lens = as_f32(doc_lens)
        self.n_docs = len(doc_lens)
        self.n_terms = V
        self.doc_base = int(doc_base)
        self.avg_doc_len = np.float32(np.mean(doc_lens) if avg_doc_len is None and self.n_docs
                                      else (avg_doc_len or 0.0))
        self.corpus_size = int(self.n_docs if corpus_size is None else corpus_size)
        self._h = ctypes.c_void_p()
        with _options.creating(self.api, opts):
            self.api.call("sa_index_create_from_file", int(device), self.n_docs, self.doc_base, self.n_terms,
                          os.fsencode(path), p_u64(src), p_u64(length), p_f32(doc_lens), self.avg_doc_len,
                          self.corpus_size, int(tile_docs), ctypes.byref(self._h))
        self._init_opts(opts)
        self._local_df = None
        self._global_df = None if global_df is None else np.asarray(global_df, dtype=np.uint64)
        return self

    def save(self, path: str) -> np.ndarray:
        """Write the resident words to ``path`` as one raw uint64 file (the reference's ``.dat``,
        phrase/memmap_arrays.py:158-161), device -> page-locked ring -> file.  Returns the term
        offsets ``uint64[V+1]`` that index it (term t = elements [off[t], off[t+1]))."""
        self._call("sa_index_save", self._h, os.fsencode(path))
        term_off = np.empty(self.n_terms + 1, dtype=np.uint64)
        self._call("sa_index_words", self._h, None, p_u64(term_off))
        return term_off

    def words(self) -> Tuple[np.ndarray, np.ndarray]:
        """(roaringish words uint64[W] term-major, term offsets uint64[V+1]) copied from the device."""
        info = self.info()
        words = np.empty(int(info.n_words), dtype=np.uint64)
        term_off = np.empty(self.n_terms + 1, dtype=np.uint64)
        self._call("sa_index_words", self._h, p_u64(words), p_u64(term_off))
        return words, term_off

    # -- lifetime
    def _track(self, batch):
        """batches hold device memory tied to this handle: they are closed before the index goes"""
        if not hasattr(self, "_batches"):
            self._batches = weakref.WeakSet()
        self._batches.add(batch)

    def close(self):
        for b in list(getattr(self, "_batches", ())):
            b.close()
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._call("sa_index_synchronize", self._h)

    def info(self) -> IndexInfo:
        out = IndexInfo()
        self._call("sa_index_info", self._h, ctypes.byref(out))
        return out

    # -- statistics
    def docfreqs(self) -> np.ndarray:
        """Shard-local document frequency of every term (uint64[n_terms])."""
        if self._local_df is None:
            out = np.empty(self.n_terms, dtype=np.uint64)
            if self.n_terms:
                self._call("sa_index_docfreqs", self._h, p_u64(out))
            self._local_df = out
        return self._local_df

    def set_idf_table(self, idf: np.ndarray):
        """one float32 idf per term, formed by the caller as the reference forms it (similarity.py:19-21);
        ``QueryBatch.step`` gathers a query set's weights from it inside the library"""
        t = as_f32(idf)
        self._call("sa_index_set_idf_table", self._h, p_f32(t), len(t))

    @staticmethod
    def comm_library_info(api=None):
        """(ncclGetVersion(), path of the shared object that provides the collectives in this process)"""
        api = api or _lib.api()
        ver = ctypes.c_int(0)
        buf = ctypes.create_string_buffer(1024)
        api.call("sa_comm_library_info", ctypes.byref(ver), buf, 1024)
        return int(ver.value), buf.value.decode(errors="replace")

    def set_global_docfreqs(self, df: np.ndarray):
        """Sharded operation: BM25 idf must use corpus-wide df (sum over shards)."""
        self._global_df = np.asarray(df, dtype=np.uint64)

    def docfreq(self, term: int) -> np.uint64:
        df = self._global_df if self._global_df is not None else self.docfreqs()
        return df[term] if 0 <= term < self.n_terms else np.uint64(0)

    # -- dense results, whole or a row subset
    def _dense(self, fn: str, rows: Optional[np.ndarray], *args, dtype=np.float32) -> np.ndarray:
        """Run a dense C-ABI call (its last argument is the float32 -- or ``dtype`` -- output).  rows
        given: only those doc ids come back (gathered on the device: the copy is proportional to the
        subset) -- through the call's `_to` twin, which takes the destination as an argument (no selection
        outlives a call)."""
        ptr = p_f32 if dtype == np.float32 else (lambda a: a.ctypes.data_as(ctypes.c_void_p))
        if rows is None:
            out = _pool(self.api).empty(self.n_docs, dtype)
            self._call(fn, self._h, *args, ptr(out))
            return out
        rows = as_u64(rows)
        out = np.empty(len(rows), dtype=dtype)
        dest = _lib.DenseDest(p_u64(rows), len(rows), None, np.float32(1.0), 0)
        self._call(fn + "_to", self._h, *args, ctypes.byref(dest), ptr(out))
        return out

    def into_vec(self, vec: "DeviceVec", boost: Optional[float], fn: str, *args) -> None:
        """Run a dense C-ABI call with its result diverted into ``vec`` (float32[n_docs] on the device),
        multiplied by ``boost`` if given (the call's `_to` twin: the destination is an argument)."""
        dest = _lib.DenseDest(None, 0, vec._h, np.float32(1.0 if boost is None else boost), 0 if boost is None else 1)
        self._call(fn + "_to", self._h, *args, ctypes.byref(dest), None)

    # -- term frequencies
    @staticmethod
    def _check_posn_range(min_posn, max_posn):
        # reference roaringish.py:270-273
        if min_posn is not None and min_posn % 18 != 0:
            raise ValueError("min_payload must be a multiple of 18")
        if max_posn is not None and max_posn % 18 != 17:
            raise ValueError("max_payload must be a multiple of 18 - 1")
        return (-1 if min_posn is None else int(min_posn)), (-1 if max_posn is None else int(max_posn))

    def termfreqs_dense(self, term: int, min_posn: Optional[int] = None, max_posn: Optional[int] = None,
                        rows: Optional[np.ndarray] = None) -> np.ndarray:
        lo, hi = self._check_posn_range(min_posn, max_posn)
        t = term if 0 <= term < self.n_terms else NO_TERM
        return self._dense("sa_index_termfreqs_dense_posn", rows, t, lo, hi)

    def termfreqs_sparse(self, term: int) -> Tuple[np.ndarray, np.ndarray]:
        if not (0 <= term < self.n_terms):
            return np.empty(0, np.uint64), np.empty(0, np.float32)
        df = int(self.docfreqs()[term])
        ids = np.empty(df, dtype=np.uint64)
        tfs = np.empty(df, dtype=np.float32)
        n = _lib.c_int64(0)
        self._call("sa_index_termfreqs_sparse", self._h, term, p_u64(ids), p_f32(tfs), n)
        return ids[:n.value], tfs[:n.value]

    # -- scoring
    def idfs(self, terms: Sequence[int]) -> np.ndarray:
        return np.asarray([compute_idf(self.corpus_size, np.asarray([self.docfreq(int(t))]))
                           if 0 <= int(t) < self.n_terms else np.float32(0) for t in terms],
                          dtype=np.float32)

    def bm25_dense(self, terms: Sequence[int], k1: float = 1.2, b: float = 0.75,
                   idf: Optional[np.ndarray] = None, rows: Optional[np.ndarray] = None) -> np.ndarray:
        """sum_t BM25(t) in query-term order, float32[n_docs] (one kernel launch)."""
        tarr = np.asarray([int(t) if 0 <= int(t) < self.n_terms else NO_TERM for t in terms],
                          dtype=np.uint32)
        idf = self.idfs(terms) if idf is None else as_f32(idf)
        return self._dense("sa_index_bm25_dense", rows, p_u32(tarr), p_f32(idf), len(tarr), np.float32(k1), np.float32(b))

    def phrase_freqs_dense(self, terms: Sequence[int], slop: int = 0, min_posn: Optional[int] = None,
                           max_posn: Optional[int] = None, rows: Optional[np.ndarray] = None) -> np.ndarray:
        """Phrase match counts, float32[n_docs] (reference PosnBitArray.phrase_freqs): exact for
        slop == 0, the reference's span search for slop > 0."""
        if len(terms) < 2:
            raise ValueError("Must have at least two terms")        # reference middle_out.py:425-426
        lo, hi = self._check_posn_range(min_posn, max_posn)
        tarr = np.asarray([int(t) if 0 <= int(t) < self.n_terms else NO_TERM for t in terms], dtype=np.uint32)
        return self._dense("sa_index_phrase_freqs_dense_posn", rows, p_u32(tarr), len(tarr), int(slop), lo, hi)

    def bm25_phrase_dense(self, terms: Sequence[int], k1: float = 1.2, b: float = 0.75, slop: int = 0,
                          idf: Optional[float] = None, min_posn: Optional[int] = None,
                          max_posn: Optional[int] = None, rows: Optional[np.ndarray] = None) -> np.ndarray:
        """BM25 of a phrase: idf summed over the phrase's terms (reference postings.py:671-679)."""
        if len(terms) < 2:
            raise ValueError("Must have at least two terms")
        tarr = np.asarray([int(t) if 0 <= int(t) < self.n_terms else NO_TERM for t in terms], dtype=np.uint32)
        if idf is None:
            dfs = np.asarray([self.docfreq(int(t)) if 0 <= int(t) < self.n_terms else 0 for t in terms])
            idf = compute_idf(self.corpus_size, dfs)
        lo, hi = self._check_posn_range(min_posn, max_posn)
        return self._dense("sa_index_bm25_phrase_dense_posn", rows, p_u32(tarr), len(tarr), int(slop), lo, hi,
                           np.float32(idf), np.float32(k1), np.float32(b))

    SIMILARITY_KINDS = {"bm25_impact": (1, np.float32), "bm25_legacy": (2, np.float64), "classic": (3, np.float64)}

    def similarity_dense(self, kind: str, terms: Sequence[int], idf: float = 0.0, k1: float = 1.2, b: float = 0.75,
                         slop: int = 0, min_posn: Optional[int] = None, max_posn: Optional[int] = None,
                         rows: Optional[np.ndarray] = None) -> np.ndarray:
        """One of the reference's other stock similarities (similarity.py:41-89) of a term (one id) or a
        phrase, computed on the device with numpy's rounding: float32 for ``bm25_impact``, float64 for
        ``bm25_legacy`` / ``classic``."""
        code, dtype = self.SIMILARITY_KINDS[kind]
        tarr = np.asarray([int(t) if 0 <= int(t) < self.n_terms else NO_TERM for t in terms], dtype=np.uint32)
        lo, hi = self._check_posn_range(min_posn, max_posn)
        return self._dense("sa_index_similarity_dense", rows, p_u32(tarr), len(tarr), int(slop), lo, hi, code,
                           float(idf), float(k1), float(b), dtype=dtype)

    def last_profile(self) -> Tuple[float, int]:
        """(kernel ms, algorithmic bytes) of the last phrase call."""
        ms = _lib.c_double(0)
        ab = _lib.c_uint64(0)
        self._call("sa_index_last_profile", self._h, ctypes.byref(ms), ctypes.byref(ab))
        return ms.value, ab.value

    def batch(self, queries: np.ndarray, k: int = 10, k1: float = 1.2, b: float = 0.75,
              idf: Optional[np.ndarray] = None, opts=None) -> "QueryBatch":
        return QueryBatch(self, queries, k=k, k1=k1, b=b, idf=idf, opts=opts)

    def queue(self, n_queries: int, n_terms: int, k: int = 10, k1: float = 1.2, b: float = 0.75, depth: int = 6, opts=None) -> "QueryQueue":
        """a query-set queue (``sa_queue_*``): ``depth`` batches of ``n_queries`` x ``n_terms`` behind one handle, stepped by a worker
        thread of the library; needs ``set_idf_table`` first"""
        return QueryQueue(self, n_queries, n_terms, k=k, k1=k1, b=b, depth=depth, opts=opts)

    def phrase_batch(self, phrases: Sequence[Sequence[int]], k: int = 10, k1: float = 1.2, b: float = 0.75,
                     idf: Optional[np.ndarray] = None, slop=0, opts=None) -> "PhraseBatch":
        return PhraseBatch(self, phrases, k=k, k1=k1, b=b, idf=idf, slop=slop, opts=opts)

    # -- multi-GPU
    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        self._call("sa_index_comm_init", self._h, rank, nranks, unique_id, len(unique_id))

    def comm_destroy(self):
        self._call("sa_index_comm_destroy", self._h)

    @staticmethod
    def comm_unique_id(api=None) -> bytes:
        """128 opaque bytes one rank creates and hands to all ranks of a communicator."""
        api = api if api is not None else _lib.api()
        buf = ctypes.create_string_buffer(128)
        api.call("sa_comm_unique_id", buf, 128)
        return buf.raw

    def comm_allreduce(self, arr: np.ndarray, op: str = "sum") -> np.ndarray:
        """Sum / max of a uint64 or float64 host array over the ranks (in place, blocking): global df,
        the sum of the doc lengths, max-over-ranks timings."""
        if arr.dtype == np.uint64:
            dt = 0
        elif arr.dtype == np.float64:
            dt = 1
        else:
            raise TypeError("comm_allreduce takes uint64 or float64 arrays")
        if not arr.flags.c_contiguous or not arr.flags.writeable:
            raise ValueError("comm_allreduce works in place on a contiguous writable array")
        self._call("sa_index_comm_allreduce", self._h, arr.ctypes.data_as(ctypes.c_void_p), arr.size, dt,
                      {"sum": 0, "max": 1}[op])
        return arr

    def comm_barrier(self):
        self._call("sa_index_comm_barrier", self._h)


class QueryBatch(_options.OptionsMixin):
    _opt_setter = "sa_batch_set_options"

    """B queries x T terms resident on the device; ``run()`` is one pass of the hot path."""

    def __init__(self, index: DeviceIndex, queries: np.ndarray, k: int = 10, k1: float = 1.2,
                 b: float = 0.75, idf: Optional[np.ndarray] = None, opts=None):
        self.index = index
        self.api = index.api
        q = np.asarray(queries, dtype=np.int64)
        if q.ndim != 2:
            raise ValueError("queries must be [B][T] term ids")
        self.B, self.T = q.shape
        self.k = int(k)
        terms = np.where((q >= 0) & (q < index.n_terms), q, NO_TERM).astype(np.uint32)
        if idf is None:
            flat = index.idfs(q.reshape(-1))
            idf = flat.reshape(self.B, self.T)
        idf = as_f32(idf)
        self._h = ctypes.c_void_p()
        base = _options.Options(index._opts_base, opts)          # a batch starts from its index's options + its own
        with _options.creating(self.api, base):
            self._create(index, terms, idf, k1, b)
        self._init_opts(base)
        index._track(self)

    def _create(self, index, terms, idf, k1, b):
        self.api.call("sa_batch_create", index._h, p_u32(as_u32(terms)), p_f32(idf), self.B, self.T,
                      self.k, np.float32(k1), np.float32(b), ctypes.byref(self._h))

    def reset(self, queries: np.ndarray, idf: Optional[np.ndarray] = None):
        """A new set of B x T queries in this batch (the caller idiom of a query STREAM: ``score()`` on queries the
        device has not seen).  One async copy + one kernel behind the runs in flight; no allocation, no
        synchronisation (``sa_batch_reset``).  Fetch the previous results first."""
        q = np.asarray(queries, dtype=np.int64)
        if q.shape != (self.B, self.T):
            raise ValueError(f"reset takes [{self.B}][{self.T}] term ids")
        terms = np.where((q >= 0) & (q < self.index.n_terms), q, NO_TERM).astype(np.uint32)
        if idf is None:
            idf = self.index.idfs(q.reshape(-1)).reshape(self.B, self.T)
        idf = as_f32(idf)
        self._call("sa_batch_reset", self._h, p_u32(as_u32(terms)), p_f32(idf))

    def step(self, queries: np.ndarray):
        """reset + run(sync=False) in ONE library call, the weights gathered from the index's idf table
        (``DeviceIndex.set_idf_table``): a step of a query stream (``sa_batch_step``)"""
        q = np.ascontiguousarray(queries, dtype=np.uint32)
        if q.shape != (self.B, self.T):
            raise ValueError(f"step takes [{self.B}][{self.T}] term ids")
        self._call("sa_batch_step", self._h, p_u32(q))

    def run(self, sync: bool = True):
        self._call("sa_batch_run", self._h, 1 if sync else 0)

    def run_local(self, local_keys_device_ptr: int = 0, sync: bool = True):
        self._call("sa_batch_run_local", self._h, ctypes.c_void_p(local_keys_device_ptr), 1 if sync else 0)

    def merge_gathered(self, gathered_device_ptr: int, nranks: int, sync: bool = True):
        self._call("sa_batch_merge_gathered", self._h, ctypes.c_void_p(gathered_device_ptr), nranks,
                      1 if sync else 0)

    def fetch(self) -> Tuple[np.ndarray, np.ndarray]:
        scores = np.empty((self.B, self.k), dtype=np.float32)
        docs = np.empty((self.B, self.k), dtype=np.uint64)
        self._call("sa_batch_fetch", self._h, p_f32(scores), p_u64(docs))
        return scores, docs

    def profile(self) -> Tuple[float, int, int]:
        ms = _lib.c_double(0)
        alg = _lib.c_uint64(0)
        post = _lib.c_uint64(0)
        self._call("sa_batch_profile", self._h, ctypes.byref(ms), ctypes.byref(alg), ctypes.byref(post))
        return ms.value, alg.value, post.value

    def group_info(self) -> dict:
        """how the exhaustive path groups this batch: groups, queries in groups, of them with a shared first term
        (the others are loose groups), queries left to the per-query kernel"""
        out = (_lib.c_uint32 * 4)()
        self._call("sa_batch_group_info", self._h, out)
        return {"groups": int(out[0]), "grouped_queries": int(out[1]), "shared_first_term": int(out[2]), "per_query_kernel": int(out[3])}

    def last_route(self) -> str:
        """the route the last run took: 'staged' (distinct terms staged in LDS per tile, sa_stage.hip), 'pruned' (dynamic
        pruning) or 'exhaustive' (grouped / per-query overlay kernels)"""
        out = ctypes.c_int(0)
        self._call("sa_batch_last_route", self._h, ctypes.byref(out))
        return {0: "exhaustive", 1: "pruned", 2: "staged"}[out.value]

    def seeds(self) -> np.ndarray:
        """the bound every query of the current set starts with (sa_batch_seeds), float32[B], 0 = none"""
        out = np.zeros(self.B, dtype=np.float32)
        self._call("sa_batch_seeds", self._h, p_f32(out))
        return out

    def host_times(self) -> dict:
        """host microseconds this batch's steps have cost so far, by part (sa_batch_host_times), and the number of query sets"""
        out = (_lib.c_uint64 * 4)()
        self._call("sa_batch_host_times", self._h, out)
        return {"fill_cpu_us": out[0] / 1e3, "fill_enqueue_us": out[1] / 1e3, "run_enqueue_us": out[2] / 1e3, "fills": int(out[3])}

    def stats(self, enable: bool = True) -> Tuple[int, int]:
        """(candidate docs scored by the sparse path since the last call, queries of the last run that
        were answered without a tile scan); diagnostics, switches the counting on / off."""
        cands = _lib.c_uint64(0)
        nq = _lib.c_uint64(0)
        self._call("sa_batch_stats", self._h, 1 if enable else 0, ctypes.byref(cands), ctypes.byref(nq))
        return cands.value, nq.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QueryQueue:
    """A stream of query SETS (each ``[B][T]`` term ids) through a ring of ``depth`` batches that a worker thread of the library feeds
    (``sa_queue_*``, csrc/sa_queue.hip): ``submit`` copies a set and returns a ticket at once -- the idf gather, the set's tables, the
    upload and the launches (``sa_batch_step``) happen on the worker --, ``fetch(ticket)`` waits for that set's results.  Tickets are
    served in order and every ticket must be fetched (``submit`` blocks while ``depth`` are outstanding).  The caller idiom of the
    reference's thread-pool drivers (test/test_msmarco.py:483-507) without the caller's thread paying for the preparation of a set.
    The batches take the options in force when the queue is created."""

    def __init__(self, index: DeviceIndex, n_queries: int, n_terms: int, k: int = 10, k1: float = 1.2, b: float = 0.75,
                 depth: int = 6, opts=None):
        self.index, self.api = index, index.api
        self.B, self.T, self.k, self.depth = int(n_queries), int(n_terms), int(k), int(depth)
        self._h = ctypes.c_void_p()
        base = _options.Options(index._opts_base, opts)
        with _options.creating(self.api, base):
            self.api.call("sa_queue_create", index._h, self.B, self.T, self.k, np.float32(k1), np.float32(b), self.depth, ctypes.byref(self._h))
        index._track(self)

    def submit(self, queries: np.ndarray) -> int:
        q = np.asarray(queries)
        if q.shape != (self.B, self.T):
            raise ValueError(f"submit takes [{self.B}][{self.T}] term ids")
        if q.dtype != np.uint32 or not q.flags.c_contiguous:
            q = as_u32(np.where((q >= 0) & (q < self.index.n_terms), q, NO_TERM).astype(np.uint32))
        t = _lib.c_uint64(0)
        self.api.call("sa_queue_submit", self._h, p_u32(q), ctypes.byref(t))
        return t.value

    def fetch(self, ticket: int) -> Tuple[np.ndarray, np.ndarray]:
        scores = np.empty((self.B, self.k), dtype=np.float32)
        docs = np.empty((self.B, self.k), dtype=np.uint64)
        self.api.call("sa_queue_fetch", self._h, _lib.c_uint64(int(ticket)), p_f32(scores), p_u64(docs))
        return scores, docs

    def last_route(self, slot: int = 0) -> str:
        """the route the last run of the batch in ``slot`` took (``QueryBatch.last_route``)"""
        bh = ctypes.c_void_p()
        self.api.call("sa_queue_batch", self._h, int(slot), ctypes.byref(bh))
        r = ctypes.c_int(0)
        self.api.call("sa_batch_last_route", bh, ctypes.byref(r))
        return {0: "exhaustive", 1: "pruned", 2: "staged"}.get(r.value, str(r.value))

    def batch_handle(self, slot: int):
        bh = ctypes.c_void_p()
        self.api.call("sa_queue_batch", self._h, int(slot), ctypes.byref(bh))
        return bh

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_queue_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PhraseBatch(QueryBatch):
    """B phrases (lists of term ids; any phrase ``score()`` takes: repeated terms, any length, ``slop`` (up to 32 terms) --
    one value or one per phrase) resident on the device; ``run()`` scores every phrase with BM25 over its match
    counts and keeps the top k docs.

    Same result contract as :class:`QueryBatch`; the dense single-phrase drop-in is
    :meth:`DeviceIndex.bm25_phrase_dense` (reference ``SearchArray.score([...])``)."""

    def __init__(self, index: DeviceIndex, phrases: Sequence[Sequence[int]], k: int = 10, k1: float = 1.2,
                 b: float = 0.75, idf: Optional[np.ndarray] = None, slop=0, opts=None):
        self.index = index
        self.api = index.api
        self.B = len(phrases)
        if self.B == 0:
            raise ValueError("empty phrase batch")
        n_terms = np.asarray([len(p) for p in phrases], dtype=np.int32)
        self.T = int(max(2, n_terms.max()))
        self.k = int(k)
        terms, n_terms, slops, idf = self._pack(phrases, idf, slop)
        self._h = ctypes.c_void_p()
        base = _options.Options(index._opts_base, opts)
        with _options.creating(self.api, base):
            self.api.call("sa_phrase_batch_create_ex", index._h, p_u32(terms),
                          n_terms.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                          slops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), p_f32(idf), self.B, self.T,
                          self.k, np.float32(k1), np.float32(b), ctypes.byref(self._h))
        self._init_opts(base)
        index._track(self)

    def _pack(self, phrases, idf, slop):
        index = self.index
        if len(phrases) != self.B:
            raise ValueError(f"this batch holds {self.B} phrases")
        n_terms = np.asarray([len(p) for p in phrases], dtype=np.int32)
        if n_terms.max() > self.T:
            raise ValueError(f"this batch holds phrases of at most {self.T} terms")
        terms = np.full((self.B, self.T), NO_TERM, dtype=np.uint32)
        for i, ph in enumerate(phrases):
            row = np.asarray(ph, dtype=np.int64)
            terms[i, :len(row)] = np.where((row >= 0) & (row < index.n_terms), row, NO_TERM)
        if idf is None:
            # idf of a phrase: float64 sum over its terms, then float32 (reference similarity.py:19-21)
            idf = np.asarray([compute_idf(index.corpus_size,
                                          np.asarray([index.docfreq(int(t)) if 0 <= int(t) < index.n_terms else 0
                                                      for t in ph])) for ph in phrases], dtype=np.float32)
        self._n_terms = n_terms
        slops = np.ascontiguousarray(np.broadcast_to(np.asarray(slop, dtype=np.int32), (self.B,)))   # one slop, or one per phrase
        if (slops < 0).any():
            raise ValueError("slop must be >= 0")
        return as_u32(terms), n_terms, slops, as_f32(idf)

    def reset(self, phrases: Sequence[Sequence[int]], idf: Optional[np.ndarray] = None, slop=0):
        """A new set of B phrases (none longer than the batch's max_terms) in this batch: ``sa_phrase_batch_reset``."""
        terms, n_terms, slops, idf = self._pack(phrases, idf, slop)
        self._call("sa_phrase_batch_reset", self._h, p_u32(terms),
                      n_terms.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                      slops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), p_f32(idf))


class DeviceVec:
    """A dense per-doc vector in HBM (float64, or 32-bit: float32 values / uint32 counters); Part 4 of
    the C ABI.  Used by ``searcharray_amd.solr`` to combine multi-field scores without leaving the device."""

    def __init__(self, api, n: int, f64: bool, device: int = 0):
        self.api, self.n, self.f64 = api, int(n), bool(f64)
        self._h = ctypes.c_void_p()
        api.call("sa_vec_create", int(device), self.n, 1 if f64 else 0, ctypes.byref(self._h))

    def zero(self):
        self.api.call("sa_vec_zero", self._h)

    def copy_from(self, other: "DeviceVec"):
        self.api.call("sa_vec_copy", self._h, other._h)

    def fetch(self, dtype=None) -> np.ndarray:
        out = np.empty(self.n, dtype=dtype if dtype is not None else (np.float64 if self.f64 else np.float32))
        self.api.call("sa_vec_fetch", self._h, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.sa_vec_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


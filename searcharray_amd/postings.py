"""Tokenized, searchable text as a pandas dtype -- the reference's public surface
(searcharray/postings.py) over the MI355X scoring path.

``SearchArray.index(strings)`` builds the positional index on the host, uploads it to HBM on
first use and answers ``termfreqs`` / ``docfreq`` / ``score`` / phrase queries with the HIP
kernels behind the C ABI (include/searcharray_hip.h).  The ExtensionArray protocol (slicing,
take, copy, setitem, concat, factorize, equality) is host bookkeeping over an immutable shared
index core plus a row selection.

Slices behave like the reference's filtered views (``FilteredPosns``, middle_out.py:291-317):
``docfreq`` -- and with it the idf of every score taken on a slice -- counts the docs of the SLICE that
contain the term, while ``corpus_size`` and ``avg_doc_length`` stay those of the whole index
(known answers from the reference in tests/test_search_api.py).
There is no CPU fallback: every query-time computation goes through the C ABI.
"""
from __future__ import annotations

import json
import numbers
import os
import warnings
from collections import Counter
from typing import Iterable, List, Optional, Tuple, Union

import numpy as np
import pandas as pd
from pandas.api.extensions import ExtensionArray, ExtensionDtype, register_extension_dtype, take
from pandas.api.types import is_list_like

from . import _lib
from . import roaringish as rz
from .device_index import DeviceIndex, NO_DOC, compute_idf
from .indexing import HostIndex, build_index_from_terms_list, build_index_from_tokenizer
from .similarity import compute_idf as similarity_idf, default_bm25
from .term_dict import TermDict, TermMissingError


class Terms:
    """One indexed doc: term -> tf, optional term -> positions, and the doc length
    (reference postings.py:57-165)."""

    def __init__(self, postings, doc_len: int = 0, posns: Optional[dict] = None, encoded=False):
        self.postings = postings
        self.encoded = encoded
        self.doc_len = doc_len
        self.posns = posns

    def termfreq(self, token):
        return self.postings[token]

    def terms(self):
        return self.postings.items()

    def positions(self, term=None):
        if self.posns is None:
            return {}
        if term is None:
            return self.posns.items()
        return self.posns[term]

    def raw_positions(self, term_dict, term=None):
        """(term id, positions) pairs, ids from ``term_dict`` (reference postings.py:94-101)."""
        if self.posns is None:
            return {}
        wanted = self.posns.items() if term is None else [(term, self.posns[term])]
        return [(term_dict.get_term_id(tok), plist) for tok, plist in wanted]

    def tf_to_dense(self, term_dict):
        """term frequencies as a vector over the dictionary (reference postings.py:103-108)."""
        dense = np.zeros(len(term_dict))
        for tok, tf in self.terms():
            dense[term_dict.get_term_id(tok)] = tf
        return dense

    def __len__(self):
        return len(self.postings)

    def __repr__(self):
        return f"Terms({set(self.postings.keys())})"

    __str__ = __repr__

    def __eq__(self, other):
        if isinstance(other, SearchArray):
            return other == self
        return isinstance(other, Terms) and self.postings == other.postings and self.doc_len == other.doc_len

    def __ne__(self, other):
        res = self.__eq__(other)
        return ~res if isinstance(res, np.ndarray) else not res

    def __lt__(self, other):
        # lexical comparison of the two sparse tf vectors
        for key in sorted(set(self.postings) | set(other.postings)):
            a, b = self.postings.get(key, 0), other.postings.get(key, 0)
            if a != b:
                return a < b
        return False

    def __le__(self, other):
        return self < other or self == other

    def __gt__(self, other):
        return not (self < other) and self != other

    def __hash__(self):
        return hash(json.dumps(self.postings, sort_keys=True))


class TermsDtype(ExtensionDtype):
    """pandas dtype ``tokenized_text`` (reference postings.py:168-203)."""
    name = 'tokenized_text'
    type = Terms
    kind = 'O'

    @classmethod
    def construct_from_string(cls, string):
        if not isinstance(string, str):
            raise TypeError(f"'construct_from_string' expects a string, got {type(string)}")
        if string == cls.name:
            return cls()
        raise TypeError(f"Cannot construct a '{cls.__name__}' from '{string}'")

    @classmethod
    def construct_array_type(cls):
        return SearchArray

    def __repr__(self):
        return 'TermsDtype()'

    @property
    def na_value(self):
        return Terms({})

    def valid_value(self, value):
        return isinstance(value, dict) or pd.isna(value) or isinstance(value, Terms)


register_extension_dtype(TermsDtype)


def ws_tokenizer(string):
    if pd.isna(string):
        return []
    if not isinstance(string, str):
        raise ValueError("Expected a string")
    return string.split()


class _IndexCore:
    """The index behind a SearchArray and all of its views: host arrays + the lazily created
    HBM-resident DeviceIndex.  Views (slices) share one core object, so an assignment through a
    view is visible to its base -- pandas' view semantics.  ``copy()`` gets its own core object
    that references the same immutable host arrays (and device handle) until one side is
    assigned to (copy-on-write at core granularity)."""

    def __init__(self, host: HostIndex, device: Optional[DeviceIndex] = None):
        self._device: Optional[DeviceIndex] = device
        self.replace(host, device)

    def replace(self, host: HostIndex, device: Optional[DeviceIndex] = None):
        self.host = host
        self.term_dict = host.term_dict
        self.doc_lens = host.doc_lens
        # reference indexing.py:282-284 / postings.py:296: np.mean over the float32 lengths
        self.avg_doc_length = np.mean(host.doc_lens) if len(host.doc_lens) else 0.0
        self.corpus_size = len(host.doc_lens)
        self._device = device

    def fork(self) -> "_IndexCore":
        return _IndexCore(self.host, self._device)

    def words_nbytes(self) -> int:
        """bytes of the roaringish words, wherever they live (host array, .dat file, HBM)"""
        h = self.host
        if h.has_words:
            return int(h._words.nbytes)
        if h.words_file is not None:
            return int(8 * np.sum(h.words_file[2], dtype=np.uint64))
        if self._device is not None:
            return int(8 * self._device.info().n_words)
        return int(h.words.nbytes)                       # nothing resident yet: host encode

    def sharded(self, devices) -> "ShardedIndex":
        """The same index as doc-range shards over several GPUs (searcharray_amd/sharded.py ->
        sa_sharded_create), kept for the next batched search on the same device list.  The library cuts the shards out
        of the encoded words on the HOST: an index that so far only lives in a .dat file or in HBM is read back once
        here (``host.words``) -- a one-time cost of asking for another device layout."""
        from .sharded import ShardedIndex
        key = tuple(int(d) for d in devices)
        cached = getattr(self, "_sharded", None)
        if cached is None or cached[0] != key or cached[1] is not self.host:
            if cached is not None:
                cached[2].close()
            h = self.host
            self._sharded = (key, h, ShardedIndex(h.words, h.term_off, h.doc_lens, key, avg_doc_len=self.avg_doc_length,
                                                   api=_lib.api()))
        return self._sharded[2]

    def device(self) -> DeviceIndex:
        if self._device is None:
            h = self.host
            if not h.has_words and h.words_file is not None:
                # file -> page-locked ring -> HBM, no host copy of the words
                path, src, length = h.words_file
                self._device = DeviceIndex.from_file(path, (src, length), h.doc_lens, n_terms=len(self.term_dict),
                                                     avg_doc_len=self.avg_doc_length,
                                                     corpus_size=self.corpus_size, api=_lib.api())
            elif not h.has_words and h.tokens is not None:
                # index build on the device: sort by term + roaringish encode in HBM
                dev = DeviceIndex.from_tokens(h.tokens, h.doc_ptr, len(self.term_dict), doc_lens=h.doc_lens,
                                              avg_doc_len=self.avg_doc_length, corpus_size=self.corpus_size,
                                              api=_lib.api())
                h.words_source = dev.words
                self._device = dev
            else:
                self._device = DeviceIndex(h.words, h.term_off, h.doc_lens, avg_doc_len=self.avg_doc_length,
                                           corpus_size=self.corpus_size, api=_lib.api())
        return self._device

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_device"] = None                    # HBM handles do not pickle; re-uploaded on demand
        state.pop("_sharded", None)
        return state

    # -- host-side doc reconstruction (scalar __getitem__, equality, positions)
    def term_words_of_doc(self, term_id: int, doc_id: int) -> np.ndarray:
        h = self.host
        w = h.words[int(h.term_off[term_id]):int(h.term_off[term_id + 1])]
        lo = np.searchsorted(w, np.uint64(doc_id) << np.uint64(rz.KEY_SHIFT), side="left")
        hi = np.searchsorted(w, np.uint64(doc_id + 1) << np.uint64(rz.KEY_SHIFT), side="left")
        return w[lo:hi]

    def doc_as_terms(self, doc_id: int) -> Terms:
        h = self.host
        a, b = int(h.doc_term_ptr[doc_id]), int(h.doc_term_ptr[doc_id + 1])
        postings, posns = {}, {}
        for k in range(a, b):
            tid = int(h.doc_term_ids[k])
            term = self.term_dict.get_term(tid)
            _, p = rz.decode_positions(self.term_words_of_doc(tid, doc_id))
            if len(p):
                postings[term] = len(p)
                posns[term] = p.astype(np.uint32)
            elif h.doc_term_tfs is not None:
                postings[term] = h.doc_term_tfs[k]
            else:
                postings[term] = 0
        return Terms(postings, doc_len=h.doc_lens[doc_id], posns=posns if posns else None)


class _PosnsAdapter:
    """``arr.posns`` of the reference (PosnBitArray) as seen by callers: df / warm / cache hooks."""

    def __init__(self, array: "SearchArray"):
        self._array = array

    def docfreq(self, term_id: int):
        return self._array._core.device().docfreq(int(term_id))

    def warm(self):
        self._array._core.device()            # upload + derive every term's tf / df on the device

    def clear_cache(self):
        pass                                   # derived postings are device-resident, nothing to clear

    @property
    def nbytes(self):
        return self._array._core.words_nbytes()


class SearchArray(ExtensionArray):
    """An array of tokenized text (reference postings.py:228-708)."""

    dtype = TermsDtype()

    def __init__(self, postings, tokenizer=ws_tokenizer, avoid_copies=True):
        if not is_list_like(postings):
            raise TypeError(f"Expected list-like object, got {type(postings)}")
        self.avoid_copies = avoid_copies
        self.tokenizer = tokenizer
        self._core = _IndexCore(build_index_from_terms_list(postings, Terms))
        self._rows: Optional[np.ndarray] = None

    @classmethod
    def index(cls, array: Iterable, tokenizer=ws_tokenizer, truncate=False, batch_size=100000,
              avoid_copies=True, workers=4, cache_gt_than=25, data_dir: Optional[str] = None,
              autowarm=True) -> 'SearchArray':
        """Index strings with ``tokenizer`` (reference postings.py:249-300).  The docs are tokenised ``batch_size`` at a
        time, ``workers`` batches ahead on a thread pool (indexing.build_index_from_tokenizer; reference
        indexing.py:235-296); ``cache_gt_than`` is accepted for signature compatibility: the derived tf/df live on the
        device.  ``data_dir``: as in the reference
        (indexing.py:291-293 -> PosnBitArray.memmap, middle_out.py:333-335) the encoded positions are
        written to ``<data_dir>/<number of files in it>.dat`` as raw uint64 and pickles of the array
        then carry the filename instead of the words; here the file is written from, and read back
        into, HBM directly (csrc/sa_io.hip)."""
        if not is_list_like(array):
            raise TypeError(f"Expected list-like object, got {type(array)}")
        host = build_index_from_tokenizer(array, tokenizer, truncate=truncate, batch_size=batch_size, workers=workers)
        arr = cls.__new__(cls)
        arr.avoid_copies = avoid_copies
        arr.tokenizer = tokenizer
        arr._core = _IndexCore(host)
        arr._rows = None
        if data_dir is not None and len(host.term_dict):
            arr.memmap(data_dir)
        return arr

    def memmap(self, data_dir: str) -> str:
        """Persist the encoded positions under ``data_dir`` (reference PosnBitArray.memmap,
        middle_out.py:333-335; file naming of memmap_arrays.py:7-12).  Returns the file name."""
        core = self._core
        filename = os.path.join(data_dir, f"{len(os.listdir(data_dir))}") + ".dat"
        term_off = core.device().save(filename)
        core.host.words_file = (filename, np.ascontiguousarray(term_off[:-1]),
                                np.ascontiguousarray(np.diff(term_off)))
        return filename

    @classmethod
    def from_memmap(cls, filename: str, metadata, terms: Iterable[str], doc_lens, tokenizer=ws_tokenizer,
                    avoid_copies=True) -> 'SearchArray':
        """Open an index persisted in the reference's on-disk format without re-indexing: ``filename``
        is the raw uint64 ``.dat`` (phrase/memmap_arrays.py:158-161), ``metadata`` the ArrayDict
        metadata ``{term_id: {'offset', 'length'}}`` (memmap_arrays.py:28-54), ``terms`` the term
        strings in id order (TermDict, reference term_dict.py) and ``doc_lens`` the doc lengths.
        The words go from the file straight to HBM on first use."""
        term_dict = TermDict()
        term_dict.add_terms(list(terms))
        V = len(term_dict)
        src = np.zeros(V, dtype=np.uint64)
        length = np.zeros(V, dtype=np.uint64)
        for k, v in metadata.items():
            if not 0 <= int(k) < V:
                raise ValueError(f"metadata names term id {k}; the dictionary has {V} terms")
            src[int(k)], length[int(k)] = int(v["offset"]), int(v["length"])
        host = HostIndex(term_dict, np.ascontiguousarray(doc_lens, dtype=np.float32))
        host.words_file = (filename, src, length)
        arr = cls.__new__(cls)
        arr.avoid_copies = avoid_copies
        arr.tokenizer = tokenizer
        arr._core = _IndexCore(host)
        arr._rows = None
        return arr

    @classmethod
    def _view(cls, core, rows, tokenizer, avoid_copies=True):
        arr = cls.__new__(cls)
        arr.avoid_copies = avoid_copies
        arr.tokenizer = tokenizer
        arr._core = core
        arr._rows = rows
        return arr

    def warm(self):
        self.posns.warm()

    # ---- attributes the reference exposes ------------------------------------------------
    @property
    def term_dict(self):
        return self._core.term_dict

    @property
    def avg_doc_length(self):
        return self._core.avg_doc_length

    @property
    def corpus_size(self):
        return self._core.corpus_size

    @property
    def doc_lens(self) -> np.ndarray:
        return self._core.doc_lens if self._rows is None else self._core.doc_lens[self._rows]

    @property
    def posns(self):
        return _PosnsAdapter(self)

    def _row_ids(self) -> np.ndarray:
        return np.arange(self._core.corpus_size, dtype=np.int64) if self._rows is None else self._rows


    # ---- ExtensionArray protocol -----------------------------------------------------------
    @classmethod
    def _from_sequence(cls, scalars, dtype=None, copy=False):
        if dtype is not None and not isinstance(dtype, TermsDtype):
            return scalars
        if isinstance(scalars, SearchArray):
            return scalars.copy() if copy else scalars
        if isinstance(scalars, np.ndarray) and scalars.dtype != object and scalars.dtype.kind not in 'US':
            return scalars
        return cls(scalars)

    @classmethod
    def _from_factorized(cls, values, original):
        return cls(values)

    def _values_for_factorize(self):
        return np.asarray(self._as_terms_list(), dtype=object), Terms({})

    @classmethod
    def _concat_same_type(cls, to_concat):
        items: List[Terms] = []
        for arr in to_concat:
            items.extend(arr._as_terms_list())
        return SearchArray(items, tokenizer=to_concat[0].tokenizer)

    def __len__(self):
        return self._core.corpus_size if self._rows is None else len(self._rows)

    @property
    def nbytes(self):
        # from SIZES, never from the data: pandas calls this from DataFrame.info() / memory_usage(), and an
        # index built on the device or opened from a .dat file must not be copied to host RAM for that
        h = self._core.host
        n_terms = len(self.term_dict)
        doc_terms = h._doc_term_ids.nbytes if h._doc_term_ids is not None else 0
        return self._core.words_nbytes() + 8 * (n_terms + 1) + h.doc_lens.nbytes + doc_terms + self.term_dict.nbytes

    def memory_usage(self, deep=False):
        return self.nbytes

    def _as_terms_list(self) -> List[Terms]:
        return [self._core.doc_as_terms(int(d)) for d in self._row_ids()]

    def __getitem__(self, key):
        key = pd.api.indexers.check_array_indexer(self, key)
        if isinstance(key, numbers.Integral):
            n = len(self)
            if key < -n or key >= n:
                raise IndexError("index out of bounds")
            return self._core.doc_as_terms(int(self._row_ids()[key]))
        rows = self._row_ids()[key]
        return SearchArray._view(self._core, np.asarray(rows, dtype=np.int64), self.tokenizer, self.avoid_copies)

    def __setitem__(self, key, value):
        key = pd.api.indexers.check_array_indexer(self, key)
        if isinstance(value, (pd.Series,)):
            value = value.values
        if isinstance(value, pd.DataFrame):
            value = value.values.flatten()
        if isinstance(value, SearchArray):
            value = np.asarray(value._as_terms_list(), dtype=object)
        if isinstance(value, list):
            value = np.asarray(value, dtype=object)
        if not isinstance(value, np.ndarray) and not self.dtype.valid_value(value):
            raise ValueError(f"Cannot set non-object array to SearchArray -- you passed type:{type(value)} -- {value}")
        if isinstance(key, numbers.Integral) and isinstance(value, np.ndarray):
            raise ValueError("Cannot set a single value to an array")

        def as_terms(v):
            if isinstance(v, Terms):
                return v
            if isinstance(v, dict):
                return Terms(v, doc_len=len(v))
            if v is None or (isinstance(v, float) and np.isnan(v)):
                return Terms({})
            raise ValueError(f"Cannot store {type(v)} in a SearchArray")

        # Assignment rewrites the shared core in place (slow path, like the reference's "this is
        # slow" warning): the docs this array's rows select are replaced and the index is rebuilt.
        core = self._core
        targets = np.atleast_1d(self._row_ids()[key])
        if isinstance(value, np.ndarray):
            new_vals = [as_terms(v) for v in value]
            if len(new_vals) != len(targets):
                if len(new_vals) == 1:
                    new_vals = new_vals * len(targets)
                else:
                    raise ValueError("cannot set using a list-like indexer with a different length than the value")
        else:
            new_vals = [as_terms(value)] * len(targets)
        items = [core.doc_as_terms(d) for d in range(core.corpus_size)]
        for d, v in zip(targets, new_vals):
            items[int(d)] = v
        core.replace(build_index_from_terms_list(items, Terms))

    def __array__(self, dtype=None, copy=None):
        # object array of Terms; always a fresh materialisation (nothing to share with numpy), so
        # copy=False gets pandas' own NumPy-2 transition warning instead of a silent copy
        if copy is False:
            warnings.warn("Starting with NumPy 2.0, the behavior of the 'copy' keyword has changed and passing "
                          "'copy=False' raises an error when returning a zero-copy NumPy array is not possible. "
                          "SearchArray always materialises a new object array.", FutureWarning, stacklevel=2)
        out = np.empty(len(self), dtype=object)
        for i, t in enumerate(self._as_terms_list()):
            out[i] = t
        return out

    def isna(self):
        return self.doc_lens == 0

    def take(self, indices, allow_fill=False, fill_value=None):
        rows = self._row_ids()
        picked = take(np.arange(len(rows)), indices, allow_fill=allow_fill, fill_value=-1)
        picked = np.asarray(picked, dtype=np.int64)
        if allow_fill and (picked == -1).any():
            if fill_value is None or pd.isna(fill_value):
                fill_value = Terms({}, encoded=True)
            items = [self._core.doc_as_terms(int(rows[i])) if i >= 0 else fill_value for i in picked]
            return SearchArray(items, tokenizer=self.tokenizer)
        return SearchArray._view(self._core.fork(), rows[picked], self.tokenizer, self.avoid_copies)   # take copies

    def copy(self):
        rows = None if self._rows is None else self._rows.copy()
        return SearchArray._view(self._core.fork(), rows, self.tokenizer, self.avoid_copies)

    def unique(self):
        return self[:]

    def value_counts(self, dropna: bool = True):
        counts = Counter(self._as_terms_list())
        if dropna:
            counts.pop(Terms({}), None)
        return pd.Series(counts)

    def __iter__(self):
        if len(self) > 10000:
            warnings.warn("Iterating over SearchArray is very slow and not recommended.")
        return super().__iter__()

    def __ne__(self, other):
        if isinstance(other, (pd.DataFrame, pd.Series, pd.Index)):
            return NotImplemented
        return ~(self == other)

    def __eq__(self, other):
        if isinstance(other, (pd.DataFrame, pd.Series, pd.Index)):
            return NotImplemented
        if isinstance(other, SearchArray):
            if len(self) != len(other):
                return False
            if len(other) == 0:
                return np.array([], dtype=bool)
            if other._core.host is self._core.host:
                # same index: equal row ids are equal docs; different rows may still hold equal CONTENT
                # (the reference compares term_mat rows and doc_lens, postings.py:463-464)
                ra, rb = self._row_ids(), other._row_ids()
                out = ra == rb
                # cheap invariants first, vectorised: equal docs have equal lengths and equally many distinct terms;
                # then the term-id slices of the doc -> term CSR; positions are only decoded (tf per term) for the
                # few pairs that survive all of that (`arr == arr.take(perm)` on a large array used to decode every
                # differing row on the host)
                h = self._core.host
                ptr = h.doc_term_ptr
                maybe = ~out & (h.doc_lens[ra] == h.doc_lens[rb]) & ((ptr[ra + 1] - ptr[ra]) == (ptr[rb + 1] - ptr[rb]))
                for i in np.flatnonzero(maybe):
                    a0, a1, b0, b1 = int(ptr[ra[i]]), int(ptr[ra[i] + 1]), int(ptr[rb[i]]), int(ptr[rb[i] + 1])
                    # (as SETS: an index built from Terms / dict postings keeps a doc's term ids in insertion order)
                    if not np.array_equal(np.sort(h.doc_term_ids[a0:a1]), np.sort(h.doc_term_ids[b0:b1])):
                        continue
                    out[i] = bool(self._core.doc_as_terms(int(ra[i])) == self._core.doc_as_terms(int(rb[i])))
                return out
            a, b = self._as_terms_list(), other._as_terms_list()
            return np.asarray([bool(x == y) for x, y in zip(a, b)], dtype=bool)
        if isinstance(other, Terms):
            return np.asarray([bool(x == other) for x in self._as_terms_list()], dtype=bool)
        if is_list_like(other):
            if len(self) != len(other):
                return False
            if len(other) == 0:
                return np.array([], dtype=bool)
            return self == SearchArray(other, tokenizer=self.tokenizer)
        return np.full(len(self), False)

    # ---- search API ---------------------------------------------------------------------------
    def _check_token_arg(self, token):
        if isinstance(token, str):
            return token
        if isinstance(token, list) and len(token) == 1:
            return token[0]
        if isinstance(token, list):
            return token
        raise TypeError("Expected a string or list of strings for phrases")

    @staticmethod
    def _check_posn_args(slop, min_posn, max_posn):
        if slop < 0:
            raise ValueError("slop must be >= 0")

    def _term_id(self, token: str) -> int:
        try:
            return self.term_dict.get_term_id(token)
        except TermMissingError:
            return -1

    def termfreqs(self, token: Union[List[str], str], slop: int = 0, min_posn: Optional[int] = None,
                  max_posn: Optional[int] = None) -> np.ndarray:
        """reference postings.py:607-638 (single term) / :689-708 (phrase)."""
        token = self._check_token_arg(token)
        self._check_posn_args(slop, min_posn, max_posn)
        if len(self._core.doc_lens) == 0:
            return np.zeros(len(self), dtype=np.float32)
        dev = self._core.device()
        # a slice asks the device for its rows only (gathered in HBM: the copy is len(self) floats)
        if isinstance(token, list):
            ids = [self._term_id(t) for t in token]
            return dev.phrase_freqs_dense(ids, slop=slop, min_posn=min_posn, max_posn=max_posn, rows=self._rows)
        tid = self._term_id(token)
        if tid < 0:
            return np.zeros(len(self), dtype=np.float32)            # unknown term: zeros, never raises
        return dev.termfreqs_dense(tid, min_posn=min_posn, max_posn=max_posn, rows=self._rows)

    def docfreq(self, token: str) -> int:
        """reference postings.py:640-647."""
        if not isinstance(token, str):
            raise TypeError("Expected a string")
        tid = self._term_id(token)
        if tid < 0 or len(self._core.doc_lens) == 0:
            return 0
        if self._rows is None:
            return self._core.device().docfreq(tid)
        # A sliced array counts the docs of the SLICE that contain the term (the reference's slices
        # carry FilteredPosns, middle_out.py:291-317, so docfreq -- and with it the idf of every score
        # on a slice -- is subset-local while corpus_size and avg_doc_length stay global).
        tf = self._core.device().termfreqs_dense(tid, rows=np.unique(self._rows))
        return np.uint64(np.count_nonzero(tf))

    def doclengths(self) -> np.ndarray:
        return self.doc_lens

    def score_device(self, token: Union[str, List[str]], similarity=default_bm25, slop: int = 0):
        """``score()`` with the result LEFT ON THE DEVICE: a ``DeviceVec`` (float32[len(index)]; ``.fetch()`` copies it
        to the host, ``.close()`` frees it).  The drop-in ``score()`` returns a dense host array as the reference does
        (postings.py:652-680), which makes a call PCIe-bound (40 MB at 10 M docs); a caller that combines or ranks scores
        on the GPU -- as ``searcharray_amd.solr.edismax`` does -- keeps them in HBM this way.  Stock BM25 similarity, whole
        (unsliced) arrays."""
        from .device_index import DeviceVec
        from ._lib import p_u32
        token = self._check_token_arg(token)
        self._check_posn_args(slop, None, None)
        if getattr(similarity, "kind", None) != "bm25" or self._rows is not None:
            raise ValueError("score_device needs a stock BM25 similarity and a whole (unsliced) array")
        tokens = [token] if isinstance(token, str) else token
        dfs = np.asarray([self.docfreq(t) for t in tokens])
        dev = self._core.device()
        idf = np.float32(compute_idf(self.corpus_size, dfs))
        unknown = dev.n_terms
        tarr = np.asarray([t if (t := self._term_id(tok)) >= 0 else unknown for tok in tokens], dtype=np.uint32)
        out = DeviceVec(dev.api, dev.n_docs, False)
        k1, b = np.float32(similarity.k1), np.float32(similarity.b)
        try:
            if len(tokens) == 1:
                dev.into_vec(out, None, "sa_index_bm25_dense", p_u32(tarr), np.asarray([idf], np.float32).ctypes.data_as(
                    _lib.ctypes.POINTER(_lib.ctypes.c_float)), 1, k1, b)
            else:
                dev.into_vec(out, None, "sa_index_bm25_phrase_dense_posn", p_u32(tarr), len(tarr), int(slop), -1, -1, idf, k1, b)
        except Exception:
            out.close()
            raise
        return out

    def score(self, token: Union[str, List[str]], similarity=default_bm25, slop: int = 0,
              min_posn: Optional[int] = None, max_posn: Optional[int] = None) -> np.ndarray:
        """BM25 (or any Similarity) of one term or one phrase for every doc (reference
        postings.py:652-680).  Tagged BM25 closures run entirely on the GPU."""
        token = self._check_token_arg(token)
        self._check_posn_args(slop, min_posn, max_posn)
        tokens = [token] if isinstance(token, str) else token
        dfs = np.asarray([self.docfreq(t) for t in tokens])
        if len(self._core.doc_lens) == 0:
            return np.zeros(len(self), dtype=np.float32)
        posn_filter = min_posn is not None or max_posn is not None
        # (k1 == 0 or b == 1 can make the BM25 denominator 0 for docs without the term: the reference
        # then returns 0/0 = NaN there, which only the dense tf -> bm25_score route reproduces)
        degenerate = getattr(similarity, "k1", 1.0) == 0 or getattr(similarity, "b", 0.0) == 1
        if getattr(similarity, "kind", None) == "bm25" and not (posn_filter and len(tokens) == 1) and not degenerate:
            dev = self._core.device()
            idf = np.float32(compute_idf(self.corpus_size, dfs))
            ids = [self._term_id(t) for t in tokens]
            if len(ids) == 1:
                return dev.bm25_dense(ids, k1=similarity.k1, b=similarity.b, idf=np.asarray([idf], np.float32),
                                      rows=self._rows)
            return dev.bm25_phrase_dense(ids, k1=similarity.k1, b=similarity.b, slop=slop, idf=idf,
                                         min_posn=min_posn, max_posn=max_posn, rows=self._rows)
        kind = getattr(similarity, "kind", None)
        if (kind in DeviceIndex.SIMILARITY_KINDS and self.avg_doc_length != 0
                and (isinstance(token, str) or len(tokens) >= 2)):
            # the reference's other stock similarities (similarity.py:41-89) as device kernels
            if kind == "classic":
                idf = np.log((self.corpus_size + 1) / (np.sum(dfs, axis=0) + 1)) + 1
                k1 = b = 0.0
            else:
                idf, k1, b = similarity_idf(self.corpus_size, dfs), similarity.k1, similarity.b   # float64

            return self._core.device().similarity_dense(kind, [self._term_id(t) for t in tokens], idf=idf, k1=k1, b=b,
                                                        slop=slop, min_posn=min_posn, max_posn=max_posn,
                                                        rows=self._rows)
        # any other Similarity (or a position-restricted single term): tf on the device, then the
        # similarity callable (the stock BM25 closure applies the BM25 kernel through the C ABI)
        tfs = self.termfreqs(token, slop=slop, min_posn=min_posn, max_posn=max_posn)
        return similarity(tfs, dfs, self.doc_lens, self.avg_doc_length, self.corpus_size)

    def memory_report(self, N=1000):
        """Where the bytes are (reference postings.py:570-602): host structures, what the index holds in
        HBM, and the ``N`` largest terms by positional words."""
        def human(n):
            for unit in ("bytes", "KB", "MB", "GB"):
                if n < 1024 or unit == "GB":
                    return f"{n:.2f} {unit}" if unit != "bytes" else f"{int(n)} bytes"
                n /= 1024.0
        h = self._core.host
        sizes = np.diff(h.term_off.astype(np.int64)) * 8
        order = np.argsort(-sizes, kind="stable")[:min(N, len(sizes))]
        lines = ["", "SearchArray Memory Report", "-------------------------",
                 f"Number of Terms: {len(self.term_dict)}", "-------------------------",
                 f"Doc -> terms CSR: {human(h.doc_term_ids.nbytes + h.doc_term_ptr.nbytes)}",
                 f"Positions:        {human(h.words.nbytes)}",
                 f"Term Dictionary:  {human(self.term_dict.nbytes)}"]
        if self._core._device is not None:
            lines.append(f"On the device:    {human(self._core._device.info().hbm_bytes)}")
        lines.append("")
        running = 0
        for rank, tid in enumerate(order):
            running += int(sizes[tid])
            lines.append(f"Term {rank}: {self.term_dict.get_term(int(tid))} - {human(int(sizes[tid]))} - Cumulative: {human(running)}")
        return "\n".join("        " + ln for ln in lines) + "\n"

    # -- batched top-k (no counterpart in the reference: its callers loop over score() + argpartition)
    def search(self, queries, k: int = 10, similarity=default_bm25, devices=None) -> Tuple[np.ndarray, np.ndarray]:
        """Top-``k`` docs for many queries at once, without materialising dense score vectors:
        ``queries`` is a list of token lists (each scored as a disjunction: the sum of its terms' BM25,
        ``np.sum([arr.score(t) for t in q], axis=0)`` in reference terms) or a list of strings (each run
        through this array's tokenizer first).  Returns ``(scores float32[B][k], doc_ids uint64[B][k])``
        sorted by score descending then doc id ascending; unused slots hold score 0 and doc id 2**64-1.
        Needs a stock BM25 similarity (``bm25_similarity(k1, b)``) and the whole array (not a slice).
        ``devices=[0, 1, ...]``: the index is cut into doc-id ranges, one per listed GPU, every shard is
        scored concurrently with the global statistics and the per-shard top-k are merged over RCCL
        (searcharray_amd/sharded.py) -- same results as on one device."""
        return self._topk(queries, k, similarity, phrases=False, devices=devices)

    def search_phrases(self, phrases, k: int = 10, similarity=default_bm25, devices=None, slop=0) -> Tuple[np.ndarray, np.ndarray]:
        """Like :meth:`search`, each query a phrase: the top-``k`` of ``arr.score(phrase, slop=slop)`` (``slop``: one
        value or one per phrase).  Any phrase ``score`` takes is fine -- repeated tokens, long phrases, slop."""
        return self._topk(phrases, k, similarity, phrases=True, devices=devices, slop=slop)

    def _topk(self, queries, k, similarity, phrases, devices=None, slop=0):
        if getattr(similarity, "kind", None) != "bm25":
            raise ValueError("batched search needs a stock BM25 similarity (bm25_similarity(k1, b))")
        if self._rows is not None:
            raise ValueError("batched search runs on the whole indexed array, not on a slice")
        toks = [list(self.tokenizer(q)) if isinstance(q, str) else [self._check_token_arg(t) for t in q] for q in queries]
        B = len(toks)
        if B == 0 or len(self._core.doc_lens) == 0:
            return np.zeros((B, k), np.float32), np.full((B, k), NO_DOC, np.uint64)
        # (an explicit device list is honoured even when it names ONE device: a one-shard handle on that GPU)
        dev = self._core.sharded(devices) if devices is not None and len(devices) >= 1 else self._core.device()
        unknown = dev.n_terms                                   # any id >= n_terms matches nothing
        ids = [[t if (t := self._term_id(tok)) >= 0 else unknown for tok in q] for q in toks]
        if phrases:
            batch = dev.phrase_batch(ids, k=k, k1=similarity.k1, b=similarity.b, slop=slop)
        else:
            T = max(1, max(len(q) for q in ids))
            mat = np.full((B, T), unknown, dtype=np.int64)
            for i, q in enumerate(ids):
                mat[i, :len(q)] = q
            batch = dev.batch(mat, k=k, k1=similarity.k1, b=similarity.b)
        try:
            batch.run()
            return batch.fetch()
        finally:
            batch.close()

    def positions(self, token: str, key=None) -> List[np.ndarray]:
        """positions of ``token`` per doc (reference postings.py:682-687)."""
        tid = self.term_dict.get_term_id(token)
        rows = self._row_ids()
        if key is not None:
            rows = np.atleast_1d(rows[key])
        out = []
        for d in rows:
            _, p = rz.decode_positions(self._core.term_words_of_doc(tid, int(d)))
            out.append(p.astype(np.uint32))
        return out

"""ctypes binding of libsearcharray_hip.so (C ABI: include/searcharray_hip.h).

The library is the ONLY compute path of this package: there is no CPU fallback.  If the
gfx950 shared object is missing or cannot be loaded, importing a compute entry point raises
``SearchArrayHipError`` with build instructions.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint32, c_uint64,
                    c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libsearcharray_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

u64p = POINTER(c_uint64)
u32p = POINTER(c_uint32)
f32p = POINTER(c_float)
i64p = POINTER(c_int64)


class SearchArrayHipError(RuntimeError):
    """Raised for any failure reported by (or while loading) the HIP library."""


class IndexInfo(ctypes.Structure):
    _fields_ = [("n_docs", c_uint64), ("doc_base", c_uint64), ("corpus_size", c_uint64),
                ("n_words", c_uint64), ("n_postings", c_uint64),
                ("n_terms", c_uint32), ("tile_docs", c_uint32), ("n_tiles", c_uint32),
                ("n_dir_terms", c_uint32), ("hbm_bytes", c_uint64), ("device", c_int),
                ("dl_packed", c_int), ("n_docdir_terms", c_uint32), ("n_tf8_terms", c_uint32)]


# name -> (restype, argtypes).  Every symbol declared in include/searcharray_hip.h.
class DenseDest(ctypes.Structure):
    """sa_dense_dest_t (include/searcharray_hip.h): where a `_to` dense call puts its result"""
    _fields_ = [("rows", POINTER(c_uint64)), ("n_rows", c_uint64), ("vec", c_void_p), ("boost", c_float), ("has_boost", c_int)]


PROTOTYPES = {
    "sa_last_error": (c_char_p, []),
    # Part 0: options (sa_options_t is passed by address: searcharray_amd/options.py builds it from sa_option_name)
    "sa_options_init": (None, [c_void_p]),
    "sa_options_set": (c_int, [c_void_p, c_char_p, c_int64]),
    "sa_options_get": (c_int, [c_void_p, c_char_p, i64p]),
    "sa_option_count": (c_int, []),
    "sa_option_name": (c_char_p, [c_int]),
    "sa_options_process_defaults_get": (c_int, [c_void_p]),
    "sa_options_set_thread_defaults": (c_int, [c_void_p]),
    "sa_index_set_options": (c_int, [c_void_p, c_void_p]),
    "sa_index_get_options": (c_int, [c_void_p, c_void_p]),
    "sa_batch_set_options": (c_int, [c_void_p, c_void_p]),
    "sa_batch_get_options": (c_int, [c_void_p, c_void_p]),
    "sa_abi_version": (c_int, []),
    "sa_device_count": (c_int, [POINTER(c_int)]),
    "sa_device_name": (c_int, [c_int, ctypes.c_char_p, c_int]),
    # Part 1
    "sa_bm25_score": (c_int, [f32p, f32p, c_float, c_float, c_float, c_float, c_int64]),
    "sa_as_dense": (c_int, [u64p, f32p, c_int64, f32p, c_int64]),
    "sa_popcount64_reduce": (c_int, [u64p, c_int64, c_uint64, c_uint64, u64p, f32p, i64p]),
    "sa_unique": (c_int, [u64p, c_int64, c_uint64, u64p, i64p]),
    "sa_popcount64": (c_int, [u64p, c_int64, u64p]),
    "sa_intersect": (c_int, [u64p, c_int64, u64p, c_int64, c_uint64, c_int, u64p, u64p, i64p, i64p]),
    "sa_adjacent": (c_int, [u64p, c_int64, u64p, c_int64, c_uint64, u64p, u64p, i64p]),
    "sa_intersect_with_adjacents": (c_int, [u64p, c_int64, u64p, c_int64, c_uint64, u64p, u64p, i64p, u64p, u64p, i64p]),
    "sa_merge": (c_int, [u64p, c_int64, u64p, c_int64, c_int, u64p, i64p]),
    "sa_sort_merge_counts": (c_int, [u64p, f32p, c_int64, u64p, f32p, c_int64, u64p, f32p, i64p]),
    "sa_popcount_reduce_at": (c_int, [u64p, u64p, c_int64, u64p, f32p, i64p]),
    "sa_key_sum_over": (c_int, [u64p, u64p, c_int64, u64p, f32p, i64p]),
    "sa_payload_slice": (c_int, [u64p, c_int64, c_uint64, c_uint64, c_uint64, u64p, i64p]),
    "sa_span_search": (c_int, [u64p, u64p, c_int, c_uint64, u64p, u64p, POINTER(c_int64)]),
    "sa_stream_probe": (c_int, [c_uint64, c_int, c_int, POINTER(c_double)]),
    # Part 2
    "sa_index_create": (c_int, [c_int, c_uint64, c_uint64, c_uint32, u64p, u64p, f32p, c_float,
                                c_uint64, c_uint32, POINTER(c_void_p)]),
    "sa_index_create_from_tokens": (c_int, [c_int, c_uint64, c_uint64, c_uint32, u32p, u64p, f32p, c_float, c_uint64,
                                            c_uint32, POINTER(c_void_p)]),
    "sa_index_create_from_file": (c_int, [c_int, c_uint64, c_uint64, c_uint32, c_char_p, u64p, u64p, f32p, c_float,
                                          c_uint64, c_uint32, POINTER(c_void_p)]),
    "sa_index_save": (c_int, [c_void_p, c_char_p]),
    "sa_index_similarity_dense": (c_int, [c_void_p, u32p, c_int, c_int, c_int64, c_int64, c_int, c_double, c_double,
                                          c_double, c_void_p]),
    "sa_index_words": (c_int, [c_void_p, u64p, u64p]),
    "sa_index_destroy": (c_int, [c_void_p]),
    "sa_index_synchronize": (c_int, [c_void_p]),
    "sa_index_docfreq": (c_int, [c_void_p, c_uint32, u64p]),
    "sa_index_docfreqs": (c_int, [c_void_p, u64p]),
    "sa_index_termfreqs_dense": (c_int, [c_void_p, c_uint32, f32p]),
    "sa_index_termfreqs_sparse": (c_int, [c_void_p, c_uint32, u64p, f32p, i64p]),
    "sa_index_bm25_dense": (c_int, [c_void_p, u32p, f32p, c_int, c_float, c_float, f32p]),
    "sa_index_phrase_freqs_dense": (c_int, [c_void_p, u32p, c_int, c_int, f32p]),
    "sa_index_bm25_phrase_dense": (c_int, [c_void_p, u32p, c_int, c_int, c_float, c_float, c_float, f32p]),
    "sa_index_phrase_freqs_dense_posn": (c_int, [c_void_p, u32p, c_int, c_int, c_int64, c_int64, f32p]),
    "sa_index_bm25_phrase_dense_posn": (c_int, [c_void_p, u32p, c_int, c_int, c_int64, c_int64, c_float, c_float, c_float, f32p]),
    "sa_index_termfreqs_dense_posn": (c_int, [c_void_p, c_uint32, c_int64, c_int64, f32p]),
    "sa_index_last_profile": (c_int, [c_void_p, POINTER(c_double), u64p]),
    "sa_index_info": (c_int, [c_void_p, POINTER(IndexInfo)]),
    "sa_batch_create": (c_int, [c_void_p, u32p, f32p, c_int, c_int, c_int, c_float, c_float,
                                POINTER(c_void_p)]),
    "sa_phrase_batch_create": (c_int, [c_void_p, u32p, POINTER(ctypes.c_int32), f32p, c_int, c_int, c_int, c_float,
                                       c_float, POINTER(c_void_p)]),
    "sa_phrase_batch_create_ex": (c_int, [c_void_p, u32p, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), f32p, c_int, c_int,
                                          c_int, c_float, c_float, POINTER(c_void_p)]),
    "sa_batch_reset": (c_int, [c_void_p, u32p, f32p]),
    "sa_phrase_batch_reset": (c_int, [c_void_p, u32p, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), f32p]),
    "sa_batch_run": (c_int, [c_void_p, c_int]),
    "sa_batch_run_local": (c_int, [c_void_p, c_void_p, c_int]),
    "sa_batch_merge_gathered": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "sa_batch_fetch": (c_int, [c_void_p, f32p, u64p]),
    "sa_batch_profile": (c_int, [c_void_p, POINTER(c_double), u64p, u64p]),
    "sa_index_set_idf_table": (c_int, [c_void_p, f32p, c_uint32]),
    "sa_batch_step": (c_int, [c_void_p, u32p]),
    "sa_comm_library_info": (c_int, [POINTER(c_int), ctypes.c_char_p, c_int]),
    "sa_batch_group_info": (c_int, [c_void_p, POINTER(c_uint32)]),
    "sa_batch_last_route": (c_int, [c_void_p, POINTER(c_int)]),
    "sa_batch_host_times": (c_int, [c_void_p, POINTER(c_uint64)]),
    "sa_queue_create": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, POINTER(c_void_p)]),
    "sa_queue_submit": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint64)]),
    "sa_queue_fetch": (c_int, [c_void_p, c_uint64, POINTER(c_float), POINTER(c_uint64)]),
    "sa_queue_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "sa_queue_destroy": (c_int, [c_void_p]),
    "sa_batch_debug_rank_table": (c_int, [c_void_p, ctypes.c_uint32, POINTER(ctypes.c_float), POINTER(ctypes.c_float)]),
    "sa_batch_seeds": (c_int, [c_void_p, POINTER(c_float)]),
    "sa_batch_stats": (c_int, [c_void_p, c_int, u64p, u64p]),
    "sa_batch_destroy": (c_int, [c_void_p]),
    "sa_sharded_create": (c_int, [POINTER(c_int), c_int, c_uint64, c_uint32, u64p, u64p, f32p, c_float, c_uint32, POINTER(c_void_p)]),
    "sa_sharded_destroy": (c_int, [c_void_p]),
    "sa_sharded_info": (c_int, [c_void_p, POINTER(c_int), u64p]),
    "sa_sharded_shard": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "sa_sharded_docfreqs": (c_int, [c_void_p, u64p]),
    "sa_sharded_batch_create": (c_int, [c_void_p, u32p, f32p, c_int, c_int, c_int, c_float, c_float, POINTER(c_void_p)]),
    "sa_sharded_phrase_batch_create": (c_int, [c_void_p, u32p, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), f32p, c_int, c_int,
                                               c_int, c_float, c_float, POINTER(c_void_p)]),
    "sa_sharded_batch_reset": (c_int, [c_void_p, u32p, f32p]),
    "sa_sharded_batch_run": (c_int, [c_void_p, c_int]),
    "sa_sharded_batch_set_options": (c_int, [c_void_p, c_void_p]),
    "sa_sharded_batch_fetch": (c_int, [c_void_p, f32p, u64p]),
    "sa_sharded_batch_destroy": (c_int, [c_void_p]),
    "sa_index_select_rows": (c_int, [c_void_p, u64p, c_uint64]),
    "sa_host_alloc": (c_int, [c_uint64, POINTER(c_void_p)]),
    "sa_host_free": (c_int, [c_void_p]),
    # Part 4
    "sa_vec_create": (c_int, [c_int, c_uint64, c_int, POINTER(c_void_p)]),
    "sa_vec_destroy": (c_int, [c_void_p]),
    "sa_vec_zero": (c_int, [c_void_p]),
    "sa_vec_copy": (c_int, [c_void_p, c_void_p]),
    "sa_vec_fetch": (c_int, [c_void_p, c_void_p]),
    "sa_index_select_vec": (c_int, [c_void_p, c_void_p, c_float, c_int]),
    "sa_vec_dismax_acc": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sa_vec_clause": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "sa_vec_mask_min_count": (c_int, [c_void_p, c_void_p, c_uint32]),
    "sa_vec_sum_count32": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sa_vec_field_row": (c_int, [c_void_p, c_void_p, c_uint32, c_float, c_int, c_int, c_void_p, c_void_p]),
    "sa_vec_field_finish": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "sa_vec_add32": (c_int, [c_void_p, c_void_p, c_int]),
    "sa_vec_add_where": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sa_vec_count_where": (c_int, [c_void_p, c_void_p, u64p]),
    # Part 3
    "sa_comm_unique_id": (c_int, [ctypes.c_char_p, c_int]),
    "sa_index_comm_init": (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p, c_int]),
    "sa_index_comm_destroy": (c_int, [c_void_p]),
    "sa_index_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "sa_index_comm_allreduce": (c_int, [c_void_p, c_void_p, c_uint64, c_int, c_int]),
    "sa_index_comm_barrier": (c_int, [c_void_p]),
}

# the dense calls with an explicit destination: `name`_to(ix, <args of `name` without out>, const sa_dense_dest_t*, out)
for _n in ("sa_index_termfreqs_dense", "sa_index_termfreqs_dense_posn", "sa_index_bm25_dense", "sa_index_phrase_freqs_dense",
           "sa_index_phrase_freqs_dense_posn", "sa_index_bm25_phrase_dense", "sa_index_bm25_phrase_dense_posn",
           "sa_index_similarity_dense"):
    _r, _a = PROTOTYPES[_n]
    PROTOTYPES[_n + "_to"] = (_r, _a[:-1] + [POINTER(DenseDest), _a[-1]])


class HipApi:
    """Bound C ABI.  ``api.sa_xxx(...)`` returns the raw status; ``api.call('sa_xxx', ...)``
    raises SearchArrayHipError on a non-zero status."""

    def __init__(self, cdll: ctypes.CDLL, path: str, allow_missing: bool = False):
        self._cdll = cdll
        self.path = path
        missing = []
        for name, (restype, argtypes) in PROTOTYPES.items():
            try:
                fn = getattr(cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        if missing and not allow_missing:       # (allow_missing: scripts/ab.py binding an OLDER build of the library for a same-box A/B)
            raise SearchArrayHipError(f"{path} does not export: {', '.join(missing)}")

    def last_error(self) -> str:
        msg = self.sa_last_error()
        return msg.decode("utf-8", "replace") if msg else ""

    def call(self, name: str, *args) -> None:
        rc = getattr(self, name)(*args)
        if rc != 0:
            raise SearchArrayHipError(f"{name} failed ({rc}): {self.last_error()}")


def bind(cdll: ctypes.CDLL, path: str = "<cdll>", allow_missing: bool = False) -> HipApi:
    return HipApi(cdll, path, allow_missing)


_api = None


def api() -> HipApi:
    """The gfx950 library, loaded on first use.  Fails loudly -- there is no other backend."""
    global _api
    if _api is None:
        if not os.path.exists(LIB_PATH):
            raise SearchArrayHipError(
                f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C searcharray_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
                f"searcharray_amd has no CPU fallback.")
        try:
            cdll = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise SearchArrayHipError(f"cannot load {LIB_PATH}: {e}") from e
        _api = bind(cdll, LIB_PATH)
    return _api


def use_api(binding) -> None:
    """Install an explicit binding as the process default (``None`` restores lazy loading of the
    gfx950 library).  Exists for the test-suite, which binds the host-emulated kernel build to
    check kernel logic on machines without a GPU; nothing in the package calls it."""
    global _api
    _api = binding


# ---- numpy <-> ctypes helpers --------------------------------------------------------------
def as_u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def as_u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def p_u64(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def p_u32(a: np.ndarray):
    return a.ctypes.data_as(u32p)


def p_f32(a: np.ndarray):
    return a.ctypes.data_as(f32p)

"""The C ABI: every function declared in include/searcharray_hip.h is exported by the gfx950 library
(loaded without touching a GPU), bound in searcharray_amd/_lib.py, and present in the host stand-in
build the CPU tests run against.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from searcharray_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "searcharray_hip.h")
LIB = os.path.join(ROOT, "searcharray_amd", "libsearcharray_hip.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for must in ("sa_index_create", "sa_index_create_from_tokens", "sa_index_bm25_dense", "sa_index_phrase_freqs_dense",
                 "sa_batch_create", "sa_phrase_batch_create", "sa_batch_run", "sa_batch_fetch", "sa_bm25_score",
                 "sa_intersect_with_adjacents", "sa_index_comm_init", "sa_last_error"):
        assert must in names
    assert len(names) >= 45


def test_every_declared_function_is_bound_in_python():
    missing = [n for n in declared_functions() if n not in _lib.PROTOTYPES]
    assert not missing, missing
    extra = [n for n in _lib.PROTOTYPES if n not in declared_functions()]
    assert not extra, extra


def test_gfx950_library_loads_and_exports_every_symbol():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "searcharray_amd", "csrc")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)                       # needs libamdhip64 / librccl only, no device
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.sa_abi_version() >= 1


def test_host_stand_in_exports_every_symbol(request):
    from tests.emu import emu_api
    api = emu_api()
    missing = [n for n in declared_functions() if not hasattr(api, n)]
    assert not missing, missing

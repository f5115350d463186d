"""Shared test helpers: golden fixture loading and oracle index construction."""
import os

import numpy as np

from oracle import refimpl as O
from searcharray_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def golden_corpus(name):
    """(golden npz, sorted triples (t, d, p), doc_lens f32, num_docs, vocab)."""
    g = load_golden(name)
    num_docs, vocab = int(g["meta"][0]), int(g["meta"][1])
    t, d, p = synth.tokens_to_triples(g["lens"], g["terms"])
    return g, (t, d, p), g["lens"].astype(np.float32), num_docs, vocab


def oracle_index(name):
    g, (t, d, p), lens, num_docs, vocab = golden_corpus(name)
    return g, O.OracleIndex.from_triples(t, d, p, num_docs, doc_lens=lens)


def dense_from_sparse(idx, val, n):
    out = np.zeros(n, dtype=np.float32)
    out[idx.astype(np.int64)] = val
    return out


# ---- library options in tests: named values, not environment variables (include/searcharray_hip.h, Part 0) -------------
# tests/conftest.py opens a searcharray_amd.options.Scope around every test (autouse); set_opt / unset_opt change it, and
# every handle the test's thread creates or uses from then on follows (the switches' former environment names --
# "SA_SPARSE" -- are accepted as names, strings as values).
_scope = None


def set_opt(name=None, value=None, **kw):
    _scope.set(name, value, **kw) if name is not None else _scope.set(**kw)


def unset_opt(*names):
    _scope.unset(*names)

"""TEST INFRASTRUCTURE -- the REFERENCE itself (softwaredoug/searcharray), built by oracle/build_ref.sh
into oracle/_ref/, imported under its own package name and fed the same synthetic corpora as the HIP path.

Used by bench.py's cpu_baseline leg (``"kind": "reference"``) and by tests that re-check the oracle /
the device against the real thing when the built tree is present.  Never imported by the product.

The corpus is injected below the Python tokenizer: the reference's own encoder
produces exactly the words searcharray_amd.synth produces (tests/golden pins that), so the reference's
``PosnBitArray`` is constructed over those words directly -- `ArrayDict.from_array_with_boundaries`
(reference phrase/memmap_arrays.py:40-54) -- and everything above it (`SearchArray.score`,
`PosnBitArray.termfreqs / docfreq / phrase_freqs`, the Cython kernels, `bm25_similarity`) is the
reference's code running unmodified.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_DIR, "searcharray", "bm25")) and any(
        f.endswith(".so") for f in os.listdir(os.path.join(REF_DIR, "searcharray", "bm25")))


_mod = None


def reference():
    """The reference package (module ``searcharray`` from oracle/_ref)."""
    global _mod
    if _mod is None:
        if not available():
            raise ImportError("oracle/_ref is not built: run `bash oracle/build_ref.sh` where /root/reference exists")
        if REF_DIR not in sys.path:
            sys.path.insert(0, REF_DIR)
        _mod = importlib.import_module("searcharray")
        got = os.path.dirname(os.path.abspath(_mod.__file__))
        if not got.startswith(REF_DIR):
            raise ImportError(f"`searcharray` resolved to {got}, not to oracle/_ref")
    return _mod


def reference_array(words: np.ndarray, term_off: np.ndarray, doc_lens: np.ndarray, avg_doc_length=None,
                    warm: bool = False):
    """A reference ``SearchArray`` over an already-encoded corpus: term t is named ``t{t}`` and owns
    ``words[term_off[t]:term_off[t+1]]`` (reference postings.py:293-300 sets the same attributes after
    indexing)."""
    reference()
    from searcharray.postings import SearchArray
    from searcharray.phrase.middle_out import PosnBitArray
    from searcharray.phrase.memmap_arrays import ArrayDict
    from searcharray.term_dict import TermDict
    from searcharray.utils.row_viewable_matrix import RowViewableMatrix
    from searcharray.utils.mat_set import SparseMatSet
    n_docs, vocab = len(doc_lens), len(term_off) - 1
    posns = PosnBitArray(ArrayDict.from_array_with_boundaries(np.ascontiguousarray(words, dtype=np.uint64),
                                                              np.arange(vocab), np.asarray(term_off, dtype=np.int64)),
                         max_doc_id=n_docs - 1)
    td = TermDict()
    for t in range(vocab):
        td.add_term(f"t{t}")
    sa = SearchArray([])
    sa.posns = posns
    sa.term_dict = td
    sa.doc_lens = np.ascontiguousarray(doc_lens, dtype=np.float32)
    sa.avg_doc_length = np.mean(sa.doc_lens) if avg_doc_length is None else avg_doc_length
    sa.corpus_size = n_docs
    # score() only asks the doc -> term matrix for len() / rows / subset
    sa.term_mat = RowViewableMatrix(SparseMatSet(cols=np.empty(0, dtype=np.uint32), rows=np.zeros(n_docs + 1, dtype=np.uint32)))
    if warm:
        sa.warm()
    return sa

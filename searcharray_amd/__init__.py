"""searcharray_amd -- MI355X-native scoring hot path behind the searcharray API.

Host code is Python; every query-time operation (term-at-a-time BM25 over TF postings,
roaringish positional phrase intersection, top-k) runs in hand-written HIP kernels for
gfx950, reached through the C ABI declared in include/searcharray_hip.h.
"""
__version__ = "0.1.0"

from .postings import SearchArray, Terms, TermsDtype, ws_tokenizer      # noqa: E402,F401
from .similarity import (bm25_similarity, bm25_impact, bm25_legacy_similarity, classic_similarity,  # noqa: E402,F401
                         default_bm25)
from .results import SetOfResults                                         # noqa: E402,F401

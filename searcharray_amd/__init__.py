"""searcharray_amd -- MI355X-native scoring hot path behind the searcharray API.

Host code is Python; every query-time operation (term-at-a-time BM25 over TF postings,
roaringish positional phrase intersection, top-k) runs in hand-written HIP kernels for
gfx950, reached through the C ABI declared in include/searcharray_hip.h.
"""
__version__ = "0.1.0"

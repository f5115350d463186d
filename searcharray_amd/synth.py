"""Seeded synthetic corpora and query sets.

``zipf-N``: V terms named ``t{rank-1}``, token ranks Zipf(s=1) (p_r ~ 1/r), doc length
max(1, Poisson(mean_len)), positions 0..L-1.  Generation is per batch of ``batch_docs``
documents with an independent seed ``(seed, batch_index)``, so any rank can build any doc
range of the same corpus without generating the rest (doc-range sharding), and batches can
be built in parallel.

Host-side only: produces roaringish words + per-term CSR offsets + doc lengths, the upload
format of ``DeviceIndex``.  The same triples feed the CPU oracle in tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import roaringish as rz


@dataclass
class EncodedCorpus:
    """A term-major roaringish index of docs [doc_base, doc_base + num_docs) (doc ids in the
    words are LOCAL, i.e. minus doc_base)."""
    words: np.ndarray          # uint64[W]   all terms back to back
    term_off: np.ndarray       # uint64[V+1] words of term t = words[term_off[t]:term_off[t+1]]
    doc_lens: np.ndarray       # float32[num_docs]
    num_docs: int
    num_terms: int
    doc_base: int = 0

    def term_words(self, t: int) -> np.ndarray:
        return self.words[int(self.term_off[t]):int(self.term_off[t + 1])]


def _zipf_cdf(vocab: int, s: float = 1.0) -> np.ndarray:
    w = 1.0 / np.arange(1, vocab + 1, dtype=np.float64) ** s
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return cdf


_QUANTILE_BITS = 24
_quantile_cache = {}


def _zipf_quantile_table(vocab: int, s: float = 1.0) -> np.ndarray:
    """Inverse-CDF lookup table with 2^24 equal-probability cells (term id per cell): sampling is
    one random integer + one gather instead of a binary search per token."""
    key = (vocab, s)
    if key not in _quantile_cache:
        cdf = _zipf_cdf(vocab, s)
        u = (np.arange(1 << _QUANTILE_BITS, dtype=np.float64) + 0.5) / float(1 << _QUANTILE_BITS)
        tab = np.searchsorted(cdf, u, side="right").astype(np.uint32)
        np.minimum(tab, vocab - 1, out=tab)
        _quantile_cache[key] = tab
    return _quantile_cache[key]


def zipf_batch_tokens(batch_index: int, n_docs: int, vocab: int = 100_000, mean_len: int = 32,
                      seed: int = 1234, s: float = 1.0, cdf: Optional[np.ndarray] = None,
                      fast: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Tokens of one batch in document order: (doc_lens int64[n_docs], terms uint32[sum lens]).

    fast=False: exact inverse-CDF sampling (binary search per token; used by the golden
    fixtures).  fast=True: the 2^24-cell quantile table (what the large bench corpora use)."""
    rng = np.random.default_rng([seed, batch_index])
    lens = np.maximum(1, rng.poisson(mean_len, n_docs)).astype(np.int64)
    total = int(lens.sum())
    if fast:
        tab = _zipf_quantile_table(vocab, s)
        terms = tab[rng.integers(0, 1 << _QUANTILE_BITS, total, dtype=np.uint32)]
        return lens, terms
    if cdf is None:
        cdf = _zipf_cdf(vocab, s)
    terms = np.searchsorted(cdf, rng.random(total), side="right").astype(np.uint32)
    np.minimum(terms, vocab - 1, out=terms)
    return lens, terms


def tokens_to_triples(lens: np.ndarray, terms: np.ndarray, doc_base_local: int = 0
                      ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Doc-ordered tokens -> (term, doc, posn) triples sorted by term, doc, posn."""
    n_docs = len(lens)
    total = len(terms)
    starts = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    doc = np.repeat(np.arange(doc_base_local, doc_base_local + n_docs, dtype=np.uint64), lens)
    pos = np.arange(total, dtype=np.uint64) - np.repeat(starts[:-1].astype(np.uint64), lens)
    # one sortable key: term | doc | posn  (18 + 28 + 18 bits)
    key = (terms.astype(np.uint64) << np.uint64(46)) | (doc << np.uint64(18)) | pos
    key.sort()
    t = (key >> np.uint64(46)).astype(np.uint32)
    d = (key >> np.uint64(18)) & np.uint64((1 << 28) - 1)
    p = key & np.uint64((1 << 18) - 1)
    return t, d, p


def encode_batch(lens: np.ndarray, terms: np.ndarray, vocab: int, doc_base_local: int = 0
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Tokens of one batch -> (words term-major, per-term word counts int64[vocab])."""
    t, d, p = tokens_to_triples(lens, terms, doc_base_local)
    words, word_terms = rz.encode_sorted(t, d, p)
    counts = np.bincount(word_terms, minlength=vocab).astype(np.int64)
    return words, counts


def concat_term_major(batches: Sequence[Tuple[np.ndarray, np.ndarray]], vocab: int
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """Regroup per-batch term-major word arrays into one term-major array (batches hold
    increasing doc ranges, so per-term concatenation in batch order stays sorted) -- the
    analogue of the reference's ArrayDict.concat (phrase/memmap_arrays.py:56-87)."""
    total_counts = np.zeros(vocab, dtype=np.int64)
    for _, c in batches:
        total_counts += c
    term_off = np.zeros(vocab + 1, dtype=np.uint64)
    np.cumsum(total_counts, out=term_off[1:])
    out = np.empty(int(term_off[-1]), dtype=np.uint64)
    cursor = term_off[:-1].astype(np.int64).copy()
    for words, c in batches:
        if len(words) == 0:
            continue
        src_start = np.zeros(vocab, dtype=np.int64)
        np.cumsum(c[:-1], out=src_start[1:])
        word_term = np.repeat(np.arange(vocab, dtype=np.int64), c)
        dest = cursor[word_term] + (np.arange(len(words), dtype=np.int64) - src_start[word_term])
        out[dest] = words
        cursor += c
    return out, term_off


def zipf_corpus(num_docs: int, vocab: int = 100_000, mean_len: int = 32, seed: int = 1234,
                doc_base: int = 0, batch_docs: int = 1_000_000, total_docs: Optional[int] = None,
                workers: int = 1, fast: bool = True) -> EncodedCorpus:
    """Encode docs [doc_base, doc_base + num_docs) of the ``zipf-N`` corpus (N = total_docs).

    doc_base must be a multiple of batch_docs-aligned shard boundaries only in the sense that
    generation always walks whole seeded batches of the global corpus and keeps the docs that
    fall in the requested range, so every shard layout sees identical documents.
    """
    if total_docs is None:
        total_docs = doc_base + num_docs
    cdf = None if fast else _zipf_cdf(vocab)
    if fast:
        _zipf_quantile_table(vocab)          # build once before the worker threads start
    first_b = doc_base // batch_docs
    last_b = (doc_base + num_docs - 1) // batch_docs if num_docs > 0 else first_b - 1
    jobs = []
    for b in range(first_b, last_b + 1):
        b_lo = b * batch_docs
        b_n = min(batch_docs, total_docs - b_lo)
        jobs.append((b, b_lo, b_n))

    def run(job):
        b, b_lo, b_n = job
        lens, terms = zipf_batch_tokens(b, b_n, vocab, mean_len, seed, cdf=cdf, fast=fast)
        lo = max(doc_base, b_lo) - b_lo
        hi = min(doc_base + num_docs, b_lo + b_n) - b_lo
        if lo > 0 or hi < b_n:
            starts = np.zeros(b_n + 1, dtype=np.int64)
            np.cumsum(lens, out=starts[1:])
            terms = terms[starts[lo]:starts[hi]]
            lens = lens[lo:hi]
        words, counts = encode_batch(lens, terms, vocab, doc_base_local=(b_lo + lo) - doc_base)
        return words, counts, lens.astype(np.float32)

    if workers > 1 and len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            results = list(ex.map(run, jobs))
    else:
        results = [run(j) for j in jobs]
    words, term_off = concat_term_major([(w, c) for w, c, _ in results], vocab)
    doc_lens = (np.concatenate([l for _, _, l in results]) if results
                else np.empty(0, np.float32))
    return EncodedCorpus(words, term_off, doc_lens, num_docs, vocab, doc_base)


def zipf_doc_lens(total_docs: int, mean_len: int = 32, seed: int = 1234, batch_docs: int = 1_000_000) -> np.ndarray:
    """float32 doc lengths of the WHOLE ``zipf-N`` corpus without generating a single token (the lengths
    are the first draw of every seeded batch): what a rank of a sharded run needs to form the reference's
    ``avg_doc_length = np.mean(doc_lens)`` (reference indexing.py:282-284) bit for bit."""
    out = []
    for b in range((total_docs + batch_docs - 1) // batch_docs):
        n = min(batch_docs, total_docs - b * batch_docs)
        rng = np.random.default_rng([seed, b])
        out.append(np.maximum(1, rng.poisson(mean_len, n)).astype(np.float32))
    return np.concatenate(out) if out else np.empty(0, np.float32)


def corpus_triples(num_docs: int, vocab: int, mean_len: int, seed: int = 1234,
                   batch_docs: int = 1_000_000):
    """Small-corpus helper for tests: the raw sorted (term, doc, posn) triples + doc lens."""
    cdf = _zipf_cdf(vocab)
    ts, ds, ps, ls = [], [], [], []
    for b in range((num_docs + batch_docs - 1) // batch_docs):
        n = min(batch_docs, num_docs - b * batch_docs)
        lens, terms = zipf_batch_tokens(b, n, vocab, mean_len, seed, cdf=cdf)
        t, d, p = tokens_to_triples(lens, terms, b * batch_docs)
        ts.append(t); ds.append(d); ps.append(p); ls.append(lens)
    t = np.concatenate(ts); d = np.concatenate(ds); p = np.concatenate(ps)
    order = np.argsort(t, kind="stable")
    return t[order], d[order], p[order], np.concatenate(ls).astype(np.float32)


# ---------------------------------------------------------------------------
# Query sets
# ---------------------------------------------------------------------------
PROBE_QUERY = (0, 9, 99, 999)          # t0 t9 t99 t999


def bm25_queries(n_queries: int = 256, vocab: int = 100_000, seed: int = 42) -> np.ndarray:
    """n_queries x 4 distinct term ids: one rank from each of 1-10, 11-100, 101-1000,
    1001-10000 (0-based ids = rank - 1); row 0 is the fixed probe query."""
    rng = np.random.default_rng(seed)
    bands = [(1, 10), (11, 100), (101, 1000), (1001, 10000)]
    cols = [rng.integers(lo, min(hi, vocab) + 1, n_queries) - 1 for lo, hi in bands]
    q = np.stack(cols, axis=1).astype(np.uint32)
    q[0] = [min(t, vocab - 1) for t in PROBE_QUERY]
    return q


def bm25_queries_distinct(n_queries: int = 256, n_terms: int = 4, vocab: int = 100_000, seed: int = 43) -> np.ndarray:
    """n_queries x n_terms term ids that are pairwise DISTINCT over the whole batch: slot j draws, without
    replacement, from ranks j*n_queries+1 .. (j+1)*n_queries (ids = rank - 1), so no posting list is
    shared between two queries and cache reuse across queries cannot flatter a bandwidth figure."""
    if n_queries * n_terms > vocab:
        raise ValueError("not enough terms for a pairwise-distinct batch")
    rng = np.random.default_rng(seed)
    cols = [rng.permutation(n_queries) + j * n_queries for j in range(n_terms)]
    return np.stack(cols, axis=1).astype(np.uint32)


def phrase_queries_from_tokens(lens: np.ndarray, terms: np.ndarray, n_queries: int = 64,
                               length: int = 3, seed: int = 7) -> np.ndarray:
    """Phrases sampled as actual consecutive n-grams of random docs (>= 1 match guaranteed),
    terms distinct within a phrase."""
    rng = np.random.default_rng(seed)
    starts = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    out: List[np.ndarray] = []
    tries = 0
    while len(out) < n_queries and tries < 100 * n_queries:
        tries += 1
        d = int(rng.integers(0, len(lens)))
        if lens[d] < length:
            continue
        o = int(rng.integers(0, lens[d] - length + 1))
        gram = terms[starts[d] + o: starts[d] + o + length]
        if len(set(gram.tolist())) == length:
            out.append(gram.astype(np.uint32))
    return np.stack(out) if out else np.empty((0, length), np.uint32)

"""Generate tests/golden/*.npz by running the REFERENCE itself (searcharray v0.0.73).

Run in the build container only (the GPU box has no /root/reference):

    cp -r /root/reference /tmp/ref && chmod -R u+w /tmp/ref
    (cd /tmp/ref && python setup.py build_ext --inplace)
    REF_BUILD=/tmp/ref python tests/golden/make_golden.py

Nothing from the reference is copied into the repo: the fixtures hold seeded synthetic
inputs (produced by searcharray_amd.synth) and the reference's OUTPUTS on them, plus the
reference's captured lhs/rhs/mask triples (data files, fixtures/*.npy) with
the reference's outputs on them.
"""
import os
import sys

import numpy as np

REF = os.environ.get("REF_BUILD", "/tmp/ref")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from searcharray.postings import SearchArray                      # noqa: E402
from searcharray.similarity import bm25_similarity               # noqa: E402
from searcharray.roaringish.intersect import intersect, adjacent, intersect_with_adjacents  # noqa: E402
from searcharray.roaringish.merge import merge                   # noqa: E402
from searcharray.roaringish.unique import unique                 # noqa: E402
from searcharray.roaringish.popcount import popcount64_reduce    # noqa: E402
from searcharray.phrase.bigram_freqs import bigram_freqs, Continuation  # noqa: E402
from searcharray_amd import synth                                 # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
HEADER_MASK = np.uint64(0xFFFFFFFFFFFC0000)
LSB_MASK = np.uint64(0x3FFFF)


SNP_TAGS = ("128", "24179", "27685", "44358")        # every COMPLETE captured triple under fixtures/ (lhs + rhs + mask)
SNP_LHS_ONLY = ("185", "45907", "90596")              # captured without an rhs: the one-array primitives only


def snp_fixture_goldens():
    """Reference set primitives on its own captured arrays (test/test_snp_ops.py:323-349)."""
    out = {}
    for tag in SNP_TAGS:
        lhs = np.load(f"/root/reference/fixtures/lhs_{tag}.npy")
        rhs = np.load(f"/root/reference/fixtures/rhs_{tag}.npy")
        mask = np.load(f"/root/reference/fixtures/mask_{tag}.npy")
        mask = np.uint64(mask) if mask.shape == () else np.uint64(mask.flatten()[0])
        out[f"{tag}_lhs"], out[f"{tag}_rhs"], out[f"{tag}_mask"] = lhs, rhs, np.asarray(mask)
        li, ri = intersect(lhs, rhs, mask=mask)
        out[f"{tag}_int_drop_l"], out[f"{tag}_int_drop_r"] = li, ri
        lk, rk = intersect(lhs, rhs, mask=mask, drop_duplicates=False)
        out[f"{tag}_int_keep_l"], out[f"{tag}_int_keep_r"] = lk, rk
        a, b, c, d = intersect_with_adjacents(lhs, rhs, mask=mask)
        out[f"{tag}_iwa_l"], out[f"{tag}_iwa_r"], out[f"{tag}_iwa_al"], out[f"{tag}_iwa_ar"] = a, b, c, d
        al, ar = adjacent(lhs, rhs, mask=mask)
        out[f"{tag}_adj_l"], out[f"{tag}_adj_r"] = al, ar
        out[f"{tag}_merge"] = merge(lhs, rhs)
        out[f"{tag}_merge_drop"] = merge(lhs, rhs, drop_duplicates=True)
        out[f"{tag}_unique36"] = unique(lhs, 36)
        k, c = popcount64_reduce(lhs, np.uint64(36), LSB_MASK)
        out[f"{tag}_pcr_keys"], out[f"{tag}_pcr_counts"] = k, c
        (ids, cnt), (_, rn) = bigram_freqs(lhs, rhs, Continuation.RHS)
        out[f"{tag}_bg_rhs_ids"], out[f"{tag}_bg_rhs_counts"], out[f"{tag}_bg_rhs_next"] = ids, cnt, rn
        (ids, cnt), (ln, _) = bigram_freqs(lhs, rhs, Continuation.LHS)
        out[f"{tag}_bg_lhs_ids"], out[f"{tag}_bg_lhs_counts"], out[f"{tag}_bg_lhs_next"] = ids, cnt, ln
    np.savez_compressed(os.path.join(OUT, "snp_fixtures.npz"), **out)
    print("snp_fixtures.npz", len(out), "arrays")
    out = {}
    for tag in SNP_LHS_ONLY:
        lhs = np.load(f"/root/reference/fixtures/lhs_{tag}.npy")
        out[f"{tag}_lhs"] = lhs
        out[f"{tag}_unique36"] = unique(lhs, 36)
        out[f"{tag}_unique18"] = unique(lhs, 18)
        k, c = popcount64_reduce(lhs, np.uint64(36), LSB_MASK)
        out[f"{tag}_pcr_keys"], out[f"{tag}_pcr_counts"] = k, c
    np.savez_compressed(os.path.join(OUT, "snp_fixtures_lhs.npz"), **out)
    print("snp_fixtures_lhs.npz", len(out), "arrays")


def sparse(a):
    idx = np.flatnonzero(a).astype(np.uint32)
    return idx, a[idx]


def corpus_goldens(name, num_docs, vocab, mean_len, seed, n_phr, slop_queries):
    lens, terms = synth.zipf_batch_tokens(0, num_docs, vocab, mean_len, seed)
    starts = np.zeros(num_docs + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    docs = [" ".join(f"t{t}" for t in terms[starts[i]:starts[i + 1]]) for i in range(num_docs)]
    sa = SearchArray.index(docs, autowarm=False)
    out = {"lens": lens, "terms": terms,
           "meta": np.asarray([num_docs, vocab, mean_len, seed], dtype=np.int64),
           "avg_doc_length": np.asarray(sa.avg_doc_length, dtype=np.float32),
           "doc_lens": sa.doc_lens.astype(np.float32)}
    # df for every term, tf (sparse) for a spread of terms
    dfs = np.asarray([sa.docfreq(f"t{t}") for t in range(vocab)], dtype=np.uint64)
    out["df"] = dfs
    tf_terms = np.unique(np.concatenate([np.arange(0, 12), np.geomspace(12, vocab - 1, 24).astype(int)]))
    out["tf_terms"] = tf_terms.astype(np.uint32)
    for t in tf_terms:
        i, v = sparse(sa.termfreqs(f"t{t}"))
        out[f"tf_{t}_idx"], out[f"tf_{t}_val"] = i, v
    # single-term BM25 (default + custom k1/b)
    sc_terms = tf_terms[::3]
    out["score_terms"] = sc_terms.astype(np.uint32)
    custom = bm25_similarity(k1=1.7, b=0.3)
    for t in sc_terms:
        out[f"score_{t}"] = sa.score(f"t{t}")
        out[f"score_custom_{t}"] = sa.score(f"t{t}", similarity=custom)
    # 4-term disjunctions (caller idiom test/test_msmarco.py:353-354)
    rng = np.random.default_rng(99)
    queries = np.stack([rng.integers(0, min(10, vocab), 16), rng.integers(0, min(60, vocab), 16),
                        rng.integers(0, vocab, 16), rng.integers(0, vocab, 16)], axis=1).astype(np.uint32)
    queries[0] = [0, 1, 2, 3]
    out["or_queries"] = queries
    out["or_scores"] = np.stack([np.sum([sa.score(f"t{t}") for t in q], axis=0) for q in queries])
    # phrases: real n-grams of length 2..7, random (mostly non-matching) ones, same-term ones
    phrases = []
    for length in (2, 3, 4, 5, 6, 7):
        rngp = np.random.default_rng(1000 + length)
        got = 0
        while got < n_phr:
            d = int(rngp.integers(0, num_docs))
            if lens[d] < length:
                continue
            o = int(rngp.integers(0, lens[d] - length + 1))
            phrases.append(terms[starts[d] + o: starts[d] + o + length].astype(np.int64))
            got += 1
        for _ in range(max(2, n_phr // 4)):
            phrases.append(rngp.integers(0, min(vocab, 12), length))
    phrases += [np.asarray(p) for p in ([0, 0], [0, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [1, 1], [0, 0, 0, 0],
                                        [0, 1, 0, 1], [2, 0, 0, 3], [0, 1, 2, 0, 1], [3, 3, 3], [0, 0, 1, 1],
                                        [5, 4, 0, 2, 1], [0, 1, 9, 2, 3], [1, 2, 0, 9, 3, 4, 0])]
    out["n_phrases"] = np.asarray(len(phrases))
    for i, p in enumerate(phrases):
        toks = [f"t{t}" for t in p]
        out[f"phr_{i}_terms"] = np.asarray(p, dtype=np.uint32)
        idx, val = sparse(sa.termfreqs(toks))
        out[f"phr_{i}_idx"], out[f"phr_{i}_val"] = idx, val
        sidx, sval = sparse(sa.score(toks))
        out[f"phr_{i}_sidx"], out[f"phr_{i}_sval"] = sidx, sval
    # slop
    out["n_slop"] = np.asarray(len(slop_queries))
    for i, (p, slop) in enumerate(slop_queries):
        toks = [f"t{t}" for t in p]
        out[f"slop_{i}_terms"] = np.asarray(p, dtype=np.uint32)
        out[f"slop_{i}_slop"] = np.asarray(slop)
        idx, val = sparse(sa.termfreqs(toks, slop=slop))
        out[f"slop_{i}_idx"], out[f"slop_{i}_val"] = idx, val
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz", len(out), "arrays,", num_docs, "docs")


def edismax_goldens():
    """Reference solr.edismax outputs on seeded multi-field frames (scores and explain strings)."""
    import json
    import pandas as pd
    from searcharray.solr import edismax
    rng = np.random.default_rng(99)
    vocab = [f"w{i}" for i in range(40)]
    probs = 1.0 / np.arange(1, 41)
    probs /= probs.sum()

    def docs(n, mean_len):
        return [" ".join(rng.choice(vocab, size=max(1, rng.poisson(mean_len)), p=probs)) for _ in range(n)]

    n = 300
    fields = {"title": docs(n, 6), "body": docs(n, 30), "tags": docs(n, 3)}

    def lower_whole(text):              # a tokenizer that yields ONE token: forces the field-centric path
        return [text.lower()]

    frame = pd.DataFrame({"title": SearchArray.index(fields["title"]), "body": SearchArray.index(fields["body"]),
                          "tags": SearchArray.index(fields["tags"], tokenizer=lower_whole)})
    cases = [
        {"q": "w0 w3", "qf": ["title", "body"]},
        {"q": "w1 w2 w5", "qf": ["title^3", "body"], "tie": 0.2},
        {"q": "w0 w1 w2 w7", "qf": ["title", "body^0.5"], "mm": "3"},
        {"q": "w0 w1 w2 w7", "qf": ["title", "body"], "mm": "2<75%", "tie": 0.1},
        {"q": "w2 w1", "qf": ["title", "body"], "pf": ["body^2"]},
        {"q": "w0 w1 w3", "qf": ["title", "body"], "pf": ["title", "body"], "pf2": ["body"], "pf3": ["body^1.5"]},
        {"q": "w0 w1 w3", "qf": ["body", "title"], "pf2": ["body", "title^4"], "q_op": "AND"},
        {"q": "w4 w0", "qf": ["title", "tags"]},                              # field-centric
        {"q": "w4 w0", "qf": ["tags^2", "title", "body"], "mm": "2", "tie": 0.3},
        {"q": "w39 w38", "qf": ["title", "body"], "mm": -1},
        {"q": "w1", "qf": ["title", "body"], "pf": ["title"], "pf2": ["title"], "pf3": ["title"]},
        {"q": "zzz w1", "qf": ["title", "body"]},                              # unknown term
    ]
    out = {"n_cases": np.asarray(len(cases)), "cases": np.asarray(json.dumps(cases))}
    for k, v in fields.items():
        out[f"field_{k}"] = np.asarray(v)
    for i, c in enumerate(cases):
        scores, explain = edismax(frame, **c)
        out[f"scores_{i}"] = np.asarray(scores)              # native dtype: float64 term-centric, float32 field-centric
        out[f"explain_{i}"] = np.asarray(explain)
    np.savez_compressed(os.path.join(OUT, "edismax.npz"), **out)


def similarity_goldens():
    """Reference scores under its other stock similarities (similarity.py:41-89): single terms, phrases,
    slop, position ranges and a slice, native dtypes (float32 for bm25_impact, float64 for the others)."""
    from searcharray.similarity import bm25_impact, bm25_legacy_similarity, classic_similarity
    rng = np.random.default_rng(31337)
    vocab = [f"w{i}" for i in range(50)]
    probs = 1.0 / np.arange(1, 51)
    probs /= probs.sum()
    docs = [" ".join(rng.choice(vocab, size=max(1, rng.poisson(25)), p=probs)) for _ in range(600)]
    arr = SearchArray.index(docs)
    sims = {"impact": bm25_impact(), "impact_b": bm25_impact(k1=0.9, b=0.3), "legacy": bm25_legacy_similarity(),
            "legacy_b": bm25_legacy_similarity(k1=1.7, b=0.3), "classic": classic_similarity()}
    cases = [("w0", {}), ("w7", {}), ("w49", {}), ("nope", {}), (["w0", "w1"], {}), (["w1", "w0", "w2"], {}),
             (["w0", "w0"], {}), (["w0", "nope"], {}), (["w3", "w1"], {"slop": 2}),
             ("w0", {"min_posn": 0, "max_posn": 17}), (["w0", "w1"], {"min_posn": 0, "max_posn": 35})]
    # (a single term with min_posn > 0 raises inside the reference's as_dense -- float64 counts -- so
    #  only ranges starting at 0 can be pinned for single terms)
    out = {"docs": np.asarray(docs), "n_cases": np.asarray(len(cases)), "sims": np.asarray(list(sims))}
    rows = np.asarray([5, 17, 300, 420, 599])
    for name, sim in sims.items():
        for i, (tok, kw) in enumerate(cases):
            out[f"{name}_{i}"] = arr.score(tok, similarity=sim, **kw)
        out[f"{name}_slice"] = arr[rows].score("w0", similarity=sim)
        out[f"{name}_slice_phrase"] = arr[rows].score(["w0", "w1"], similarity=sim)
    out["rows"] = rows
    np.savez_compressed(os.path.join(OUT, "similarities.npz"), **out)


def encoder_goldens():
    """The reference's RoaringishEncoder (roaringish/roaringish.py:54-282) on seeded inputs, for key widths
    28 (default), 32 and 20: encode with and without boundaries, decode, per-key counts, header / key
    intersections, key partitions and slices."""
    from searcharray.roaringish.roaringish import RoaringishEncoder, convert_keys
    out = {}
    rng = np.random.default_rng(808)
    for kb in (28, 32, 20):
        enc = RoaringishEncoder(np.uint64(kb))
        L = int(enc.payload_lsb_bits)

        def make(n_keys, max_key, mean):
            keys = np.sort(rng.choice(max_key, size=n_keys, replace=False)).astype(np.uint64)
            ks, ps = [], []
            for k in keys:
                m = max(1, rng.poisson(mean))
                p = np.sort(rng.choice(6 * L, size=min(m, 6 * L), replace=False))
                ks.append(np.full(len(p), k, dtype=np.uint64))
                ps.append(p.astype(np.uint64))
            return np.concatenate(ks), np.concatenate(ps)

        tag = f"k{kb}_"
        # three "terms" back to back with boundaries
        parts = [make(40, 200, 5), make(25, 200, 9), make(60, 200, 2)]
        keys = np.concatenate([p[0] for p in parts])
        posns = np.concatenate([p[1] for p in parts])
        bounds = np.cumsum([0] + [len(p[0]) for p in parts])[:-1].astype(np.uint64)
        out[tag + "keys"], out[tag + "posns"], out[tag + "bounds"] = keys, posns, bounds
        words, nb = enc.encode(keys=keys, payload=posns, boundaries=bounds)
        out[tag + "enc_b"], out[tag + "enc_b_bounds"] = words, nb
        a = words[int(nb[0]):int(nb[1])]
        b = words[int(nb[1]):int(nb[2])]
        single, none = enc.encode(keys=parts[0][0], payload=parts[0][1])
        assert none is None and np.array_equal(single, a)
        out[tag + "enc_nokeys"] = enc.encode(payload=parts[1][1][:12])[0]
        dec = enc.decode(a)
        out[tag + "dec_keys"] = np.asarray([k for k, _ in dec], dtype=np.uint64)
        out[tag + "dec_lens"] = np.asarray([len(v) for _, v in dec], dtype=np.int64)
        out[tag + "dec_vals"] = np.concatenate([v for _, v in dec]).astype(np.uint64)
        out[tag + "dec_nokeys_n"] = np.asarray(len(enc.decode(a, get_keys=False)))
        k, c = enc.num_values_per_key(a)
        out[tag + "nvpk_k"], out[tag + "nvpk_c"] = k, c
        out[tag + "keys_of"], out[tag + "keys_unique"] = enc.keys(a), enc.keys_unique(a)
        out[tag + "msb"], out[tag + "lsb"], out[tag + "hdr"] = enc.payload_msb(a), enc.payload_lsb(a), enc.header(a)
        for name, res in (("cand", enc.intersect_candidates(a, b)), ("rshift", enc.intersect_rshift(a, b)),
                          ("isect", enc.intersect(a, b))):
            for j, r in enumerate(res):
                out[f"{tag}{name}_{j}"] = r
        out[tag + "part2"] = enc.key_partition(a, np.uint64(200), 2)
        out[tag + "part8"] = enc.key_partition(a, np.uint64(200), 8)
        some = convert_keys([int(x) for x in np.unique(parts[0][0])[::3]] + [9999])
        out[tag + "slice_keys_in"] = some
        out[tag + "slice_keys"] = enc.slice(a, keys=some)
        out[tag + "slice_hdr"] = enc.slice(a, header=enc.header(b))
        out[tag + "slice_posn"] = enc.slice(a, min_payload=L, max_payload=3 * L - 1)
        out[tag + "slice_keys_posn"] = enc.slice(a, keys=some, max_payload=2 * L - 1)
    out["convert"] = np.concatenate([convert_keys(5), convert_keys([3, 1]), convert_keys(range(2, 6)), convert_keys(range(0))])
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)


def memmap_goldens():
    """The reference's on-disk index: SearchArray.index(..., data_dir=...) writes one raw uint64 .dat
    (phrase/memmap_arrays.py:158-161) and keeps {term_id: {offset, length}} metadata.  The fixture holds
    the file's content, the metadata, the term dictionary, doc lengths and reference scores."""
    import tempfile
    rng = np.random.default_rng(2024)
    vocab = [f"w{i}" for i in range(60)]
    probs = 1.0 / np.arange(1, 61)
    probs /= probs.sum()
    docs = [" ".join(rng.choice(vocab, size=max(1, rng.poisson(12)), p=probs)) for _ in range(400)]
    with tempfile.TemporaryDirectory() as d:
        arr = SearchArray.index(docs, data_dir=d)
        files = sorted(os.listdir(d))
        assert len(files) == 1 and files[0].endswith(".dat"), files
        dat = np.fromfile(os.path.join(d, files[0]), dtype=np.uint64)
        md = arr.posns.encoded_term_posns.arrays.metadata
        ids = np.asarray(sorted(md), dtype=np.int64)
        out = {"dat": dat, "ids": ids,
               "offsets": np.asarray([md[int(i)]["offset"] for i in ids], dtype=np.uint64),
               "lengths": np.asarray([md[int(i)]["length"] for i in ids], dtype=np.uint64),
               "terms": np.asarray([arr.term_dict.get_term(int(i)) for i in range(len(arr.term_dict))]),
               "doc_lens": np.asarray(arr.doc_lens, dtype=np.float32), "docs": np.asarray(docs)}
        queries = ["w0", "w3", "w17", "w59"]
        phrases = [["w0", "w1"], ["w1", "w0", "w2"], ["w2", "w2"], ["w5", "w0"]]
        for i, q in enumerate(queries):
            out[f"score_{i}"] = arr.score(q)
        for i, q in enumerate(phrases):
            out[f"phrase_{i}"] = arr.score(q)
        out["queries"] = np.asarray(queries)
        out["phrases"] = np.asarray(["|".join(p) for p in phrases])
    np.savez_compressed(os.path.join(OUT, "memmap.npz"), **out)


if __name__ == "__main__":
    only = os.environ.get("ONLY", "")          # "" = everything, or one of: snp, core, edismax, memmap, similarity, encoder
    if only in ("", "core", "snp"):
        snp_fixture_goldens()
    if only in ("", "core"):
        slopq = [([3, 7], 1), ([3, 7], 2), ([0, 1], 2), ([5, 2, 9], 2), ([10, 4], 3), ([1, 0], 1),
                 ([20, 30], 5), ([2, 2], 2), ([8, 1, 3], 4), ([40, 6], 2)]
        corpus_goldens("zipf_small", 1500, 200, 40, 4321, 10, slopq)
        corpus_goldens("zipf_sparse", 4000, 5000, 24, 77, 6, slopq[:4])
    if only in ("", "edismax"):
        edismax_goldens()
    if only in ("", "memmap"):
        memmap_goldens()
    if only in ("", "similarity"):
        similarity_goldens()
    if only in ("", "encoder"):
        encoder_goldens()

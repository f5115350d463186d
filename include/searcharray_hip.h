/*
 * searcharray_hip.h -- C ABI of libsearcharray_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for searcharray's scoring hot path.  The reference has no FFI
 * registry: its native boundary is the set of Cython `def` functions that take numpy
 * buffers (searcharray/roaringish/__init__.py:1-7, searcharray/bm25/bm25.pyx:28).  Part 1
 * mirrors those entry points one to one (host pointers in, host pointers out, caller-owned
 * buffers with the reference's worst-case sizes), so a maintainer can rebind them with
 * ctypes/cffi.  Part 2 is the HBM-resident index the Python classes
 * (SearchArray / PosnBitArray) sit on: the index is uploaded once and every query runs on
 * the device (index build from a token stream, dense drop-in results, resident top-k batches
 * of BM25 term disjunctions and of exact phrases).  Part 3 is the doc-range-sharded multi-GPU
 * top-k exchange over RCCL.  Part 4 keeps dense per-doc vectors on the device for the
 * combination step of multi-field (Solr edismax style) queries.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SA_ERR_* code on failure; the message
 *     is available from sa_last_error() (thread-local).  Nothing throws across the boundary.
 *   - plain C types only; pointers are HOST pointers owned by the caller unless a parameter
 *     is documented as a device pointer.
 *   - calls on different index handles may run concurrently from different threads; calls
 *     on one handle are serialised internally (the reference's callers score from thread
 *     pools, test/test_tmdb.py:285-312).  ctypes releases the GIL around every call.
 */
#ifndef SEARCHARRAY_HIP_H
#define SEARCHARRAY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_ABI_VERSION 1

typedef struct sa_index sa_index_t;        /* opaque HBM-resident index (one doc-range shard) */
typedef struct sa_batch sa_batch_t;        /* opaque device-resident query batch              */
typedef struct sa_vec sa_vec_t;            /* opaque dense per-doc vector in HBM (Part 4)     */

const char* sa_last_error(void);
int sa_abi_version(void);
int sa_device_count(int* out_count);
int sa_device_name(int device, char* buf, int buf_len);

/* ------------------------------------------------------------------------------------- */
/* Part 0 -- options                                                                       */
/* ------------------------------------------------------------------------------------- */
/* The library's switches travel as DATA, per index handle and per batch -- the counterpart of the keyword arguments of
 * the reference's objects (searcharray/postings.py:250-258 SearchArray.index(...), :652-656 score(...)); nothing the scoring
 * path does depends on the process environment.  Every field is an int64; SA_OPT_UNSET = "the library decides" (its default
 * or its automatic rule).  A handle starts from the process defaults: all unset, plus the debug override
 * SA_OPTS="name=value,name=value" that is read ONCE when the library first needs it.  Options at CREATION (an index's
 * directory thresholds, a batch's capacities): sa_options_set_thread_defaults -- every index handle the calling thread creates
 * afterwards starts from them, every batch too (without thread defaults a batch starts from its index's options); other threads
 * are not affected.  sa_index_set_options / sa_batch_set_options replace a handle's options later (index: the switches of its
 * dense calls and what batches created afterwards start from; batch: effective from its next reset / run; the switches read at
 * creation keep what they were).  Bindings that do not want to mirror the struct fill it by name:
 * sa_options_init + sa_options_set(&o, "sparse", 0); sa_option_count / sa_option_name enumerate the fields in struct order. */
#define SA_OPT_UNSET INT64_MIN
typedef struct sa_options {
    uint64_t struct_size;           /* sizeof(sa_options_t) of the caller's header */
    int64_t sparse;          /* 1: dynamic pruning (MaxScore), 0: exhaustive scoring; unset: the rule of sa_batch_run_shard */
    int64_t group;           /* 0: no grouped kernel (queries that share their first term are scored one by one) */
    int64_t group_loose;     /* 0: no loose groups */
    int64_t group_side;      /* 0: ungrouped rows on the batch's own stream instead of the side stream */
    int64_t group_dense;     /* 0: the grouped kernel builds its base from postings even where a dense factor row exists */
    int64_t group_one;       /* left-over queries (too dense for a loose group, first term shared with nobody) as groups of one: 2 all (default), 1 only over a dense factor row, 0 none (per-query kernel) */
    int64_t group_min;       /* smallest group (default 2) */
    int64_t group_maxq;      /* queries per table pass of a grouped item (default and at most 16) */
    int64_t group_item;      /* queries per grouped item: passes of group_maxq queries over ONE base (at most 64; default 32 for big launches, else 16) */
    int64_t group_warm;      /* tiles scored by the per-query kernel first to establish bounds (default: none with starting bounds) */
    int64_t loose_postings;  /* loose groups: expected postings of a query per tile at most (default 400) */
    int64_t xcd_range;       /* 0: tiles dealt round-robin to the XCDs instead of ranges */
    int64_t term_seed;       /* 0: no starting bounds from the terms' rank tables */
    int64_t topf_slice;      /* TEST HOOK: postings per workgroup of a long list's rank-table histogram (default 65536; lists of 4 slices and more) */
    int64_t seed_scale_pct;  /* TEST HOOK: starting bounds scaled by this percentage (> 100 makes them too high: the redo path) */
    int64_t merge_small;     /* 0: the 1024-thread merge also for k <= 64 */
    int64_t impact;          /* 0: score the TF postings, no impact stream */
    int64_t pruned_topk;     /* 0: block-level selection (no bounds) */
    int64_t no_topk;         /* timing experiments: skip the per-tile selection */
    int64_t topk_hist;       /* 0: slot bound instead of the histogram bound */
    int64_t topk_hist_mink;  /* smallest k that takes the histogram bound */
    int64_t cand_cap;        /* TEST HOOK: candidate-list capacity per query (forces the overflow handling) */
    int64_t sparse_div;      /* pruning: a lead term has at most n_docs / this postings (default 8) */
    int64_t sparse_lazy;     /* 0: pruning tables derived at every reset, needed or not */
    int64_t bloom_floor;     /* TEST HOOK: smallest Bloom buffer in bytes */
    int64_t sp_chunk1;       /* pruning: postings per lead work item */
    int64_t stage;           /* staged-tile route (sa_stage.hip): 1 force where eligible, 0 off; unset: on where eligible and `sparse` is unset */
    int64_t stage_docs;      /* docs per stage tile (multiple of 64; default: what fits the LDS stage for the query set's terms) */
    int64_t stage_wgs;       /* staged-tile route: resident workgroups per CU (default 2) */
    int64_t stage_cw;        /* staged-tile route: workgroups of an XCD that co-walk a range of tiles (default 32; 1: private ranges) */
    int64_t stage_probe;     /* 0: the staged-tile route streams EVERY term of the batch; default: terms that cannot be essential are probed in dense rows */
    int64_t probe_div;       /* probe rows (dense factor rows the staged-tile route probes) for terms with df >= n_docs / this (default 128; 0: none) */
    int64_t dense_direct;    /* 0: sa_index_bm25_dense scores the TF postings into scratch and copies (rounds 1-5); default: one launch over the impact stream, straight into the destination */
    int64_t batch_stream;    /* 0: batches share the index stream */
    int64_t res_xs;          /* 0: result copies on the batches' own streams */
    int64_t dense_div;       /* dense factor rows for terms with df >= n_docs / this (default 4) */
    int64_t dir_div;         /* tile directory rows for terms with df >= n_tiles / this */
    int64_t docdir_div;      /* doc directory rows for terms with >= n_docs / this words (0: none) */
    int64_t tf8_div;         /* dense tf rows for terms with df >= n_docs / this */
    int64_t tf8_maxrows;     /* ... at most this many */
    int64_t seg_words;       /* TEST HOOK: words per segment of the posting derivation */
    int64_t phrase_mode;     /* 0 auto, 1 general chain, 2 fused kernel */
    int64_t phrase_docdir;   /* 0: no doc directory probes */
    int64_t phrase_docs;     /* 0: no chain per document */
    int64_t phrase_lanes;    /* phrase tiles: lanes per phrase */
    int64_t ptile;           /* docs per phrase tile (2048 / 4096) */
    int64_t span_doc;        /* 0: no doc-parallel route */
    int64_t span_docdir;     /* 0: no doc directory */
    int64_t span_doc_multi;  /* 0: one launch per phrase */
    int64_t span_doc_rank;   /* 0: ranking in its own launch */
    int64_t span_fast;       /* 0: general state machine only */
    int64_t span_multi;      /* 0: no multi-phrase launch of the general route */
    int64_t span_sort;       /* 1 / 0: force / forbid sorting the docs by work */
    int64_t span_lds_pad;    /* MEASUREMENT HOOK: unused dynamic LDS per block of the doc-parallel batch launch (fewer resident blocks per CU) */
    int64_t span_bundle;     /* doc-parallel batch launch: phrases whose blocks take turns in the launch (default 32; 1: a phrase's blocks back to back) */
    int64_t span_tab_waves;  /* doc-parallel batch launch: waves of a block that hold span tables, 2 / 4 (unset: 2 when the launch has more blocks than four per CU, else 4) */
    int64_t span_threads;    /* TEST HOOK: grid cap (forces the stride loop) */
    int64_t io_piece_bytes;  /* bytes per staged piece */
    int64_t io_threads;      /* file threads */
    int64_t trace;           /* 1: route decisions to stderr */
} sa_options_t;
void sa_options_init(sa_options_t* opts);                                  /* all unset */
int sa_options_set(sa_options_t* opts, const char* name, int64_t value);
int sa_options_get(const sa_options_t* opts, const char* name, int64_t* value_out);
int sa_option_count(void);
const char* sa_option_name(int i);                                          /* NULL past the end */
int sa_options_process_defaults_get(sa_options_t* out);                    /* all unset + the SA_OPTS override */
int sa_options_set_thread_defaults(const sa_options_t* opts);              /* NULL: this thread creates from the process defaults again */
int sa_index_set_options(sa_index_t* ix, const sa_options_t* opts);
int sa_index_get_options(sa_index_t* ix, sa_options_t* out);
int sa_batch_set_options(sa_batch_t* batch, const sa_options_t* opts);
int sa_batch_get_options(sa_batch_t* batch, sa_options_t* out);

/* ------------------------------------------------------------------------------------- */
/* Part 1 -- kernel-level mirrors of the reference's native entry points                   */
/* ------------------------------------------------------------------------------------- */

/* bm25_score(term_freqs inout, doc_lens, avg_doc_lens, idf, k1, b)
 * reference searcharray/bm25/bm25.pyx:28-41 (in-place, fp32, no FMA contraction). */
int sa_bm25_score(float* term_freqs, const float* doc_lens, float avg_doc_lens, float idf,
                  float k1, float b, int64_t n);

/* as_dense(indices, values, size) -> float32[size]
 * reference searcharray/roaringish/roaringish_ops.pyx:84-98, scatter_assign.h:6-29. */
int sa_as_dense(const uint64_t* indices, const float* values, int64_t n, float* out, int64_t size);

/* popcount64_reduce(arr, key_shift, value_mask) -> (keys u64[g], counts f32[g])
 * reference searcharray/roaringish/popcount.pyx:212-237,271-278.  Outputs sized n. */
int sa_popcount64_reduce(const uint64_t* arr, int64_t n, uint64_t key_shift, uint64_t value_mask,
                         uint64_t* keys_out, float* counts_out, int64_t* n_out);

/* unique(arr, rshift) -> distinct (arr >> rshift) of a sorted array
 * reference searcharray/roaringish/unique.pyx:87-104,139-145.  Output sized n. */
int sa_unique(const uint64_t* arr, int64_t n, uint64_t rshift, uint64_t* out, int64_t* n_out);

/* popcount64(arr) -> u64[n]      reference searcharray/roaringish/popcount.pyx:71-81,119-121 */
int sa_popcount64(const uint64_t* arr, int64_t n, uint64_t* out);

/* intersect(lhs, rhs, mask, drop_duplicates) -> index pairs into lhs / rhs of equal masked values
 * reference searcharray/roaringish/intersect.pyx:278-320 (drop: first member of each equal run,
 * outputs sized min(nl, nr); keep: every member of both runs, outputs sized max(nl, nr)).
 * mask == 0 is an error ("Mask cannot be zero", intersect.pyx:290-291). */
int sa_intersect(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, uint64_t mask,
                 int drop_duplicates, uint64_t* lhs_idx, uint64_t* rhs_idx,
                 int64_t* n_lhs_out, int64_t* n_rhs_out);

/* adjacent(lhs, rhs, mask): pairs with lhs&mask + delta == rhs&mask, delta = lowest set bit of mask
 * reference intersect.pyx:131-190,323-343.  Outputs sized min(nl, nr). */
int sa_adjacent(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, uint64_t mask,
                uint64_t* lhs_idx, uint64_t* rhs_idx, int64_t* n_out);

/* intersect_with_adjacents(lhs, rhs, mask): equal pairs and adjacent pairs in one call
 * reference intersect.pyx:213-275,346-390.  All four outputs sized min(nl, nr). */
int sa_intersect_with_adjacents(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr,
                                uint64_t mask, uint64_t* lhs_idx, uint64_t* rhs_idx, int64_t* n_out,
                                uint64_t* adj_lhs_idx, uint64_t* adj_rhs_idx, int64_t* n_adj_out);

/* merge(lhs, rhs, drop_duplicates) of two sorted arrays; reference merge.pyx:54-158. Output sized nl+nr. */
int sa_merge(const uint64_t* lhs, int64_t nl, const uint64_t* rhs, int64_t nr, int drop_duplicates,
             uint64_t* out, int64_t* n_out);

/* sort_merge_counts: union of two sorted id lists (ids unique within each list) with float counts
 * added on equal ids; reference merge.pyx:161-232.  Outputs sized nl+nr. */
int sa_sort_merge_counts(const uint64_t* lhs_ids, const float* lhs_counts, int64_t nl,
                         const uint64_t* rhs_ids, const float* rhs_counts, int64_t nr,
                         uint64_t* ids_out, float* counts_out, int64_t* n_out);

/* popcount_reduce_at(ids, payload) / key_sum_over(ids, count): groups of consecutive equal ids,
 * summing popcount(payload) / count; reference popcount.pyx:124-204.  Outputs sized n. */
int sa_popcount_reduce_at(const uint64_t* ids, const uint64_t* payload, int64_t n, uint64_t* ids_out,
                          float* counts_out, int64_t* n_out);
int sa_key_sum_over(const uint64_t* ids, const uint64_t* count, int64_t n, uint64_t* ids_out,
                    float* counts_out, int64_t* n_out);

/* payload_slice(arr, payload_msb_mask, min_payload, max_payload): words whose UNSHIFTED
 * (word & msb_mask) lies in [min, max] -- the reference's comparison, reference
 * roaringish_ops.pyx:46-68.  Output sized n. */
int sa_payload_slice(const uint64_t* arr, int64_t n, uint64_t payload_msb_mask, uint64_t min_payload,
                     uint64_t max_payload, uint64_t* out, int64_t* n_out);

/* span_search(posns, lengths, phrase_freqs, slop, key_mask, header_mask, key_bits, lsb_bits)
 * reference searcharray/roaringish/spans.pyx:189-330: the slop > 0 span state machine over the
 * terms' candidate words, term t = posns[lengths[t] : lengths[t + 1]] (what phrase/spans.py:171-187
 * passes after _intersect_all; default 28 / 18 / 18 layout, at most 32 terms).  Returns the documents
 * whose count the walk raised, ascending, and the increments -- the values the reference adds into its
 * Counter.  Outputs sized by the number of distinct doc ids in the input (<= lengths[n_terms] - lengths[0]). */
int sa_span_search(const uint64_t* posns, const uint64_t* lengths, int n_terms, uint64_t slop,
                   uint64_t* docs_out, uint64_t* counts_out, int64_t* n_out);

/* HBM read-bandwidth probe (roofline calibration): streams `bytes` of device memory `reps`
 * times with 8-byte (mode 0) or 16-byte (mode 1) loads per lane; best GB/s. */
int sa_stream_probe(uint64_t bytes, int mode, int reps, double* gbps_out);

/* ------------------------------------------------------------------------------------- */
/* Part 2 -- HBM-resident index                                                            */
/* ------------------------------------------------------------------------------------- */

/* Upload one doc-range shard: roaringish words of all terms back to back (term-major, the
 * reference's ArrayDict layout, searcharray/phrase/memmap_arrays.py:15-56), CSR offsets
 * term_off[n_terms + 1], doc lengths of the shard's n_docs documents (doc ids inside `words`
 * are shard-local), and the GLOBAL statistics avg_doc_len / corpus_size (BM25 must use
 * global, not shard-local, statistics -- reference postings.py:293-299).
 * Derives on the device what PosnBitArray.warm() caches on the host (reference
 * middle_out.py:337-342): per-term TF postings and document frequencies.
 * tile_docs: docs per scoring tile (0 = default 2048; allowed 1024, 2048, 4096, 8192, 16384, 32768). */
int sa_index_create(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                    const uint64_t* words, const uint64_t* term_off, const float* doc_lens,
                    float avg_doc_len, uint64_t corpus_size, uint32_t tile_docs,
                    sa_index_t** out);
/* The same index read from the reference's ON-DISK format: one raw file of uint64 roaringish words
 * (ArrayDict.data.tofile, searcharray/phrase/memmap_arrays.py:158-161) whose per-term
 * {offset, length} metadata (element units, memmap_arrays.py:28-54; it travels in the reference's
 * pickle, memmap_arrays.py:196-208) is passed as term_src_off[n_terms] / term_len[n_terms]; a term
 * without an entry has length 0.  Terms may lie in any order in the file; the device keeps them
 * back to back in id order.  The file is streamed file -> page-locked ring -> HBM (reads and H2D
 * copies overlapped); no pageable host copy is made -- this replaces the np.memmap the reference
 * faults in page by page (memmap_arrays.py:163-165).  I/O failures are SA_ERR_IO, a term outside
 * the file is SA_ERR_ARG. */
int sa_index_create_from_file(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                              const char* path, const uint64_t* term_src_off, const uint64_t* term_len,
                              const float* doc_lens, float avg_doc_len, uint64_t corpus_size,
                              uint32_t tile_docs, sa_index_t** out);
/* Write the resident words to `path` in that same format (terms back to back in id order, i.e.
 * offset = term_off[t], length = term_off[t+1] - term_off[t] as returned by sa_index_words):
 * what PosnBitArray.memmap(data_dir) does for the host arrays (middle_out.py:333-335). */
int sa_index_save(sa_index_t* ix, const char* path);
/* The same index built from the TOKEN STREAM instead of pre-encoded words (index build on the
 * device): tokens[doc_ptr[d] .. doc_ptr[d+1]) are the term ids of doc d in position order
 * (position = index within the doc).  Replaces the host part of the reference indexer -- the stable
 * sort by term (indexing.py:102-115) and RoaringishEncoder.encode (roaringish.py:93-142); the
 * resulting words are byte-identical.  A term id >= n_terms or a doc longer than 18 * 2^18 tokens
 * ("Document length exceeds maximum", indexing.py:141-142) is SA_ERR_ARG. */
int sa_index_create_from_tokens(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                                const uint32_t* tokens, const uint64_t* doc_ptr, const float* doc_lens,
                                float avg_doc_len, uint64_t corpus_size, uint32_t tile_docs, sa_index_t** out);
/* copy the resident roaringish words (u64[n_words], see sa_index_info) and / or term offsets
 * (u64[n_terms + 1]) back to the host; either pointer may be null */
int sa_index_words(sa_index_t* ix, uint64_t* words_out, uint64_t* term_off_out);
int sa_index_destroy(sa_index_t* ix);
/* block until everything enqueued on the index's stream has finished */
int sa_index_synchronize(sa_index_t* ix);

/* shard-local document frequency of one term / of all terms (u64[n_terms])
 * reference SearchArray.docfreq postings.py:640-647 -> PosnBitArray.docfreq middle_out.py:521 */
int sa_index_docfreq(sa_index_t* ix, uint32_t term, uint64_t* out);
int sa_index_docfreqs(sa_index_t* ix, uint64_t* out);

/* dense term frequencies float32[n_docs] of one term (zeros for term >= n_terms)
 * reference SearchArray.termfreqs single-term branch, postings.py:607-638 */
int sa_index_termfreqs_dense(sa_index_t* ix, uint32_t term, float* out);

/* sparse TF postings of one term: doc ids u64[df] (global = local + doc_base), tfs f32[df]
 * reference PosnBitArray.termfreqs middle_out.py:481-509 */
int sa_index_termfreqs_sparse(sa_index_t* ix, uint32_t term, uint64_t* doc_ids_out,
                              float* tfs_out, int64_t* n_out);

/* Term-at-a-time BM25 over n_query_terms terms, summed in query-term order in fp32 -- the
 * caller idiom np.sum([arr.score(t) for t in q], axis=0) (reference test/test_msmarco.py:353)
 * with similarity = bm25_similarity(k1, b) (similarity.py:24-38).  idf[i] is the float32 idf of
 * term i computed by the host exactly as similarity.py:19-21 does.  terms[i] == 0xFFFFFFFF
 * (unknown term) contributes nothing.  out = float32[n_docs]. */
int sa_index_bm25_dense(sa_index_t* ix, const uint32_t* terms, const float* idf,
                        int n_query_terms, float k1, float b, float* out);

/* Exact-phrase match counts (slop == 0) as a dense float32[n_docs]: positions p with
 * terms[0]@p, terms[1]@p+1, ... under the reference's bigram-chain semantics, including its
 * same-term rule and plan selection (reference PosnBitArray.phrase_freqs middle_out.py:418-441,
 * compute_phrase_freqs :154-168, phrase/bigram_freqs.py:213-307).  Unknown terms give zeros
 * (postings.py:705-708); n_terms < 2 is an error (middle_out.py:425-426).
 * slop > 0 runs the reference's span search (phrase/spans.py:71-187, roaringish/spans.pyx:189-319):
 * header-set candidate selection + one thread per document replaying the 512-span state machine,
 * with the reference's observable quirks; at most 32 terms. */
int sa_index_phrase_freqs_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, float* out);

/* The same with the reference's min_posn / max_posn restriction (-1 = None): every term's words are
 * sliced with payload_slice first (reference middle_out.py:434-437, roaringish.py:266-282 -- the
 * reference compares the UNSHIFTED (word & msb_mask) with posn // 18, kept as is).
 * min_posn must be a multiple of 18 and max_posn a multiple of 18 minus 1, else SA_ERR_ARG. */
int sa_index_phrase_freqs_dense_posn(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                     int64_t min_posn, int64_t max_posn, float* out);
int sa_index_bm25_phrase_dense_posn(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                    int64_t min_posn, int64_t max_posn, float idf, float k1, float b, float* out);
/* single-term dense tf restricted to the position range (reference middle_out.py:489-497) */
int sa_index_termfreqs_dense_posn(sa_index_t* ix, uint32_t term, int64_t min_posn, int64_t max_posn, float* out);

/* The reference's other stock similarities (searcharray/similarity.py:41-89) of one term (n_terms == 1)
 * or one phrase (n_terms >= 2, slop as above), positions restricted like the _posn calls (-1 = open):
 * what SearchArray.score(token, similarity=bm25_impact() | bm25_legacy_similarity() | classic_similarity())
 * returns (postings.py:652-680).  Evaluated on the device with numpy's operation order and rounding, so
 * the result is bit-identical to the reference's: float32[n_docs] for SA_SIM_BM25_IMPACT, float64[n_docs]
 * for SA_SIM_BM25_LEGACY and SA_SIM_CLASSIC (their np.float64 idf promotes the product).  idf is what the
 * similarity computes from the doc frequencies (compute_idf, or log((N + 1) / (df + 1)) + 1 for classic);
 * it is ignored by SA_SIM_BM25_IMPACT, k1 / b are ignored by SA_SIM_CLASSIC.  Honours
 * sa_index_select_rows.  The caller handles avg_doc_len == 0 (the reference returns zeros). */
#define SA_SIM_BM25_IMPACT 1
#define SA_SIM_BM25_LEGACY 2
#define SA_SIM_CLASSIC 3
int sa_index_similarity_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                              int64_t min_posn, int64_t max_posn, int kind, double idf, double k1, double b,
                              void* out);

/* SearchArray.score(phrase): BM25 over the phrase counts, idf = float32 sum over the phrase's
 * terms computed by the host (reference postings.py:652-680, similarity.py:19-38). */
int sa_index_bm25_phrase_dense(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                               float idf, float k1, float b, float* out);

/* HIP-event time (ms) of the device work of the last phrase call on this index (D2H copy
 * excluded) and its algorithmic bytes: sum_t 8 * W_t, every word of every term once. */
int sa_index_last_profile(sa_index_t* ix, double* kernel_ms_out, uint64_t* alg_bytes_out);

/* Device-resident batch of B queries x T terms with top-k selection (k <= 1024).
 * terms/idf are [B][T] row-major.  Results per query: k (score, doc) pairs sorted by score
 * descending then doc id ascending; slots beyond the number of matching docs hold
 * score 0 and doc 0xFFFFFFFFFFFFFFFF.  Doc ids are global (local + doc_base).
 * The first batch created with a given (k1, b) also derives the index's impact stream for those
 * parameters (8 bytes per posting of HBM, shared by later batches with the same parameters and
 * released with the last of them; SA_IMPACT=0 or a failed allocation: batches score the TF
 * postings instead, same results). */
int sa_batch_create(sa_index_t* ix, const uint32_t* terms, const float* idf, int n_queries,
                    int n_query_terms, int k, float k1, float b, sa_batch_t** out);
/* Device-resident batch of B exact phrases (slop 0) with BM25 scoring and top-k selection: the
 * caller loop `for phrase in phrases: top_k(arr.score(phrase))` around reference
 * SearchArray.score (postings.py:652-680) -> PosnBitArray.phrase_freqs (middle_out.py:418-446) ->
 * compute_phrase_freqs (middle_out.py:73-168) -> bm25 (similarity.py:24-38).  terms is
 * [B][max_terms] row-major, phrase i uses its first n_terms[i] entries (n_terms[i] >= 2, no upper limit; fewer
 * than two terms is SA_ERR_ARG like the reference's ValueError, middle_out.py:425-426).  Exact phrases of
 * up to 18 pairwise-distinct terms are scored tile by tile; the others take the dense route described at
 * sa_phrase_batch_create_ex.  idf[B] is the per-phrase idf the host sums over the phrase's
 * terms (similarity.py:19-21).  An unknown term (id >= n_terms) makes the phrase match nothing.
 * The result is a sa_batch_t: run / run_local / merge_gathered / fetch / profile / destroy below
 * apply unchanged (profile: alg_bytes = postings_bytes = sum over phrases of 8 * words of its terms). */
int sa_phrase_batch_create(sa_index_t* ix, const uint32_t* terms, const int32_t* n_terms, const float* idf,
                           int n_phrases, int max_terms, int k, float k1, float b, sa_batch_t** out);
/* The same with a slop per phrase (slop == NULL: all exact): slop[i] > 0 scores phrase i with the reference's span
 * search (phrase/spans.py:71-187, roaringish/spans.pyx:189-319; at most 32 terms), as SearchArray.score(phrase,
 * slop=...) does.  Any phrase score() accepts is accepted: phrases with repeated terms, with more than 18 terms
 * or with slop > 0 are counted over the whole shard -- the slop phrases of a batch in SHARED launches -- and ranked on
 * the device; only the B x k results leave it. */
int sa_phrase_batch_create_ex(sa_index_t* ix, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                              const float* idf, int n_phrases, int max_terms, int k, float k1, float b, sa_batch_t** out);
/* A NEW set of queries in an existing batch (same n_queries, n_query_terms, k, k1, b as at creation): what
 * SearchArray.score does per call on a fresh query (reference postings.py:652-680; the loop timed by
 * test/test_msmarco.py:345-395), for a stream of batches.  The host derives the query-dependent tables into a
 * page-locked image; ONE asynchronous copy and one kernel (the slice table) are enqueued on the index stream behind the
 * runs still in flight.  Nothing is allocated and nothing is waited for: with two batches used alternately the host
 * prepares query set i+1 while the device scores query set i.  Fetch a batch's results before resetting it. */
int sa_batch_reset(sa_batch_t* batch, const uint32_t* terms, const float* idf);
/* the same for a phrase batch (same n_phrases and max_terms; slop == NULL: all exact) */
int sa_phrase_batch_reset(sa_batch_t* batch, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                          const float* idf);
/* one pass of the hot path over the batch; asynchronous on the index stream unless sync != 0.
 * If the index has a communicator (Part 3) the per-shard top-k are exchanged and merged. */
int sa_batch_run(sa_batch_t* batch, int sync);
/* External-collective variant of sa_batch_run for callers that own the exchange (e.g.
 * torch.distributed over RCCL): run this shard and copy its B*k ranking keys (u64, score bits
 * << 32 | ~global_doc) to a DEVICE buffer; then merge `nranks` gathered key blocks
 * [nranks][B][k] (DEVICE pointer) into the final top-k. */
int sa_batch_run_local(sa_batch_t* batch, void* local_keys_out_device, int sync);
int sa_batch_merge_gathered(sa_batch_t* batch, const void* gathered_keys_device, int nranks, int sync);
/* One step of a query STREAM in ONE call: a new query set (terms u32[B][T]) is weighted from the index's idf table
 * (sa_index_set_idf_table: one float32 per term, formed by the host exactly as similarity.py:19-21 forms it -- the
 * per-step gather is the only arithmetic-free part and moves into the library), then sa_batch_reset + sa_batch_run
 * (sync = 0).  What the caller loop score() -> top-k is per query (postings.py:652-680), per batch. */
int sa_index_set_idf_table(sa_index_t* ix, const float* idf_per_term, uint32_t n_terms);
int sa_batch_step(sa_batch_t* batch, const uint32_t* terms);

/* ---- Part 2c: a query-set QUEUE (csrc/sa_queue.hip).  A ring of `depth` batches of the same shape behind one handle, fed by a WORKER
 * THREAD of the library: sa_queue_submit copies a set of B x T term ids (weights come from the index's idf table, sa_index_set_idf_table,
 * as for sa_batch_step) and returns a ticket -- it blocks only while `depth` tickets are outstanding; the worker runs sa_batch_step for
 * the tickets in order; sa_queue_fetch(ticket) waits for that set's device work without holding any lock the worker needs and returns
 * its top-k (scores float32[B][k], doc ids uint64[B][k], as sa_batch_fetch).  Every ticket must be fetched; its slot is free again then.
 * The counterpart of the reference's callers that drive score() from a thread pool (test/test_msmarco.py:483-507): the host cost of
 * preparing a query set leaves the caller's thread.  sa_queue_batch exposes a slot's batch for diagnostics (sa_batch_last_route ...). */
typedef struct sa_queue sa_queue_t;
int sa_queue_create(sa_index_t* index, int n_queries, int n_query_terms, int k, float k1, float b, int depth, sa_queue_t** out);
int sa_queue_submit(sa_queue_t* queue, const uint32_t* terms, uint64_t* ticket_out);
int sa_queue_fetch(sa_queue_t* queue, uint64_t ticket, float* scores_out, uint64_t* docs_out);
int sa_queue_batch(sa_queue_t* queue, int slot, sa_batch_t** out);
int sa_queue_destroy(sa_queue_t* queue);
/* Results to the host: scores f32[B][k], docs u64[B][k].  Every sa_batch_run ends with an asynchronous copy of its
 * B*k keys into a page-locked buffer of the batch; fetch waits for THAT copy only (not for the streams), so other
 * batches of the index keep running behind it. */
int sa_batch_fetch(sa_batch_t* batch, float* scores_out, uint64_t* docs_out);
/* Mean HIP-event time (ms, events recorded on the index stream around the scoring kernel) over
 * the runs since the previous call, and the algorithmic bytes of one run: sum over queries of
 * (sum_t 8*df_t + 4*n_docs); postings_bytes is the sum_t 8*df_t part alone. */
int sa_batch_profile(sa_batch_t* batch, double* kernel_ms_out, uint64_t* alg_bytes_out,
                     uint64_t* postings_bytes_out);
/* Diagnostics of the dynamic pruning (BM25 batches): enable != 0 starts counting the candidate docs
 * the sparse path scores (one extra atomic each: not for timed runs); returns the count accumulated
 * since the previous call and how many queries of the last run were answered without a tile scan. */
int sa_batch_stats(sa_batch_t* batch, int enable, uint64_t* sparse_candidates_out, uint64_t* sparse_queries_out);
/* How the exhaustive path groups the batch (csrc/sa_bm25.hip, sa_k_bm25_group_tiles): out[0] = groups, out[1] = queries
 * in groups, out[2] = of them in groups that share their first term (the others are loose groups), out[3] = queries
 * left to the per-query kernel.  Diagnostics for benchmarks and tests; no reference counterpart. */
int sa_batch_group_info(sa_batch_t* batch, uint32_t out[4]);
/* Which route the batch's last run took: *pruned_out = 2 the staged-tile route (csrc/sa_stage.hip), 1 dynamic pruning, 0 exhaustive scoring
 * (the options `stage` / `sparse`, or -- unset -- the library's rule: csrc/sa_stage.hip sa_stage_plan, csrc/sa_bm25.hip sa_batch_run_shard;
 * measured: profiles/route_rule_r06*.jsonl).  All three are exact.  Diagnostics; no reference counterpart. */
int sa_batch_last_route(sa_batch_t* batch, int* pruned_out);
/* Diagnostics: the rank table (22 lower bounds of a term's r-th largest BM25 factor, r = 1, 2, ... 1024) and the exact largest factor of
 * `term` in this batch's impact stream -- what starting bounds are formed from.  No reference counterpart. */
int sa_batch_debug_rank_table(sa_batch_t* batch, uint32_t term, float* ranks22_out, float* maxf_out);
/* Host time this batch's steps have cost, cumulative nanoseconds by part: out[0] = sa_batch_reset / _step up to the upload
 * (grouping + pruning tables: CPU work only), out[1] = its enqueues (the upload copy, the slice-table launch), out[2] =
 * sa_batch_run's enqueues, out[3] = number of query sets filled.  Diagnostics (scripts/host_cost.py); no reference counterpart. */
int sa_batch_host_times(sa_batch_t* batch, uint64_t out[4]);
/* The bound every query of the current set STARTS with (a score; 0: none), in caller order: the best, over the query's terms, of
 * weight x a lower bound of the term's k-th largest BM25 factor in this shard (the rank tables built with the impact stream,
 * csrc/sa_bm25.hip sa_k_make_topf) -- never above the query's k-th best score.  Diagnostics / tests; no reference counterpart. */
int sa_batch_seeds(sa_batch_t* batch, float* out);
int sa_batch_destroy(sa_batch_t* batch);

typedef struct sa_index_info {
    uint64_t n_docs, doc_base, corpus_size, n_words, n_postings;
    uint32_t n_terms, tile_docs, n_tiles, n_dir_terms;
    uint64_t hbm_bytes;          /* device memory held by the index */
    int device;
    int dl_packed;
    uint32_t n_docdir_terms;     /* frequent terms with a doc directory (phrase probes) */
    uint32_t n_tf8_terms;        /* frequent terms with a dense tf row (dynamic pruning) */
} sa_index_info_t;
int sa_index_info(sa_index_t* ix, sa_index_info_t* out);

/* Subset output for the dense calls above (sliced arrays, rerank-top-N: reference arr[mask].score(...),
 * postings.py:619-627,702-703): after sa_index_select_rows the NEXT dense call on `ix` made by the same
 * thread writes out[i] = dense[rows[i]] for i < n_rows (n_rows floats; rows >= n_docs give 0) instead of
 * the whole vector -- the rows are gathered on the device.  The selection is consumed by that call;
 * rows must stay valid until it returns.  rows == NULL clears a pending selection. */
int sa_index_select_rows(sa_index_t* ix, const uint64_t* rows, uint64_t n_rows);

/* The same WITHOUT state between calls: every dense call above has a `_to` twin that takes its destination as an argument --
 * `rows` != NULL: out[i] = dense[rows[i]] for i < n_rows (the subset of sa_index_select_rows); `vec` != NULL: the result
 * (x boost if has_boost) goes into that device vector (Part 4, sa_index_select_vec) and `out` is not written; both NULL (or
 * dest == NULL): the whole vector into `out`.  A binder in any language can use these alone; the select calls stay for
 * callers that already use them. */
typedef struct sa_dense_dest {
    const uint64_t* rows; uint64_t n_rows;      /* subset of the docs, gathered on the device */
    struct sa_vec* vec; float boost; int has_boost;   /* or: a float32 device vector */
} sa_dense_dest_t;
int sa_index_termfreqs_dense_to(sa_index_t* ix, uint32_t term, const sa_dense_dest_t* dest, float* out);
int sa_index_termfreqs_dense_posn_to(sa_index_t* ix, uint32_t term, int64_t min_posn, int64_t max_posn,
                                     const sa_dense_dest_t* dest, float* out);
int sa_index_bm25_dense_to(sa_index_t* ix, const uint32_t* terms, const float* idf, int n_query_terms, float k1, float b,
                           const sa_dense_dest_t* dest, float* out);
int sa_index_phrase_freqs_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop,
                                   const sa_dense_dest_t* dest, float* out);
int sa_index_phrase_freqs_dense_posn_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                        int64_t max_posn, const sa_dense_dest_t* dest, float* out);
int sa_index_bm25_phrase_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, float idf, float k1, float b,
                                  const sa_dense_dest_t* dest, float* out);
int sa_index_bm25_phrase_dense_posn_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                       int64_t max_posn, float idf, float k1, float b, const sa_dense_dest_t* dest, float* out);
int sa_index_similarity_dense_to(sa_index_t* ix, const uint32_t* terms, int n_terms, int slop, int64_t min_posn,
                                 int64_t max_posn, int kind, double idf, double k1, double b, const sa_dense_dest_t* dest,
                                 void* out);

/* Page-locked host buffers for dense results (optional): any host pointer works as `out` of the
 * dense calls above; a buffer from sa_host_alloc is filled at full PCIe rate and can be recycled by
 * the binding without first-touch page faults (searcharray_amd/device_index.py keeps a small pool). */
int sa_host_alloc(uint64_t bytes, void** out);
int sa_host_free(void* p);

/* ------------------------------------------------------------------------------------- */
/* Part 3 -- doc-range sharding: per-shard top-k exchange over RCCL (xGMI)                 */
/* ------------------------------------------------------------------------------------- */
/* The reference is single-process; there is no counterpart to cite.  One process per GPU, each
 * holding the shard [doc_base, doc_base + n_docs) built with GLOBAL avg_doc_len / corpus_size /
 * idf.  The only collective is an all-gather of B*k 8-byte ranking keys per batch. */
#define SA_COMM_ID_BYTES 128
int sa_comm_unique_id(char* id_out, int len);                    /* rank 0: ncclGetUniqueId */
/* which collective library this process resolved: ncclGetVersion() and the path of the shared object that provides it
 * (benchmarks record both: an environment can put another RCCL in front of /opt/rocm's) */
int sa_comm_library_info(int* version_out, char* path_out, int path_len);
int sa_index_comm_init(sa_index_t* ix, int rank, int nranks, const char* id_bytes, int len);
int sa_index_comm_destroy(sa_index_t* ix);
/* rank / number of ranks of the index's communicator (0 / 1 without one) */
int sa_index_comm_info(sa_index_t* ix, int* rank_out, int* nranks_out);
/* Host-side reductions over the communicator, in place and blocking: the index-time statistics of a
 * doc-range-sharded corpus -- df per term and the sum of the doc lengths (-> avgdl) are summed over the
 * shards once, BM25 then uses the global values on every shard (the reference computes them over the
 * whole corpus, indexing.py:282-284, middle_out.py:521-528) -- and max-over-ranks timings.  With these
 * a one-process-per-GPU caller needs no other collective library (bench.py uses nothing else). */
#define SA_DT_U64 0
#define SA_DT_F64 1
#define SA_OP_SUM 0
#define SA_OP_MAX 1
int sa_index_comm_allreduce(sa_index_t* ix, void* host_inout, uint64_t n, int dtype, int op);
/* all ranks arrived and everything enqueued on this index's streams has finished */
int sa_index_comm_barrier(sa_index_t* ix);

/* Part 3b -- N devices behind ONE handle.  The corpus (an encoded index: words / term_off as for sa_index_create,
 * doc_lens for all docs) is cut into n_dev doc-id ranges [g * n_docs / n_dev, (g + 1) * n_docs / n_dev) -- a
 * contiguous run of every term's word list, what the reference's key_partition computes (roaringish.py:227-243) --,
 * shard g is built on device_ids[g] with the GLOBAL corpus size and avg_doc_len, the shards join one RCCL
 * communicator (one host thread per device inside the library) and the document frequencies are summed over the
 * shards.  avg_doc_len is the caller's (the reference forms it as np.mean over the float32 lengths of the whole
 * corpus, indexing.py:282-284).  A batch of the handle is one resident batch per shard: run scores every shard's doc
 * range, all-gathers the per-shard top-k keys and merges them on every device; fetch returns the merged result
 * (identical to a single index over the whole corpus).  Dense drop-in calls go to the shards (sa_sharded_shard
 * lends their handles; they stay owned by the sharded handle): a dense result is the shard vectors in doc order. */
typedef struct sa_sharded sa_sharded_t;
typedef struct sa_sharded_batch sa_sharded_batch_t;
int sa_sharded_create(const int* device_ids, int n_dev, uint64_t n_docs, uint32_t n_terms, const uint64_t* words,
                      const uint64_t* term_off, const float* doc_lens, float avg_doc_len, uint32_t tile_docs,
                      sa_sharded_t** out);
int sa_sharded_destroy(sa_sharded_t* sh);
/* number of shards and their doc-id cuts (bounds_out: n_shards + 1 entries, or NULL) */
int sa_sharded_info(sa_sharded_t* sh, int* n_shards_out, uint64_t* bounds_out);
int sa_sharded_shard(sa_sharded_t* sh, int g, sa_index_t** out);
/* corpus-wide document frequency of every term (what idf must be computed from) */
int sa_sharded_docfreqs(sa_sharded_t* sh, uint64_t* df_out);
/* the batch entry points of Part 2 over all shards: same arguments, same results as on a single index */
int sa_sharded_batch_create(sa_sharded_t* sh, const uint32_t* terms, const float* idf, int n_queries, int n_query_terms,
                            int k, float k1, float b, sa_sharded_batch_t** out);
int sa_sharded_phrase_batch_create(sa_sharded_t* sh, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                                   const float* idf, int n_phrases, int max_terms, int k, float k1, float b,
                                   sa_sharded_batch_t** out);
int sa_sharded_batch_reset(sa_sharded_batch_t* batch, const uint32_t* terms, const float* idf);
int sa_sharded_batch_run(sa_sharded_batch_t* batch, int sync);
/* replace the options of every shard's batch (as sa_batch_set_options does for one) */
int sa_sharded_batch_set_options(sa_sharded_batch_t* batch, const sa_options_t* opts);
int sa_sharded_batch_fetch(sa_sharded_batch_t* batch, float* scores_out, uint64_t* docs_out);
int sa_sharded_batch_destroy(sa_sharded_batch_t* batch);

/* ------------------------------------------------------------------------------------- */
/* Part 4 -- dense vectors on the device: the combine step of Solr-style multi-field queries */
/* ------------------------------------------------------------------------------------- */
/* reference searcharray/solr.py:112-355 combines the per-field, per-term score() vectors with numpy on
 * the host.  Here the vectors stay in HBM: a score is computed straight into a float32 vector
 * (sa_index_select_vec + any dense call of Part 2) and combined by the calls below, which reproduce
 * numpy's arithmetic (float64 accumulators fed by float32 scores on the term-centric path, float32 on
 * the field-centric path, one rounding per operation).  All calls are synchronous.
 * Vector kinds: f64 (is_f64 != 0) or 32-bit (float32 values, or uint32 counters). */
int sa_vec_create(int device, uint64_t n, int is_f64, sa_vec_t** out);        /* zero-initialised */
int sa_vec_destroy(sa_vec_t* v);
int sa_vec_zero(sa_vec_t* v);
int sa_vec_copy(sa_vec_t* dst, const sa_vec_t* src);                          /* same kind and length */
int sa_vec_fetch(sa_vec_t* v, void* host_out);                               /* n * 8 or n * 4 bytes */
/* the next dense call on `ix` from this thread writes (result * boost if has_boost) into out32 on the
 * device instead of to its host `out`; out32 == NULL clears a pending selection */
int sa_index_select_vec(sa_index_t* ix, sa_vec_t* out32, float boost, int has_boost);
/* term-centric (solr.py:124-143): sum += s; max = maximum(max, s)  |  clause = max + (sum - max) * tie;
 * total += clause; cnt += clause > 0  |  total[cnt < need] = 0 */
int sa_vec_dismax_acc(sa_vec_t* sum64, sa_vec_t* max64, const sa_vec_t* s32);
int sa_vec_clause(const sa_vec_t* sum64, const sa_vec_t* max64, double tie, sa_vec_t* total64, sa_vec_t* cnt32);
int sa_vec_mask_min_count(sa_vec_t* v64, const sa_vec_t* cnt32, uint32_t need);
/* field-centric (solr.py:156-176): sum += s; cnt += s > 0  |  row = (cnt >= need ? sum : 0) * boost;
 * fsum (+)= row; fmax = max(fmax, row)  |  out = fmax + (fsum - fmax) * tie */
int sa_vec_sum_count32(sa_vec_t* sum32, sa_vec_t* cnt32, const sa_vec_t* s32);
int sa_vec_field_row(const sa_vec_t* sum32, const sa_vec_t* cnt32, uint32_t need, float boost, int has_boost, int first,
                     sa_vec_t* fsum32, sa_vec_t* fmax32);
int sa_vec_field_finish(const sa_vec_t* fsum32, const sa_vec_t* fmax32, float tie, sa_vec_t* out32);
/* phrase boosts (solr.py:320-353): dst32 (+)= src32  |  dst[i] += extra32[i] where mask[i] != 0  |
 * docs with tf32 > 0 and mask > 0 (the subset-local docfreq of the matched docs) */
int sa_vec_add32(sa_vec_t* dst32, const sa_vec_t* src32, int first);
int sa_vec_add_where(sa_vec_t* dst, const sa_vec_t* extra32, const sa_vec_t* mask);
int sa_vec_count_where(const sa_vec_t* tf32, const sa_vec_t* mask, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* SEARCHARRAY_HIP_H */

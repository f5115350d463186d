// sa_options.hpp -- the library's switches as DATA: one sa_options_t per index handle and per batch (include/searcharray_hip.h,
// Part 0), instead of process environment variables read inside the kernels' launch code (rounds 1-4).  The reference configures
// the same way -- keyword arguments on the objects (searcharray/postings.py:250-258 `SearchArray.index(..., batch_size, avoid_copies,
// data_dir)`, :652-656 `score(token, similarity, min_posn, max_posn)`), never the environment.
//
// Every field is an int64; SA_OPT_UNSET means "the library decides" (its default or its automatic rule), exactly what an unset
// environment variable meant.  SA_OPTION_LIST is the one place a switch is declared: the public struct's fields
// (include/searcharray_hip.h lists them in the same order; a static_assert pins the size), the name table of sa_options_set /
// sa_option_name, and the debug override SA_OPTS="name=value,..." -- read ONCE, when the library first needs its process defaults.
#pragma once
#include <stdint.h>
#include "../../include/searcharray_hip.h"

// X(field)  -- order = order of the fields in sa_options_t
#define SA_OPTION_LIST(X)                                                                                                          \
    /* ---- BM25 top-k batches (sa_bm25.hip, sa_sparse.hip) */                                                                     \
    X(sparse)          /* 1: dynamic pruning (MaxScore), 0: exhaustive scoring; unset: the rule of sa_batch_run_shard */           \
    X(group)           /* 0: no grouped kernel (queries that share their first term are scored one by one) */                       \
    X(group_loose)     /* 0: no loose groups */                                                                                    \
    X(group_side)      /* 0: ungrouped rows on the batch's own stream instead of the side stream */                                \
    X(group_dense)     /* 0: the grouped kernel builds its base from postings even where a dense factor row exists */              \
    X(group_one)       /* left-over queries (too dense for a loose group, first term shared with nobody) as groups of one: 2 all (default), 1 only over a dense factor row, 0 none (per-query kernel) */ \
    X(group_min)       /* smallest group (default 2) */                                                                            \
    X(group_maxq)      /* queries per table pass of a grouped item (default and at most 16) */                                      \
    X(group_item)      /* queries per grouped item: passes of group_maxq queries over ONE base (at most 64; default 32 for big launches, else 16) */ \
    X(group_warm)      /* tiles scored by the per-query kernel first to establish bounds (default: none with starting bounds) */   \
    X(loose_postings)  /* loose groups: expected postings of a query per tile at most (default 400) */                             \
    X(xcd_range)       /* 0: tiles dealt round-robin to the XCDs instead of ranges */                                              \
    X(term_seed)       /* 0: no starting bounds from the terms' rank tables */                                                     \
    X(topf_slice)      /* TEST HOOK: postings per workgroup of a long list's rank-table histogram (default 65536; lists of 4 slices and more) */ \
    X(seed_scale_pct)  /* TEST HOOK: starting bounds scaled by this percentage (> 100 makes them too high: the redo path) */       \
    X(merge_small)     /* 0: the 1024-thread merge also for k <= 64 */                                                             \
    X(impact)          /* 0: score the TF postings, no impact stream */                                                            \
    X(pruned_topk)     /* 0: block-level selection (no bounds) */                                                                  \
    X(no_topk)         /* timing experiments: skip the per-tile selection */                                                       \
    X(topk_hist)       /* 0: slot bound instead of the histogram bound */                                                          \
    X(topk_hist_mink)  /* smallest k that takes the histogram bound */                                                             \
    X(cand_cap)        /* TEST HOOK: candidate-list capacity per query (forces the overflow handling) */                           \
    X(sparse_div)      /* pruning: a lead term has at most n_docs / this postings (default 8) */                                   \
    X(sparse_lazy)     /* 0: pruning tables derived at every reset, needed or not */                                               \
    X(bloom_floor)     /* TEST HOOK: smallest Bloom buffer in bytes */                                                             \
    X(sp_chunk1)       /* pruning: postings per lead work item */                                                                  \
    X(stage)           /* staged-tile route (sa_stage.hip): 1 force where eligible, 0 off; unset: on where eligible and `sparse` is unset */ \
    X(stage_docs)      /* docs per stage tile (multiple of 64; default: what fits the LDS stage for the query set's terms) */                 \
    X(stage_wgs)       /* staged-tile route: resident workgroups per CU (default 2) */                                              \
    X(stage_cw)        /* staged-tile route: workgroups of an XCD that co-walk a range of tiles (default 32; 1: private ranges) */                \
    X(stage_probe)     /* 0: the staged-tile route streams EVERY term of the batch; default: terms that cannot be essential are probed in dense rows */ \
    X(probe_div)       /* probe rows (dense factor rows the staged-tile route probes) for terms with df >= n_docs / this (default 128; 0: none) */ \
    X(dense_direct)    /* 0: sa_index_bm25_dense scores the TF postings into scratch and copies (rounds 1-5); default: one launch over the impact stream, straight into the destination */ \
    X(batch_stream)    /* 0: batches share the index stream */                                                                     \
    X(res_xs)          /* 0: result copies on the batches' own streams */                                                          \
    /* ---- index creation (sa_index.hip, sa_bm25.hip) */                                                                          \
    X(dense_div)       /* dense factor rows for terms with df >= n_docs / this (default 4) */                                      \
    X(dir_div)         /* tile directory rows for terms with df >= n_tiles / this */                                               \
    X(docdir_div)      /* doc directory rows for terms with >= n_docs / this words (0: none) */                                    \
    X(tf8_div)         /* dense tf rows for terms with df >= n_docs / this */                                                      \
    X(tf8_maxrows)     /* ... at most this many */                                                                                 \
    X(seg_words)       /* TEST HOOK: words per segment of the posting derivation */                                                \
    /* ---- phrases (sa_phrase.hip, sa_phrase_batch.hip) */                                                                        \
    X(phrase_mode)     /* 0 auto, 1 general chain, 2 fused kernel */                                                               \
    X(phrase_docdir)   /* 0: no doc directory probes */                                                                            \
    X(phrase_docs)     /* 0: no chain per document */                                                                              \
    X(phrase_lanes)    /* phrase tiles: lanes per phrase */                                                                        \
    X(ptile)           /* docs per phrase tile (2048 / 4096) */                                                                    \
    /* ---- slop (sa_spans.hip) */                                                                                                 \
    X(span_doc)        /* 0: no doc-parallel route */                                                                              \
    X(span_docdir)     /* 0: no doc directory */                                                                                   \
    X(span_doc_multi)  /* 0: one launch per phrase */                                                                              \
    X(span_doc_rank)   /* 0: ranking in its own launch */                                                                          \
    X(span_fast)       /* 0: general state machine only */                                                                         \
    X(span_multi)      /* 0: no multi-phrase launch of the general route */                                                        \
    X(span_sort)       /* 1 / 0: force / forbid sorting the docs by work */                                                        \
    X(span_lds_pad)    /* MEASUREMENT HOOK: unused dynamic LDS per block of the doc-parallel batch launch (fewer resident blocks per CU) */   \
    X(span_bundle)     /* doc-parallel batch launch: phrases whose blocks take turns in the launch (default 32; 1: a phrase's blocks back to back) */   \
    X(span_tab_waves)  /* doc-parallel batch launch: waves of a block that hold span tables, 2 / 4 (unset: 2 when the launch has more blocks than four per CU, else 4) */   \
    X(span_threads)    /* TEST HOOK: grid cap (forces the stride loop) */                                                          \
    /* ---- index files (sa_io.hip) */                                                                                             \
    X(io_piece_bytes)  /* bytes per staged piece */                                                                                \
    X(io_threads)      /* file threads */                                                                                          \
    /* ---- diagnostics */                                                                                                         \
    X(trace)           /* 1: route decisions to stderr */

struct sa_options_fields {
    uint64_t struct_size;
#define SA_X(f) int64_t f;
    SA_OPTION_LIST(SA_X)
#undef SA_X
};
static_assert(sizeof(sa_options_fields) == sizeof(sa_options_t), "include/searcharray_hip.h: sa_options_t out of step with SA_OPTION_LIST");

// the value of a switch, or what the library does when nobody set it
static inline long long sa_opt(int64_t v, long long dflt) { return v == SA_OPT_UNSET ? dflt : (long long)v; }
static inline bool sa_opt_is_set(int64_t v) { return v != SA_OPT_UNSET; }

// the process defaults: all unset + the SA_OPTS override, parsed once
const sa_options_t& sa_options_process_defaults();
// what a handle created by the calling thread starts from: the thread's defaults (sa_options_set_thread_defaults), else `fallback`
// (a batch: its index's options), else the process defaults
sa_options_t sa_options_for_new_handle(const sa_options_t* fallback);
bool sa_options_thread_defaults(sa_options_t* out);            // the calling thread's defaults, if it has set any

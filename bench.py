#!/usr/bin/env python
"""bench.py -- the reference's headline workload on MI355X (no PyTorch anywhere: numpy + the C ABI).

Metric (BASELINE.json): queries/sec (+ GB/s of postings scanned) of 4-term disjunctive BM25 with
top-k over a 10M-doc synthetic Zipf corpus, 1/2/4/8 GPUs.

  python bench.py --gpus N --steps K --warmup W          # N > 1: spawns one rank per GPU itself
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # or is launched as ranks

Either way every rank is one process on one GPU that reads RANK / LOCAL_RANK / WORLD_SIZE from the
environment; the ranks find each other through a 128-byte id file and everything collective -- global df,
max-over-ranks timing, barriers, the per-batch exchange of the per-shard top-k -- goes through
libsearcharray_hip.so's own RCCL communicator (include/searcharray_hip.h Part 3).

A "step" = one pass of the hot path over one batch of 256 queries THE DEVICE HAS NOT SEEN, ONE library call
(sa_batch_step): the batch's idf weights are gathered from the index's per-term table (float64 numpy arithmetic of the
reference, formed once like df itself), sa_batch_reset (the query set's tables -- the staged-tile route's plan, or grouping /
pruning tables -- into a page-locked image, one async copy), then sa_batch_run: the scoring kernel of the route the library
picks for the batch's shape (staged tiles, exhaustive overlay or dynamic pruning -- all exact, identical results)
+ per-shard merge (+ RCCL all-gather of the per-shard
top-k keys and a final merge when N > 1) + an async copy of the B x k results to the host, which the loop fetches
two steps later.  Eight seeded query sets rotate through two batch objects, so nothing is replayed: what is timed is
what a query stream gets (the reference's unit of work is score() on a fresh query, postings.py:652-680, timed as
test/test_msmarco.py:345-395).  The 10M-doc corpus is sharded by doc-id range (10M / N docs per GPU, global BM25
statistics); every rank scores every query of a batch on its docs.  The main region takes 256 queries per step at EVERY N
(`"scaling": "strong"`: value(N) / value(1) is a like-for-like speed-up), and a WIDE batch of 2048 queries per step is timed
beside it at every N, N = 1 included (`wide_batch`: wide_batch(N) / wide_batch(1) is the same-batch speed-up of a
deployment that batches wider -- a rank's device work per step shrinks with N, the host's cost per batch does not).  The
timed region of --steps steps is repeated --repeats times (5): `value` is the median region, min / max are in `repeats`.
The index is resident in HBM before the timed region.  Rank 0 prints one JSON line.

Legs (all on the same resident index; only the first is `value`):
  main (fresh)      8 rotating BASELINE-shaped query sets (256 x 4 terms, one rank from each of 1-10 / 11-100 /
                    101-1000 / 1001-10000; set 0 is THE BASELINE set, seed 42) through the library's DEFAULT route for
                    the batch's shape (round 6, k <= 32: the staged-tile route; `roofline.route` names it): the exact top-k of
                    every query -- identical to scoring every posting of every query term like the reference does, which
                    parity_check verifies against the reference itself; reset + run + fetch per step
  replay            set 0 resident, run only (what rounds 1-2 reported as `value`); its HIP-event kernel time is the roofline's
  exhaustive_overlay  set 0 with every posting scored (option sparse = 0: rounds 2-5's headline route), replayed
  dynamic_pruning   set 0 through sa_sparse.hip (option sparse = 1), replayed
  distinct_terms    256 x 4 pairwise-distinct terms (ranks 1..1024), default route (the exhaustive overlay): no posting list is
                    shared between queries, so cache reuse cannot flatter the bandwidth figure
  dense_score       BASELINE config 2's literal call: single-term BM25 -> float32[n_docs], to the host and left on the device
  phrase_batch      BASELINE config 3: 256 sampled 3-token phrases -> BM25 -> top-10 on a resident zipf-1M index
  slop_batch        BASELINE config 5 (synthetic stand-in): 256 two-token slop-2 phrases (half on ranks 1-50) -> top-10, same index
Roofline blocks are per leg; see roofline_block().
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
MARKER = "sa_k_stream8"    # PMC child: two dispatches of this kernel separate the legs
CALIB = "sa_k_stream16"    # PMC child: a known byte count read with 16-byte loads (FETCH_SIZE calibration)
CALIB_BYTES = 1 << 30
LOADS_8B = ("sa_k_bm25_group_tiles",)   # kernels whose posting loads are 8 bytes per lane


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--tile", type=int, default=0, help="docs per scoring tile (0 = library default)")
    ap.add_argument("--corpus-cache", default="", help="directory to cache the encoded corpus shard (.npz)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (traffic / L2 hit rate)")
    ap.add_argument("--driver", choices=["ring", "queue"], default="ring",
                    help="who feeds the fresh-batch stream: 'ring' = this process steps a ring of batch objects (sa_batch_step + sa_batch_fetch per step: "
                         "rounds 3-6), 'queue' = the library's query-set queue (sa_queue_*: a worker thread of the library steps the ring)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="a library option for every batch of the run (e.g. stage=0: the main region on round 5's routes); measurement only")
    ap.add_argument("--pruned", action="store_true",
                    help="time dynamic pruning in the main region (default: the exhaustive kernel, which scores every "
                         "posting like the reference; the other mode is always reported beside it)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of each CPU baseline leg")
    ap.add_argument("--query-sets", type=int, default=8, help="seeded query sets the main leg rotates through")
    ap.add_argument("--scaled-queries", type=int, default=2048,
                    help="queries per step of the wide_batch leg, timed at EVERY N beside the main stream (0: off)")
    ap.add_argument("--batch-mult", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = the median region")
    ap.add_argument("--weak", action="store_true", help="N > 1: N x --queries per step in the main region (default: --queries at every N)")
    ap.add_argument("--strong", action="store_true", help=argparse.SUPPRESS)     # (round 3's switch; the default now)
    ap.add_argument("--pipeline", type=int, default=6, help="batch objects (= batches in flight, one stream each) of the main leg")
    ap.add_argument("--no-phrase-legs", action="store_true", help="skip the zipf-1M phrase / slop legs")
    ap.add_argument("--phrase-docs", type=int, default=1_000_000)
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)      # internal: run under rocprofv3, print nothing
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launching: plain `python bench.py --gpus N` spawns the ranks itself
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    idfile = os.path.join(tempfile.gettempdir(), f"sa_bench_id_{os.getpid()}_{int(time.time())}")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), SA_BENCH_ID_FILE=idfile,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:                                   # a rank that dies must not leave the others waiting in a collective
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
            if code != 0:
                for other in live:
                    other.kill()
        time.sleep(0.05)
    if os.path.exists(idfile):
        os.unlink(idfile)
    sys.exit(rc)


def id_file_path():
    if os.environ.get("SA_BENCH_ID_FILE"):
        return os.environ["SA_BENCH_ID_FILE"]
    # launched by torch.distributed.run (or anything else that sets RANK): all ranks share the launcher
    # as parent and the rendezvous port, which names a file no other job on this node uses
    return os.path.join(tempfile.gettempdir(),
                        f"sa_bench_id_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}")


def exchange_unique_id(rank, make_id):
    path = id_file_path()
    if rank == 0:
        uid = make_id()
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)                 # atomic: a reader sees nothing or all 128 bytes
        return uid, path
    t0 = time.time()
    while time.time() - t0 < 600:
        try:
            if os.path.getmtime(path) > START_TIME - 600:      # (a leftover of a crashed earlier job is ignored)
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid, path
        except OSError:
            pass
        time.sleep(0.05)
    raise RuntimeError(f"rank {rank}: no communicator id at {path} after 600 s")


START_TIME = time.time()


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        if self.world != args.gpus:
            log(self.rank, f"warning: --gpus {args.gpus} but WORLD_SIZE {self.world}; using WORLD_SIZE")
        self.api = None
        self.index = None
        self.collective = "none"
        # SA_BENCH_FORCE_COMM=1: take the communicator path with a single rank too (exercises the RCCL
        # bootstrap, all-reduce, barrier and exchange on a one-GPU box)
        self.use_comm = self.world > 1 or os.environ.get("SA_BENCH_FORCE_COMM") == "1"
        self.idf_table = None

    # -- corpus (host only: nothing here touches the GPU) -----------------------------------------
    def generate(self):
        from searcharray_amd import synth
        a = self.args
        D, V = a.docs, a.vocab
        self.lo = (D * self.rank) // self.world
        self.hi = (D * (self.rank + 1)) // self.world
        t0 = time.time()
        workers = max(1, min(8, (os.cpu_count() or 8) // self.world))
        corpus = None
        cpath = os.path.join(a.corpus_cache, f"zipf_{D}_{V}_{self.lo}_{self.hi}.npz") if a.corpus_cache else ""
        if cpath and os.path.exists(cpath):
            z = np.load(cpath)
            corpus = synth.EncodedCorpus(z["words"], z["term_off"], z["doc_lens"], self.hi - self.lo, V, self.lo)
        if corpus is None:
            corpus = synth.zipf_corpus(self.hi - self.lo, vocab=V, doc_base=self.lo, total_docs=D, workers=workers)
            if cpath:
                os.makedirs(a.corpus_cache, exist_ok=True)
                np.savez(cpath, words=corpus.words, term_off=corpus.term_off, doc_lens=corpus.doc_lens)
        self.corpus = corpus
        log(self.rank, f"shard docs [{self.lo},{self.hi}) words={len(corpus.words)} generated in {time.time()-t0:.1f}s "
                       f"({workers} host threads)")
        # global statistics (index-time constants, replicated).  avgdl exactly as the reference forms it,
        # np.mean over the float32 lengths of the WHOLE corpus (indexing.py:282-284): the lengths are the
        # first draw of every seeded batch, so every rank can produce all of them without the tokens
        all_lens = corpus.doc_lens if self.world == 1 else synth.zipf_doc_lens(D)
        self.avgdl = np.float32(np.mean(all_lens))

    # -- index --------------------------------------------------------------------------------------
    def build(self):
        from searcharray_amd import _lib
        from searcharray_amd.device_index import DeviceIndex
        self.api = _lib.api()                 # fails loudly if the gfx950 library is missing
        a, corpus, D = self.args, self.corpus, self.args.docs
        t0 = time.time()
        self.index = DeviceIndex(corpus.words, corpus.term_off, corpus.doc_lens, avg_doc_len=self.avgdl,
                                 corpus_size=D, doc_base=self.lo, device=self.local_rank, tile_docs=a.tile, api=self.api)
        self.info = self.index.info()
        log(self.rank, f"index resident: {self.info.n_postings} postings, {self.info.hbm_bytes/1e9:.2f} GB HBM, "
                       f"tile_docs={self.info.tile_docs} n_tiles={self.info.n_tiles} "
                       f"({time.time()-t0:.1f}s incl. H2D + derive)")
        df = self.index.docfreqs().astype(np.uint64)
        if self.use_comm:
            uid, path = exchange_unique_id(self.rank, lambda: DeviceIndex.comm_unique_id(self.api))
            self.index.comm_init(self.rank, self.world, uid)
            self.collective = "rccl"
            df = self.index.comm_allreduce(np.ascontiguousarray(df), "sum")      # global df = sum over the shards
            self.index.comm_barrier()
            if self.rank == 0 and os.path.exists(path):
                os.unlink(path)
        self.df = df
        self.index.set_global_docfreqs(df)

    def idf_of(self, queries):
        """idf weights [B][T] as the reference forms them per term -- float64 numpy, then float32 (similarity.py:19-21,
        bm25.pyx:31) -- for a whole batch at once: this is host work of every NEW batch and is inside the timed step"""
        if self.idf_table is None:
            # one float64 log per TERM of the vocabulary, once (index-time statistics, as df itself); a batch gathers
            dfs = self.df
            self.idf_table = np.log(1 + (self.args.docs - dfs + 0.5) / (dfs + 0.5)).astype(np.float32)
            self.index.set_idf_table(self.idf_table)         # (sa_batch_step gathers a query set's weights from it)
        return self.idf_table[queries]

    def make_batch(self, queries, check=True):
        from searcharray_amd.device_index import QueryBatch, compute_idf
        D = self.args.docs
        idf = self.idf_of(queries)
        if check:                                            # (the vectorised table against the reference's per-term arithmetic)
            ref = np.asarray([[compute_idf(D, np.asarray([self.df[t]])) for t in q] for q in queries], dtype=np.float32)
            if not np.array_equal(ref, idf):
                raise AssertionError("vectorised idf differs from the reference's per-term compute_idf")
        return QueryBatch(self.index, queries, k=self.args.k, idf=idf)

    # -- collectives over the library's communicator -----------------------------------------------
    def barrier(self):
        self.index.synchronize()
        if self.use_comm:
            self.index.comm_barrier()

    def allmax(self, x):
        if not self.use_comm:
            return x
        return float(self.index.comm_allreduce(np.asarray([x], dtype=np.float64), "max")[0])

    def allsum(self, x):
        if not self.use_comm:
            return x
        return float(self.index.comm_allreduce(np.asarray([x], dtype=np.float64), "sum")[0])

    def timed(self, batch, n_warm, n_steps):
        """n_warm untimed steps, then n_steps bracketed by barrier + synchronize; max over ranks"""
        for _ in range(n_warm):
            batch.run(sync=False)
        self.barrier()
        batch.profile()                                  # reset the kernel-event ring
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            batch.run(sync=False)
        self.barrier()
        return self.allmax(time.perf_counter() - t0)

    def timed_fresh_queue(self, queue, sets, n_warm, n_steps, repeats=1):
        """The same query stream through the library's QUEUE (sa_queue_*: a ring of `depth` batches fed by a worker thread of the library):
        step i fetches ticket i - depth (its slot is the one ticket i takes) and submits set i mod len(sets); the worker runs sa_batch_step.
        Same bracketing and the same EXACTLY n_steps per timed region as timed_fresh.  -> ([seconds per region], {set: (scores, docs)})"""
        P = queue.depth
        pending = []
        results = {}
        sets_u32 = [np.ascontiguousarray(q, dtype=np.uint32) for q in sets]

        def drain_one():
            tk, si = pending.pop(0)
            results[si] = queue.fetch(tk)

        def step(i):
            if len(pending) == P:
                drain_one()
            si = i % len(sets)
            pending.append((queue.submit(sets_u32[si]), si))

        for i in range(n_warm):
            step(i)
        while pending:
            drain_one()
        self.barrier()
        dts, i0 = [], n_warm
        for _ in range(max(1, repeats)):
            self.barrier()
            t0 = time.perf_counter()
            for i in range(n_steps):
                step(i0 + i)
            while pending:
                drain_one()
            self.barrier()
            dts.append(self.allmax(time.perf_counter() - t0))
            i0 += n_steps
        return dts, results

    def timed_fresh(self, ring, sets, n_warm, n_steps, repeats=1):
        """The query stream: step i hands batch object i mod P the query set i mod len(sets) -- ONE library call per step
        (sa_batch_step: the set's idf weights gathered from the index's table, sa_batch_reset, sa_batch_run) -- and, P steps
        later, before that batch object is used again, fetches its results: P batches are in flight, each on its own stream.
        n_warm untimed steps, then `repeats` timed regions of EXACTLY n_steps each, every one bracketed by barrier +
        synchronize; max over ranks per region.  -> ([seconds per region], {set: (scores, docs)})"""
        P = len(ring)
        pending = [None] * P
        results = {}
        sets_u32 = [np.ascontiguousarray(q, dtype=np.uint32) for q in sets]

        def drain(b):
            if pending[b] is not None:
                results[pending[b]] = ring[b].fetch()
                pending[b] = None

        def step(i):
            b = i % P
            drain(b)
            si = i % len(sets)
            ring[b].step(sets_u32[si])
            pending[b] = si

        for i in range(n_warm):
            step(i)
        for b in range(P):
            drain(b)
        self.barrier()
        for b in ring:
            b.profile()                                       # reset the kernel-event rings
        dts, i0 = [], n_warm
        for _ in range(max(1, repeats)):
            self.barrier()
            t0 = time.perf_counter()
            for i in range(n_steps):
                step(i0 + i)
            for b in range(P):
                drain(b)
            self.barrier()
            dts.append(self.allmax(time.perf_counter() - t0))
            i0 += n_steps
        return dts, results

    def close(self):
        if self.index is not None:
            if self.use_comm:
                self.index.comm_barrier()
                self.index.comm_destroy()
            self.index.close()


# ------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 5: phrase / slop batches on a resident zipf-1M index (one GPU)
# ------------------------------------------------------------------------------------------------
class PhraseSide:
    """zipf-`docs` index beside the main one + the two phrase workloads of SURVEY 8d: 256 consecutive trigrams
    sampled from random docs (config 3) and 256 two-token slop-2 queries, half on mid-frequency terms (ranks 50-5000:
    config 5's stand-in for MSMARCO), half on ranks 1-50."""

    def __init__(self, api, docs, vocab, device=0):
        from searcharray_amd import synth
        from searcharray_amd.device_index import DeviceIndex
        t0 = time.time()
        lens, terms = synth.zipf_batch_tokens(0, docs, vocab, fast=True)
        words, counts = synth.encode_batch(lens, terms, vocab)
        words, term_off = synth.concat_term_major([(words, counts)], vocab)
        self.docs, self.vocab = docs, vocab
        self.words, self.term_off, self.doc_lens = words, term_off, lens.astype(np.float32)
        self.index = DeviceIndex(words, term_off, self.doc_lens, device=device, api=api)
        self.trigrams = [[int(t) for t in p] for p in synth.phrase_queries_from_tokens(lens, terms, 256, 3, seed=77)]
        rng = np.random.default_rng(5)
        # 256 two-token slop-2 queries: half on mid-frequency terms (ranks 50-5000, SURVEY 8d's stand-in for MSMARCO), half on
        # the most frequent terms (ranks 1-50: lists of 10^5 .. 10^6 words, the ones that load the device)
        self.slop2 = []
        for i in range(256):
            lo, hi = (49, min(5000, vocab - 1)) if i % 2 == 0 else (0, min(50, vocab - 1))
            a, b = (int(x) for x in rng.integers(lo, hi, 2))
            self.slop2.append([a, b + 1 if a == b else b])
        self.pb = self.index.phrase_batch(self.trigrams, k=10)
        self.sb = self.index.phrase_batch(self.slop2, k=10, slop=2)
        self.build_s = time.time() - t0

    def legs(self):
        return [("phrase_batch", self.pb), ("slop_batch", self.sb)]

    def word_bytes(self, phrases):
        return int(sum(8 * int(self.term_off[t + 1] - self.term_off[t]) for p in phrases for t in p))

    def timed(self, batch, n_warm, n_steps):
        for _ in range(n_warm):
            batch.run(sync=False)
        self.index.synchronize()
        batch.profile()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            batch.run(sync=False)
        self.index.synchronize()
        dt = time.perf_counter() - t0
        return dt, batch.profile()[0]

    def check(self, phrases, slop, batch, budget_s):
        """device counts (dense drop-in call) bit-exact vs the REFERENCE's termfreqs for as many phrases as `budget_s`
        allows, the rest vs the oracle's C restatement; device top-10 of every checked phrase vs the CPU scores"""
        from oracle import refimpl as O
        from oracle import ref_loader
        ps, pd_ = batch.fetch()
        orc = O.OracleIndex(self.words, np.arange(self.vocab), self.term_off, self.doc_lens, self.docs)
        sa = ref_loader.reference_array(self.words, self.term_off, self.doc_lens) if ref_loader.available() else None
        ok_counts, ok_top, n_ref, n_orc = True, True, 0, 0
        # slop-2 phrases whose REFERENCE outputs are committed (tests/golden/slop_1m.npz, written offline by tests/golden/make_slop_1m.py
        # from oracle/_ref on this very corpus): compared first, at no CPU cost
        gold, gold_at = None, {}
        gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "slop_1m.npz")
        if slop == 2 and os.path.exists(gpath):
            gold = np.load(gpath)
            if [int(x) for x in gold["meta"]] == [self.docs, self.vocab]:
                gold_at = {int(i): j for j, i in enumerate(gold["b_index"]) if [int(x) for x in gold["b_queries"][j]] == list(phrases[int(i)])}
        for i, j in sorted(gold_at.items()):
            import hashlib
            got = self.index.phrase_freqs_dense(phrases[i], slop=slop)
            sha = np.frombuffer(hashlib.sha1(np.ascontiguousarray(got, dtype=np.float32).tobytes()).digest(), dtype=np.uint8)
            ok_counts &= bool(np.array_equal(sha, gold[f"b{j}_sha1"]))
            ws = gold[f"b{j}_top_scores"]
            n = int((ws > 0).sum())
            ok_top &= bool(np.allclose(ps[i, :n], ws[:n], rtol=1e-5, atol=0))
            n_ref += 1
        t0 = time.perf_counter()
        for i, ph in enumerate(phrases):
            if i in gold_at:
                continue
            if time.perf_counter() - t0 > 2.5 * budget_s and i >= 8:
                break
            got = self.index.phrase_freqs_dense(ph, slop=slop)
            if sa is not None and time.perf_counter() - t0 < budget_s:
                want = sa.termfreqs([f"t{t}" for t in ph], slop=slop)
                want_scores = sa.score([f"t{t}" for t in ph], slop=slop)
                n_ref += 1
            else:
                want = orc.phrase_freqs(ph, slop=slop)
                want_scores = orc.score(ph, slop=slop)
                n_orc += 1
            ok_counts &= bool(np.array_equal(got, want))
            ws, wd = O.topk(np.asarray(want_scores, dtype=np.float32), 10)
            n = int((ws > 0).sum())
            if slop == 0:
                ok_top &= bool(np.array_equal(ps[i, :n], ws[:n])) and bool(np.array_equal(pd_[i, :n], wd[:n]))
            else:                                        # north_star: slop scores within 1e-5 relative
                ok_top &= bool(np.allclose(ps[i, :n], ws[:n], rtol=1e-5, atol=0))
        return {"counts_bit_exact": ok_counts, "top10_matches": ok_top, "phrases_vs_reference": n_ref,
                "of_which_from_committed_reference_outputs": len(gold_at), "phrases_vs_oracle_port": n_orc, "of": len(phrases)}

    def single_queries(self, cpu_s):
        """One phrase per call (the reference's own unit: PosnBitArray.phrase_freqs): device ms (HIP events) of the heaviest
        two-term slop-2 query and of a phrase with a repeated term, each through its one-launch route and through the
        general route; counts of the slop query bit-exact against the C oracle when the CPU budget allows it."""
        from oracle import refimpl as O
        out = {}

        def best_ms(ph, slop, opts):
            from searcharray_amd import options as sa_options
            with sa_options.scoped(**opts):                  # (the route is an option of the call's handle, not of the process)
                ms = []
                for _ in range(4):
                    r = self.index.phrase_freqs_dense(ph, slop=slop)
                    ms.append(self.index.last_profile()[0])
            return round(min(ms[1:]), 4), r

        ms, got = best_ms([0, 1], 2, {})
        ms_g, got_g = best_ms([0, 1], 2, {"span_doc": 0})
        row = {"phrase": "t0 t1", "slop": 2, "device_ms": ms, "general_route_device_ms": ms_g, "matches": int(got.sum()),
               "routes_agree": bool(np.array_equal(got, got_g)),
               "algorithmic_bytes": self.word_bytes([[0, 1]]) + 4 * self.docs}
        row["algorithmic_GBps"] = round(row["algorithmic_bytes"] / ms / 1e6, 1)
        if cpu_s >= 5:
            orc = O.OracleIndex(self.words, np.arange(self.vocab), self.term_off, self.doc_lens, self.docs)
            t0 = time.perf_counter()
            want = orc.phrase_freqs([0, 1], slop=2)
            row["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            row["counts_bit_exact"] = bool(np.array_equal(got, want))
        out["slop_heaviest_two_terms"] = row
        ms, got = best_ms([0, 0, 1], 0, {})
        ms_g, got_g = best_ms([0, 0, 1], 0, {"phrase_docs": 0})
        out["phrase_repeated_term"] = {"phrase": "t0 t0 t1", "device_ms": ms, "general_chain_device_ms": ms_g, "matches": int(got.sum()),
                                       "routes_agree": bool(np.array_equal(got, got_g))}
        return out

    def close(self):
        self.pb.close()
        self.sb.close()
        self.index.close()


def phrase_leg_block(side, name, phrases, slop, batch, pmc, K, cpu_s):
    dt, kms = side.timed(batch, 2, K)
    wb = side.word_bytes(phrases) + 8 * len(phrases) * 10
    d = dominant(pmc.get(name), ("sa_k_",)) if pmc else None
    # the all-kernel sum: a phrase batch has no single dominant kernel on the dense (slop) route
    note = ("algorithmic bytes = sum over phrases of 8 * words of its terms (SURVEY 8d: every word of every term once) + results; "
            "a phrase of the tile route probes through the doc directory and can read LESS than that; kernel_ms = HIP events "
            "around the batch's scoring kernels (all lanes joined), mean over the timed runs")
    blk = roofline_block("sa_k_phrase_tiles (+ merge)" if slop == 0 else "sa_k_span_* per phrase + sa_k_dense_topk_tiles (+ merge)",
                         kms, wb, wb, d, note)
    # (for a phrase batch the byte model IS the compulsory one -- every word of every phrase term once: wb is passed as both --
    #  so `achieved` / `frac` are algorithmic bytes / kernel time as the bench contract asks; the doc directory lets the tile
    #  route read LESS than that, which the counter traffic beside it shows)
    blk["frac_basis"] = "algorithmic_bytes (= compulsory for a phrase batch)"
    out = {"value": round(len(phrases) * K / dt, 1), "unit": "phrases/s", "steps": K, "ms_per_step": round(dt / K * 1e3, 4),
           "workload": (f"zipf-{side.docs}: {len(phrases)} consecutive trigrams sampled from random docs -> BM25 -> top-10 (one resident phrase batch)"
                        if slop == 0 else
                        f"zipf-{side.docs}: {len(phrases)} two-token slop-{slop} phrases, half on terms of ranks 50-5000, half on ranks 1-50 -> BM25 -> top-10 (one resident phrase batch)"),
           "roofline": blk}
    if cpu_s > 0:
        out["parity"] = side.check(phrases, slop, batch, cpu_s)
    return out


def compulsory_bytes(df, queries, B, k):
    """What one launch cannot avoid moving: the posting stream (8 bytes per posting) of every DISTINCT term
    of the batch once, plus the result keys."""
    terms = np.unique(np.asarray(queries).reshape(-1))
    return int(8 * int(df[terms].astype(np.int64).sum()) + 8 * B * k)


def roofline_block(kernel, kernel_ms, alg_bytes, compulsory, pmc, note):
    """bound / achieved / peak / unit / frac / traffic per the bench contract.  The headline `frac` is what the design
    cannot avoid moving over the time it takes -- a fraction of the HBM peak that cannot exceed 1 and that wasted traffic
    cannot raise:
      compulsory_bytes   distinct posting lists of the batch once + outputs
      achieved / frac    compulsory_bytes / kernel time (GB/s; / the 8 TB/s peak)
      traffic            HBM bytes per launch from the PMC counters (FETCH_SIZE x the calibration of the kernel's load
                         width + WRITE_SIZE), or null; traffic_GBps / traffic_frac = the same over the kernel time;
                         wasted = traffic / compulsory_bytes (re-reads across the XCDs' L2s)
      logical_*          SURVEY 8d's per-query bytes (sum_t 8 df_t + 4 n_docs, every query counted in full) over the
                         kernel time: a LOGICAL rate -- it counts bytes that the shared first terms and the caches serve --
                         not a bandwidth and not a fraction of anything
    """
    sec = kernel_ms * 1e-3
    gbps = (lambda x: round(x / sec / 1e9, 1) if sec > 0 else 0.0)
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
         "compulsory_bytes": int(compulsory),
         "achieved": gbps(compulsory), "frac": round(gbps(compulsory) / HBM_PEAK_GBS, 4), "frac_basis": "compulsory_bytes",
         "logical_bytes_per_launch": int(alg_bytes), "logical_GBps": gbps(alg_bytes)}
    if pmc and pmc.get("hbm_bytes"):
        hb = pmc["hbm_bytes"]
        r.update({"traffic": int(hb), "traffic_source": pmc["source"], "traffic_GBps": gbps(hb),
                  "traffic_frac": round(gbps(hb) / HBM_PEAK_GBS, 4),
                  "wasted": round(hb / compulsory, 3) if compulsory else None,
                  "l2_hit_rate": pmc.get("l2_hit_rate"), "fetch_calibration": pmc.get("fetch_calibration"),
                  "pmc_kernel_ms": pmc.get("kernel_ms")})
    else:
        r.update({"traffic": None, "traffic_source": None})
    r["note"] = note
    return r


# ------------------------------------------------------------------------------------------------
# PMC child runs: bench.py re-invokes itself under rocprofv3, twice (FETCH_SIZE costs 3 of the 4 TCC
# slots: MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel trace only
# ------------------------------------------------------------------------------------------------
PMC_PASSES = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])]
PMC_STEPS = 3


def pmc_child(r, legs):
    """legs: list of (name, batch, sparse_env).  Dispatch order written for the parent's parser:
    calibration stream, then per leg: marker, warm-up run, marker, PMC_STEPS counted runs; a last marker."""
    import ctypes
    g = ctypes.c_double(0)
    r.api.call("sa_stream_probe", CALIB_BYTES, 1, 1, ctypes.byref(g))
    # the same byte count read with 8-byte loads (the width of the scoring kernels' posting loads): the marker kernel with a
    # big argument, right in front of the first marker (consecutive markers count as one)
    r.api.call("sa_stream_probe", CALIB_BYTES, 0, 1, ctypes.byref(g))
    for name, batch, sparse in legs:
        if sparse is not None:
            batch.set_options(sparse=int(sparse))
        r.api.call("sa_stream_probe", 1 << 20, 0, 1, ctypes.byref(g))     # marker: the leg's untimed warm-up run follows
        batch.run(sync=True)
        r.api.call("sa_stream_probe", 1 << 20, 0, 1, ctypes.byref(g))     # marker: the leg's PMC_STEPS counted runs follow
        for _ in range(PMC_STEPS):
            batch.run(sync=True)
    r.api.call("sa_stream_probe", 1 << 20, 0, 1, ctypes.byref(g))


def run_pmc_children(args, leg_names, corpus):
    """-> {leg: {"hbm_bytes", "l2_hit_rate", "kernels": {...}, ...}} or {} (no rocprofv3 / failure)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {}, "rocprofv3 not found"
    if not os.path.exists("/dev/kfd"):
        return {}, "no GPU"
    out_root = tempfile.mkdtemp(prefix="sa_pmc_")
    per_pass = {}
    cache = args.corpus_cache
    if not cache:                                             # hand the generated corpus to the children
        cache = os.path.join(out_root, "corpus")
        os.makedirs(cache)
        np.savez(os.path.join(cache, f"zipf_{args.docs}_{args.vocab}_0_{args.docs}.npz"), words=corpus.words,
                 term_off=corpus.term_off, doc_lens=corpus.doc_lens)
    try:
        for pname, counters in PMC_PASSES:
            d = os.path.join(out_root, pname)
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", pname, "--docs", str(args.docs),
                   "--vocab", str(args.vocab), "--queries", str(args.queries), "--k", str(args.k), "--tile", str(args.tile),
                   "--no-cpu-baseline", "--no-pmc", "--corpus-cache", cache, "--phrase-docs", str(args.phrase_docs)]
            if args.no_phrase_legs:
                cmd.append("--no-phrase-legs")
            env = dict(os.environ, TMPDIR="/tmp")
            for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(v, None)
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=420)
            except subprocess.TimeoutExpired:
                return {}, f"rocprofv3 pass '{pname}' timed out"
            if p.returncode != 0:
                err = p.stderr.decode(errors="replace")
                keep = os.path.join(ROOT, "gpurun_out")
                if os.path.isdir(keep):                       # (scratch dir of the GPU box: keep the evidence)
                    open(os.path.join(keep, f"pmc_child_{pname}.stderr"), "w").write(err)
                lines = [ln for ln in err.splitlines() if ln.strip() and not ln.lstrip().startswith("@")]
                return {}, f"rocprofv3 pass '{pname}' failed rc={p.returncode}: {' | '.join(lines[-6:])[-600:]}"
            files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getsize)
            if not files:
                return {}, f"rocprofv3 pass '{pname}' wrote no counter_collection.csv"
            per_pass[pname] = parse_pmc_csv(files[-1], leg_names)
    finally:
        shutil.rmtree(out_root, ignore_errors=True)
    return merge_pmc(per_pass, leg_names), None


def parse_pmc_csv(path, leg_names):
    """-> {"calib": {counter: mean}, leg: {kernel: {counter: [values]}, "_dur": {kernel: [ns]}}}"""
    rows = {}
    for r in csv.DictReader(open(path)):
        key = int(r["Dispatch_Id"])
        e = rows.setdefault(key, {"kernel": r["Kernel_Name"], "c": {}, "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = [rows[k] for k in sorted(rows)]
    out = {"calib": {}}
    seg, in_marker = -1, False
    calib = [e for e in seq if e["kernel"].startswith(CALIB)]
    if calib:
        for cname in calib[-1]["c"]:
            out["calib"][cname] = calib[-1]["c"][cname]           # the second (warm) dispatch
    # 8-byte-load calibration: the marker kernel's dispatches over CALIB_BYTES (the plain markers read 1 MiB)
    # (recognised by their duration: a 1 GiB read takes >= 0.1 ms, a 1 MiB marker a few microseconds -- the pass that does
    #  not collect FETCH_SIZE must skip them too)
    calib8 = [e for e in seq if e["kernel"].startswith(MARKER) and e["dur"] > 60_000]
    out["calib8"] = dict(calib8[-1]["c"]) if calib8 else {}
    started = False
    big = [id(e) for e in calib8]
    for e in seq:
        name = e["kernel"]
        if id(e) in big:                                          # (the 8-byte calibration reads: not markers)
            continue
        if name.startswith(MARKER):
            if not in_marker:
                seg += 1
            in_marker, started = True, True
            continue
        in_marker = False
        # segments alternate: warm-up run of leg i (even), counted runs of leg i (odd)
        if not started or seg < 0 or seg % 2 == 0 or seg // 2 >= len(leg_names):
            continue
        leg = out.setdefault(leg_names[seg // 2], {"_dur": {}})
        short = name.split("(")[0].replace("void ", "")
        for cname, v in e["c"].items():
            leg.setdefault(short, {}).setdefault(cname, []).append(v)
        leg["_dur"].setdefault(short, []).append(e["dur"])
    return out


def merge_pmc(per_pass, leg_names):
    fetch, write = per_pass.get("fetch", {}), per_pass.get("write", {})
    # FETCH_SIZE is in KiB and under-reports wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM
    # section: exactly 1/2 for 16 B per lane): calibrated here on a known 1 GiB read with 16-byte loads
    raw = fetch.get("calib", {}).get("FETCH_SIZE")
    factor = (CALIB_BYTES / (raw * 1024.0)) if raw else 2.0
    if not (1.0 <= factor <= 4.0):
        factor = 2.0
    # ... and on the same read with 8-byte loads: the width of the posting loads of the grouped / head-group kernels (their
    # dense base rows are 16-byte loads: the smaller part of their traffic)
    raw8 = fetch.get("calib8", {}).get("FETCH_SIZE")
    factor8 = (CALIB_BYTES / (raw8 * 1024.0)) if raw8 else factor
    if not (0.5 <= factor8 <= 4.0):
        factor8 = factor
    res = {}
    for leg in leg_names:
        f, w = fetch.get(leg, {}), write.get(leg, {})
        kernels = {}
        for kname in sorted(set(list(f) + list(w)) - {"_dur"}):
            fv = f.get(kname, {}).get("FETCH_SIZE", [])
            wv = w.get(kname, {}).get("WRITE_SIZE", [])
            hit, miss = sum(w.get(kname, {}).get("TCC_HIT_sum", [])), sum(w.get(kname, {}).get("TCC_MISS_sum", []))
            n = max(len(fv), len(wv), 1)
            per_step = n / PMC_STEPS                               # dispatches of this kernel per step
            kf = factor8 if kname.startswith(LOADS_8B) else factor
            fb = (sum(fv) / len(fv) * 1024.0 * kf) if fv else 0.0
            wb = (sum(wv) / len(wv) * 1024.0) if wv else 0.0
            durs = f.get("_dur", {}).get(kname, [])
            kernels[kname] = {"hbm_bytes_per_dispatch": int(fb + wb), "dispatches_per_step": round(per_step, 2),
                              "fetch_KiB_raw": round(sum(fv) / len(fv), 1) if fv else None, "fetch_factor": round(kf, 3),
                              "write_KiB_raw": round(sum(wv) / len(wv), 1) if wv else None,
                              "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss > 0 else None,
                              "ms_under_pmc": round(sum(durs) / len(durs) / 1e6, 4) if durs else None}
        res[leg] = {"kernels": kernels, "fetch_calibration": {"16B_loads": round(factor, 3), "8B_loads": round(factor8, 3)},
                    "source": "rocprofv3 --pmc child runs of this bench.py invocation (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum "
                              "TCC_MISS_sum, kernel trace only); FETCH_SIZE x a calibration measured on a 1 GiB stream read with "
                              "the kernel's load width (8-byte loads: grouped / head-group kernels; 16-byte loads: the others)"}
    return res


def dominant(pmc_leg, prefixes):
    """sum over the kernels whose name starts with one of `prefixes` (per step), and the hit rate / time of the largest"""
    if not pmc_leg:
        return None
    tot, best, best_b = 0.0, None, -1
    for kname, kv in pmc_leg["kernels"].items():
        if any(kname.startswith(p) for p in prefixes):
            b = kv["hbm_bytes_per_dispatch"] * kv["dispatches_per_step"]
            tot += b
            if b > best_b:
                best, best_b = kv, b
    if best is None:
        return None
    return {"hbm_bytes": int(tot), "l2_hit_rate": best["l2_hit_rate"], "kernel_ms": best["ms_under_pmc"],
            "source": pmc_leg["source"], "fetch_calibration": pmc_leg["fetch_calibration"]}


# ------------------------------------------------------------------------------------------------
# CPU baseline: the REFERENCE itself (oracle/_ref) when built, else the oracle port
# ------------------------------------------------------------------------------------------------
def cpu_baseline(r, queries, scores, docs, extra=()):
    """extra: [(label, queries, scores, docs, rows)] -- rotated query sets whose device results are checked against the
    reference on a few rows each (untimed)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import refimpl as O                    # CPU restatement: the checker (and the fallback baseline)
    from oracle import ref_loader
    a = r.args
    D, V, B, k = a.docs, a.vocab, len(queries), a.k
    corpus = r.corpus
    budget = a.cpu_seconds
    n_cores = os.cpu_count() or 1
    kind = "reference" if ref_loader.available() else "port"
    if kind == "reference":
        sa = ref_loader.reference_array(corpus.words, corpus.term_off, corpus.doc_lens, avg_doc_length=r.avgdl)
        names = [[f"t{int(t)}" for t in q] for q in queries]

        def dense(qi):                                  # the caller idiom of test/test_msmarco.py:353-354
            return np.sum([sa.score(tok) for tok in names[qi]], axis=0)

        def clear():
            sa.posns.clear_cache()
        what = "the reference itself (oracle/_ref: its Python + Cython built by its own setup.py), SearchArray.score per term + np.sum"
    else:
        orc = O.OracleIndex(corpus.words, np.arange(V), corpus.term_off, corpus.doc_lens, D)
        orc.avg_doc_length = r.avgdl

        def dense(qi):
            return orc.score_terms_sum([int(t) for t in queries[qi]])

        def clear():
            orc.clear_cache() if hasattr(orc, "clear_cache") else None
        what = "oracle/ C+numpy port of score()+np.sum (oracle/_ref not built)"

    def topk_ref(s):                                     # reference utils/sort.py:24
        return np.argpartition(s, -k)[-k:]

    # -- cold: caches cleared before every query (test_msmarco.py:362-379), bounded sample
    t0 = time.perf_counter()
    n_cold = 0
    for qi in range(B):
        clear()
        topk_ref(dense(qi))
        n_cold += 1
        if time.perf_counter() - t0 > budget * 0.4 and n_cold >= 4:
            break
    cold_dt = time.perf_counter() - t0
    # -- warm, one thread: one pass to fill the tf/df caches (test_msmarco.py:384-395), then timed
    n_warm = B
    tw = time.perf_counter()
    for qi in range(B):
        dense(qi)
        if time.perf_counter() - tw > budget and qi + 1 >= 8:
            n_warm = qi + 1
            break
    ok = True
    t0 = time.perf_counter()
    for qi in range(n_warm):
        s = dense(qi)
        topk_ref(s)
        ws, wd = O.topk(s, k)                            # parity of the GPU result against the CPU result: BIT-exact
        ok &= bool(np.array_equal(scores[qi], ws)) and bool(np.array_equal(docs[qi][ws > 0], wd[ws > 0]))
    warm_dt = time.perf_counter() - t0
    n_extra = 0
    for label, q2, s2, d2, rows in extra:                # rotated sets of the fresh-batch leg
        for qi in rows:
            if kind == "reference":
                dv = np.sum([sa.score(f"t{int(t)}") for t in q2[qi]], axis=0)
            else:
                dv = orc.score_terms_sum([int(t) for t in q2[qi]])
            ws, wd = O.topk(dv, k)
            good = bool(np.array_equal(s2[qi], ws)) and bool(np.array_equal(d2[qi][ws > 0], wd[ws > 0]))
            if not good:
                log(0, f"parity MISMATCH: {label} query {qi}")
            ok &= good
            n_extra += 1
    # (the deterministic top-k used for the parity check is timed too; it is a few % of a query)
    out = {"value": round(n_warm / warm_dt, 3), "unit": "queries/s", "cores": 1, "kind": kind,
           "sample": f"queries 0..{n_warm - 1} of the {B} on the same {D}-doc corpus, tf/df caches warm (one untimed pass first), "
                     f"single thread: {what} + np.argpartition top-{k}; host has {n_cores} hardware threads",
           "cold": {"value": round(n_cold / cold_dt, 3), "unit": "queries/s", "cores": 1,
                    "sample": f"first {n_cold} queries, posns.clear_cache() before each (tf/df recomputed from the roaringish words)"}}
    # -- warm, thread pool (test_msmarco.py:483-507 runs its queries through a ThreadPoolExecutor)
    n_thr = int(max(2, min(n_cores, 64)))                # each in-flight query holds ~5 dense float32[D] vectors
    qs = [i % n_warm for i in range(max(n_warm, 2 * n_thr))]

    def one(qi):
        return int(topk_ref(dense(qi))[0])
    with ThreadPoolExecutor(n_thr) as ex:
        t0 = time.perf_counter()
        list(ex.map(one, qs))
        mt_dt = time.perf_counter() - t0
    out["threaded"] = {"value": round(len(qs) / mt_dt, 3), "unit": "queries/s", "cores": n_thr,
                       "sample": f"{len(qs)} warm queries through ThreadPoolExecutor({n_thr}) "
                                 f"(capped at 64 of the {n_cores} hardware threads: every in-flight query holds several float32[{D}] vectors)"}
    verdict = (f"ok (bit-exact scores and docs: {n_warm} queries of set 0 + {n_extra} of the rotated sets vs the {kind})"
               if ok else "MISMATCH")
    return out, verdict, n_warm


def sharded_parity(r, queries, scores, docs, n_check=8):
    """N > 1: every rank scores the first n_check queries on ITS doc range with the CPU oracle (global df / avgdl /
    corpus size, the reference's op order), the per-shard top-k lists travel through the library's all-reduce (one slot
    per rank in a zero array, summed), and rank 0 merges them -- score descending, doc id ascending -- and compares with
    the device result.  Collective: every rank calls it."""
    a, c = r.args, r.corpus
    k, D, V = a.k, a.docs, a.vocab
    n_local = r.hi - r.lo
    Q = min(n_check, len(queries))
    # one extra element per rank says "my part is there": a rank whose oracle cannot run (library not built on that box)
    # must still take part in the all-reduce, or the others would wait for it forever
    slots = np.zeros((r.world, Q * k * 2 + 1), dtype=np.float64)
    err = ""
    try:
        from oracle import refimpl as O                # the checker
        orc = O.OracleIndex(c.words, np.arange(V), c.term_off, c.doc_lens, n_local)
        mine = slots[r.rank, :Q * k * 2].reshape(Q, k, 2)
        for qi in range(Q):
            vecs = [O.bm25(orc.termfreqs(int(t)).copy(), np.asarray([r.df[int(t)]]), c.doc_lens, r.avgdl, D) for t in queries[qi]]
            ws, wd = O.topk(np.sum(vecs, axis=0), k)
            mine[qi, :len(ws), 0] = ws
            mine[qi, :len(ws), 1] = wd.astype(np.float64) + r.lo
        slots[r.rank, -1] = 1.0
    except Exception as e:                             # noqa: BLE001 -- reported in the JSON line, never fatal for the bench
        err = f"{type(e).__name__}: {e}"
        slots[r.rank, :] = 0.0
    if r.use_comm:
        slots = r.index.comm_allreduce(slots.reshape(-1), "sum").reshape(r.world, Q * k * 2 + 1)
    if r.rank != 0:
        return "n/a"
    if not (slots[:, -1] == 1.0).all():
        return f"skipped (the CPU oracle did not run on every rank{': ' + err if err else ''})"
    lists = slots[:, :Q * k * 2].reshape(r.world, Q, k, 2)
    ok = True
    for qi in range(Q):
        cs, cd = lists[:, qi, :, 0].reshape(-1), lists[:, qi, :, 1].reshape(-1)
        order = np.lexsort((cd, -cs))[:k]
        ws, wd = cs[order].astype(np.float32), cd[order].astype(np.uint64)
        ok &= bool(np.array_equal(scores[qi], ws)) and bool(np.array_equal(docs[qi][ws > 0], wd[ws > 0]))
    return f"ok (bit-exact, {Q} queries, per-shard CPU oracle merged over {r.world} rank(s))" if ok else "MISMATCH"


# ------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1 and not args.pmc_child:
        spawn_ranks(args.gpus)
    r = Rank(args)
    rank, world = r.rank, r.world
    from searcharray_amd import synth
    D, V, Bq, K, W = args.docs, args.vocab, args.queries, args.steps, args.warmup
    # N > 1: every rank scores EVERY query of a batch on its 1/N of the docs.  The main region keeps --queries per step at
    # every N (the same stream as N = 1: value(N) / value(1) is a like-for-like speed-up), and a WIDE batch (--scaled-queries,
    # 2048) is timed beside it at every N too -- N = 1 included -- because a rank's device work per step shrinks with N while
    # the host's cost per batch does not: wide_batch(N) / wide_batch(1) is the same-batch speed-up of a deployment that
    # batches wider.  Nothing is ever divided across batch sizes.  (--weak: N x --queries in the main region.)
    mult = max(world, args.batch_mult)                      # (--batch-mult: the N > 1 batch logic on one GPU, for testing)
    weak = mult > 1 and args.weak
    B = Bq * mult if weak else Bq
    r.generate()
    phrase_legs_on = world == 1 and not r.use_comm and not args.no_phrase_legs
    leg_names = ["main", "exhaustive_overlay", "dynamic_pruning"] + (["distinct_terms"] if 4 * B <= V else []) + \
                (["phrase_batch", "slop_batch"] if phrase_legs_on else [])
    # HBM traffic / L2 hit rate: bench.py profiles ITSELF under rocprofv3 --pmc, in child processes that run
    # BEFORE this process touches the GPU (one process on the device at a time, as in a stand-alone rocprofv3
    # run); the children read the corpus this process just generated instead of generating it again.
    pmc, pmc_err = {}, "skipped"
    if rank == 0 and world == 1 and not args.no_pmc and not args.pmc_child:
        t0 = time.time()
        pmc, pmc_err = run_pmc_children(args, leg_names, r.corpus)
        log(rank, f"PMC child runs: {'ok' if pmc else pmc_err} ({time.time()-t0:.0f}s)")
    r.build()
    n_sets = max(2, args.query_sets)

    def query_set(i, n):
        """n queries = n / --queries seeded BASELINE-shaped blocks; block 0 of set 0 is THE BASELINE set (seed 42)"""
        blocks = [synth.bm25_queries(Bq, vocab=V) if (i == 0 and j == 0) else synth.bm25_queries(Bq, vocab=V, seed=1000 + 131 * i + j)
                  for j in range(max(1, n // Bq))]
        return np.concatenate(blocks)[:n]
    sets = [query_set(i, B) for i in range(n_sets)]
    queries = sets[0]
    batch = r.make_batch(queries)
    q_distinct = synth.bm25_queries_distinct(B, vocab=V) if 4 * B <= V else None
    batch_d = r.make_batch(q_distinct) if q_distinct is not None else None
    side = None
    if phrase_legs_on:
        side = PhraseSide(r.api, args.phrase_docs, V, device=r.local_rank)
        log(rank, f"phrase side index: zipf-{args.phrase_docs} resident, {len(side.trigrams)} trigrams + {len(side.slop2)} slop-2 queries "
                  f"({side.build_s:.1f}s)")

    if args.pmc_child:
        legs = [("main", batch, None), ("exhaustive_overlay", batch, "0"), ("dynamic_pruning", batch, "1")]
        if batch_d is not None:
            legs.append(("distinct_terms", batch_d, None))
        if side is not None:
            legs += [(name, b, None) for name, b in side.legs()]
        pmc_child(r, legs)
        batch.close()
        if batch_d is not None:
            batch_d.close()
        if side is not None:
            side.close()
        r.close()
        return

    # The main region times the library's DEFAULT route for the batch shape -- since round 6 the staged-tile route
    # (csrc/sa_stage.hip: the batch's distinct posting lists staged in LDS once per tile, candidates from the essential
    # terms, exact scores in query-term order; results identical to scoring every posting) -- on FRESH batches: 8 seeded
    # query sets rotating through the batch objects, reset + run + fetch per step.  Timed right after on the resident
    # set 0, for comparison: the grouped overlay kernel that scores every posting of every query term (option sparse = 0;
    # rounds 2-5's headline) and dynamic pruning (sparse = 1).  --pruned makes dynamic pruning the main region.
    exhaustive = not args.pruned
    from searcharray_amd import options as sa_options
    route = sa_options.Scope()                              # the scoring route of the legs below: an OPTION of the batches (thread-scoped here)
    for kv in args.opt:
        route.set(kv.split("=")[0], int(kv.split("=")[1]))
    if args.pruned:
        route.set(sparse=1)
    P = max(1, args.pipeline if B <= 1024 else min(args.pipeline, 4))
    pair = [r.make_batch(sets[i % len(sets)], check=(i == 0)) for i in range(P)]
    R = max(1, args.repeats)
    if args.driver == "queue":
        qq = r.index.queue(B, sets[0].shape[1], k=args.k, depth=P)
        dts, fresh_results = r.timed_fresh_queue(qq, sets, max(W, P), K, repeats=R)
        qq.close()
    else:
        dts, fresh_results = r.timed_fresh(pair, sets, max(W, P), K, repeats=R)
    dt = float(np.median(dts))
    # (HIP events around one batch's scoring kernels: with P batches in flight on P streams they overlap the other
    #  batches' kernels, so this is a batch's latency share, not the device time per step -- the replay leg's is)
    kernel_ms_fresh = float(np.mean([b.profile()[0] for b in pair])) if args.driver == "ring" else float("nan")     # (the queue's batches are its own)
    scores, docs = fresh_results[0] if 0 in fresh_results else (None, None)

    # replay of the resident set 0 (rounds 1-2 reported this as `value`)
    dt_r = r.timed(batch, max(W, 1), K)
    kernel_ms_r, alg_bytes, post_bytes = batch.profile()
    kernel_ms = kernel_ms_r
    post_total = r.allsum(float(post_bytes))
    scores_r, docs_r = batch.fetch()
    if scores is None:                                   # (fewer timed + warm steps than sets: cannot happen with the defaults)
        scores, docs = scores_r, docs_r
    fresh_equals_replay = bool(np.array_equal(scores, scores_r) and np.array_equal(docs, docs_r))

    # the WIDE batch as a fresh stream too, at every N (and, with --weak, the fixed batch beside the scaled one)
    scaled = None
    B2 = Bq if weak else args.scaled_queries
    if B2 and B2 != B:
        sets2 = [query_set(20 + i, B2) for i in range(4)]
        # (a batch's candidate lists: B x min(tiles x k, 2^20) keys of 8 bytes -- 16 GiB at 2048 queries and k = 1000; the batches
        #  in flight of this leg stay within 40 GiB)
        inf = r.index.info()
        per_batch = B2 * min(max(int(inf.n_docs) // 2048, 1) * args.k, 1 << 20) * 8
        n_ring2 = min(args.pipeline, 4) if B2 > 1024 else max(1, args.pipeline)
        n_ring2 = max(1, min(n_ring2, int((40 << 30) // max(per_batch, 1))))
        ring2 = [r.make_batch(sets2[i % len(sets2)], check=False) for i in range(n_ring2)]
        Ks = K if B2 <= B else max(4, K // 4)
        dts2, _ = r.timed_fresh(ring2, sets2, len(ring2), Ks, repeats=3)
        d2 = float(np.median(dts2))
        scaled = {"value": round(B2 * Ks / d2, 2), "unit": "queries/s", "queries_per_step": B2, "steps": Ks,
                  "ms_per_step": round(d2 / Ks * 1e3, 4), "ms_per_step_min_max": [round(min(dts2) / Ks * 1e3, 4), round(max(dts2) / Ks * 1e3, 4)],
                  "batches_in_flight": len(ring2), "n_gpus": world,
                  "note": "the same fresh-batch stream (rotating sets, one sa_batch_step + fetch per step) with the other batch size, "
                          "same index; compare with the SAME leg of the N = 1 line, never with `value`"}
        for b in ring2:
            b.close()

    main_route = batch.last_route()
    K2 = max(3, min(K, 10))
    # the grouped overlay kernel: every posting of every query term scored (rounds 2-5's headline)
    route.set(sparse=0)
    dt_o = r.timed(batch, 2, K2)
    kernel_ms_o, _, _ = batch.profile()
    scores_o, docs_o = batch.fetch()
    same_o = bool(np.array_equal(scores_r, scores_o) and np.array_equal(docs_r, docs_o))
    gi_overlay = batch.group_info()
    # dynamic pruning
    route.set(sparse=1)
    dt2 = r.timed(batch, 2, K2)
    kernel_ms2, _, _ = batch.profile()
    scores2, docs2 = batch.fetch()
    same = bool(np.array_equal(scores_r, scores2) and np.array_equal(docs_r, docs2))

    dt3 = kernel_ms3 = alg3 = None
    distinct_route = None
    if batch_d is not None:
        route.unset("sparse")
        dt3 = r.timed(batch_d, 2, K2)
        kernel_ms3, alg3, post3 = batch_d.profile()
        batch_d.fetch()
        distinct_route = batch_d.last_route()
    if args.pruned:
        route.set(sparse=1)
    else:
        route.unset("sparse")

    qps = B * K / dt

    cpu, parity = None, "skipped"
    if rank == 0 and world == 1 and not r.use_comm and not args.no_cpu_baseline:
        extra = [(f"set {si}", sets[si], fresh_results[si][0], fresh_results[si][1], [1, B // 2, B - 1])
                 for si in sorted(fresh_results) if si != 0]
        cpu, parity, _ = cpu_baseline(r, queries, scores, docs, extra)
    elif r.use_comm:
        parity = sharded_parity(r, queries, scores, docs)            # (collective)

    # the drop-in call itself: SearchArray.score's device call for ONE term with the dense float32[n_docs] result copied to
    # the host (what the reference returns) -- PCIe-bound -- and the same with the result left in HBM (score_device)
    dense_out = None
    if rank == 0 and world == 1 and not r.use_comm:
        from searcharray_amd.device_index import DeviceVec
        terms = [int(t) for t in queries[0]]
        idf1 = r.idf_of(np.asarray([terms]))[0]
        nd = 20
        r.index.bm25_dense(terms[:1], idf=idf1[:1])
        t0 = time.perf_counter()
        for i in range(nd):
            keep = r.index.bm25_dense([terms[i % 4]], idf=idf1[i % 4: i % 4 + 1])
        dt_host = (time.perf_counter() - t0) / nd
        del keep
        vec = DeviceVec(r.api, r.index.n_docs, False)
        from searcharray_amd._lib import p_u32, p_f32
        tarr = [np.asarray([t], dtype=np.uint32) for t in terms]
        warr = [np.asarray([w], dtype=np.float32) for w in idf1]
        r.index.into_vec(vec, None, "sa_index_bm25_dense", p_u32(tarr[0]), p_f32(warr[0]), 1, np.float32(1.2), np.float32(0.75))
        r.index.synchronize()
        t0 = time.perf_counter()
        for i in range(nd):
            r.index.into_vec(vec, None, "sa_index_bm25_dense", p_u32(tarr[i % 4]), p_f32(warr[i % 4]), 1, np.float32(1.2), np.float32(0.75))
        r.index.synchronize()
        dt_dev = (time.perf_counter() - t0) / nd
        vec.close()
        dfs = [int(r.df[t]) for t in terms]                 # (docs holding the term = postings of the impact stream)
        alg = sum(8 * x for x in dfs) / len(dfs) + 4 * D       # every posting of the term once + the float32[n_docs] result
        dense_out = {"to_host_ms_per_call": round(dt_host * 1e3, 4), "to_host_GBps": round(4 * D / dt_host / 1e9, 1),
                     "device_resident_ms_per_call": round(dt_dev * 1e3, 4), "calls": nd,
                     "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "sa_k_bm25_dense_direct (one launch per call)",
                                  "algorithmic_bytes_per_call": int(alg), "achieved": round(alg / dt_dev / 1e9, 1),
                                  "frac": round(alg / dt_dev / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                  "note": f"bytes = 8 x df + 4 x n_docs, mean over the 4 terms of the calls (df {dfs}); time = wall clock of {nd} back-to-back "
                                          "asynchronous calls / their number (launch gaps included), the result left in a device vector"},
                     "note": f"single-term BM25 over {D} docs: sa_index_bm25_dense with the float32[{D}] result copied into a page-locked "
                             "host buffer (the drop-in SearchArray.score: PCIe-bound) vs left in a device vector (SearchArray.score_device)"}

    phrase_out = {}
    if side is not None and rank == 0:
        route.unset("sparse")
        cpu_s = 0.0 if args.no_cpu_baseline else 8.0
        phrase_out["phrase_batch"] = phrase_leg_block(side, "phrase_batch", side.trigrams, 0, side.pb, pmc, K2, cpu_s)
        phrase_out["slop_batch"] = phrase_leg_block(side, "slop_batch", side.slop2, 2, side.sb, pmc, K2, cpu_s)
        phrase_out["single_phrase_queries"] = side.single_queries(cpu_s)

    if cpu is not None:
        cpu["parity"] = parity                               # (inside cpu_baseline too: the checker that timed it is the one that compared)
    if rank == 0:
        from searcharray_amd.device_index import DeviceIndex
        try:
            comm_lib = DeviceIndex.comm_library_info(r.api)
        except Exception as e:                               # noqa: BLE001
            comm_lib = (None, f"{type(e).__name__}: {e}")
        n_tiles = int(r.info.n_tiles)
        comp = compulsory_bytes(r.df if world == 1 else r.index.docfreqs(), queries, B, args.k)
        main_note = ("the library's default route for this batch shape -- " + main_route + " -- exact top-k, results identical to scoring every "
                     "posting (fresh_equals_replay, same_results of the other legs, parity_check).  staged: the posting lists of the terms that "
                     "can be essential for a query are streamed from HBM once per batch and staged in LDS tile by tile; a term that cannot be "
                     "essential for any query of the batch is NOT streamed: it enters the bounds through its per-tile block maximum, its "
                     "presence bitmap is staged, and the few documents that pass the bound test read its factor from a probe row.  "
                     "compulsory_bytes counts EVERY distinct list of the batch once (the exhaustive design's unavoidable bytes), so achieved / "
                     "frac = that over the kernel time is the like-for-like figure of rounds 2-5; `traffic` is what actually moved.  "
                     "kernel_ms = HIP events on the batch's stream around the scoring kernel(s) of a step, mean over the steps of the "
                     "REPLAY leg (one batch alone on the device: in the fresh-batch leg the batches in flight overlap, so per-batch event "
                     "times are latencies, reported as fresh_batch_latency_ms); rank 0's shard")
        exh_note = ("every posting of every query term scored (reference behaviour; option sparse = 0: rounds 2-5's headline route); "
                    "kernel_ms = HIP events around the scoring kernels of a step of the replayed set 0")
        prn_note = ("dynamic pruning: postings of non-essential terms are never read (by design traffic < compulsory_bytes of the "
                    "exhaustive leg is possible); byte model = the posting lists the routing keeps ESSENTIAL plus probes, so the "
                    "bound is gather latency / sector traffic, reported as traffic-based GB/s; results identical (same_results)")
        main_block = roofline_block("sa_k_bm25_stage (+ sa_k_topk_merge)" if main_route == "staged" else "sa_k_bm25_* / sa_k_sparse_* (scoring kernels of one step)",
                                    kernel_ms, alg_bytes, comp, dominant(pmc.get("main"), ("sa_k_bm25", "sa_k_sparse")), main_note)
        main_block["route"] = main_route
        exh_block = roofline_block("sa_k_bm25_group_tiles (+ work list, per-query kernel)", kernel_ms_o, alg_bytes, comp,
                                   dominant(pmc.get("exhaustive_overlay"), ("sa_k_bm25",)), exh_note)
        prn_block = roofline_block("sa_k_sparse_lead + route + scan + rest + score (+ sa_k_bm25_tiles_list)", kernel_ms2, alg_bytes, comp,
                                   dominant(pmc.get("dynamic_pruning"), ("sa_k_sparse", "sa_k_bm25")), prn_note)
        exh_block["workgroups_per_launch"] = B * n_tiles
        # how the exhaustive path grouped the batch; the grouped kernel runs one WAVE per (tile, group) item
        gi = gi_overlay
        exh_block["grouping"] = dict(gi, items_per_launch=(gi["groups"] + gi["per_query_kernel"]) * n_tiles,
                                     ns_per_pair=round(kernel_ms_o * 1e6 / max(1, B * n_tiles), 3) if kernel_ms_o else None,
                                     note="items: one wave per (tile, group) + one workgroup per (tile, ungrouped query); "
                                          "ns_per_pair = scoring-kernel time / (tile, query) pairs")
        overlay = {"value": round(B * K2 / dt_o, 2), "unit": "queries/s", "steps": K2, "ms_per_step": round(dt_o / K2 * 1e3, 4),
                   "roofline": exh_block, "same_results": same_o,
                   "note": "the resident set 0 replayed (compare with `replay`, not with `value`)"}
        other = {"value": round(B * K2 / dt2, 2), "unit": "queries/s", "steps": K2, "ms_per_step": round(dt2 / K2 * 1e3, 4),
                 "roofline": prn_block, "same_results": same}
        out = {
            "metric": "queries/sec, 4-term disjunctive BM25 + top-k over 10M synthetic Zipf docs",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "repeats": {"regions": R, "steps_per_region": K, "value_from": "median region",
                        "ms_per_step_min": round(min(dts) / K * 1e3, 4), "ms_per_step_median": round(dt / K * 1e3, 4),
                        "ms_per_step_max": round(max(dts) / K * 1e3, 4)},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"zipf-{D} (V={V}, Poisson(32) doc lengths, seed 1234) sharded by doc-id range, "
                                   f"FRESH batches: {len(sets)} rotating seeded sets of {B} x 4-term disjunctive BM25 queries "
                                   f"(k1=1.2 b=0.75; {'the first ' + str(Bq) + ' of ' if B > Bq else ''}set 0 = the BASELINE set"
                                   f"{'; ' + str(Bq) + ' queries per GPU and step: every rank scores all of them on its docs' if weak else ''}), "
                                   f"one sa_batch_step (idf gathered from the index table, reset, run) + fetch per step, "
                                   f"top-{args.k}, the library's default route for the shape: {main_route}",
                       "docs": D, "queries_per_step": B, "terms_per_query": 4, "k": args.k, "query_sets": len(sets),
                       "batches_in_flight": P, "driver": args.driver,
                       "distinct_terms_in_batch": int(len(np.unique(queries))),
                       "tile_docs": int(r.info.tile_docs), "parallelism": f"doc-range shards x{world}",
                       "collective": r.collective, "collective_library": dict(zip(("nccl_version", "path"), comm_lib)),
                       "launcher": "torch-free: ranks rendezvous through an id file, "
                                                               "collectives = libsearcharray_hip.so's RCCL communicator"},
            "postings_scanned_GBps": round(post_total * K / dt / 1e9, 2),
            "collective_library": dict(zip(("nccl_version", "path"), comm_lib)),
            "fresh_batch_latency_ms": None if kernel_ms_fresh != kernel_ms_fresh else round(kernel_ms_fresh, 4),
            "replay": {"value": round(B * K / dt_r, 2), "unit": "queries/s", "ms_per_step": round(dt_r / K * 1e3, 4),
                       "kernel_ms": round(kernel_ms_r, 4), "fresh_over_replay": round(dt_r / dt, 4),
                       "fresh_equals_replay": fresh_equals_replay,
                       "note": "set 0 resident, sa_batch_run only -- no reset, no fetch (rounds 1-2 reported this as `value`)"},
            "roofline": main_block,
            ("fixed_batch" if weak else "wide_batch"): scaled,
            "exhaustive_overlay": overlay,
            "dynamic_pruning": other,
            "cpu_baseline": cpu,
            "parity_check": parity,
        }
        if batch_d is not None:
            comp_d = compulsory_bytes(r.df if world == 1 else r.index.docfreqs(), q_distinct, B, args.k)
            out["distinct_terms"] = {
                "value": round(B * K2 / dt3, 2), "unit": "queries/s", "steps": K2, "ms_per_step": round(dt3 / K2 * 1e3, 4),
                "workload": f"{B} x 4 pairwise-distinct terms (ranks 1..{4 * B}, one per quarter per query), the library's default route ({distinct_route}), top-{args.k}",
                "grouping": batch_d.group_info(),
                "roofline": roofline_block("sa_k_bm25_* (" + str(distinct_route) + ")", kernel_ms3, alg3, comp_d,
                                           dominant(pmc.get("distinct_terms"), ("sa_k_bm25",)),
                                           "no posting list is shared between queries: compulsory_bytes = all posting bytes of the batch")}
        out["dense_score"] = dense_out
        out.update(phrase_out)
        if pmc:
            out["pmc_kernels"] = {leg: pmc[leg]["kernels"] for leg in pmc}
        elif not args.no_pmc:
            out["pmc_error"] = pmc_err
        print(json.dumps(out), flush=True)
    for b in pair:
        b.close()
    batch.close()
    if batch_d is not None:
        batch_d.close()
    if side is not None:
        side.close()
    r.close()


if __name__ == "__main__":
    main()

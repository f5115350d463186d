// sa_sharded.hip -- Part 3b of the C ABI: N devices behind ONE handle.
//
// The reference is single-process and single-device; its partition primitive is key_partition
// (searcharray/roaringish/roaringish.py:227-243): the roaringish key of a word is its doc id and every term's
// words are doc-sorted, so a doc-id range is one contiguous run of every term's word list.  sa_sharded_create cuts
// an encoded index at the doc ids g * N / G, builds shard g on device_ids[g] with GLOBAL corpus size / average doc
// length, joins the shards in one RCCL communicator (one host thread per device: ncclCommInitRank and the
// collectives of a run must be entered concurrently) and sums the document frequencies over the shards.  A batch
// of the handle is one resident batch per shard; run = every shard scores its doc range, the per-shard top-k keys
// are all-gathered over xGMI and merged on every device (sa_batch_run), fetch reads shard 0's copy.
//
// Host code only: everything below goes through the single-device entry points of this library (Part 2 / Part 3), so a
// binder in any language gets N GPUs behind one handle -- what searcharray_amd/sharded.py did in Python until round 3.
#include "sa_common.hpp"
#include "sa_options.hpp"
#include "../../include/searcharray_hip.h"

#include <condition_variable>
#include <algorithm>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

// one persistent host thread per shard: tasks of a call run on all of them at once
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;
    bool has = false, done = false, quit = false;
    int rc = SA_OK;
    std::string err;

    void loop() {
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has || quit; });
                if (quit) return;
                f = task;
                has = false;
            }
            const int r = f();
            const char* msg = r != SA_OK ? sa_last_error() : "";       // (thread-local: copied for the caller's thread)
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                err = msg ? msg : "";
                done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> f) {
        {
            std::lock_guard<std::mutex> lk(m);
            task = std::move(f);
            has = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}   // namespace

struct sa_sharded {
    int G = 0;
    u64 n_docs = 0;
    u32 n_terms = 0;
    float avgdl = 0.f;
    std::vector<int> devices;
    std::vector<u64> bounds;                 // [G + 1] doc-id cuts
    std::vector<sa_index_t*> shards;
    std::vector<u64> df;                     // global document frequencies
    bool comm = false;
    std::vector<std::unique_ptr<Worker>> workers;
    // batches created on this handle and not destroyed yet: sa_sharded_destroy takes their per-shard parts down with the
    // shards and orphans them (sh = null: every later call on such a batch is refused, its destroy only frees the shell)
    std::mutex live_mu;
    std::vector<struct sa_sharded_batch*> live;

    // fn(g) on every shard's thread; the first failure's code and message are the call's
    int all(const std::function<int(int)>& fn) {
        // (the shard threads create handles on behalf of the caller: they start from the CALLER's thread defaults)
        sa_options_t caller_opts;
        const bool have_opts = sa_options_thread_defaults(&caller_opts);
        for (int g = 0; g < G; g++)
            workers[(size_t)g]->post([fn, g, have_opts, caller_opts] {
                sa_options_set_thread_defaults(have_opts ? &caller_opts : nullptr);
                const int r = fn(g);
                sa_options_set_thread_defaults(nullptr);
                return r;
            });
        int rc = SA_OK;
        std::string err;
        for (int g = 0; g < G; g++) {
            const int r = workers[(size_t)g]->wait();
            if (r != SA_OK && rc == SA_OK) { rc = r; err = workers[(size_t)g]->err; }
        }
        if (rc != SA_OK) sa_set_error("shard: %s", err.c_str());
        return rc;
    }
};

struct sa_sharded_batch {
    sa_sharded* sh = nullptr;
    std::vector<sa_batch_t*> parts;
    u32 B = 0, k = 0;
    std::vector<std::vector<float>> scores;          // per-shard fetch buffers (every shard fetches: the redo after an
    std::vector<std::vector<uint64_t>> docs;         // overflow is collective); shard 0's is handed out
};

extern "C" int sa_sharded_destroy(sa_sharded_t* sh) {
    if (!sh) return SA_OK;
    if (!sh->workers.empty()) {
        std::vector<sa_sharded_batch*> live;
        { std::lock_guard<std::mutex> g(sh->live_mu); live.swap(sh->live); }
        for (sa_sharded_batch* bt : live) {
            sh->all([&](int g) { return bt->parts[(size_t)g] ? sa_batch_destroy(bt->parts[(size_t)g]) : SA_OK; });
            bt->parts.assign(bt->parts.size(), nullptr);
            bt->sh = nullptr;
        }
        if (sh->comm) sh->all([sh](int g) { return sh->shards[(size_t)g] ? sa_index_comm_destroy(sh->shards[(size_t)g]) : SA_OK; });
        sh->all([sh](int g) { return sh->shards[(size_t)g] ? sa_index_destroy(sh->shards[(size_t)g]) : SA_OK; });
        for (auto& w : sh->workers) w->stop();
    }
    delete sh;
    return SA_OK;
}

extern "C" int sa_sharded_create(const int* device_ids, int n_dev, uint64_t n_docs, uint32_t n_terms, const uint64_t* words,
                                 const uint64_t* term_off, const float* doc_lens, float avg_doc_len, uint32_t tile_docs,
                                 sa_sharded_t** out) {
    SA_ARG(out && device_ids && term_off && (doc_lens || n_docs == 0), "null argument");
    SA_ARG(n_dev >= 1 && n_dev <= 64, "need 1 .. 64 devices");
    SA_ARG(words || term_off[n_terms] == 0, "null words");
    sa_sharded* sh = new (std::nothrow) sa_sharded();
    if (!sh) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    const int G = n_dev;
    sh->G = G; sh->n_docs = n_docs; sh->n_terms = n_terms; sh->avgdl = avg_doc_len;
    sh->devices.assign(device_ids, device_ids + G);
    sh->bounds.resize((size_t)G + 1);
    for (int g = 0; g <= G; g++) sh->bounds[(size_t)g] = n_docs * (u64)g / (u64)G;
    sh->shards.assign((size_t)G, nullptr);
    for (int g = 0; g < G; g++) {
        sh->workers.emplace_back(new Worker());
        Worker* w = sh->workers.back().get();
        w->th = std::thread([w] { w->loop(); });
    }
    // every shard's thread cuts its own doc range out of every term's list (a lower-bound search per term and cut: the
    // lists are doc-sorted), rebases the doc ids and builds its index
    int rc = sh->all([&](int g) -> int {
        const u64 lo = sh->bounds[(size_t)g], hi = sh->bounds[(size_t)g + 1];
        std::vector<u64> off((size_t)n_terms + 1, 0), w;
        std::vector<std::pair<u64, u64>> cut((size_t)n_terms);
        u64 total = 0;
        auto first_at_least = [&](u64 a, u64 b, u64 doc) {              // first word of [a, b) with doc id >= doc
            while (a < b) { const u64 m = a + ((b - a) >> 1); if ((words[m] >> SA_KEY_SHIFT) < doc) a = m + 1; else b = m; }
            return a;
        };
        for (u32 t = 0; t < n_terms; t++) {
            const u64 a = first_at_least(term_off[t], term_off[t + 1], lo), b = first_at_least(a, term_off[t + 1], hi);
            cut[t] = {a, b};
            total += b - a;
            off[(size_t)t + 1] = total;
        }
        w.resize((size_t)total);
        const u64 rebase = lo << SA_KEY_SHIFT;
        for (u32 t = 0; t < n_terms; t++) {
            u64* dst = w.data() + off[t];
            for (u64 i = cut[t].first; i < cut[t].second; i++) *dst++ = words[i] - rebase;
        }
        return sa_index_create(sh->devices[(size_t)g], hi - lo, lo, n_terms, w.data(), off.data(), doc_lens ? doc_lens + lo : nullptr,
                               avg_doc_len, n_docs, tile_docs, &sh->shards[(size_t)g]);
    });
    if (rc == SA_OK && G > 1) {
        char id[SA_COMM_ID_BYTES];
        rc = sa_comm_unique_id(id, SA_COMM_ID_BYTES);
        if (rc == SA_OK) rc = sh->all([&](int g) { return sa_index_comm_init(sh->shards[(size_t)g], g, G, id, SA_COMM_ID_BYTES); });
        if (rc == SA_OK) sh->comm = true;
    }
    if (rc == SA_OK) {
        // global document frequencies: shard-local df summed by the library's own all-reduce (every shard ends with the sum)
        std::vector<std::vector<u64>> dfs((size_t)G, std::vector<u64>((size_t)n_terms));
        rc = sh->all([&](int g) -> int {
            if (n_terms == 0) return SA_OK;
            SA_TRY(sa_index_docfreqs(sh->shards[(size_t)g], dfs[(size_t)g].data()));
            if (sh->comm) SA_TRY(sa_index_comm_allreduce(sh->shards[(size_t)g], dfs[(size_t)g].data(), n_terms, SA_DT_U64, SA_OP_SUM));
            return SA_OK;
        });
        if (rc == SA_OK) sh->df = dfs[0];
    }
    if (rc != SA_OK) {
        const std::string keep = sa_last_error();
        sa_sharded_destroy(sh);
        sa_set_error("%s", keep.c_str());
        return rc;
    }
    *out = sh;
    return SA_OK;
}

extern "C" int sa_sharded_info(sa_sharded_t* sh, int* n_shards_out, uint64_t* bounds_out) {
    SA_ARG(sh, "null handle");
    if (n_shards_out) *n_shards_out = sh->G;
    if (bounds_out) for (int g = 0; g <= sh->G; g++) bounds_out[g] = sh->bounds[(size_t)g];
    return SA_OK;
}

extern "C" int sa_sharded_shard(sa_sharded_t* sh, int g, sa_index_t** out) {
    SA_ARG(sh && out && g >= 0 && g < sh->G, "bad shard");
    *out = sh->shards[(size_t)g];
    return SA_OK;
}

extern "C" int sa_sharded_docfreqs(sa_sharded_t* sh, uint64_t* df_out) {
    SA_ARG(sh && (df_out || sh->n_terms == 0), "null argument");
    for (u32 t = 0; t < sh->n_terms; t++) df_out[t] = sh->df[t];
    return SA_OK;
}

static int sa_sharded_batch_finish(sa_sharded* sh, sa_sharded_batch* bt, int rc, int B, int k, sa_sharded_batch_t** out) {
    if (rc != SA_OK) {
        const std::string keep = sa_last_error();
        sh->all([&](int g) { return bt->parts[(size_t)g] ? sa_batch_destroy(bt->parts[(size_t)g]) : SA_OK; });
        delete bt;
        sa_set_error("%s", keep.c_str());
        return rc;
    }
    bt->B = (u32)B; bt->k = (u32)k;
    bt->scores.assign((size_t)sh->G, std::vector<float>((size_t)B * (size_t)k));
    bt->docs.assign((size_t)sh->G, std::vector<uint64_t>((size_t)B * (size_t)k));
    { std::lock_guard<std::mutex> g(sh->live_mu); sh->live.push_back(bt); }
    *out = bt;
    return SA_OK;
}

extern "C" int sa_sharded_batch_create(sa_sharded_t* sh, const uint32_t* terms, const float* idf, int n_queries, int n_query_terms,
                                       int k, float k1, float b, sa_sharded_batch_t** out) {
    SA_ARG(sh && out && terms && idf, "null argument");
    sa_sharded_batch* bt = new (std::nothrow) sa_sharded_batch();
    if (!bt) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    bt->sh = sh;
    bt->parts.assign((size_t)sh->G, nullptr);
    const int rc = sh->all([&](int g) {
        return sa_batch_create(sh->shards[(size_t)g], terms, idf, n_queries, n_query_terms, k, k1, b, &bt->parts[(size_t)g]);
    });
    return sa_sharded_batch_finish(sh, bt, rc, n_queries, k, out);
}

extern "C" int sa_sharded_phrase_batch_create(sa_sharded_t* sh, const uint32_t* terms, const int32_t* n_terms, const int32_t* slop,
                                              const float* idf, int n_phrases, int max_terms, int k, float k1, float b,
                                              sa_sharded_batch_t** out) {
    SA_ARG(sh && out && terms && n_terms && idf, "null argument");
    sa_sharded_batch* bt = new (std::nothrow) sa_sharded_batch();
    if (!bt) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    bt->sh = sh;
    bt->parts.assign((size_t)sh->G, nullptr);
    const int rc = sh->all([&](int g) {
        return sa_phrase_batch_create_ex(sh->shards[(size_t)g], terms, n_terms, slop, idf, n_phrases, max_terms, k, k1, b,
                                         &bt->parts[(size_t)g]);
    });
    return sa_sharded_batch_finish(sh, bt, rc, n_phrases, k, out);
}

extern "C" int sa_sharded_batch_reset(sa_sharded_batch_t* bt, const uint32_t* terms, const float* idf) {
    SA_ARG(bt && bt->sh && terms && idf, "null argument");
    return bt->sh->all([&](int g) { return sa_batch_reset(bt->parts[(size_t)g], terms, idf); });
}

// every shard's run enqueues its scoring kernels and then the collective, each from its own thread, so the all-gathers
// of the ranks meet (a single thread would block in the first one)
extern "C" int sa_sharded_batch_run(sa_sharded_batch_t* bt, int sync) {
    SA_ARG(bt && bt->sh, "null batch");
    return bt->sh->all([&](int g) { return sa_batch_run(bt->parts[(size_t)g], sync); });
}

// replace the options of every shard's batch (the sharded counterpart of sa_batch_set_options)
extern "C" int sa_sharded_batch_set_options(sa_sharded_batch_t* bt, const sa_options_t* o) {
    SA_ARG(bt && bt->sh && o, "null argument");
    return bt->sh->all([&](int g) { return bt->parts[(size_t)g] ? sa_batch_set_options(bt->parts[(size_t)g], o) : SA_OK; });
}

extern "C" int sa_sharded_batch_fetch(sa_sharded_batch_t* bt, float* scores_out, uint64_t* docs_out) {
    SA_ARG(bt && bt->sh && scores_out && docs_out, "null argument");
    // collective: after a candidate-list overflow all ranks redo the batch together (sa_batch_fetch)
    SA_TRY(bt->sh->all([&](int g) { return sa_batch_fetch(bt->parts[(size_t)g], bt->scores[(size_t)g].data(), bt->docs[(size_t)g].data()); }));
    const size_t n = (size_t)bt->B * bt->k;
    memcpy(scores_out, bt->scores[0].data(), n * sizeof(float));
    memcpy(docs_out, bt->docs[0].data(), n * sizeof(uint64_t));
    return SA_OK;
}

extern "C" int sa_sharded_batch_destroy(sa_sharded_batch_t* bt) {
    if (!bt) return SA_OK;
    if (bt->sh) {
        { std::lock_guard<std::mutex> g(bt->sh->live_mu); auto& l = bt->sh->live; l.erase(std::remove(l.begin(), l.end(), bt), l.end()); }
        bt->sh->all([&](int g) { return bt->parts[(size_t)g] ? sa_batch_destroy(bt->parts[(size_t)g]) : SA_OK; });
    }
    delete bt;
    return SA_OK;
}

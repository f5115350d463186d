// sa_queue.hip -- a QUERY-SET QUEUE in front of a ring of top-k batches (header, Part 2c).
//
// The reference's unit of work is score() on a query it has not seen (postings.py:652-680), and its callers drive it from a thread
// pool (test/test_msmarco.py:483-507).  The library's counterpart for a stream of query SETS is a ring of batch objects fed in turn:
// sa_batch_step (idf gather, tables, upload, launches: 45-75 us of host time per set) and, `depth` sets later, sa_batch_fetch.  From a
// single caller thread -- Python's, with the ctypes marshalling on top -- that host time is serial with the caller's own work and, on a
// rank-sized shard whose device step is ~70 us, it is the bottleneck.  The queue moves it to a worker thread of the library:
//
//   sa_queue_submit(q, terms)  copies the B x T term ids into the next slot of the ring and returns a ticket (blocks only while `depth`
//                              tickets are outstanding); the WORKER thread runs sa_batch_step for it;
//   sa_queue_fetch(q, ticket)  waits for that set's device work -- outside every lock, so the worker keeps stepping -- and returns
//                              its top-k.
//
// Tickets are served in order; a slot is free again once its ticket has been fetched.  With one rank per process the worker is also the
// only thread that issues the rank's collectives, in ticket order on every rank.  No kernels here: host code over sa_batch_*.
#include "sa_index.hpp"
#include "sa_batch.hpp"
#include "../../include/searcharray_hip.h"

#include <condition_variable>
#include <new>
#include <string>
#include <thread>

struct sa_queue {
    sa_index* ix = nullptr;
    u32 B = 0, T = 0, k = 0;
    int depth = 0;
    std::vector<sa_batch_t*> batches;
    std::vector<std::vector<u32>> terms;        // [depth][B * T]
    // slot state: 0 free, 1 submitted (the worker has not taken it), 2 stepped (device work enqueued), 3 the step failed
    std::vector<int> state, rc;
    std::vector<u64> ticket_of;
    std::vector<std::string> err;
    u64 next_ticket = 0, next_run = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::thread worker;
    bool stop = false;
};

static void sa_queue_worker(sa_queue* q) {
    (void)hipSetDevice(q->ix->device);
    std::unique_lock<std::mutex> lk(q->mu);
    for (;;) {
        const int slot = (int)(q->next_run % (u64)q->depth);
        q->cv.wait(lk, [&] { return q->stop || (q->state[slot] == 1 && q->ticket_of[slot] == q->next_run); });
        if (q->stop) return;
        lk.unlock();
        const int rc = sa_batch_step(q->batches[(size_t)slot], q->terms[(size_t)slot].data());
        std::string e = rc == SA_OK ? std::string() : std::string(sa_last_error());
        lk.lock();
        q->rc[slot] = rc; q->err[slot] = e;
        q->state[slot] = rc == SA_OK ? 2 : 3;
        q->next_run++;
        q->cv.notify_all();
    }
}

extern "C" int sa_queue_create(sa_index_t* ix, int n_queries, int n_query_terms, int k, float k1, float b, int depth, sa_queue_t** out) {
    SA_ARG(ix && out, "null argument");
    SA_ARG(n_queries > 0 && n_query_terms > 0 && k > 0, "empty query sets");
    SA_ARG(depth >= 1 && depth <= 64, "depth must be 1 .. 64 batches");
    sa_queue* q = new (std::nothrow) sa_queue();
    if (!q) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    q->ix = ix; q->B = (u32)n_queries; q->T = (u32)n_query_terms; q->k = (u32)k; q->depth = depth;
    const size_t n = (size_t)q->B * q->T;
    // (the batches are created over a set of unknown terms: every submitted set replaces it)
    std::vector<u32> none(n, 0xFFFFFFFFu);
    std::vector<float> zeros(n, 0.f);
    for (int i = 0; i < depth; i++) {
        sa_batch_t* bt = nullptr;
        const int rc = sa_batch_create(ix, none.data(), zeros.data(), n_queries, n_query_terms, k, k1, b, &bt);
        if (rc != SA_OK) {
            const std::string keep = sa_last_error();
            for (sa_batch_t* x : q->batches) sa_batch_destroy(x);
            delete q;
            sa_set_error("%s", keep.c_str());
            return rc;
        }
        q->batches.push_back(bt);
    }
    q->terms.assign((size_t)depth, std::vector<u32>(n, 0xFFFFFFFFu));
    q->state.assign((size_t)depth, 0); q->rc.assign((size_t)depth, SA_OK);
    q->ticket_of.assign((size_t)depth, ~0ull); q->err.assign((size_t)depth, std::string());
    q->worker = std::thread(sa_queue_worker, q);
    *out = q;
    return SA_OK;
}

extern "C" int sa_queue_submit(sa_queue_t* q, const uint32_t* terms, uint64_t* ticket_out) {
    SA_ARG(q && terms && ticket_out, "null argument");
    std::unique_lock<std::mutex> lk(q->mu);
    const u64 ticket = q->next_ticket;
    const int slot = (int)(ticket % (u64)q->depth);
    q->cv.wait(lk, [&] { return q->state[slot] == 0; });        // (free again once ticket - depth has been fetched)
    memcpy(q->terms[(size_t)slot].data(), terms, (size_t)q->B * q->T * sizeof(u32));
    q->ticket_of[slot] = ticket;
    q->state[slot] = 1;
    q->next_ticket++;
    *ticket_out = ticket;
    q->cv.notify_all();
    return SA_OK;
}

extern "C" int sa_queue_fetch(sa_queue_t* q, uint64_t ticket, float* scores_out, uint64_t* docs_out) {
    SA_ARG(q && scores_out && docs_out, "null argument");
    int slot;
    {
        std::unique_lock<std::mutex> lk(q->mu);
        SA_ARG(ticket < q->next_ticket, "sa_queue_fetch: a ticket that was never handed out");
        slot = (int)(ticket % (u64)q->depth);
        SA_ARG(q->ticket_of[slot] == ticket && q->state[slot] != 0, "sa_queue_fetch: the ticket has been fetched already (or is older than the ring)");
        q->cv.wait(lk, [&] { return q->state[slot] >= 2; });
        if (q->state[slot] == 3) {
            const int rc = q->rc[slot];
            sa_set_error("%s", q->err[slot].c_str());
            q->state[slot] = 0;
            q->cv.notify_all();
            return rc;
        }
    }
    // the set's device work: waited for outside every lock (the worker keeps stepping the following sets meanwhile)
    sa_batch* bt = q->batches[(size_t)slot];
    SA_HIP(hipSetDevice(q->ix->device));
    if (bt->res_pending && bt->ev_res) SA_HIP(hipEventSynchronize(bt->ev_res));
    const int rc = sa_batch_fetch(bt, scores_out, docs_out);
    {
        std::lock_guard<std::mutex> lk(q->mu);
        q->state[slot] = 0;
        q->cv.notify_all();
    }
    return rc;
}

extern "C" int sa_queue_destroy(sa_queue_t* q) {
    if (!q) return SA_OK;
    {
        std::lock_guard<std::mutex> lk(q->mu);
        q->stop = true;
        q->cv.notify_all();
    }
    if (q->worker.joinable()) q->worker.join();
    for (sa_batch_t* bt : q->batches) sa_batch_destroy(bt);
    delete q;
    return SA_OK;
}

// the batch behind a ticket's slot (diagnostics: sa_batch_last_route, sa_batch_host_times, kernel profiles of the queue's batches)
extern "C" int sa_queue_batch(sa_queue_t* q, int slot, sa_batch_t** out) {
    SA_ARG(q && out && slot >= 0 && slot < q->depth, "bad slot");
    *out = q->batches[(size_t)slot];
    return SA_OK;
}

// TEST INFRASTRUCTURE: the host-emulated build has no RCCL.  Part 3 of the C ABI is served by an
// IN-PROCESS communicator instead: the "ranks" are threads of one process (one per shard, exactly how
// searcharray_amd/sharded.py drives N devices from one process), they meet at a rendezvous keyed by
// the unique id, and a collective is two barriers around plain memory copies ("device" memory is heap
// memory here and every emulated stream is synchronous).  This lets the CPU suite run the whole
// N-devices-behind-one-handle path: shard split, global statistics, all-gather of the per-shard top-k
// and the cross-rank merge.
#include "sa_index.hpp"
#include "../../include/searcharray_hip.h"
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {
struct Group {
    int nranks = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    u64 gen = 0;
    std::vector<const void*> src;
    void barrier(std::unique_lock<std::mutex>& lk) {
        const u64 my = gen;
        if (++arrived == nranks) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != my; });
    }
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Group>> g_groups;
}   // namespace

extern "C" int sa_comm_library_info(int* version_out, char* path_out, int path_len) {
    if (version_out) *version_out = 0;
    if (path_out && path_len > 0) { strncpy(path_out, "in-process test communicator", (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return SA_OK;
}

struct sa_comm {
    std::shared_ptr<Group> g;
    std::string key;
    int rank = 0, nranks = 1;
};

// every rank publishes `mine`, then runs fn(all pointers) while nobody may touch the published buffers
template <class F>
static void sa_emu_collective(sa_comm* c, const void* mine, F fn) {
    Group& g = *c->g;
    std::unique_lock<std::mutex> lk(g.m);
    g.src[(size_t)c->rank] = mine;
    g.barrier(lk);
    std::vector<const void*> all = g.src;
    lk.unlock();
    fn(all);
    lk.lock();
    g.barrier(lk);
}

int sa_comm_allgather_topk(sa_index* ix, const u64* d_local, u64* d_gather, size_t count, int* nranks_out, hipStream_t) {
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    *nranks_out = ix->comm->nranks;
    if (count == 0) return SA_OK;
    sa_emu_collective(ix->comm, d_local, [&](const std::vector<const void*>& all) {
        for (size_t r = 0; r < all.size(); r++) memcpy(d_gather + r * count, all[r], count * sizeof(u64));
    });
    return SA_OK;
}

int sa_comm_allreduce_max_u32(sa_index* ix, u32* d_val, hipStream_t) {
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    u32 m = 0;
    sa_emu_collective(ix->comm, d_val, [&](const std::vector<const void*>& all) {
        for (const void* p : all) { const u32 v = *(const u32*)p; m = v > m ? v : m; }
    });
    // (written after the second barrier: every rank has read the old values)
    *d_val = m;
    return SA_OK;
}

extern "C" int sa_comm_unique_id(char* id_out, int len) {
    SA_ARG(id_out && len >= SA_COMM_ID_BYTES, "id buffer must hold 128 bytes");
    static std::mutex mu;
    static std::mt19937_64 rng(std::random_device{}());
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < SA_COMM_ID_BYTES; i += 8) {
        const u64 v = rng();
        memcpy(id_out + i, &v, 8);
    }
    return SA_OK;
}

extern "C" int sa_index_comm_init(sa_index* ix, int rank, int nranks, const char* id_bytes, int len) {
    SA_ARG(ix && id_bytes && len >= SA_COMM_ID_BYTES, "bad argument");
    SA_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
    std::lock_guard<std::mutex> gi(ix->mu);
    SA_ARG(!ix->comm, "communicator already initialised");
    sa_comm* c = new sa_comm();
    c->key.assign(id_bytes, SA_COMM_ID_BYTES);
    c->rank = rank; c->nranks = nranks;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto& slot = g_groups[c->key];
        if (!slot) { slot = std::make_shared<Group>(); slot->nranks = nranks; slot->src.resize((size_t)nranks); }
        if (slot->nranks != nranks) { delete c; sa_set_error("ranks disagree about the communicator size"); return SA_ERR_COMM; }
        slot->joined++;
        c->g = slot;
    }
    ix->comm = c;
    return SA_OK;
}

extern "C" int sa_index_comm_destroy(sa_index* ix) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> gi(ix->mu);
    if (!ix->comm) return SA_OK;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_groups.find(ix->comm->key);
        if (it != g_groups.end() && --it->second->joined <= 0) g_groups.erase(it);
    }
    delete ix->comm;
    ix->comm = nullptr;
    return SA_OK;
}

extern "C" int sa_index_comm_info(sa_index* ix, int* rank_out, int* nranks_out) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> g(ix->mu);
    if (rank_out) *rank_out = ix->comm ? ix->comm->rank : 0;
    if (nranks_out) *nranks_out = ix->comm ? ix->comm->nranks : 1;
    return SA_OK;
}

extern "C" int sa_index_comm_allreduce(sa_index* ix, void* host_inout, uint64_t n, int dtype, int op) {
    SA_ARG(ix && (host_inout || n == 0), "null argument");
    SA_ARG(dtype == SA_DT_U64 || dtype == SA_DT_F64, "dtype must be SA_DT_U64 or SA_DT_F64");
    SA_ARG(op == SA_OP_SUM || op == SA_OP_MAX, "op must be SA_OP_SUM or SA_OP_MAX");
    std::lock_guard<std::mutex> gi(ix->mu);
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    if (n == 0) return SA_OK;
    std::vector<u64> res((size_t)n);
    sa_emu_collective(ix->comm, host_inout, [&](const std::vector<const void*>& all) {
        for (uint64_t i = 0; i < n; i++) {
            if (dtype == SA_DT_U64) {
                u64 a = ((const u64*)all[0])[i];
                for (size_t r = 1; r < all.size(); r++) {          // rank order, like a ring would not be -- sums of u64 commute
                    const u64 v = ((const u64*)all[r])[i];
                    a = op == SA_OP_SUM ? a + v : (v > a ? v : a);
                }
                res[(size_t)i] = a;
            } else {
                double a = ((const double*)all[0])[i];
                for (size_t r = 1; r < all.size(); r++) {
                    const double v = ((const double*)all[r])[i];
                    a = op == SA_OP_SUM ? a + v : (v > a ? v : a);
                }
                memcpy(&res[(size_t)i], &a, 8);
            }
        }
    });
    memcpy(host_inout, res.data(), (size_t)n * 8);
    return SA_OK;
}

extern "C" int sa_index_comm_barrier(sa_index* ix) {
    uint64_t one = 1;
    return sa_index_comm_allreduce(ix, &one, 1, SA_DT_U64, SA_OP_SUM);
}

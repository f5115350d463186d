// sa_comm.hip -- Part 3 of the C ABI: the one exchange step of the doc-range-sharded path.
//
// The reference is single-process (no NCCL/MPI anywhere); sharding is new work.  Documents are
// independent, so GPU g scores its doc range with GLOBAL statistics and the only data-path
// collective is one all-gather of the per-shard top-k keys (B * k * 8 bytes per rank -- 20 KiB for
// B = 256, k = 10), followed by a local k-way merge on every rank.  Latency-bound on xGMI, so one
// collective per query BATCH, enqueued on the index's exchange stream so it (and the cross-rank
// merge) overlaps the next batch's scoring kernels.
#include "sa_index.hpp"
#include "../../include/searcharray_hip.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <new>

struct sa_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    void* d_buf = nullptr;          // device staging of the host-side reductions (grown on demand, kept)
    size_t buf_bytes = 0;
};

#define SA_NCCL(expr)                                                                        \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            sa_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
            return SA_ERR_COMM;                                                              \
        }                                                                                    \
    } while (0)

extern "C" int sa_comm_unique_id(char* id_out, int len) {
    SA_ARG(id_out && len >= (int)sizeof(ncclUniqueId), "id buffer must hold 128 bytes");
    ncclUniqueId id;
    SA_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return SA_OK;
}

extern "C" int sa_comm_library_info(int* version_out, char* path_out, int path_len) {
    if (version_out) {
        int v = 0;
        SA_NCCL(ncclGetVersion(&v));
        *version_out = v;
    }
    if (path_out && path_len > 0) {
        path_out[0] = 0;
        Dl_info info;
        if (dladdr((void*)&ncclGetVersion, &info) && info.dli_fname) {
            strncpy(path_out, info.dli_fname, (size_t)path_len - 1);
            path_out[path_len - 1] = 0;
        }
    }
    return SA_OK;
}

extern "C" int sa_index_comm_init(sa_index_t* ix, int rank, int nranks, const char* id_bytes, int len) {
    SA_ARG(ix && id_bytes && len >= (int)sizeof(ncclUniqueId), "bad argument");
    SA_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
    std::lock_guard<std::mutex> g(ix->mu);
    SA_ARG(!ix->comm, "communicator already initialised");
    SA_HIP(hipSetDevice(ix->device));
    sa_comm* c = new (std::nothrow) sa_comm();
    if (!c) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        sa_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return SA_ERR_COMM;
    }
    c->rank = rank; c->nranks = nranks;
    if (!ix->xstream && hipStreamCreateWithFlags(&ix->xstream, hipStreamNonBlocking) != hipSuccess) {
        sa_set_error("hipStreamCreate (exchange stream) failed");
        ncclCommDestroy(c->comm);
        delete c;
        return SA_ERR_HIP;
    }
    ix->comm = c;
    return SA_OK;
}

extern "C" int sa_index_comm_destroy(sa_index_t* ix) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->comm) return SA_OK;
    hipSetDevice(ix->device);
    hipStreamSynchronize(ix->stream);
    if (ix->xstream) hipStreamSynchronize(ix->xstream);
    ncclCommDestroy(ix->comm->comm);
    if (ix->comm->d_buf) hipFree(ix->comm->d_buf);
    delete ix->comm;
    ix->comm = nullptr;
    return SA_OK;
}

// count == 0: only report nranks.
int sa_comm_allgather_topk(sa_index* ix, const u64* d_local, u64* d_gather, size_t count, int* nranks_out,
                           hipStream_t st) {
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    *nranks_out = ix->comm->nranks;
    if (count == 0) return SA_OK;
    SA_NCCL(ncclAllGather(d_local, d_gather, count, ncclUint64, ix->comm->comm, st));
    return SA_OK;
}

// max over the ranks of one device-resident u32 (in place)
int sa_comm_allreduce_max_u32(sa_index* ix, u32* d_val, hipStream_t st) {
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    SA_NCCL(ncclAllReduce(d_val, d_val, 1, ncclUint32, ncclMax, ix->comm->comm, st));
    return SA_OK;
}

// Host-side reductions over the communicator: the index-time statistics of a sharded corpus (df per term,
// sum of doc lengths -> avgdl) and the driver's "max over ranks" timing, so a multi-process caller needs no
// other collective library.  dtype: SA_DT_U64 / SA_DT_F64, op: SA_OP_SUM / SA_OP_MAX; in place, blocking.
extern "C" int sa_index_comm_allreduce(sa_index_t* ix, void* host_inout, uint64_t n, int dtype, int op) {
    SA_ARG(ix && (host_inout || n == 0), "null argument");
    SA_ARG(dtype == SA_DT_U64 || dtype == SA_DT_F64, "dtype must be SA_DT_U64 or SA_DT_F64");
    SA_ARG(op == SA_OP_SUM || op == SA_OP_MAX, "op must be SA_OP_SUM or SA_OP_MAX");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
    if (n == 0) return SA_OK;
    SA_HIP(hipSetDevice(ix->device));
    sa_comm* c = ix->comm;
    if (c->buf_bytes < n * 8) {                       // (barriers and timing reductions reuse the same few bytes)
        if (c->d_buf) SA_HIP(hipFree(c->d_buf));
        c->d_buf = nullptr; c->buf_bytes = 0;
        const size_t want = n * 8 < 4096 ? 4096 : n * 8;
        SA_HIP(hipMalloc(&c->d_buf, want));
        c->buf_bytes = want;
    }
    void* d = c->d_buf;
    hipStream_t xs = ix->xstream;
    int rc = SA_OK;
    if (hipMemcpyAsync(d, host_inout, n * 8, hipMemcpyHostToDevice, xs) != hipSuccess) rc = SA_ERR_HIP;
    if (rc == SA_OK) {
        ncclResult_t r = ncclAllReduce(d, d, n, dtype == SA_DT_U64 ? ncclUint64 : ncclDouble, op == SA_OP_SUM ? ncclSum : ncclMax,
                                       ix->comm->comm, xs);
        if (r != ncclSuccess) { sa_set_error("ncclAllReduce failed: %s", ncclGetErrorString(r)); rc = SA_ERR_COMM; }
    }
    if (rc == SA_OK && hipMemcpyAsync(host_inout, d, n * 8, hipMemcpyDeviceToHost, xs) != hipSuccess) rc = SA_ERR_HIP;
    if (hipStreamSynchronize(xs) != hipSuccess && rc == SA_OK) rc = SA_ERR_HIP;
    if (rc == SA_ERR_HIP) sa_set_error("sa_index_comm_allreduce: HIP copy / synchronize failed");
    return rc;
}

// every rank has reached this call and the work enqueued before it on both of the index's streams is done
extern "C" int sa_index_comm_barrier(sa_index_t* ix) {
    SA_ARG(ix, "null index");
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->comm) { sa_set_error("index has no communicator"); return SA_ERR_STATE; }
        SA_HIP(hipSetDevice(ix->device));
        SA_HIP(hipStreamSynchronize(ix->stream));
    }
    uint64_t one = 1;
    return sa_index_comm_allreduce(ix, &one, 1, SA_DT_U64, SA_OP_SUM);
}

extern "C" int sa_index_comm_info(sa_index_t* ix, int* rank_out, int* nranks_out) {
    SA_ARG(ix, "null index");
    std::lock_guard<std::mutex> g(ix->mu);
    if (rank_out) *rank_out = ix->comm ? ix->comm->rank : 0;
    if (nranks_out) *nranks_out = ix->comm ? ix->comm->nranks : 1;
    return SA_OK;
}

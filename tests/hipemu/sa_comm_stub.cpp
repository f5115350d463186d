// TEST INFRASTRUCTURE: the host-emulated build has no RCCL; Part 3 entry points report that.
#include "sa_index.hpp"
int sa_comm_allgather_topk(sa_index*, const u64*, u64*, size_t, int*, hipStream_t) {
    sa_set_error("emulated build has no RCCL communicator");
    return SA_ERR_UNSUPPORTED;
}
int sa_comm_allreduce_max_u32(sa_index*, u32*, hipStream_t) {
    sa_set_error("emulated build has no RCCL communicator");
    return SA_ERR_UNSUPPORTED;
}
extern "C" int sa_comm_unique_id(char*, int) { sa_set_error("emulated build has no RCCL"); return SA_ERR_UNSUPPORTED; }
extern "C" int sa_index_comm_init(sa_index*, int, int, const char*, int) { sa_set_error("emulated build has no RCCL"); return SA_ERR_UNSUPPORTED; }
extern "C" int sa_index_comm_destroy(sa_index*) { return SA_OK; }

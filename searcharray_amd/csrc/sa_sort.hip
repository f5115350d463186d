// sa_sort.hip -- the one library primitive of the index build (sa_build.hip): a stable LSD radix sort of the
// token stream's (term, doc << 24 | position) pairs by term, rocPRIM's device-wide radix_sort_pairs.  The index
// BUILD is a "next" row of the scope table, not the scoring hot path; the hot path has no library
// kernels.
#include "sa_common.hpp"
#include <rocprim/rocprim.hpp>

int sa_sort_pairs_by_key(u32* keys_in, u32* keys_out, u64* vals_in, u64* vals_out, u32 n, int bits, hipStream_t st) {
    size_t temp_bytes = 0;
    SA_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                     (unsigned int)bits, st));
    void* temp = nullptr;
    SA_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                             (unsigned int)bits, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(temp);
    if (e != hipSuccess || e2 != hipSuccess) {
        sa_set_error("radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return SA_ERR_HIP;
    }
    return SA_OK;
}

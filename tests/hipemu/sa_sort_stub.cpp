// TEST INFRASTRUCTURE: the host-emulated build has no rocPRIM; "device" memory is host memory there, so
// the stable sort by key of csrc/sa_sort.hip is a std::stable_sort.
#include "sa_common.hpp"
#include <algorithm>
#include <numeric>
#include <vector>

int sa_sort_pairs_by_key(u32* keys_in, u32* keys_out, u64* vals_in, u64* vals_out, u32 n, int, hipStream_t) {
    std::vector<u32> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys_in[a] < keys_in[b]; });
    for (u32 i = 0; i < n; i++) { keys_out[i] = keys_in[order[i]]; vals_out[i] = vals_in[order[i]]; }
    return SA_OK;
}

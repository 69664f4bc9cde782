// sort_emu.cpp -- TEST-ONLY host stand-in for the sort backend (sort_rocprim.hip) used by the emulated
// build of the kernels (rocPRIM cannot run on the host).  std::stable_sort on the masked key.
#include <hip/hip_runtime.h>

#include <numeric>
#include <vector>

#include "../../activesplat_amd/csrc/gs_common.h"

namespace gs {
size_t sort_temp_bytes(int64_t, int) { return 256; }
hipError_t sort_pairs(void*, size_t, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                      uint32_t* vals_out, int64_t D, int end_bit, hipStream_t)
{
    std::vector<int64_t> idx((size_t)D);
    std::iota(idx.begin(), idx.end(), 0);
    const uint64_t mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
    for (int64_t i = 0; i < D; i++) { keys_out[i] = keys_in[idx[i]]; vals_out[i] = vals_in[idx[i]]; }
    return 0;
}
}  // namespace gs

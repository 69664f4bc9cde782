// sort_rocprim.hip -- (key64, value32) stable radix sort backend.
// Round-1 backend: rocPRIM's device radix sort restricted to the significant key bits
// (32 depth bits + ceil(log2(#tiles)) tile bits).  Kept behind this two-function interface so the
// hand-written per-tile LDS sort can replace it without touching the callers.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "gs_common.h"

namespace gs {

size_t sort_temp_bytes(int64_t D, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)(D > 0 ? D : 1), 0u, (unsigned)end_bit, (hipStream_t)0);
    return bytes;
}

hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, int64_t D, int end_bit, hipStream_t st)
{
    if (D <= 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)D, 0u,
                                     (unsigned)end_bit, st);
}

}  // namespace gs

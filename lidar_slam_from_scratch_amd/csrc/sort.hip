// sort.hip -- stable radix sort of (Morton key, index) pairs for the target pre-pass.
// Kept in its own translation unit: rocPRIM's templates take most of the build time and
// never change with the ICP kernels.  Not on the per-iteration path (once per call).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace icpmi {

// temp == nullptr: only *temp_bytes is set.  30-bit keys.
hipError_t sort_pairs_u32(void *temp, size_t *temp_bytes, const unsigned *keys_in, unsigned *keys_out,
                          const unsigned *vals_in, unsigned *vals_out, unsigned n, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 30u, stream);
}

} // namespace icpmi

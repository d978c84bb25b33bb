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

// ---- voxel pre-filter helpers (file_utils.cpp:148-196 on the GPU): 63-bit voxel keys ----------
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

namespace icpmi {

hipError_t sort_pairs_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys_in,
                          unsigned long long *keys_out, const unsigned *vals_in, unsigned *vals_out,
                          unsigned n, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 63u, stream);
}

// runs of equal keys: counts[r] and *runs_out (device) ; unique keys are written to unique_out
hipError_t run_lengths_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys, unsigned n,
                           unsigned long long *unique_out, unsigned *counts_out, unsigned *runs_out,
                           hipStream_t stream)
{
    return rocprim::run_length_encode(temp, *temp_bytes, keys, n, unique_out, counts_out, runs_out, stream);
}

// all 64 bits, keys only: the occupancy set (occupancy.h)
hipError_t sort_keys_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys_in,
                         unsigned long long *keys_out, unsigned n, hipStream_t stream)
{
    return rocprim::radix_sort_keys(temp, *temp_bytes, keys_in, keys_out, n, 0u, 64u, stream);
}

} // namespace icpmi
#include <rocprim/device/device_merge.hpp>
namespace icpmi {
// two sorted key arrays -> one (the occupancy set and a frame's new cells)
hipError_t merge_keys_u64(void *temp, size_t *temp_bytes, const unsigned long long *a, const unsigned long long *b,
                          unsigned long long *out, unsigned na, unsigned nb, hipStream_t stream)
{
    return rocprim::merge(temp, *temp_bytes, a, b, out, na, nb, rocprim::less<unsigned long long>(), stream);
}

hipError_t exclusive_sum_u32(void *temp, size_t *temp_bytes, const unsigned *in, unsigned *out, unsigned n,
                             hipStream_t stream)
{
    return rocprim::exclusive_scan(temp, *temp_bytes, in, out, 0u, n, rocprim::plus<unsigned>(), stream);
}

} // namespace icpmi

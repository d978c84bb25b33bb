// voxel.h -- voxel-grid downsampling on the GPU: what
// slam::voxel_downsample does before every ICP call (src/core/file_utils.cpp:148-196,
// slam_node.cpp:122).  Key = floor(coord / voxel) per axis as int64 (file_utils.cpp:177-179),
// centroid = (sum of the voxel's points in INPUT order) / count (file_utils.cpp:187-192).
//
// Sort-based instead of the reference's unordered_map: 63-bit key (21 bits per axis, offset
// by the cloud's minimum key) + stable radix sort keeps the input order inside a voxel, so the
// fp64 sums run in the reference's order and the centroids are bit-identical.  Voxels come
// out sorted by (kx, ky, kz); the reference's order is std::unordered_map iteration order
// (implementation-defined), so parity is as a set.  HBM-bound: 24 B read + 12 B key/value
// per point, a radix sort of n pairs, 24 B per point gathered once more.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpmi {

// float32 records (x, y, z first, `stride` floats apart: 4 for a KITTI .bin, file_utils.cpp:126-138)
// -> the library's N x 3 fp64 layout.  The widening that load_bin does on the host
// (static_cast<double>, exact) done where the points are going to live: 16 B per point cross the
// PCIe link instead of 24.
__global__ __launch_bounds__(256) void k_widen_f32(const float *__restrict__ rec, int n, int stride,
                                                   double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[3 * i] = (double)rec[(size_t)i * stride];
    out[3 * i + 1] = (double)rec[(size_t)i * stride + 1];
    out[3 * i + 2] = (double)rec[(size_t)i * stride + 2];
}

__global__ __launch_bounds__(256) void k_voxel_keys(const double *__restrict__ pts, int n, double voxel,
                                                    long long kx0, long long ky0, long long kz0,
                                                    unsigned long long *__restrict__ keys,
                                                    unsigned *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long kx = (long long)floor(pts[3 * i] / voxel) - kx0;       // file_utils.cpp:177
    const long long ky = (long long)floor(pts[3 * i + 1] / voxel) - ky0;   // file_utils.cpp:178
    const long long kz = (long long)floor(pts[3 * i + 2] / voxel) - kz0;   // file_utils.cpp:179
    keys[i] = ((unsigned long long)kx << 42) | ((unsigned long long)ky << 21) | (unsigned long long)kz;
    vals[i] = (unsigned)i;
}

// one thread per voxel: points summed in input order (the sort is stable), then / count
__global__ __launch_bounds__(256) void k_voxel_centroids(const double *__restrict__ pts,
                                                         const unsigned *__restrict__ order,
                                                         const unsigned *__restrict__ offsets,
                                                         const unsigned *__restrict__ counts, int runs,
                                                         double *__restrict__ out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= runs) return;
    const unsigned o = offsets[r], c = counts[r];
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (unsigned t = 0; t < c; ++t) { // file_utils.cpp:188-190
        const unsigned i = order[o + t];
        cx += pts[3 * i];
        cy += pts[3 * i + 1];
        cz += pts[3 * i + 2];
    }
    const double k = (double)c;        // file_utils.cpp:191
    out[3 * r] = cx / k;
    out[3 * r + 1] = cy / k;
    out[3 * r + 2] = cz / k;
}

} // namespace icpmi

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

// The key origin (the cloud's minimum key per axis) comes from the bounding box the device has
// just reduced: no round trip to the host for it.  `flag` (one word, read back with the run count)
// is set (OR-ed into a word the caller has cleared) when the grid cannot be keyed: more than 2^21 cells on
// an axis (an infinite coordinate included: it is in the box), or a point with a NaN coordinate -- the box's
// comparisons skip a NaN, and the cast of floor(NaN / voxel) is undefined (in the reference too,
// file_utils.cpp:177-179): its key could land in another voxel's bit fields and poison that centroid.
struct VoxelBox { // same layout as NnFrame (nn_mfma.h)
    double lo[3], hi[3];
};
__global__ __launch_bounds__(256) void k_voxel_keys(const double *__restrict__ pts, int n, double voxel,
                                                    const VoxelBox *__restrict__ box, unsigned *__restrict__ flag,
                                                    unsigned long long *__restrict__ keys,
                                                    unsigned *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    long long k0[3];
    bool bad = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = floor(box->lo[a] / voxel), hi = floor(box->hi[a] / voxel);
        bad |= !(lo == lo) || !(hi == hi) || hi - lo >= 2097152.0 || fabs(lo) > 4.0e18 || fabs(hi) > 4.0e18;
        k0[a] = bad ? 0 : (long long)lo;
    }
    if (i == 0 && bad) atomicOr(flag, 1u);
    if (i >= n) return;
    unsigned long long key = 0ull;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    if (!(px == px) || !(py == py) || !(pz == pz)) {
        atomicOr(flag, 1u);
    } else if (!bad) {
        const long long kx = (long long)floor(px / voxel) - k0[0];   // file_utils.cpp:177
        const long long ky = (long long)floor(py / voxel) - k0[1];   // file_utils.cpp:178
        const long long kz = (long long)floor(pz / voxel) - k0[2];   // file_utils.cpp:179
        key = ((unsigned long long)kx << 42) | ((unsigned long long)ky << 21) | (unsigned long long)kz;
    }
    keys[i] = key;
    vals[i] = (unsigned)i;
}

// (readlane_f64: lane `l` (wave-uniform) of a double, through the scalar unit -- icp_small.h)

// One WAVE per voxel.  The centroid must add the voxel's points in input order (the sort is
// stable) like file_utils.cpp:188-190, so the three sums are serial chains -- but the loads
// need not be: 64 points of the run are fetched at once (index, then coordinates), and the
// additions walk over them through wave-uniform lane reads (v_readlane: the operand arrives in
// an SGPR).  A thread per voxel, two dependent loads per point, took 103 us on a 115k-point scan
// (the voxels next to the sensor hold hundreds of points); this form is bound by the adds.
__global__ __launch_bounds__(256) void k_voxel_centroids(const double *__restrict__ pts,
                                                         const unsigned *__restrict__ order,
                                                         const unsigned *__restrict__ offsets,
                                                         const unsigned *__restrict__ counts, int runs,
                                                         double *__restrict__ out,
                                                         const unsigned *__restrict__ runs_dev = nullptr)
{
    // runs_dev: the number of voxels is still on the device (the prefetch worker queues the whole filter
    // without a host round trip); `runs` is then the bound the grid was sized for (rows `out` holds)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= runs || (runs_dev && (unsigned)r >= *runs_dev)) return; // wave-uniform
    const unsigned o = offsets[r], c = counts[r];
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (unsigned base = 0; base < c; base += 64) {
        double x = 0.0, y = 0.0, z = 0.0;
        if (base + lane < c) {
            const unsigned i = order[o + base + lane];
            x = pts[3 * (size_t)i];
            y = pts[3 * (size_t)i + 1];
            z = pts[3 * (size_t)i + 2];
        }
        const int nb = c - base < 64u ? (int)(c - base) : 64;
        for (int u = 0; u < nb; ++u) { // file_utils.cpp:188-190, every lane forms the same sums
            cx += readlane_f64(x, u);
            cy += readlane_f64(y, u);
            cz += readlane_f64(z, u);
        }
    }
    if (lane == 0) {
        const double k = (double)c;    // file_utils.cpp:191
        out[3 * (size_t)r] = cx / k;
        out[3 * (size_t)r + 1] = cy / k;
        out[3 * (size_t)r + 2] = cz / k;
    }
}

} // namespace icpmi

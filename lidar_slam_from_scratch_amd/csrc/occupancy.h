// occupancy.h -- occupancy-grid insert on the GPU (SURVEY section 8f, row N3's add-on): what
// SlamNode::update_occupancy_grid does with every registered frame (src/ros/slam_node.cpp:153,
// :211-221; config slam_node.hpp:35-40):
//     if (z < height_min || z > height_max) continue;                          :214
//     r = sqrt((x - sensor.x)^2 + (y - sensor.y)^2);                            :215
//     if (r > max_range || r < 0.5) continue;                                   :216
//     occupied_cells_.insert({(int)floor(x / resolution), (int)floor(y / resolution)});   :217-219
// The reference's set is a std::unordered_set<GridCell>; here it is a sorted array of unique 64-bit
// keys in device memory ((x, y) biased to unsigned, x in the upper half: ascending keys = ascending
// (x, y)), and an update is key formation, one radix sort of {set, new keys} and a run-length pass.
// Integer work, HBM-bound: 24 B read + 8 B written per point, then the sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpmi {

struct GridParams {
    double sx, sy;                 // sensor position (new_pose.t(), slam_node.cpp:153)
    double resolution, height_min, height_max, max_range;
};

constexpr unsigned long long kGridNone = ~0ull; // sorts behind every cell; never a cell (see below)

__device__ __forceinline__ unsigned long long grid_key(int x, int y)
{
    return ((unsigned long long)((unsigned)x ^ 0x80000000u) << 32) | (unsigned long long)((unsigned)y ^ 0x80000000u);
}

// keys[i] = the cell point i marks, kGridNone where the reference `continue`s.  A quotient that is
// not finite or does not fit an int is undefined in the reference's static_cast: it marks nothing
// here (|cell| <= 2^31 - 2, so kGridNone = (INT_MAX, INT_MAX) is never a cell).
__global__ __launch_bounds__(256) void k_grid_keys(const double *__restrict__ pts, int n, GridParams g,
                                                   unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    unsigned long long key = kGridNone;
    if (!(z < g.height_min || z > g.height_max)) {
        const double dx = x - g.sx, dy = y - g.sy;
        const double r = __dsqrt_rn(dx * dx + dy * dy);
        if (!(r > g.max_range || r < 0.5)) {
            const double cx = floor(x / g.resolution), cy = floor(y / g.resolution);
            if (fabs(cx) <= 2147483646.0 && fabs(cy) <= 2147483646.0) key = grid_key((int)cx, (int)cy);
        }
    }
    keys[i] = key;
}

// after the run-length pass over the sorted keys: the set's size is the number of runs, less the
// run of kGridNone at the end when there is one
__global__ void k_grid_count(const unsigned long long *__restrict__ unique, const unsigned *__restrict__ runs,
                             unsigned *__restrict__ count)
{
    const unsigned r = *runs;
    *count = r > 0 && unique[r - 1] == kGridNone ? r - 1 : r;
}

__global__ __launch_bounds__(256) void k_grid_decode(const unsigned long long *__restrict__ keys, int n,
                                                     int *__restrict__ cells_xy)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    cells_xy[2 * i] = (int)((unsigned)(k >> 32) ^ 0x80000000u);
    cells_xy[2 * i + 1] = (int)((unsigned)k ^ 0x80000000u);
}

} // namespace icpmi

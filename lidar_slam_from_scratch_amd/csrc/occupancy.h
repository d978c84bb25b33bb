// occupancy.h -- occupancy-grid insert on the GPU (SURVEY section 8f, row N3's add-on): what
// SlamNode::update_occupancy_grid does with every registered frame (src/ros/slam_node.cpp:153,
// :211-221; config slam_node.hpp:35-40):
//     if (z < height_min || z > height_max) continue;                          :214
//     r = sqrt((x - sensor.x)^2 + (y - sensor.y)^2);                            :215
//     if (r > max_range || r < 0.5) continue;                                   :216
//     occupied_cells_.insert({(int)floor(x / resolution), (int)floor(y / resolution)});   :217-219
// The reference's set is a std::unordered_set<GridCell>; here it is a sorted array of unique 64-bit
// keys in device memory ((x, y) biased to unsigned, x in the upper half: ascending keys = ascending
// (x, y)).  An update forms the keys of the frame's points and drops those already in the set (a
// binary search each), sorts what is left -- a few hundred new cells and a lot of "none" -- makes it
// unique, and MERGES it into the set: the work per frame follows the frame, not the map (a re-sort of
// {set, new keys} grew with the drive: 300k cells after a KITTI-length sequence).
// Integer work, HBM-bound: 24 B read per point, then small sorts and one pass over the set.
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

// keys already in the (sorted, unique) set become kGridNone
__global__ __launch_bounds__(256) void k_grid_drop_known(unsigned long long *__restrict__ keys, int n,
                                                         const unsigned long long *__restrict__ set, int set_n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    if (k == kGridNone) return;
    int lo = 0, hi = set_n; // first position with set[pos] >= k
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (set[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    if (lo < set_n && set[lo] == k) keys[i] = kGridNone;
}

// After the run-length pass over the frame's sorted keys: `unique` holds `*runs` keys, the last of
// them kGridNone if any point marked nothing or a known cell.  Everything from the first kGridNone on
// is set to kGridNone (the merge then leaves those n - new entries behind the cells), and the set's
// size after the merge is published: set_n + new.
__global__ __launch_bounds__(256) void k_grid_pad(unsigned long long *__restrict__ unique, int n,
                                                  const unsigned *__restrict__ runs, unsigned set_n,
                                                  unsigned *__restrict__ count)
{
    const unsigned r = *runs;
    const unsigned fresh = r > 0 && unique[r - 1] == kGridNone ? r - 1 : r;
    const int i = blockIdx.x * 256 + threadIdx.x;
    // (the read of unique[r - 1] above and the writes below touch the same word only with the same value)
    if (i < n && (unsigned)i >= fresh) unique[i] = kGridNone;
    if (i == 0) *count = set_n + fresh;
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

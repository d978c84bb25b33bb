// nn_mfma.h -- nearest-neighbour search, engine 2: bf16 MFMA coarse pass over ALL
// (query, target) pairs + certified fp64 resolve.  Returns exactly what k_nn_f64 returns
// (the fp64 nearest neighbour in the reference's arithmetic, kdtree.hpp:112-142).
//
// Why bf16 matrix cores.  On gfx950 the per-pair min-tracking VALU work does not hide under
// the MFMA: measured (scripts/micro), time ~ MFMA cycles + VALU cycles.  Ranking of the
// candidates for 1e10 pairs, loop only:
//     v_mfma_f32_16x16x4_f32   (fp32 operands, K = 4)            0.72-0.78 ms
//     v_mfma_f32_16x16x32_bf16 (3 bf16 pieces per coordinate)    0.37-0.42 ms
//     v_mfma_f32_32x32x16_bf16 (2 bf16 pieces per coordinate)    0.27-0.28 ms   <- used
// Each centred fp32 coordinate is cut into two bf16 pieces (8 + 8 significand bits, the
// second rounded to nearest: relative error <= 2^-16) and the contraction carries the 4
// cross products per axis, plus |Q|^2 as three exact pieces against a constant 1:
//     |P-Q|^2 ~ |P|^2 + sum_axis sum_{i,j<2} P_i * (-2 Q_j) + 1 * |Q|^2
// and the leading bf16 piece of |P|^2 against a constant 1 (what is left of |P|^2, < 2^-7 |P|^2,
// is added in the epilogue): K = 12 + 3 + 1 = 16: ONE v_mfma_f32_32x32x16_bf16 per 32 queries x
// 32 targets, every product exact in fp32, fp32 accumulation, two v_min3_f32 per 256 pairs.
//
// Geometry.  Targets are Morton-sorted once per call (the target does not move during the
// ICP loop) and cut into splits of 2048; every split has its own centre c_s and radius
// rho_s, and both operands of a (query block, split) workgroup are expressed about c_s, so
// the representation error scales with (|p-c_s| + rho_s): small for the splits near the
// query, and large only where the distance itself is large.  Lane l / register r of a wave
// keeps the running minimum of query row (r&3)+8(r>>2)+4(l>>5) against column l&31 of every
// tile: a "slot" = 64 targets CONTIGUOUS in the sorted array (tile t, column c of split s is
// sorted position s*2048 + c*64 + t).  No index is tracked in the loop.  The epilogue
// transposes through LDS, adds |P|^2 and writes either (column-tagged min, second min)
// [1-NN] or all 32 column minima [k-NN] per (query, split).
//
// Resolve (k_nn_resolve / k_knn_resolve), fp64 in the reference's operation order: exact
// scan of the winning slot -> D; every target with exact distance <= D has a coarse value
// <= tau_s = D + E_s(D) in its split s, so every slot whose recorded minimum is under its
// split's tau_s is scanned exactly too (a split whose SECOND minimum is under it as a whole:
// scan_split, the slots culled by their bounding boxes).  Result: exact minimum, ties to the lowest ORIGINAL
// index -- bit-identical to k_nn_f64 and to the oracle.
//
// Error bound E_s(d) for a pair in split s.  a >= |p-c_s| + rho_s, u = 2^-24.
//   representation: each centred coordinate is off by <= (2^-16 + 2^-24)|x| (fp32 rounding,
//     then two bf16 pieces); BOTH norms are those of the represented points, so the
//     contraction is |P~ - Q~|^2 of two slightly moved points and, with
//     eps = (2^-16 + 2^-24) a (1 + 1e-6):   | |P~-Q~|^2 - |p-q|^2 | <= eps (2 sqrt(d) + eps)
//   arithmetic: the 16 products are exact; their fp32 accumulation inside the MFMA is not
//     specified, so every one of the <= 18 additions is charged a full truncation
//     (2u x the largest magnitude, <= a^2): 36 u a^2; plus fl32(|Q|^2), fl32(|P|^2) formed
//     with 5 roundings, and the final add: 43 u a^2.
//   column tag (1-NN epilogue): the column number replaces the 5 low mantissa bits of the
//     contraction BEFORE the rest of |P|^2 is added; the contraction already holds the leading
//     piece of |P|^2, so it is within 2^-7 a^2 of the squared distance d: the tag moves it by
//     < 2^-19 (d + 2^-7 a^2) = 2^-19 d + u a^2 / 4; the sum is truncated and tagged again, another
//     2^-19 d (both relative parts, 3.8e-6 together, are inside the 5e-6 inflation of tau).
//   52 u a^2 is used.
//
// Round 3: where a row comes WITH a bound on the distance it looks within, the coarse pass keeps no minima at all -- its
// epilogue (MODE 2) lists the slots under the bound's per-split threshold and the resolve scans the listed slots:
// knn_lists.h (normal estimation: the bound from the target's Morton order) and nn_bounded.h (the ICP loop from its second
// pass on: the bound from each row's previous match; round 4: the first pass too, the bound from the row's place in the target's
// Morton order, nn_culled.h).  The forms described above remain for the stand-alone searches and as the A/B reference
// (ICPMI_NN_BOUNDED=0).
//
// Engine 3 (ICPMI_SEARCH_MFMA_PRUNED; what AUTO takes on targets of more than 16 splits) runs the same coarse unit and
// the bounded resolve on the (64-row group, split) pairs whose boxes are within the group's bound of each other:
// nn_culled.h.  The cull is conservative (strict inequality, margins for its own roundings), so the result is the
// same, ties included.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace icpmi {

constexpr int kTile = 32;                         // queries / targets per MFMA tile
constexpr int kCols = 32;                         // columns (slots) per split
#ifndef ICPMI_SPLIT_TILES
#define ICPMI_SPLIT_TILES 64
#endif
constexpr int kSplitTiles = ICPMI_SPLIT_TILES;    // target tiles per split (build-time tunable)
static_assert(kSplitTiles % 64 == 0, "slots (kSplitTiles targets) are scanned in runs of 64; 32 was tried: slower and not supported");
constexpr int kSplitTargets = kSplitTiles * kTile;// targets per split
constexpr int kSlotTargets = kSplitTiles;         // targets per (split, column) slot
constexpr int kChunkTiles = 32;                   // tiles staged in LDS at a time (32 KiB)
constexpr int kCoarseQT = 2;                      // 32-query tiles per wave
constexpr int kCoarseWaves = 8;                   // waves per workgroup
constexpr int kCoarseThreads = 64 * kCoarseWaves;
constexpr int kCoarseQueries = kTile * kCoarseQT * kCoarseWaves; // queries per workgroup
constexpr float kBig = 3.0e38f;
#ifndef ICPMI_ARITH_BOUND
#define ICPMI_ARITH_BOUND 52.0
#endif
constexpr double kArithBound = ICPMI_ARITH_BOUND; // x u a^2, see above (build-time override: A/B timing only)
constexpr double kReprEps = 1.52587890625e-05 + 5.9604644775390625e-08; // 2^-16 + 2^-24

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct NnFrame {     // bounding box of the whole target (Morton quantisation)
    double lo[3], hi[3];
};
// The Morton-sorted copy of the target is stored SoA -- x[0..m), y[0..m), z[0..m) with the
// component stride `ms` (m rounded up to 64) -- so that a run of consecutive sorted positions
// is read as three fully coalesced streams.
#define ICPMI_SX(sorted, ms, j) (sorted)[(j)]
#define ICPMI_SY(sorted, ms, j) (sorted)[(size_t)(ms) + (j)]
#define ICPMI_SZ(sorted, ms, j) (sorted)[2 * (size_t)(ms) + (j)]

// The coarse minima of the 1-NN pass are two planes of one buffer: x[split][n], fp32 (smallest column
// minimum, tagged with its column), and y[split][n], bf16 (second smallest, ROUNDED DOWN: the upper
// half of the fp32 word of a non-negative value, negative ones stored as 0).  The resolve reads all
// of x (phase 1) and a handful of y (certificate candidates), and y is only ever asked "<= bound?"
// with a non-negative bound, which a value that never exceeds the true one answers with a superset:
// 6 bytes per (query, split) instead of 8.
#define ICPMI_CX(coarse, n, nsplits, s, i) (reinterpret_cast<const float *>(coarse))[(size_t)(s) * (n) + (i)]
#define ICPMI_CY(coarse, n, nsplits, s, i)                                                                                   \
    __uint_as_float((unsigned)(reinterpret_cast<const unsigned short *>(reinterpret_cast<const float *>(coarse) +            \
                                                                        (size_t)(nsplits) * (size_t)(n)))[(size_t)(s) * (n) + (i)] \
                    << 16)
__host__ __device__ inline size_t coarse_bytes(int nsplits, int n)
{
    return (sizeof(float) + sizeof(unsigned short)) * (size_t)nsplits * (size_t)n + 16;
}

struct SplitFrame {  // per split of 2048 sorted targets
    double c[3];     // centre the split's operands are expressed about
    double rho;      // >= max |q - c| over the split (inflated)
    double lo[3], hi[3]; // exact bounding box of the split's targets
};
// ---- bounding box -----------------------------------------------------------------------------
// A target with a non-finite coordinate is never anybody's nearest neighbour (kdtree.hpp:125: no
// `dist_sq < best` holds for an infinite or NaN distance), so the search structures are built
// from the finite targets only (`finite_only`): boxes, centres and radii stay finite, and such
// a target is packed as padding.  The voxel filter keeps every point in its box (and then
// rejects a non-finite one).
__device__ __forceinline__ bool finite3(double x, double y, double z)
{
    return __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
}

__global__ __launch_bounds__(256) void k_bbox_partial(const double *__restrict__ pts, int m,
                                                      double *__restrict__ part /*[grid][6]*/, int finite_only)
{
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        if (finite_only && !finite3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])) continue;
        for (int a = 0; a < 3; ++a) {
            const double v = pts[3 * i + a];
            lo[a] = v < lo[a] ? v : lo[a];
            hi[a] = v > hi[a] ? v : hi[a];
        }
    }
    __shared__ double red[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) {
        double l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) {
            const double l2 = __shfl_down(l, off, 64), h2 = __shfl_down(h, off, 64);
            l = l2 < l ? l2 : l;
            h = h2 > h ? h2 : h;
        }
        if (lane == 0) {
            red[wave][a] = l;
            red[wave][3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        double v = red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? (red[w][a] < v ? red[w][a] : v) : (red[w][a] > v ? red[w][a] : v);
        part[blockIdx.x * 6 + a] = v;
    }
}

__global__ __launch_bounds__(64) void k_bbox_final(const double *__restrict__ part, int nblocks, NnFrame *frame)
{
    const int lane = threadIdx.x;
    for (int a = 0; a < 6; ++a) {
        double v = a < 3 ? 1.7e308 : -1.7e308;
        for (int b = lane; b < nblocks; b += 64) {
            const double x = part[b * 6 + a];
            v = a < 3 ? (x < v ? x : v) : (x > v ? x : v);
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double x = __shfl_down(v, off, 64);
            v = a < 3 ? (x < v ? x : v) : (x > v ? x : v);
        }
        if (lane == 0) (a < 3 ? frame->lo[a] : frame->hi[a - 3]) = v;
    }
}

// Both steps in ONE launch for clouds a single workgroup gets through in a few microseconds (the calls of the
// reference's real callers are chains of such small launches: every one saved is ~4 us of a 0.3-0.5 ms registration).
// `clear3` (may be null): three 64-bit statistics words to zero for the coming call (a hipMemsetAsync less).
constexpr int kBboxSingleMax = 65536;
__global__ __launch_bounds__(1024) void k_bbox_single(const double *__restrict__ pts, int m, NnFrame *frame, int finite_only,
                                                      unsigned long long *__restrict__ clear3)
{
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308};
    // eight points per thread and round, their loads in flight together (as one point per trip the kernel was a chain of
    // m / 1024 memory round trips: 13.9 us for 20k points where the two-launch form took 9)
    constexpr int U = 8;
    for (int base = threadIdx.x; base < m; base += 1024 * U) {
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * 1024, ic = i < m ? i : m - 1; // (clamped: the last point twice changes no box)
            x[u] = pts[3 * ic], y[u] = pts[3 * ic + 1], z[u] = pts[3 * ic + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = !finite_only || finite3(x[u], y[u], z[u]);
            lo[0] = ok && x[u] < lo[0] ? x[u] : lo[0], hi[0] = ok && x[u] > hi[0] ? x[u] : hi[0];
            lo[1] = ok && y[u] < lo[1] ? y[u] : lo[1], hi[1] = ok && y[u] > hi[1] ? y[u] : hi[1];
            lo[2] = ok && z[u] < lo[2] ? z[u] : lo[2], hi[2] = ok && z[u] > hi[2] ? z[u] : hi[2];
        }
    }
    __shared__ double red[16][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double l = lo[a], h = hi[a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double l2 = __shfl_down(l, off, 64), h2 = __shfl_down(h, off, 64);
            l = l2 < l ? l2 : l;
            h = h2 > h ? h2 : h;
        }
        if (lane == 0) {
            red[wave][a] = l;
            red[wave][3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        double v = red[0][a];
        for (int w = 1; w < 16; ++w) v = a < 3 ? (red[w][a] < v ? red[w][a] : v) : (red[w][a] > v ? red[w][a] : v);
        (a < 3 ? frame->lo[a] : frame->hi[a - 3]) = v;
    }
    if (clear3 && threadIdx.x >= 64 && threadIdx.x < 67) clear3[threadIdx.x - 64] = 0ull;
}

// ---- Morton keys, gather, split frames ------------------------------------------------------------
__device__ __forceinline__ unsigned spread10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_morton_keys(const double *__restrict__ pts, int m,
                                                     const NnFrame *__restrict__ frame,
                                                     unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    // one scale for all axes (the largest extent): Morton cells are cubes in space, so 128
    // consecutive sorted points form a compact blob whatever the cloud's aspect ratio
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) ext = frame->hi[a] - frame->lo[a] > ext ? frame->hi[a] - frame->lo[a] : ext;
    unsigned q[3];
    for (int a = 0; a < 3; ++a) {
        double f = ext > 0.0 ? (pts[3 * i + a] - frame->lo[a]) / ext : 0.0;
        f = !(f >= 0.0) ? 0.0 : (f > 1.0 ? 1.0 : f); // (a NaN coordinate sorts with the low corner)
        const int qi = (int)(f * 1023.0);
        q[a] = (unsigned)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
    }
    keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_gather_points(const double *__restrict__ pts,
                                                       const unsigned *__restrict__ perm, int m, int ms,
                                                       double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned j = perm[i];
    ICPMI_SX(out, ms, i) = pts[3 * j];
    ICPMI_SY(out, ms, i) = pts[3 * j + 1];
    ICPMI_SZ(out, ms, i) = pts[3 * j + 2];
}

// The sorted copy a second time as RECORDS behind the three planes (round 4): 32 bytes per target -- x, y, z, original
// index, pad -- so that an exact scan fetches a candidate with two 16-byte loads instead of four (three planes + the
// permutation).  The texture-address unit takes 16 cycles for a wave's load instruction whatever its width
// (TA_TA_BUSY / TA_FLAT_READ_WAVEFRONTS = 15.2 on the bounded resolve), and that unit was the resolve's bound: busy 75 % of
// the kernel with 81 load instructions per wave, 64 of them the scan's.
__host__ __device__ inline size_t sorted_doubles(int ms) { return 7 * (size_t)ms; } // 3 planes + 4 doubles' worth of record per target
__device__ __forceinline__ const uint4 *sorted_records(const double *sorted, int ms) { return reinterpret_cast<const uint4 *>(sorted + 3 * (size_t)ms); }
__global__ __launch_bounds__(256) void k_gather_points_rec(const double *__restrict__ pts, const unsigned *__restrict__ perm, int m, int ms,
                                                           double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned j = perm[i];
    const double x = pts[3 * j], y = pts[3 * j + 1], z = pts[3 * j + 2];
    ICPMI_SX(out, ms, i) = x;
    ICPMI_SY(out, ms, i) = y;
    ICPMI_SZ(out, ms, i) = z;
    uint4 *rec = reinterpret_cast<uint4 *>(out + 3 * (size_t)ms) + 2 * (size_t)i;
    rec[0] = make_uint4((unsigned)__double2loint(x), (unsigned)__double2hiint(x), (unsigned)__double2loint(y), (unsigned)__double2hiint(y));
    rec[1] = make_uint4((unsigned)__double2loint(z), (unsigned)__double2hiint(z), j, 0u);
}

// one workgroup per split: bounding box of its sorted points -> centre and radius; and, behind the
// frames, the bounding box of each of its 32 SLOTS (64 targets contiguous in the sorted array:
// a compact blob), six doubles per slot, which let the resolve cull a whole-split scan down to the
// few slots that can hold a target within the current distance (scan_split).  An empty slot keeps
// lo = +1.7e308 > hi = -1.7e308: infinitely far from every query.
constexpr bool kSlotBoxes = kSlotTargets == 64; // one wave-wide load per slot; other slot sizes scan whole splits
__host__ __device__ inline size_t frames_bytes(int splits)
{
    return sizeof(SplitFrame) * (size_t)splits + (kSlotBoxes ? sizeof(double) * 6 * kCols * (size_t)splits : 0);
}
__global__ __launch_bounds__(256) void k_split_frames(const double *__restrict__ sorted, int m, int ms,
                                                      SplitFrame *__restrict__ frames)
{
    const int s = blockIdx.x;
    const int j0 = s * kSplitTargets, j1 = min(m, j0 + kSplitTargets);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *slot_boxes = reinterpret_cast<double *>(frames + gridDim.x);
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308};
    if (kSlotBoxes) {
        // eight threads per slot: thread t takes targets (t % 8) + 8 i, i = 0..7, of slot t / 8 -- minima
        // and maxima gather in registers and three exchange steps inside the group of eight finish
        // a slot (a wave per slot needed six steps on six doubles for every 64 targets: 2.5x the time)
        static_assert(!kSlotBoxes || kCols * 8 == 256, "eight threads per slot");
        const int c = threadIdx.x >> 3, sub = threadIdx.x & 7;
#pragma unroll
        for (int i = 0; i < kSlotTargets / 8; ++i) {
            const int j = j0 + c * kSlotTargets + sub + 8 * i;
            const bool ok = j < j1 && finite3(ICPMI_SX(sorted, ms, j), ICPMI_SY(sorted, ms, j), ICPMI_SZ(sorted, ms, j));
            for (int a = 0; a < 3; ++a) {
                const double v = ok ? sorted[(size_t)a * ms + j] : 0.0;
                lo[a] = ok && v < lo[a] ? v : lo[a];
                hi[a] = ok && v > hi[a] ? v : hi[a];
            }
        }
        double bl[3] = {lo[0], lo[1], lo[2]}, bh[3] = {hi[0], hi[1], hi[2]};
        for (int a = 0; a < 3; ++a)
            for (int off = 4; off > 0; off >>= 1) {
                const double l2 = __shfl_xor(bl[a], off, 64), h2 = __shfl_xor(bh[a], off, 64);
                bl[a] = l2 < bl[a] ? l2 : bl[a];
                bh[a] = h2 > bh[a] ? h2 : bh[a];
            }
        if (sub < 6) { // (selects, not a run-time index into the arrays: that would put them in scratch)
            const double v = sub == 0 ? bl[0] : sub == 1 ? bl[1] : sub == 2 ? bl[2] : sub == 3 ? bh[0] : sub == 4 ? bh[1] : bh[2];
            slot_boxes[((size_t)s * kCols + c) * 6 + sub] = v;
        }
    } else {
        for (int j = j0 + threadIdx.x; j < j1; j += 256) {
            if (!finite3(ICPMI_SX(sorted, ms, j), ICPMI_SY(sorted, ms, j), ICPMI_SZ(sorted, ms, j))) continue;
            for (int a = 0; a < 3; ++a) {
                const double v = sorted[(size_t)a * ms + j];
                lo[a] = v < lo[a] ? v : lo[a];
                hi[a] = v > hi[a] ? v : hi[a];
            }
        }
    }
    __shared__ double red[4][6];
    for (int a = 0; a < 3; ++a) {
        double l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) {
            const double l2 = __shfl_down(l, off, 64), h2 = __shfl_down(h, off, 64);
            l = l2 < l ? l2 : l;
            h = h2 > h ? h2 : h;
        }
        if (lane == 0) {
            red[wave][a] = l;
            red[wave][3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double h2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            double l = red[0][a], h = red[0][3 + a];
            for (int w = 1; w < 4; ++w) {
                l = red[w][a] < l ? red[w][a] : l;
                h = red[w][3 + a] > h ? red[w][3 + a] : h;
            }
            const bool empty = l > h; // no finite target in this split: everything in it is padding
            frames[s].c[a] = empty ? 0.0 : 0.5 * (l + h);
            frames[s].lo[a] = l;      // (an empty box is infinitely far from every block: always culled)
            frames[s].hi[a] = h;
            const double half = empty ? 0.0 : 0.5 * (h - l);
            h2 += half * half;
        }
        frames[s].rho = sqrt(h2) * (1.0 + 1e-6) + 1e-300;
    }
}

// ---- bf16 pieces of an fp32 value -----------------------------------------------------------------
// exact three-way split: x == h + m + l
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    const unsigned bh = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(bh);  // exact
    const unsigned bm = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(bm); // exact, <= 8 significant bits left
    h = bh >> 16;
    m = bm >> 16;
    l = __float_as_uint(r2) >> 16;
}
// two-way split, second piece rounded to nearest even: |x - (h + m)| <= 2^-16 |x|
__device__ __forceinline__ void split2(float x, unsigned &h, unsigned &m)
{
    const unsigned bh = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(bh); // exact
    const unsigned b1 = __float_as_uint(r1);
    h = bh >> 16;
    m = (b1 + 0x7FFFu + ((b1 >> 16) & 1u)) >> 16;
}

// K-slot k of the contraction (K = 16):
//   k = 4*axis + 2*i + j (k < 12): A = P_axis piece i, B = -2 * Q_axis piece j
//   k = 12..14: A = 1, B = piece (k-12) of fl32(|Q|^2);  k = 15: A = leading bf16 piece of |P|^2, B = 1

// ---- targets -> bf16 B operands: Bpack[(s*64 + t)*64 + lane] = 8 bf16 (k = 8*(lane>>5)+e) ----------
// tile t, column c = lane&31 of split s is sorted position s*2048 + c*64 + t
__global__ __launch_bounds__(256) void k_pack_targets(const double *__restrict__ sorted, int m, int ms,
                                                      const SplitFrame *__restrict__ frames,
                                                      uint4 *__restrict__ Bpack, int splits)
{
    const int gidx = blockIdx.x * 256 + threadIdx.x;
    if (gidx >= splits * kSplitTiles * 64) return;
    const int lane = gidx & 63, t = (gidx >> 6) & (kSplitTiles - 1), s = gidx / (kSplitTiles * 64);
    const int col = lane & 31, half = lane >> 5;
    const long j = (long)s * kSplitTargets + col * kSlotTargets + t;
    unsigned piece[3][2], np[3];
    if (j < m && finite3(ICPMI_SX(sorted, ms, j), ICPMI_SY(sorted, ms, j), ICPMI_SZ(sorted, ms, j))) {
        double n2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const float q = (float)(sorted[(size_t)a * ms + j] - frames[s].c[a]);
            split2(q, piece[a][0], piece[a][1]);
            // the norm must be that of the REPRESENTED point (h + m), or the cross-term error
            // 2 (dP.Q + P.dQ) ~ 2^-16 a^2 would not cancel
            const double qt = (double)__uint_as_float(piece[a][0] << 16) + (double)__uint_as_float(piece[a][1] << 16);
            n2 += qt * qt;
        }
        split3((float)n2, np[0], np[1], np[2]);
    } else { // padding (past the end, or a non-finite target): never the minimum
        for (int a = 0; a < 3; ++a) piece[a][0] = piece[a][1] = 0u;
        split3(kBig, np[0], np[1], np[2]);
    }
    unsigned w[8];
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * half + e;
        unsigned v = 0u;
        if (k < 12) {
            const unsigned b = piece[k >> 2][k & 1];
            v = __float_as_uint(-2.0f * __uint_as_float(b << 16)) >> 16; // exact
        } else if (k < 15) {
            v = np[k - 12];
        } else {
            v = 0x3f80u; // 1.0: carries the leading piece of |P|^2 (also for padding: kBig swallows it)
        }
        w[e] = v & 0xFFFFu;
    }
    Bpack[gidx] = make_uint4(w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16));
}

// ---- coarse pass ---------------------------------------------------------------------------------
__device__ __forceinline__ float min3f(float a, float b, float c)
{
    return __builtin_fminf(__builtin_fminf(a, b), c); // -> v_min3_f32
}

// Raw single-instruction forms for the epilogue: the C-level fminf / fmaxf on values that went
// through integer bit operations make the compiler add a canonicalising v_max_f32 v, v, v in
// front of every use (it must quiet a signalling NaN it cannot rule out): 16 more VALU per tile.
// IEEE mode: v_min_f32 returns the other operand for a NaN, v_med3_f32 the minimum of the rest.
__device__ __forceinline__ float min_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float med3_raw(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// the 5 low mantissa bits of x replaced by `tag` (< 32): one v_and_or_b32
__device__ __forceinline__ float tag_low5(float x, unsigned tag)
{
    return __uint_as_float((__float_as_uint(x) & 0xFFFFFFE0u) | tag);
}

// MODE 0: 1-NN epilogue -> coarse[split][n] = (tagged min, second min over the 32 columns)
// MODE 1: k-NN epilogue  -> slotmin[query][split*32 + column], every column minimum kept (bf16, rounded down)
// MODE 2: epilogue for rows that come WITH a bound on the distance they look within (normal estimation, knn_lists.h; the
//         ICP loop from its second pass on, nn_bounded.h): the columns whose minimum is <= tau_s(bound) are LISTED per
//         row, 16 columns to a word; nothing else is written.  The threshold is formed per (row, SPLIT) from the row's
//         bound and the split's own frame term: the bound that holds for every split carries 52 u a^2 with a = the
//         whole target's extent, which on a 100 m cloud is a third of a nearest-neighbour distance squared and listed a
//         second slot for a third of the rows.
// QT = 32-query tiles per wave (even); WAVES = waves per workgroup.  (Tried and dropped, see
// scripts/micro/README.md: issuing a tile's min3 one tile behind its MFMAs, keeping the next B chunk
// in flight in registers, QT = 4: none beat this form, all cost occupancy.)
//
// The three parts of a wave's work on 32*QT queries against one split, shared by the kernels below.

// MODE 2's output: per row a count and up to `cap` words (split << 17 | half << 16 | mask of the 16 columns
// split*32 + half*16 + bit whose minimum is under the threshold); rows are numbered from the launch's first row.
constexpr int kKnnEntCap = 32;
struct KnnLists {
    const float *thr; // [rows] the row's distance bound, fp32 rounded up
    const float *sq;  // [rows] its square root, rounded up
    int *cnt;         // [rows] words appended (may exceed `cap`: the row then takes its reader's exhaustive path)
    unsigned *ent;    // [rows][cap]
    int cap;          // words per row: kKnnEntCap (normal estimation), kNnEntCap (the ICP loop's bounded 1-NN pass)
};
constexpr int kNnEntCap = 8;

// A operands.  Each lane builds the 16-slot bf16 row of ONE query (lane-per-query: coalesced fp64
// loads, pieces computed once), rows go through `rows` (64 rows x 32 B of LDS private to the
// wave), and every lane picks up its fragment: row l&31 of the tile, slots 8*(l>>5) .. +7.
// QSOA: the queries are an SoA array (x[0..n) | y | z with component stride `qstride`, the layout of
// the sorted target) instead of the rows of an N x 3 array.
// `q0b` >= 0 (QT == 2): the wave's second 32-row tile starts at row q0b instead of q0 + 32 (the culled engine hands a
// wave any two tiles of a split's list, nn_culled.h).
template <int QT, bool QSOA>
__device__ __forceinline__ void coarse_build_a(uint4 *rows, const int lane, const int q0,
                                               const double *__restrict__ qry, const int n, const size_t qstride,
                                               const double c0, const double c1, const double c2,
                                               bf16x8 (&afrag)[QT], float (&pn)[(QT + 1) / 2], float (&p2)[(QT + 1) / 2],
                                               const int q0b = -1)
{
    static_assert(QT == 1 || QT % 2 == 0, "operands are staged 64 queries at a time (QT = 1: 32, by both half-waves)");
#pragma unroll
    for (int gq = 0; gq < (QT + 1) / 2; ++gq) {
        const int ql = QT == 1 ? (lane & 31) : lane; // QT = 1: lanes 32.. repeat the rows of lanes 0..31
        const int ir = (QT == 2 && q0b >= 0 && lane >= 32) ? q0b + (lane - 32) : q0 + gq * 64 + ql;
        const int iq = ir < n ? ir : n - 1;
        const float px = (float)((QSOA ? qry[iq] : qry[3 * iq]) - c0),
                    py = (float)((QSOA ? qry[qstride + iq] : qry[3 * iq + 1]) - c1),
                    pz = (float)((QSOA ? qry[2 * qstride + iq] : qry[3 * iq + 2]) - c2);
        unsigned xh, xm, yh, ym, zh, zm;
        split2(px, xh, xm);
        split2(py, yh, ym);
        split2(pz, zh, zm);
        {   // |P|^2 of the REPRESENTED point (h + m: exact in fp32), see k_pack_targets
            const float tx = __uint_as_float(xh << 16) + __uint_as_float(xm << 16);
            const float ty = __uint_as_float(yh << 16) + __uint_as_float(ym << 16);
            const float tz = __uint_as_float(zh << 16) + __uint_as_float(zm << 16);
            pn[gq] = (tx * tx + ty * ty) + tz * tz;
            p2[gq] = pn[gq]; // |P|^2 whole (MODE 2's frame term)
        }
        // leading piece of |P|^2 (truncated: exact difference) goes through the matrix core,
        // the epilogue adds the rest
        const unsigned pnh = __float_as_uint(pn[gq]) >> 16;
        pn[gq] -= __uint_as_float(pnh << 16);
        const unsigned one = 0x3f80u;
        // slots: x: h h m m, y: h h m m | z: h h m m, 1 1 1 |P|^2
        uint4 u0 = make_uint4(xh | (xh << 16), xm | (xm << 16), yh | (yh << 16), ym | (ym << 16));
        uint4 u1 = make_uint4(zh | (zh << 16), zm | (zm << 16), one | (one << 16), one | (pnh << 16));
        // The MFMA's A fragment of tile t: lane l < 32 holds the first half (u0) of row 32 t + l, lane l >= 32 the second
        // (u1) of row 32 t + l - 32.  Row r was just formed by lane r, so tile 0 = [u0 of the lower half-wave | u1 of the
        // lower half-wave], tile 1 = [u0 of the upper | u1 of the upper]: exactly what v_permlane32_swap leaves in its two
        // operands (it exchanges the first's upper half with the second's lower half).  Until round 4 the rows went
        // through the workgroup's LDS buffer and back -- which also kept that buffer from being filled meanwhile.
        (void)rows;
        if (QT == 1) { // (both half-waves formed the same 32 rows: each lane keeps the half it needs)
            afrag[0] = __builtin_bit_cast(bf16x8, lane < 32 ? u0 : u1);
        } else {
#define ICPMI_SWAP_HALVES(A, B) { const auto r_ = __builtin_amdgcn_permlane32_swap((A), (B), false, false); (A) = r_[0]; (B) = r_[1]; }
            ICPMI_SWAP_HALVES(u0.x, u1.x) ICPMI_SWAP_HALVES(u0.y, u1.y) ICPMI_SWAP_HALVES(u0.z, u1.z) ICPMI_SWAP_HALVES(u0.w, u1.w)
#undef ICPMI_SWAP_HALVES
            afrag[gq * 2 + 0] = __builtin_bit_cast(bf16x8, u0);
            afrag[gq * 2 + 1] = __builtin_bit_cast(bf16x8, u1);
        }
    }
}

// NT target tiles (1 KiB each, at `tiles`) against the wave's QT query tiles: one MFMA per (query
// tile, target tile), running minimum per (row, column) with one v_min3_f32 per two results.
// Fully unrolled: immediate LDS offsets.
template <int QT, int NT>
__device__ __forceinline__ void coarse_tiles(const uint4 *tiles, const int lane, const bf16x8 (&afrag)[QT],
                                             f32x16 (&m)[QT], const f32x16 &zero)
{
#pragma unroll
    for (int tt = 0; tt < NT; tt += 2) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, tiles[tt * 64 + lane]);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, tiles[(tt + 1) * 64 + lane]);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[t], b0, zero, 0, 0, 0);
            const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[t], b1, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) m[t][r] = min3f(m[t][r], da[r], db[r]);
        }
    }
}

// Epilogue.  Transpose through `sc` (32 x 36 floats of LDS private to the wave; row stride 36:
// conflict-free ds_read_b128), one 32-query tile at a time: lane l then owns query l&31 and
// columns 16*(l>>5).. +15; + |P|^2, per-lane top-2, the two halves merge with one cross-lane step.
template <int MODE, int QT>
__device__ __forceinline__ void coarse_epilogue(float *sc, const int lane, const int q0, const int s, const int nsplits,
                                                const int n, const f32x16 (&m)[QT], const float (&pn)[(QT + 1) / 2],
                                                float2 *__restrict__ coarse, float *__restrict__ slotmin,
                                                const KnnLists &kl, const float (&thr)[(QT + 1) / 2], const int q0b = -1)
{
    const int ql = lane & 31, half = lane >> 5;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float thr_q = 0.f;
        // (formed lane-per-query like pn: the value of lane (t & 1) * 32 + ql = the lane's own or its partner's in the other
        // half-wave -- a v_permlane32_swap, not an LDS crossbar trip
        // -- fetched by EVERY lane before the select: inside one arm of a conditional expression the swap would run with
        // half the wave masked off)
        const float thr_o = MODE == 2 ? lane_xor<32>(thr[t >> 1]) : 0.f;
        if (MODE == 2) thr_q = half == (t & 1) ? thr[t >> 1] : thr_o;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + ql] = m[t][r];
        __builtin_amdgcn_wave_barrier();
        float v[16];
        const float pn_o = lane_xor<32>(pn[t >> 1]);
        const float pnq = half == (t & 1) ? pn[t >> 1] : pn_o;
        {
            const float4 *rowp = reinterpret_cast<const float4 *>(sc + ql * 36 + half * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 x = rowp[e];
                v[4 * e] = x.x;
                v[4 * e + 1] = x.y;
                v[4 * e + 2] = x.z;
                v[4 * e + 3] = x.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int iq = (QT == 2 && q0b >= 0 && t == 1) ? q0b + ql : q0 + t * 32 + ql;
        if (MODE == 2) {
            // Nearly every (row, split) pair has nothing under the row's bound: one minimum over the lane's 16
            // columns, one compare, one ballot.  (+Inf / NaN of a far-away or NaN row become kBig like in MODE 1;
            // such a row's bound is FLT_MAX, so it lists everything and is handed to the exact kernel.)
            float mn = min3f(v[0], v[1], v[2]);
#pragma unroll
            for (int c = 3; c < 15; c += 2) mn = min3f(mn, v[c], v[c + 1]);
            mn = min_raw(mn, v[15]);
            const bool pass = iq < n && min_raw(mn + pnq, kBig) <= thr_q;
            if (__ballot(pass) != 0ull) {
                if (pass) {
                    unsigned mask = 0u;
#pragma unroll
                    for (int c = 0; c < 16; ++c) mask |= min_raw(v[c] + pnq, kBig) <= thr_q ? (1u << c) : 0u;
#if defined(ICPMI_TIMING_NO_ATOMIC) /* timing experiment only (WRONG results): what the returning atomic costs */
                    const int pos = 0;
#else
                    const int pos = atomicAdd(kl.cnt + iq, 1);
#endif
                    if (pos < kl.cap) kl.ent[(size_t)iq * kl.cap + pos] = ((unsigned)s << 17) | ((unsigned)half << 16) | mask;
                }
            }
        } else if (MODE == 1) {
            if (iq < n) {
                // stored as bf16 ROUNDED DOWN (the upper half of the fp32 word of a non-negative value):
                // half the bytes of the one buffer that grows with rows x slots; a stored minimum never
                // exceeds the true one, so every "<= bound" test on it stays a superset (k_knn_resolve
                // inflates the one place that needs an upper bound).  A far-away or NaN row's +Inf / NaN
                // become kBig, tiny negative values (coincident points) zero.
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = max_raw(min_raw(v[2 * e] + pnq, kBig), 0.f), b = max_raw(min_raw(v[2 * e + 1] + pnq, kBig), 0.f);
                    w[e] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
                }
                uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(slotmin) +
                                                       (size_t)iq * (nsplits * kCols) + s * kCols + half * 16);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        } else {
            // top-2 of this lane's 16 columns on the tagged contraction values (3 VALU per value:
            // tag, minimum, median = second smallest of {v1 <= v2, x}); |P|^2 is the same for every
            // column of a query and fl(x + p) is monotone in x, so it is added to the two survivors only
            float v1 = kBig, v2 = kBig;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float x = tag_low5(v[c], (unsigned)c);
                v2 = med3_raw(v1, v2, x);
                v1 = min_raw(v1, x);
            }
            const unsigned col = (__float_as_uint(v1) & 15u) | ((unsigned)half << 4);
            // a query so far away that its squares overflow fp32 (or with a NaN coordinate) ends up
            // with +Inf or NaN here: recorded as kBig, which every bound that saturates at FLT_MAX
            // still covers, so that the resolve falls back to exhaustive scans instead of missing slots
            v1 = tag_low5(min_raw(__uint_as_float(__float_as_uint(v1) & 0xFFFFFFE0u) + pnq, kBig), col);
            v2 = min_raw(__uint_as_float(__float_as_uint(v2) & 0xFFFFFFE0u) + pnq, kBig);
            const float o1 = lane_xor<32>(v1), o2 = lane_xor<32>(v2);
            const float hi = __builtin_fmaxf(v1, o1);
            v1 = __builtin_fminf(v1, o1);
            v2 = min3f(hi, v2, o2);
            if (half == 0 && iq < n) {
                float *plane = reinterpret_cast<float *>(coarse);
                plane[(size_t)s * n + iq] = v1;
                unsigned short *yplane = reinterpret_cast<unsigned short *>(plane + (size_t)nsplits * (size_t)n);
                yplane[(size_t)s * n + iq] = (unsigned short)(__float_as_uint(max_raw(v2, 0.f)) >> 16); // bf16, rounded down
            }
        }
    }
}

// One (query block, split) unit of the coarse pass: 512 queries against 2048 targets, the targets
// staged through one 32 KiB LDS buffer in two chunks.
// `q0`: the wave's first row; `active` (wave-uniform): a wave without rows of its own (the culled engine's last chunk
// of a split's list, nn_culled.h) still stages operands and meets the barriers, and does nothing else.
template <int MODE, int QT, int WAVES, bool QSOA = false>
__device__ __forceinline__ void coarse_unit_rows(uint4 *lds, const int q0, const bool active, const int s, const int nsplits,
                                                 const double *__restrict__ qry, const int n, const size_t qstride,
                                                 const uint4 *__restrict__ Bpack,
                                                 const SplitFrame *__restrict__ frames,
                                                 float2 *__restrict__ coarse, float *__restrict__ slotmin,
                                                 const KnnLists kl = KnnLists{nullptr, nullptr, nullptr, nullptr, 0}, const int bx = 0,
                                                 const int q0b = -1 /* >= 0: the wave's second tile starts there (coarse_build_a) */,
                                                 const bool early_stage = true /* the first chunk of operands requested before the rows */)
{
    constexpr int THREADS = 64 * WAVES;
    constexpr int CHUNK16 = kChunkTiles * 64;  // uint4 per staged chunk (32 KiB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef ICPMI_COARSE_CLOCKS /* diagnostic build only (scripts/coarse_clock.py): the clock the chip holds inside this
                              kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz; the 1-NN pass gets a stamp buffer
                              through the otherwise unused `slotmin` argument */
    unsigned long long ck0 = 0, rk0 = 0;
    if (MODE == 0 && slotmin && threadIdx.x == 0) {
        ck0 = __builtin_amdgcn_s_memtime();
        rk0 = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // B operands: 2 chunks of 32 tiles through one 32 KiB LDS buffer, by LDS DMA (global_load_lds_dwordx4: a wave's 64
    // 16-byte pieces land at consecutive LDS slots, no register in between).  With `early_stage` the FIRST chunk is requested
    // here, before the rows are even loaded -- it depends on the split alone, and since the A operands no longer pass through
    // this buffer (coarse_build_a) nothing else needs it: as the code stood -- rows, A operands, barrier, THEN the chunk's
    // loads into registers and on to LDS -- a unit began with two dependent round trips.  Same box, C3 (scripts/ab_kernels.sh):
    // the grid kernels (one unit per workgroup) 299.3 -> 291.7 us with it, 296-299 with the DMA behind the A operands, 299.7
    // with the permlane A build alone; the culled kernel (a workgroup walks over several chunks) 37.1 -> 38.4 with it and
    // 37.2 with the DMA behind the A operands: k_nn_coarse_groups passes false.
#if defined(ICPMI_TIMING_B0) /* timing experiment only (WRONG results): every unit reads split 0's operands -- what L2 misses on them cost */
    const uint4 *src = Bpack;
#else
    const uint4 *src = Bpack + (size_t)s * (kSplitTiles * 64);
#endif
#ifndef ICPMI_COARSE_DMA
#define ICPMI_COARSE_DMA 1 /* 1: LDS DMA; 0: through registers (A/B) */
#endif
#ifndef ICPMI_COARSE_DMA_EARLY
#define ICPMI_COARSE_DMA_EARLY 1 /* 1: the first chunk requested before the rows; 0: behind the A operands as before (A/B) */
#endif
    auto stage = [&](const int chunk) {
#pragma unroll
        for (int e = 0; e < CHUNK16 / THREADS; ++e) {
            if (ICPMI_COARSE_DMA)
                __builtin_amdgcn_global_load_lds(src + (size_t)chunk * CHUNK16 + threadIdx.x + e * THREADS, lds + e * THREADS + wave * 64, 16, 0, 0);
            else lds[threadIdx.x + e * THREADS] = src[(size_t)chunk * CHUNK16 + threadIdx.x + e * THREADS];
        }
    };
    if (ICPMI_COARSE_DMA_EARLY && early_stage) stage(0);
    const double c0 = frames[s].c[0], c1 = frames[s].c[1], c2 = frames[s].c[2];

    bf16x8 afrag[QT];
    float pn[(QT + 1) / 2] = {}, p2[(QT + 1) / 2] = {}, thr[(QT + 1) / 2];
    if (active) coarse_build_a<QT, QSOA>(lds + wave * (64 * 2), lane, q0, qry, n, qstride, c0, c1, c2, afrag, pn, p2, q0b);
#pragma unroll
    for (int gq = 0; gq < (QT + 1) / 2; ++gq) {
        thr[gq] = 0.f;
        if (MODE == 2 && active) {
            // tau_s(d) of nn_mfma.h's header for d = the row's bound, in fp32 with every input rounded up: a >= |p - c_s| + rho_s
            // from the represented point (within 2^-16 of the true one) and the split's radius, 1e-4 over; the last factor
            // covers this evaluation's own roundings and tau_from_a's 5e-6.  A NaN bound (a row with a non-finite coordinate)
            // gives a NaN threshold, under which nothing is; an infinite one (no previous match) lists everything.
            const int ql = QT == 1 ? (lane & 31) : lane;
            const int ir = (QT == 2 && q0b >= 0 && lane >= 32) ? q0b + (lane - 32) : q0 + gq * 64 + ql;
            const int iq = ir < n ? ir : n - 1;
            const float ubf = kl.thr[iq], sqf = kl.sq[iq];
            const float a = (__builtin_amdgcn_sqrtf(p2[gq]) + (float)frames[s].rho) * 1.0001f;
            const float eps = 1.5260e-05f * a;
            thr[gq] = ((ubf + eps * (2.f * sqf + eps)) + (float)(kArithBound * 5.9604644775390625e-08) * (a * a)) * 1.00002f;
        }
    }
    f32x16 m[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[t][r] = kBig;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;

#if defined(ICPMI_COARSE_PRIO) && ICPMI_COARSE_PRIO == 1 /* A/B: static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4) */
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
#elif defined(ICPMI_COARSE_PRIO) && ICPMI_COARSE_PRIO == 2
    if (wave & 1) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll 1
    for (int chunk = 0; chunk < kSplitTiles / kChunkTiles; ++chunk) {
        if (chunk > 0 || !(ICPMI_COARSE_DMA_EARLY && early_stage)) {
            if (chunk > 0) __syncthreads(); // the previous chunk is no longer needed
            stage(chunk);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces have landed
        __syncthreads();
        if (active) coarse_tiles<QT, kChunkTiles>(lds, lane, afrag, m, zero);
    }

    __syncthreads(); // every wave is done with the B operands
#if defined(ICPMI_TIMING_SKIP_EPILOGUE) /* timing experiment only (WRONG results) */
    if (active && lane == 99)
#else
    if (active)
#endif
        coarse_epilogue<MODE, QT>(reinterpret_cast<float *>(lds) + wave * (32 * 36), lane, q0, s, nsplits, n, m, pn, coarse,
                                  slotmin, kl, thr, q0b);
#ifdef ICPMI_COARSE_CLOCKS
    if (MODE == 0 && slotmin && threadIdx.x == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(slotmin) + 4 * ((size_t)bx * nsplits + s);
        o[0] = ck0;
        o[1] = rk0;
        o[2] = __builtin_amdgcn_s_memtime();
        o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// One (query block, split) unit: wave w of block bx takes rows (bx * WAVES + w) * 32 QT ...
template <int MODE, int QT, int WAVES, bool QSOA = false>
__device__ __forceinline__ void coarse_unit(uint4 *lds, const int bx, const int s, const int nsplits,
                                            const double *__restrict__ qry, const int n, const size_t qstride,
                                            const uint4 *__restrict__ Bpack,
                                            const SplitFrame *__restrict__ frames,
                                            float2 *__restrict__ coarse, float *__restrict__ slotmin,
                                            const KnnLists kl = KnnLists{nullptr, nullptr, nullptr, nullptr, 0})
{
    coarse_unit_rows<MODE, QT, WAVES, QSOA>(lds, (bx * WAVES + (int)(threadIdx.x >> 6)) * (kTile * QT), true, s, nsplits, qry, n, qstride,
                                            Bpack, frames, coarse, slotmin, kl, bx);
}

template <int WAVES>
struct CoarseLds {
    static constexpr int CHUNK16 = kChunkTiles * 64;
    static constexpr int EPI16 = 32 * 36 / 4;  // uint4 per wave for the epilogue transpose
    static constexpr int SCRATCH16 = WAVES * EPI16 > CHUNK16 ? WAVES * EPI16 : CHUNK16;
};

// all pairs: grid (query blocks, splits)
template <int MODE, int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_coarse(
    const double *__restrict__ qry, int n, const uint4 *__restrict__ Bpack,
    const SplitFrame *__restrict__ frames, float2 *__restrict__ coarse /*[split][n]*/,
    float *__restrict__ slotmin /*[n][splits*32]*/, const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    __shared__ uint4 lds[CoarseLds<WAVES>::SCRATCH16];
    coarse_unit<MODE, QT, WAVES>(lds, blockIdx.x, blockIdx.y, gridDim.y, qry, n, 0, Bpack, frames, coarse, slotmin);
}

// all pairs, rows = `n` consecutive positions of the Morton-sorted target itself (SoA, component stride
// `qstride`), MODE 2: the k-NN pass of normal estimation (knn_lists.h)
template <int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_coarse_rows(
    const double *__restrict__ rows, int n, size_t qstride, const uint4 *__restrict__ Bpack,
    const SplitFrame *__restrict__ frames, KnnLists kl)
{
    __shared__ uint4 lds[CoarseLds<WAVES>::SCRATCH16];
    coarse_unit<2, QT, WAVES, true>(lds, blockIdx.x, blockIdx.y, gridDim.y, rows, n, qstride, Bpack, frames, nullptr, nullptr, kl);
}

// all pairs, the moved source rows of an ICP pass that come with a bound (nn_bounded.h), MODE 2
template <int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_coarse_bounded(
    const double *__restrict__ qry, int n, const uint4 *__restrict__ Bpack,
    const SplitFrame *__restrict__ frames, KnnLists kl, const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    __shared__ uint4 lds[CoarseLds<WAVES>::SCRATCH16];
    coarse_unit<2, QT, WAVES>(lds, blockIdx.x, blockIdx.y, gridDim.y, qry, n, 0, Bpack, frames, nullptr, nullptr, kl);
}

// (Measured and not kept, scripts/micro/README.md: a RESIDENT form -- one 16-wave workgroup per CU stages a
// split's 64 KiB of operands once and walks over blocks of 1,024 queries with no barrier after the
// staging, A rows and epilogue transposes in wave-private LDS.  315 us against 311 us on C3: the
// per-unit form's staging phases and barriers are not what its time goes to.)

// (The culled engine's kernels -- box test per 64-row group, coarse pass over the surviving (group, split) pairs -- are in
// nn_culled.h.)

// ---- resolve ---------------------------------------------------------------------------------------
__device__ __forceinline__ double px_sel(int a, double x, double y, double z) { return a == 0 ? x : (a == 1 ? y : z); }

// bound on the coarse value of any target whose exact distance is <= d, in a split whose frame
// term is `a` (>= |p - c_s| + rho_s)
__device__ __forceinline__ float tau_from_a(double a, double d, double sqrt_d)
{
    const double u = 5.9604644775390625e-08; // 2^-24
    const double eps = kReprEps * a * (1.0 + 1e-6);
    double tau = d + eps * (2.0 * sqrt_d + eps) + kArithBound * u * a * a;
    tau = tau * (1.0 + 5e-6) + 1e-300; // the two column tags (2 x 2^-19 relative) + slack for this fp64 evaluation
    if (!(tau < 3.0e38)) return 3.4028235e38f; // beyond fp32 (or NaN): every recorded value is inside the bound
    return __uint_as_float(__float_as_uint((float)tau) + 1u); // round up (tau > 0)
}

// the same bound for split `f`
__device__ __forceinline__ float split_tau(double px, double py, double pz, const SplitFrame &f, double d,
                                           double sqrt_d)
{
    const double dx = px - f.c[0], dy = py - f.c[1], dz = pz - f.c[2];
    // |p - c| only has to be an upper bound good to ~1e-6: fp32 square root (v_sqrt_f32, 1 ulp;
    // the conversion adds 2^-24) under the 1e-6 inflation instead of a ~35-instruction fp64 one
    const double a = (double)__builtin_amdgcn_sqrtf((float)((dx * dx + dy * dy) + dz * dz)) * (1.0 + 1e-6) + f.rho;
    return tau_from_a(a, d, sqrt_d);
}

// ... and one that holds for EVERY split: a split's centre lies inside the bounding box of the
// whole target (it is the centre of the box of a subset), so |p - c_s| is at most the distance
// from p to the farthest corner of that box, and rho_s at most its half diagonal.  tau is
// increasing in a, hence tau_s(d) <= this value for all s: a recorded minimum above it needs
// no look at its split's frame (the resolve's cheap first filter).
__device__ __forceinline__ float all_splits_tau(double px, double py, double pz, const NnFrame &g, double d,
                                                double sqrt_d)
{
    double f2 = 0.0, h2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double e1 = fabs(px_sel(a, px, py, pz) - g.lo[a]), e2 = fabs(px_sel(a, px, py, pz) - g.hi[a]);
        const double e = e1 > e2 ? e1 : e2;
        f2 += e * e;
        const double half = 0.5 * (g.hi[a] - g.lo[a]);
        h2 += half * half;
    }
    // (1 + 1e-5): covers the 1e-6 inflations and the fp32 square root on split_tau's side
    const double a = ((double)__builtin_amdgcn_sqrtf((float)f2) + (double)__builtin_amdgcn_sqrtf((float)h2)) * (1.0 + 1e-5) + 1e-300;
    return tau_from_a(a, d, sqrt_d);
}

// wave-wide argmin of (d, j): smaller d, then smaller j; result valid in every lane
__device__ __forceinline__ void wave_argmin(double &d, int &j)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(d, off, 64);
        const int oj = __shfl_xor(j, off, 64);
        if (od < d || (od == d && oj < j)) {
            d = od;
            j = oj;
        }
    }
}

// exact scan of sorted positions [j0, j0+LEN) for the query (px,py,pz), all lanes cooperate;
// (bd, bj) is updated with the smaller (distance, ORIGINAL index).  The loads of up to MAXBATCH
// rounds are in flight together (7 registers per round: the caller says what it can afford;
// clamped addresses, the bound applied to the comparison only): a
// whole split is 32 rounds, and as a chain of 32 dependent round trips one such scan was the tail
// of the whole resolve kernel on LiDAR frames (a few dozen queries per pass need one).
template <int LEN, int MAXBATCH>
__device__ __forceinline__ void scan_range(const double *__restrict__ sorted,
                                           const unsigned *__restrict__ perm, int m, int ms, int j0,
                                           double px, double py, double pz, int lane, double &bd, int &bj)
{
    static_assert(LEN % 64 == 0, "whole rounds of the wave");
    constexpr int ROUNDS = LEN / 64, BATCH = ROUNDS < MAXBATCH ? ROUNDS : MAXBATCH;
    double d = 1.7976931348623157e308;
    int j = 0x7fffffff;
#pragma unroll 1
    for (int r0 = 0; r0 < ROUNDS; r0 += BATCH) {
        double x[BATCH], y[BATCH], z[BATCH];
        int oj[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int jj = j0 + (r0 + u) * 64 + lane;
            const int jc = jj < m ? jj : m - 1;
            x[u] = ICPMI_SX(sorted, ms, jc), y[u] = ICPMI_SY(sorted, ms, jc), z[u] = ICPMI_SZ(sorted, ms, jc);
            oj[u] = (int)perm[jc];
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int jj = j0 + (r0 + u) * 64 + lane;
            const double dd = sqdist(x[u], y[u], z[u], px, py, pz);
            if (jj < m && (dd < d || (dd == d && oj[u] < j))) {
                d = dd;
                j = oj[u];
            }
        }
    }
    wave_argmin(d, j);
    if (d < bd || (d == bd && j < bj)) {
        bd = d;
        bj = j;
    }
}

// exact scan of ALL of split sL for one query, by the whole wave, slots culled by their bounding
// boxes: lane c < 32 forms the squared distance from the query to slot c's box (a lower bound of
// the distance to every target in it; the factor 1 - 1e-9 covers this evaluation's own roundings)
// and the slot is read only if that does not exceed `qbd`, the query's best distance so far --
// a target at distance <= the final minimum <= qbd cannot sit in a culled slot, ties included.
// Morton-contiguous slots are compact: 2-5 of 32 survive, and their loads go out MAXBATCH slots
// at a time.  Without slot boxes (slot sizes other than 64) every slot is read.
template <int MAXBATCH>
__device__ __forceinline__ void scan_split(const double *__restrict__ sorted, const unsigned *__restrict__ perm,
                                           int m, int ms, const double *__restrict__ slot_boxes, int sL,
                                           double qx, double qy, double qz, double qbd, int lane, double &bd, int &bj)
{
    if (!kSlotBoxes) {
        scan_range<kSplitTargets, MAXBATCH>(sorted, perm, m, ms, sL * kSplitTargets, qx, qy, qz, lane, bd, bj);
        return;
    }
    bool keep = false;
    if (lane < kCols) {
        const double *b = slot_boxes + ((size_t)sL * kCols + lane) * 6;
        double lb = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double q = px_sel(a, qx, qy, qz);
            const double g1 = b[a] - q, g2 = q - b[3 + a];
            const double g = g1 > g2 ? (g1 > 0.0 ? g1 : 0.0) : (g2 > 0.0 ? g2 : 0.0);
            lb += g * g;
        }
        keep = !(lb * (1.0 - 1e-9) > qbd);
    }
    unsigned mask = (unsigned)__ballot(keep); // wave-uniform
    double d = 1.7976931348623157e308;
    int j = 0x7fffffff;
    while (mask) {
        double x[MAXBATCH], y[MAXBATCH], z[MAXBATCH];
        int oj[MAXBATCH], jj[MAXBATCH];
#pragma unroll
        for (int u = 0; u < MAXBATCH; ++u) {
            const int c = mask ? __ffs((int)mask) - 1 : -1;
            mask &= mask - 1u; // (0 stays 0)
            jj[u] = c >= 0 ? sL * kSplitTargets + c * kSlotTargets + lane : m;
            const int jc = jj[u] < m ? jj[u] : m - 1;
            x[u] = ICPMI_SX(sorted, ms, jc), y[u] = ICPMI_SY(sorted, ms, jc), z[u] = ICPMI_SZ(sorted, ms, jc);
            oj[u] = (int)perm[jc];
        }
#pragma unroll
        for (int u = 0; u < MAXBATCH; ++u) {
            const double dd = sqdist(x[u], y[u], z[u], qx, qy, qz);
            if (jj[u] < m && (dd < d || (dd == d && oj[u] < j))) {
                d = dd;
                j = oj[u];
            }
        }
    }
    wave_argmin(d, j);
    if (d < bd || (d == bd && j < bj)) {
        bd = d;
        bj = j;
    }
}

// Phase 3 of both resolve kernels: the certificate.  A query's list of evaluated splits is
// shared by GROUP lanes (the 64 / Q sub-lanes of a query in k_nn_resolve<Q>, lane offset `sub`;
// the 16 lanes of a quarter in k_nn_resolve4: GROUP = 16, sub = ql, Q = 4 queries per wave).  First a
// cheap pass: one 4-byte load and one compare against all_splits_tau per (query, split), the
// survivors remembered as bits.  Only they -- the query's own split and the odd neighbour, 1-3
// of 49 on the 100k cloud -- get the per-split bound (frame loads + ~40 fp64 operations), and
// the slots or splits under it are scanned exactly by the whole wave.
template <int GROUP, int Q, int KEEP, int SCANBATCH>
__device__ __forceinline__ void resolve_certify(const float (&pv)[KEEP > 0 ? KEEP : 1], // phase 1's first KEEP values per lane
                                                const int lane, const int sub, const bool valid, const int ic, const int n,
                                                const double px, const double py, const double pz,
                                                const float2 *__restrict__ coarse, const int nsplits,
                                                const int *__restrict__ slist,
                                                const int nact, const SplitFrame *__restrict__ frames,
                                                const NnFrame *__restrict__ gframe, const int bs,
                                                const double *__restrict__ sorted, const unsigned *__restrict__ perm,
                                                const int m, const int ms, double &bd, int &bj,
                                                unsigned &extra_slots, unsigned &extra_splits)
{
    const double sq = sqrt(bd);
#ifdef ICPMI_NO_FIRST_FILTER /* A/B timing only */
    const float tmax = 3.4028235e38f;
#else
    const float tmax = all_splits_tau(px, py, pz, *gframe, bd, sq);
#endif
    // a query with a NaN or infinite coordinate has no neighbour whatever is scanned (kdtree.hpp:125
    // never holds): it takes no part
    const bool look = valid && finite3(px, py, pz);
    for (int base = 0; base < nact; base += 32 * GROUP) { // 32 list entries per lane and round
        unsigned cmask = 0u;
        const int left = nact - base;
        const int kmax = left >= 32 * GROUP ? 32 : (left + GROUP - 1) / GROUP; // wave-uniform
        int k0 = 0;
        if (KEEP > 0 && base == 0) { // the values phase 1 already loaded: no second trip to memory for them
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int e = k * GROUP + sub;
                cmask |= (e < nact && look && pv[k] <= tmax) ? (1u << k) : 0u;
            }
            k0 = KEEP;
        }
        for (int k = k0; k < kmax; ++k) {
            const int e = base + k * GROUP + sub;
            if (e < nact && look) {
                const int s = slist ? slist[e] : e;
                cmask |= ICPMI_CX(coarse, n, nsplits, s, ic) <= tmax ? (1u << k) : 0u;
            }
        }
        while (__ballot(cmask != 0u)) {
            const bool act = cmask != 0u;
            const int k = act ? __ffs((int)cmask) - 1 : 0;
            cmask &= cmask - 1u; // (0 stays 0)
            int s = 0;
            bool whole = false, slot = false;
            float2 v = make_float2(kBig, kBig);
            if (act) {
                const int e = base + k * GROUP + sub;
                s = slist ? slist[e] : e;
                v = make_float2(ICPMI_CX(coarse, n, nsplits, s, ic), ICPMI_CY(coarse, n, nsplits, s, ic));
                const float tauf = split_tau(px, py, pz, frames[s], bd, sq);
                whole = v.y <= tauf;                       // a second column is inside the bound
                slot = !whole && s != bs && v.x <= tauf;
            }
#if defined(ICPMI_TIMING_SKIP_SCANS) /* timing experiment only (WRONG results): 1 = no whole-split scans, 2 = nothing of the
                                        certificate survives (its results are unused) */
            if (ICPMI_TIMING_SKIP_SCANS >= 1) whole = false;
            if (ICPMI_TIMING_SKIP_SCANS >= 2) slot = false;
#endif
            unsigned long long pend = __ballot(whole || slot);
            while (pend) {                                 // rare; wave-uniform loop
                const int L = __ffsll((long long)pend) - 1;
                pend &= pend - 1;
                const double qx = __shfl(px, L, 64), qy = __shfl(py, L, 64), qz = __shfl(pz, L, 64);
                const int w = __shfl((int)whole, L, 64);
                const int c = __shfl((int)(__float_as_uint(v.x) & 31u), L, 64);
                const int sL = __shfl(s, L, 64);
                double d = 1.7976931348623157e308;
                int j = 0x7fffffff;
                if (w)
                    scan_split<SCANBATCH>(sorted, perm, m, ms, reinterpret_cast<const double *>(frames + nsplits), sL, qx, qy, qz,
                                          __shfl(bd, L, 64), lane, d, j);
                else scan_range<kSlotTargets, 1>(sorted, perm, m, ms, sL * kSplitTargets + c * kSlotTargets, qx, qy, qz, lane, d, j);
                // every lane that holds this query takes the result
                // (Q queries per wave laid out as lane % Q, or one query per quarter-wave when Q == 4)
                const bool mine = Q == 4 ? (lane >> 4) == (L >> 4) : (lane & (Q - 1)) == (L & (Q - 1));
                if (mine && (d < bd || (d == bd && j < bj))) {
                    bd = d;
                    bj = j;
                }
                if (lane == L) {
                    if (w) ++extra_splits;
                    else ++extra_slots;
                }
            }
        }
    }
}

#ifndef ICPMI_RESOLVE_WW
#define ICPMI_RESOLVE_WW 4 /* waves per workgroup = Q * WW queries per partial row of normal-equation terms */
#endif
constexpr int kResolveWW = ICPMI_RESOLVE_WW;
// The end of the Q-queries-per-wave resolve kernels (k_nn_resolve, k_nn_resolve_bounded): results out, counters, and the
// fused residual + normal-equation terms.  (jspec, q*, n*): the matched target and normal gathered ahead for target
// `jspec` (< 0: nothing was gathered).
// One lane's share of a slot -- sorted positions j0 + 16 o, o < kSlotTargets / 16 -- requested in ONE batch: the loads of
// all of them are issued before the first is waited for.  (Written as load-then-use per candidate, the compiler kept each
// candidate's four loads next to their use -- fewer live registers -- and a slot became kSlotTargets / 16 dependent memory
// round trips, a wave's four rounds sixteen: with every wave of a C3 pass resident at once the kernel's time IS one wave's
// chain of trips.)  Addresses are a scalar base + a 32-bit byte offset per candidate (positions < 2^27).
struct SlotBatch {
    static constexpr int N = kSlotTargets / 16;
    double x[N], y[N], z[N];
    int oj[N];
    __device__ __forceinline__ void load(const double *__restrict__ sorted, const unsigned *__restrict__ perm, const int m, const int ms,
                                         const int j0)
    {
        // the candidates' RECORDS (nn_mfma.h: x, y, z, original index in 32 bytes): two 16-byte loads each, where the
        // three planes and the permutation are four
        const char *rec = reinterpret_cast<const char *>(sorted_records(sorted, ms));
        (void)perm;
#pragma unroll
        for (int o = 0; o < N; ++o) {
            // (clamped: with a bound beyond the padding's stand-in distance a row lists padding slots too, whose positions
            // lie outside the sorted copy; dropped by j < m in the evaluation)
            const unsigned jj = (unsigned)(j0 + 16 * o), jc = jj < (unsigned)m ? jj : (unsigned)(m - 1);
            const uint4 a = *reinterpret_cast<const uint4 *>(rec + (jc << 5)), b = *reinterpret_cast<const uint4 *>(rec + (jc << 5) + 16);
            x[o] = __hiloint2double((int)a.y, (int)a.x);
            y[o] = __hiloint2double((int)a.w, (int)a.z);
            z[o] = __hiloint2double((int)b.y, (int)b.x);
            oj[o] = (int)b.z;
        }
        __builtin_amdgcn_sched_barrier(0); // (the scheduler moves nothing across: all requested before any is used; the waits stay progressive)
    }
    // (distance, original index) minimum with the candidates of this batch; selects, not branches: written with `if` and
    // short-circuit operators every update of a lane became an exec-mask save, a branch and a restore
    __device__ __forceinline__ void eval(const int m, const int j0, const bool act, const double qx, const double qy, const double qz,
                                         double &d, int &jo) const
    {
#pragma unroll
        for (int o = 0; o < N; ++o) {
            const double dd = sqdist(x[o], y[o], z[o], qx, qy, qz);
            const bool take = act & (j0 + 16 * o < m) & ((dd < d) | ((dd == d) & (oj[o] < jo)));
            d = take ? dd : d;
            jo = take ? oj[o] : jo;
        }
    }
};

template <int Q>
__device__ __forceinline__ void resolve_finish(const int lane, const int wave, const int ql, const bool owner /* ONE lane of each query */, const int i,
                                               const bool valid, const double bd, const int bj, const double px,
                                               const double py, const double pz, const int m, int *__restrict__ idx,
                                               double *__restrict__ d2out, unsigned long long *__restrict__ counters,
                                               const unsigned extra_slots, const unsigned extra_splits,
                                               const double *__restrict__ tgt_orig, const double *__restrict__ nrm,
                                               double *__restrict__ partials, const int jspec, double q0, double q1,
                                               double q2, double n0, double n1, double n2)
{
    if (valid && owner) {
        idx[i] = bj == 0x7fffffff ? -1 : bj; // NaN/Inf query: nothing compares less (kdtree.hpp:53)
        if (d2out) d2out[i] = bd;
    }
    // statistics: a wave's counts; with the normal-equation terms below they meet in LDS first and leave as ONE pair of
    // atomics per workgroup (thousands of waves adding to the same two words were a third of the bounded resolve's time)
    __shared__ unsigned wave_cnt[kResolveWW][2];
    if (counters) {
        unsigned es = extra_slots, ef = extra_splits;
        for (int off = 32; off > 0; off >>= 1) {
            es += __shfl_down(es, off, 64);
            ef += __shfl_down(ef, off, 64);
        }
        if (lane == 0) {
            if (partials) {
                wave_cnt[wave][0] = es;
                wave_cnt[wave][1] = ef;
            } else if (es | ef) {
                atomicAdd(&counters[0], (unsigned long long)es);
                atomicAdd(&counters[1], (unsigned long long)ef);
            }
        }
    }
    // fused residual + normal equations (what k_reduce does, icp.hpp:99-120,198-206): the Q
    // owners of a wave form their J row and b, the 28 sums go wave -> LDS -> one partial row
    // per workgroup, summed later in a fixed order by k_finish_step
    if (partials) {
        // Q rows of 28 terms per wave -> LDS (row stride 29: conflict-free), then lane l sums
        // column l & 31 over the Q / 2 rows of half l >> 5 and the halves meet with one exchange
        __shared__ double jrow[kResolveWW][Q][29];
        __shared__ double red[kResolveWW][28];
        if (owner) { // each term goes to LDS as it is formed (28 live doubles would cost 2 waves per SIMD)
            double J[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, b = 0.0;
            if (valid) {
                const int j = (unsigned)bj < (unsigned)m ? bj : 0;
                if (j != jspec) { // the certificate found a nearer target
                    q0 = tgt_orig[3 * j], q1 = tgt_orig[3 * j + 1], q2 = tgt_orig[3 * j + 2];
                    n0 = nrm[3 * j], n1 = nrm[3 * j + 1], n2 = nrm[3 * j + 2];
                }
                J[0] = py * n2 - pz * n1; // p x n, icp.hpp:105
                J[1] = pz * n0 - px * n2;
                J[2] = px * n1 - py * n0;
                J[3] = n0;
                J[4] = n1;
                J[5] = n2;
                const double e0 = q0 - px, e1 = q1 - py, e2 = q2 - pz;
                b = (e0 * n0 + e1 * n1) + e2 * n2; // icp.hpp:116
            }
            double *row = jrow[wave][ql];
            int o = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) row[o++] = J[r] * J[c];
#pragma unroll
            for (int r = 0; r < 6; ++r) row[21 + r] = J[r] * b;
            row[27] = b * b;
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int c = lane & 31, h = lane >> 5;
            double v = 0.0;
            if (c < 28) {
#pragma unroll
                for (int r = 0; r < Q / 2; ++r) v += jrow[wave][h * (Q / 2) + r][c];
            }
            v += lane_xor<32>(v);
            if (lane < 28) red[wave][lane] = v;
        }
        __syncthreads();
        if (counters && threadIdx.x == 32) {
            unsigned es = 0, ef = 0;
            for (int w = 0; w < kResolveWW; ++w) es += wave_cnt[w][0], ef += wave_cnt[w][1];
            if (es) atomicAdd(&counters[0], (unsigned long long)es);
            if (ef) atomicAdd(&counters[1], (unsigned long long)ef);
        }
        if (threadIdx.x < 28) {
            const int e = threadIdx.x;
            double v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
#pragma unroll
            for (int w = 4; w < kResolveWW; ++w) v += red[w][e];
            partials[(size_t)blockIdx.x * kSumsStride + e] = v;
        }
    }
}

// One wave resolves 16 (or 32) queries.  Lane = (query ql, sub-lane): the sub-lanes share the
// bookkeeping of a query (each looks at its share of the splits) and each quarter-wave scans one
// winning slot at a time (lane l16 takes sorted positions l16, l16+16, ... of the slot: three
// coalesced streams), so 4 slots are in flight per wave.
constexpr int kResolveQ = 16;
// Waves per SIMD the register allocation must allow (A/B knob).  The kernel waits on memory three
// quarters of the time (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.77) and, at 64 queries per workgroup, C3
// has 1,563 workgroups: at 5 per CU (82 VGPRs) 283 of them form a second round.  Forcing 6 / 7
// waves per SIMD (80 / 72 VGPRs, 11 spilled at 7) measured 52 / 54 us against 48 us at 5: kept at 5.
#ifndef ICPMI_RESOLVE_OCC
#define ICPMI_RESOLVE_OCC 5
#endif
#ifndef ICPMI_RESOLVE_KEEP
#define ICPMI_RESOLVE_KEEP 16 /* phase-1 values per lane kept for the certificate (64 splits at 16 queries per wave) */
#endif
// slots of a whole-split scan whose loads are in flight together (scan_split): what each kernel's
// register budget allows
#ifndef ICPMI_RESOLVE_SCANBATCH
#define ICPMI_RESOLVE_SCANBATCH 1
#endif
#ifndef ICPMI_RESOLVE4_SCANBATCH
#define ICPMI_RESOLVE4_SCANBATCH 2
#endif
#ifndef ICPMI_RESOLVE_UNROLL
#define ICPMI_RESOLVE_UNROLL 4
#endif
#ifndef ICPMI_RESOLVE_RUNROLL
#define ICPMI_RESOLVE_RUNROLL 4
#endif

// Q = queries per wave, 16 or 32: lane = (query ql = lane % Q, sub = lane / Q); the 64 / Q sub-lanes
// of a query share its bookkeeping.  Q = 32 halves the waves of a pass (C3: 3,125, all resident
// at once, where the 6,250 of Q = 16 need a second round at 5 waves per SIMD) at the price of a
// longer chain per wave (8 scan rounds instead of 4).
template <int Q>
__global__ __launch_bounds__(64 * kResolveWW) __attribute__((amdgpu_waves_per_eu(ICPMI_RESOLVE_OCC, 8))) void k_nn_resolve(
    const double *__restrict__ qry, int n,
                                                    const double *__restrict__ sorted,
                                                    const unsigned *__restrict__ perm, int m, int ms,
                                                    const float2 *__restrict__ coarse, int splits,
                                                    const SplitFrame *__restrict__ frames,
                                                    const NnFrame *__restrict__ gframe,
                                                    int *__restrict__ idx, double *__restrict__ d2out,
                                                    unsigned long long *__restrict__ counters,
                                                    const double *__restrict__ tgt_orig,
                                                    const double *__restrict__ nrm,
                                                    double *__restrict__ partials,
                                                    const int *__restrict__ blk_cnt,
                                                    const int *__restrict__ blk_list,
                                                    const IcpState *__restrict__ st)
{
    static_assert(Q == 16 || Q == 32, "queries per wave");
    constexpr int SUBS = 64 / Q;   // lanes per query
    constexpr int ROUNDS = Q / 4;  // phase 2: four quarter-waves scan four slots per round
    if (st && st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & (Q - 1), sub = lane / Q;
    const int l16 = lane & 15, quarter = lane >> 4;
    const int qbase = (blockIdx.x * kResolveWW + wave) * Q;
    // pruned engine: only the splits on the query block's list were evaluated (the queries
    // of a wave share a block); otherwise all of them
    static_assert(kCoarseQueries % Q == 0, "a wave's queries share a coarse block");
    const int *slist = blk_list ? blk_list + (size_t)(qbase / kCoarseQueries) * splits : nullptr;
    const int nact = blk_list ? (qbase < n ? blk_cnt[qbase / kCoarseQueries] : 0) : splits;
    const int i = qbase + ql; // waves past the end run on a clamped query and write nothing
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = qry[3 * ic], py = qry[3 * ic + 1], pz = qry[3 * ic + 2];

    // phase 1: smallest coarse value over the splits (each sub-lane takes every SUBS-th split); the
    // first KEEP values of a lane stay in registers for the certificate's first filter
    constexpr int KEEP = ICPMI_RESOLVE_KEEP;
    static_assert(KEEP <= 32, "one bit of the certificate's candidate mask per kept value");
    float pv[KEEP > 0 ? KEEP : 1];
    float best = kBig;
    int bs = 0;
    // (the loads unconditional -- a clamped split, the value dropped by a select -- so that they are all in flight together:
    // with a guard per load the compiler makes each a load-wait-compare of its own, KEEP dependent round trips)
    int ps[KEEP > 0 ? KEEP : 1];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int e = sub + SUBS * k, ec = e < nact ? e : 0;
        ps[k] = slist ? slist[nact > 0 ? ec : 0] : ec;
    }
#pragma unroll
    for (int k = 0; k < KEEP; ++k) pv[k] = ICPMI_CX(coarse, n, splits, ps[k], ic);
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const bool in = sub + SUBS * k < nact;
        pv[k] = in ? pv[k] : kBig;
        const bool take = in & (pv[k] < best);
        best = take ? pv[k] : best;
        bs = take ? ps[k] : bs;
    }
    for (int e = sub + SUBS * KEEP; e < nact; e += SUBS) {
        const int s = slist ? slist[e] : e;
        const float v = ICPMI_CX(coarse, n, splits, s, ic);
        if (v < best) {
            best = v;
            bs = s;
        }
    }
#pragma unroll
    for (int x = Q; x < 64; x <<= 1) {
        const float ov = __shfl_xor(best, x, 64);
        const int os = __shfl_xor(bs, x, 64);
        if (ov < best || (ov == best && os < bs)) {
            best = ov;
            bs = os;
        }
    }
    const int bcol = (int)(__float_as_uint(best) & 31u);

    // phase 2: exact evaluation of the winning slots, one query per quarter-wave and round
    double bd = 1.7976931348623157e308;
    int bj = 0x7fffffff;
#pragma unroll ICPMI_RESOLVE_RUNROLL
    for (int r = 0; r < ROUNDS; ++r) {
        const int src = quarter * ROUNDS + r; // the query this quarter scans now (a lane with sub == 0)
        const double qx = __shfl(px, src, 64), qy = __shfl(py, src, 64), qz = __shfl(pz, src, 64);
        const int s = __shfl(bs, src, 64), c = __shfl(bcol, src, 64);
        const int j0 = s * kSplitTargets + c * kSlotTargets + l16; // lane takes l16, l16+16, ...: coalesced
        double d = 1.7976931348623157e308;
        int j = 0x7fffffff;
        {   // (the slot's candidates requested as one batch, from the 32-byte records: SlotBatch)
            SlotBatch sb;
            sb.load(sorted, perm, m, ms, j0);
            sb.eval(m, j0, true, qx, qy, qz, d, j);
        }
        row16_argmin(d, j);
        // query ql was scanned by quarter ql / ROUNDS in round ql % ROUNDS
        const double rd = __shfl(d, (ql / ROUNDS) * 16, 64);
        const int rj = __shfl(j, (ql / ROUNDS) * 16, 64);
        if ((ql % ROUNDS) == r) {
            bd = rd;
            bj = rj;
        }
    }

    // The matched target and its normal are gathered for the phase-2 winner NOW, so that the round
    // trip runs under the certificate; the certificate changes the winner for a few queries in a
    // thousand, and those gather again.
    const bool owner = partials && sub == 0 && valid;
    const int jspec = (unsigned)bj < (unsigned)m ? bj : 0;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (owner) {
        q0 = tgt_orig[3 * jspec], q1 = tgt_orig[3 * jspec + 1], q2 = tgt_orig[3 * jspec + 2];
        n0 = nrm[3 * jspec], n1 = nrm[3 * jspec + 1], n2 = nrm[3 * jspec + 2];
    }

    // phase 3: certificate (resolve_certify)
    unsigned extra_slots = 0, extra_splits = 0;
    resolve_certify<SUBS, Q, KEEP, ICPMI_RESOLVE_SCANBATCH>(pv, lane, sub, valid, ic, n, px, py, pz, coarse, splits, slist, nact, frames, gframe, bs, sorted, perm, m, ms,
                             bd, bj, extra_slots, extra_splits);
    resolve_finish<Q>(lane, wave, ql, sub == 0, i, valid, bd, bj, px, py, pz, m, idx, d2out, counters, extra_slots, extra_splits,
                      tgt_orig, nrm, partials, jspec, q0, q1, q2, n0, n1, n2);
}

// The end of the quarter-wave resolve kernels (k_nn_resolve4, k_nn_resolve4_bounded): results out, counters, and the fused
// residual + normal-equation terms.  (jspec, q*, n*): the matched target and normal gathered ahead for target `jspec`
// (< 0: nothing was gathered).
template <int WAVES>
__device__ __forceinline__ void resolve4_finish(const int lane, const int wave, const int ql, const int quarter, const int i,
                                                const bool valid, const double bd, const int bj, const double px,
                                                const double py, const double pz, const int m, int *__restrict__ idx,
                                                double *__restrict__ d2out, unsigned long long *__restrict__ counters,
                                                const unsigned extra_slots, const unsigned extra_splits,
                                                const double *__restrict__ tgt_orig, const double *__restrict__ nrm,
                                                double *__restrict__ partials, const int jspec, double q0, double q1,
                                                double q2, double n0, double n1, double n2)
{
    if (valid && ql == 0) {
        idx[i] = bj == 0x7fffffff ? -1 : bj; // NaN/Inf query: nothing compares less (kdtree.hpp:53)
        if (d2out) d2out[i] = bd;
    }
    // statistics: a wave's counts; with the normal-equation terms below they meet in LDS first and leave as ONE pair of
    // atomics per workgroup (thousands of waves adding to the same two words were a third of the bounded resolve's time)
    __shared__ unsigned wave_cnt[WAVES][2];
    if (counters) {
        unsigned es = extra_slots, ef = extra_splits;
        for (int off = 32; off > 0; off >>= 1) {
            es += __shfl_down(es, off, 64);
            ef += __shfl_down(ef, off, 64);
        }
        if (lane == 0) {
            if (partials) {
                wave_cnt[wave][0] = es;
                wave_cnt[wave][1] = ef;
            } else if (es | ef) {
                atomicAdd(&counters[0], (unsigned long long)es);
                atomicAdd(&counters[1], (unsigned long long)ef);
            }
        }
    }
    // fused residual + normal equations (icp.hpp:99-120,198-206): the 4 owners of a wave (lane 0
    // of each quarter) form their J row and b; rows -> LDS, each wave sums its four by column,
    // the waves' sums meet in LDS in wave order: one partial row per workgroup
    if (partials) {
        __shared__ double jrow[WAVES * 4][29];
        __shared__ double red[WAVES][28];
        if (ql == 0) {
            double acc[28];
#pragma unroll
            for (int e = 0; e < 28; ++e) acc[e] = 0.0;
            if (valid) {
                const int j = (unsigned)bj < (unsigned)m ? bj : 0;
                if (j != jspec) { // the certificate found a nearer target
                    q0 = tgt_orig[3 * j], q1 = tgt_orig[3 * j + 1], q2 = tgt_orig[3 * j + 2];
                    n0 = nrm[3 * j], n1 = nrm[3 * j + 1], n2 = nrm[3 * j + 2];
                }
                double J[6];
                J[0] = py * n2 - pz * n1; // p x n, icp.hpp:105
                J[1] = pz * n0 - px * n2;
                J[2] = px * n1 - py * n0;
                J[3] = n0;
                J[4] = n1;
                J[5] = n2;
                const double e0 = q0 - px, e1 = q1 - py, e2 = q2 - pz;
                const double b = (e0 * n0 + e1 * n1) + e2 * n2; // icp.hpp:116
                int o = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = r; c < 6; ++c) acc[o++] = J[r] * J[c];
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[21 + r] = J[r] * b;
                acc[27] = b * b;
            }
#pragma unroll
            for (int e = 0; e < 28; ++e) jrow[wave * 4 + quarter][e] = acc[e];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 28)
            red[wave][lane] = ((jrow[wave * 4][lane] + jrow[wave * 4 + 1][lane]) + jrow[wave * 4 + 2][lane]) + jrow[wave * 4 + 3][lane];
        __syncthreads();
        if (counters && threadIdx.x == 32) {
            unsigned es = 0, ef = 0;
            for (int w = 0; w < WAVES; ++w) es += wave_cnt[w][0], ef += wave_cnt[w][1];
            if (es) atomicAdd(&counters[0], (unsigned long long)es);
            if (ef) atomicAdd(&counters[1], (unsigned long long)ef);
        }
        if (threadIdx.x < 28) {
            const int e = threadIdx.x;
            double v = red[0][e];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) v += red[w][e];
            partials[(size_t)blockIdx.x * kSumsStride + e] = v;
        }
    }
}

// Variant with one query per QUARTER-wave (4 queries per wave): the 16 lanes of a quarter share
// the splits of their query (lane ql takes list entries ql, ql+16, ...) and scan its winning
// slot together, in one round.  Same result as k_nn_resolve; a wave's chain of dependent memory
// round trips is ~4x shorter, which is what matters when there is about one wave per SIMD or
// less (clouds of ~10k points, one shard of a multi-GPU job, the pruned engine).
// One partial row of normal-equation terms per workgroup = per 4*WAVES queries.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_resolve4(const double *__restrict__ qry, int n,
                                                           const double *__restrict__ sorted,
                                                           const unsigned *__restrict__ perm, int m, int ms,
                                                           const float2 *__restrict__ coarse, int splits,
                                                           const SplitFrame *__restrict__ frames,
                                                           const NnFrame *__restrict__ gframe,
                                                           int *__restrict__ idx, double *__restrict__ d2out,
                                                           unsigned long long *__restrict__ counters,
                                                           const double *__restrict__ tgt_orig,
                                                           const double *__restrict__ nrm,
                                                           double *__restrict__ partials,
                                                           const int *__restrict__ blk_cnt,
                                                           const int *__restrict__ blk_list,
                                                           const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, quarter = lane >> 4;
    const int qbase = (blockIdx.x * WAVES + wave) * 4;
    const int i = qbase + quarter; // waves past the end run on a clamped query and write nothing
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = qry[3 * ic], py = qry[3 * ic + 1], pz = qry[3 * ic + 2];
    static_assert(kCoarseQueries % 4 == 0, "a wave's queries share a coarse block");
    const int *slist = blk_list ? blk_list + (size_t)(qbase / kCoarseQueries) * splits : nullptr;
    const int nact = blk_list ? (qbase < n ? blk_cnt[qbase / kCoarseQueries] : 0) : splits;

    // phase 1: smallest coarse value over the splits; ties to the lowest split.  The first KEEP4
    // values of a lane (64 splits) stay in registers for the certificate's first filter.
    constexpr int KEEP4 = ICPMI_RESOLVE_KEEP > 0 ? 4 : 0;
    float pv[KEEP4 > 0 ? KEEP4 : 1];
    float best = kBig;
    int bs = 0;
    int ps[KEEP4 > 0 ? KEEP4 : 1]; // (unconditional loads, all in flight together: see k_nn_resolve)
#pragma unroll
    for (int k = 0; k < KEEP4; ++k) {
        const int e = ql + 16 * k, ec = e < nact ? e : 0;
        ps[k] = slist ? slist[nact > 0 ? ec : 0] : ec;
    }
#pragma unroll
    for (int k = 0; k < KEEP4; ++k) pv[k] = ICPMI_CX(coarse, n, splits, ps[k], ic);
#pragma unroll
    for (int k = 0; k < KEEP4; ++k) {
        const bool in = ql + 16 * k < nact;
        pv[k] = in ? pv[k] : kBig;
        const bool take = in & ((pv[k] < best) | ((pv[k] == best) & (ps[k] < bs)));
        best = take ? pv[k] : best;
        bs = take ? ps[k] : bs;
    }
    for (int e = ql + 16 * KEEP4; e < nact; e += 16) {
        const int s = slist ? slist[e] : e;
        const float v = ICPMI_CX(coarse, n, splits, s, ic);
        if (v < best || (v == best && s < bs)) {
            best = v;
            bs = s;
        }
    }
#pragma unroll
    for (int x = 1; x < 16; x <<= 1) {
        const float ov = __shfl_xor(best, x, 64);
        const int os = __shfl_xor(bs, x, 64);
        if (ov < best || (ov == best && os < bs)) {
            best = ov;
            bs = os;
        }
    }
    const int bcol = (int)(__float_as_uint(best) & 31u);

    // phase 2: exact evaluation of the winning slot by the quarter (lane ql: positions ql, ql+16, ...)
    double bd = 1.7976931348623157e308;
    int bj = 0x7fffffff;
    {
        const int j0 = bs * kSplitTargets + bcol * kSlotTargets + ql;
#pragma unroll
        for (int o = 0; o < kSlotTargets / 16; ++o) {
            const int jj = j0 + 16 * o;
            const int jc = jj < m ? jj : m - 1;
            const double dd = sqdist(ICPMI_SX(sorted, ms, jc), ICPMI_SY(sorted, ms, jc), ICPMI_SZ(sorted, ms, jc), px, py, pz);
            const int oj = (int)perm[jc];
            if (jj < m && (dd < bd || (dd == bd && oj < bj))) {
                bd = dd;
                bj = oj;
            }
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
            const double od = __shfl_xor(bd, x, 64);
            const int oj = __shfl_xor(bj, x, 64);
            if (od < bd || (od == bd && oj < bj)) {
                bd = od;
                bj = oj;
            }
        }
    }

    // speculative gather of the phase-2 winner's target and normal (see k_nn_resolve)
    const bool owner = partials && ql == 0 && valid;
    const int jspec = (unsigned)bj < (unsigned)m ? bj : 0;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (owner) {
        q0 = tgt_orig[3 * jspec], q1 = tgt_orig[3 * jspec + 1], q2 = tgt_orig[3 * jspec + 2];
        n0 = nrm[3 * jspec], n1 = nrm[3 * jspec + 1], n2 = nrm[3 * jspec + 2];
    }

    // phase 3: certificate (resolve_certify)
    unsigned extra_slots = 0, extra_splits = 0;
    resolve_certify<16, 4, KEEP4, ICPMI_RESOLVE4_SCANBATCH>(pv, lane, ql, valid, ic, n, px, py, pz, coarse, splits, slist, nact, frames, gframe, bs, sorted, perm, m, ms,
                        bd, bj, extra_slots, extra_splits);
    resolve4_finish<WAVES>(lane, wave, ql, quarter, i, valid, bd, bj, px, py, pz, m, idx, d2out, counters, extra_slots, extra_splits,
                           tgt_orig, nrm, partials, jspec, q0, q1, q2, n0, n1, n2);
}

// ---- k-NN on the same coarse pass ---------------------------------------------------------------
// Resolve for the k nearest neighbours of point i among all targets (icp.hpp:32,
// kdtree.hpp:65-78), one wave per row.  Two upper bounds T on the k-th neighbour's exact
// squared distance are formed and the smaller one used:
//   (a) each slot minimum belongs to a distinct target, so the k-th smallest of the 64
//       per-lane minima tS (lane l looks at slots l, l+64, ...) gives k targets with coarse
//       value <= tS, hence exact distance <= dmax, solved from d <= tS + E(d) with the
//       largest frame term among those k slots;
//   (b) the slot with the smallest minimum (the row's own neighbourhood: targets are Morton
//       sorted) is scanned exactly and the k-th smallest of its per-lane minima taken.
// Every true k-neighbour then has coarse value <= T + E_s(T) in its split: the slots under
// their split's bound are scanned exactly (fp64, reference operation order) and every target
// with exact distance <= T is collected; the k smallest by (distance, original index) are
// written closest first -- the order kdtree.hpp:72-76 returns and icp.hpp:41-51 sums in.
constexpr int kKnnCap = 256;        // candidates per row held in LDS (more: the bound is tightened and the row redone)
#ifndef ICPMI_KNN_BATCH
#define ICPMI_KNN_BATCH 2 /* 1 / 2 / 4: 431 / 435 / 432 us (the kernel is issue bound, not latency bound) */
#endif
constexpr int kKnnBatch = ICPMI_KNN_BATCH; // flagged slots scanned per round (their loads overlap)
constexpr int kKnnFlagCap = 192;    // slots listed for scanning per row and attempt (a sane bound flags ~10)
constexpr int kKnnMaxSplits = 256;  // per-split bounds cached in LDS (512k targets); beyond: recomputed
#ifndef ICPMI_KNN_REGSLOTS
#define ICPMI_KNN_REGSLOTS 26 /* as fp32: 26 -> 119 VGPRs = 4 waves per SIMD, 32 -> 129 = 3 waves (430 vs 529 us on C3);
                                 as bf16 pairs 107 / 115 VGPRs, 382 / 395 us: 26 kept */
#endif
constexpr int kKnnRegSlots = ICPMI_KNN_REGSLOTS;    // slot minima per lane kept in registers, two per dword (1664 slots = 106k targets)

// Ascending bitonic sort of one value per lane (21 compare-exchange steps); lane i ends up
// with the i-th smallest.  Used for "k-th smallest of 64": a rank-by-counting loop costs 64
// broadcast + compare rounds, several times this.
template <typename T>
__device__ __forceinline__ T wave_sort_asc(T v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const T o = lane_xor_n(v, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            const T mn = o < v ? o : v, mx = o < v ? v : o;
            v = (lower == up) ? mn : mx;
        }
    }
    return v;
}

// Rank a wave's `total` candidates (distance cd[], original index cj[], in LDS) by (distance, original index);
// the k smallest go to out[0..k) closest first -- the order kdtree.hpp:72-76 returns and icp.hpp:41-51 sums in.
// Fast pass: count strictly smaller distances only (one compare per pair).  Without equal
// distances that count IS the rank; candidates with equal distance get the same count,
// which the owner table exposes -- then the full (distance, index) order is evaluated.
// With `perm`, cj[] holds SORTED positions (the scan did not wait for the index of every candidate): translated for the
// k winners only, or for all candidates when the index has to break a tie.
__device__ __forceinline__ void knn_rank_write(const double *cd, int *cj, int *cr, int *own, const int total,
                                               const int k, const int lane, int *__restrict__ out,
                                               const unsigned *__restrict__ perm = nullptr)
{
    // fewer candidates than k (a row with a NaN coordinate keeps none: no distance compares): the places left over say "no
    // neighbour", as k_knn_exact_rows writes them -- never what an earlier call left in the buffer
    for (int r = total + lane; r < k; r += 64) out[r] = -1;
    bool clash = false;
    for (int e = lane; e < total; e += 64) {
        const double d = cd[e];
        int r = 0;
        for (int f = 0; f < total; ++f) r += cd[f] < d ? 1 : 0;
        cr[e] = r;
        own[r] = e;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < total; e += 64) clash |= own[cr[e]] != e;
    if (__ballot(clash) == 0ull) {
        for (int e = lane; e < total; e += 64) {
            const int r = cr[e];
            if (r < k) out[r] = perm ? (int)perm[cj[e]] : cj[e];
        }
        return;
    }
    if (perm) {
        for (int e = lane; e < total; e += 64) cj[e] = (int)perm[cj[e]];
        __builtin_amdgcn_wave_barrier();
    }
    for (int e = lane; e < total; e += 64) {
        const double d = cd[e];
        const int j = cj[e];
        int r = 0;
        for (int f = 0; f < total; ++f) {
            const double df = cd[f];
            const int jf = cj[f];
            r += (df < d || (df == d && jf < j)) ? 1 : 0;
        }
        if (r < k) out[r] = j;
    }
}

// LISTED (pruned engine): the rows are SORTED positions row0.. of the target itself, only the
// splits on the row's block list (k_knn_block_bounds) were evaluated, and the neighbour lists
// are stored by sorted position.  Slots are then numbered locally, 32 per list entry.
template <bool LISTED>
__global__ __launch_bounds__(256) void k_knn_resolve(const double *__restrict__ pts, int row0, int nrows,
                                                     const double *__restrict__ sorted,
                                                     const unsigned *__restrict__ perm, int m, int ms, int k,
                                                     const float *__restrict__ slotmin, int nslots,
                                                     const SplitFrame *__restrict__ frames,
                                                     const NnFrame *__restrict__ gframe,
                                                     int *__restrict__ knn_idx /*[m][k]*/,
                                                     int *__restrict__ fb_list, int *__restrict__ fb_count,
                                                     const int *__restrict__ blk_cnt,
                                                     const int *__restrict__ blk_list)
{
    __shared__ double cand_d[4][kKnnCap];
    __shared__ int cand_j[4][kKnnCap];
    __shared__ float tau_sp[4][kKnnMaxSplits]; // per-split bound on the coarse value, per row
    __shared__ int cand_r[4][kKnnCap], owner[4][kKnnCap]; // distance-only rank of a candidate; who claimed a rank
    __shared__ int flist[4][kKnnFlagCap];                 // global numbers of the slots to scan
    __shared__ int s_list[kKnnMaxSplits];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = blockIdx.x * 4 + wave;
    // the four rows of a workgroup share a query block, hence its list of evaluated splits
    static_assert(kCoarseQueries % 4 == 0, "a workgroup's rows share a coarse block");
    const int *slist = LISTED ? blk_list + (size_t)(blockIdx.x * 4 / kCoarseQueries) * (nslots / kCols) : nullptr;
    const int nact = LISTED ? blk_cnt[blockIdx.x * 4 / kCoarseQueries] : 0;
    if (LISTED) {
        for (int e = threadIdx.x; e < nact && e < kKnnMaxSplits; e += 256) s_list[e] = slist[e];
        __syncthreads();
    }
    if (local >= nrows) return; // wave-uniform
    const int i = row0 + local;
    if (LISTED && nact == 0) { // cannot happen for a block with points (its own split is always listed)
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    const int ic = LISTED ? (i < m ? i : m - 1) : i;
    const double px = LISTED ? ICPMI_SX(sorted, ms, ic) : pts[3 * i], py = LISTED ? ICPMI_SY(sorted, ms, ic) : pts[3 * i + 1],
                 pz = LISTED ? ICPMI_SZ(sorted, ms, ic) : pts[3 * i + 2];
    // local slot e -> global slot; local split (list entry) -> global split
    auto gsplit = [&](int sp) -> int { return LISTED ? (sp < kKnnMaxSplits ? s_list[sp] : slist[sp]) : sp; };
    auto gslot = [&](int e) -> int { return LISTED ? gsplit(e / kCols) * kCols + (e % kCols) : e; };
    const int nloc = LISTED ? nact * kCols : nslots; // slots this row looks at
    const unsigned short *mine16 = reinterpret_cast<const unsigned short *>(slotmin) + (size_t)local * nslots;
    auto mine = [&](int g) -> float { return __uint_as_float((unsigned)mine16[g] << 16); }; // bf16, rounded down
    const int kk = k < 64 ? k : 64;
    const double kInf = 1.7976931348623157e308;

    // per-lane minimum of the slot minima, with its slot.  The first kKnnRegSlots values per
    // lane stay in registers for the flagging pass below (the slot minima are the bulk of this
    // kernel's HBM traffic: read them once).
    // (kept and compared as the raw 16-bit patterns: non-negative bf16 values order like unsigned
    // integers; a lane owns PAIRS of neighbouring slots, 2 * lane + 128 u + {0, 1}, so that it loads
    // whole dwords -- 2-byte loads came out of the compiler serialised through one register)
    constexpr int kPairs = kKnnRegSlots / 2;
    static_assert(kKnnRegSlots % 2 == 0 && kKnnRegSlots <= 32, "pairs of slots, one flag bit each");
    unsigned sv[kPairs];
    unsigned lmin16 = 0x7F61u; // kBig as stored
    int lslot = 2 * lane < nloc ? 2 * lane : 0; // LOCAL slot number
#pragma unroll
    for (int u = 0; u < kPairs; ++u) {
        const int e = 2 * lane + 128 * u;
        sv[u] = e < nloc ? *reinterpret_cast<const unsigned *>(mine16 + gslot(e)) : 0x7F617F61u;
    }
#pragma unroll
    for (int u = 0; u < kPairs; ++u) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned v = h ? sv[u] >> 16 : sv[u] & 0xFFFFu;
            if (v < lmin16) {
                lmin16 = v;
                lslot = 2 * lane + 128 * u + h;
            }
        }
    }
    // slots beyond the register-resident ones (> 106k targets): pairs again, dword loads, four in flight
#pragma unroll 4
    for (int e = 128 * kPairs + 2 * lane; e < nloc; e += 128) { // (nloc is a multiple of 32: e + 1 < nloc)
        const unsigned w = *reinterpret_cast<const unsigned *>(mine16 + gslot(e));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned v = h ? w >> 16 : w & 0xFFFFu;
            if (v < lmin16) {
                lmin16 = v;
                lslot = e + h;
            }
        }
    }
    const float lmin = __uint_as_float(lmin16 << 16);
    // best slot: lane with the smallest minimum (ties: lowest lane)
    int bslot;
    {
        float bv = lmin;
        int bl = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int ol = __shfl_xor(bl, off, 64);
            if (ov < bv || (ov == bv && ol < bl)) {
                bv = ov;
                bl = ol;
            }
        }
        bslot = __shfl(lslot, bl, 64);
    }
#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 1 /* timing experiment only: stop after the slot minima */
    if (bslot >= 0) { if (lane == 0) knn_idx[(size_t)i * k] = bslot; return; }
#endif
    // bound (b): exact scan of the best slot (kSlotTargets / 64 targets per lane); when a slot
    // is a single 64-run the 64 sorted positions after it (the next slot) also tighten the
    // bound -- those are collected later through the normal path
    constexpr int kOwn = kSlotTargets / 64 > 0 ? kSlotTargets / 64 : 1;
    constexpr int kBnd = kOwn > 1 ? kOwn : 2; // runs of 64 used for the bound
    double dloc[kBnd];
    int oloc[kBnd];
    {
        const int gb = gslot(bslot), j0 = (gb / kCols) * kSplitTargets + (gb % kCols) * kSlotTargets;
#pragma unroll
        for (int c = 0; c < kBnd; ++c) {
            const int jj = j0 + 64 * c + lane;
            dloc[c] = kInf;
            oloc[c] = 0;
            if (jj < m) {
                const double d = sqdist(ICPMI_SX(sorted, ms, jj), ICPMI_SY(sorted, ms, jj), ICPMI_SZ(sorted, ms, jj), px, py, pz);
                dloc[c] = d == d ? d : kInf; // a NaN would break the order the sort below relies on
                oloc[c] = (int)perm[jj];
            }
        }
    }
    double lbest = dloc[0];
#pragma unroll
    for (int c = 1; c < kBnd; ++c) lbest = dloc[c] < lbest ? dloc[c] : lbest;
#ifndef ICPMI_KNN_MORE_RUNS
#define ICPMI_KNN_MORE_RUNS 1 /* measured on C3: 0 -> 668 us, 1 -> 555, 2 -> 564, 4 -> 588 */
#endif
    { // further runs of 64 sorted neighbours on both sides, for the bound only
        const int gb = gslot(bslot), j0 = (gb / kCols) * kSplitTargets + (gb % kCols) * kSlotTargets;
#pragma unroll
        for (int c = 0; c < ICPMI_KNN_MORE_RUNS; ++c) {
            const int ja = j0 - 64 * (c + 1) + lane, jb = j0 + 64 * (kBnd + c) + lane;
            if (ja >= 0) {
                const double d = sqdist(ICPMI_SX(sorted, ms, ja), ICPMI_SY(sorted, ms, ja), ICPMI_SZ(sorted, ms, ja), px, py, pz);
                lbest = d < lbest ? d : lbest;
            }
            if (jb < m) {
                const double d = sqdist(ICPMI_SX(sorted, ms, jb), ICPMI_SY(sorted, ms, jb), ICPMI_SZ(sorted, ms, jb), px, py, pz);
                lbest = d < lbest ? d : lbest;
            }
        }
    }
    // k-th smallest of the 64 per-lane minima, sorted as fp32 rounded UP (an upper bound stays one;
    // the fp32 network is 5 instead of 8 instructions per compare-exchange)
    float lbf = (float)lbest;
    lbf = (double)lbf < lbest ? __uint_as_float(__float_as_uint(lbf) + 1u) : lbf; // (lbest >= 0; Inf / NaN pass through)
    const double t1 = (double)__shfl(wave_sort_asc(lbf, lane), kk - 1, 64);
    double T = t1;
    { // both bounds matter: on the 100k uniform cloud either one alone makes this kernel 2-3x slower
        // bound (a): k-th smallest of the 64 lane minima; the lanes at or under it (k of them, more
        // on ties: only makes the frame term larger) set the frame term
        const float tS = __shfl(wave_sort_asc(lmin, lane), kk - 1, 64);
        double a;
        {
            const SplitFrame &f = frames[gsplit(lslot / kCols)];
            const double dx = px - f.c[0], dy = py - f.c[1], dz = pz - f.c[2];
            a = lmin <= tS ? sqrt((dx * dx + dy * dy) + dz * dz) * (1.0 + 1e-6) + f.rho : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(a, off, 64);
                a = o > a ? o : a;
            }
        }
        const double u = 5.9604644775390625e-08;
        const double eps = kReprEps * a * (1.0 + 1e-6);
        const double A = kArithBound * u * a * a;
        // (the stored minima are bf16 rounded down: the true one is within 2^-7 above)
        const double ts = tS > 0.f ? (double)tS * (1.0 + 0.0078125 + 1e-4) : 0.0;
        const double xr = eps + sqrt(eps * eps + (ts + eps * eps + A)); // sqrt(dmax)
        const double dmax = tS >= 2.9e38f ? 1.0e300 : xr * xr * (1.0 + 1e-9); // (kBig as stored: bf16, rounded down)
        T = T < dmax ? T : dmax; // (NaN rows: T becomes dmax)
    }

#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 2 /* timing experiment only: stop after the bounds */
    if (T >= 0.0) { if (lane == 0) knn_idx[(size_t)i * k] = (int)T; return; }
#endif
    // candidates: the best slot's targets under T, then every other slot under its split's
    // bound.  If more than kKnnCap turn up, the k-th smallest of those already held is a
    // tighter valid bound: collect again with it (a few rows per cloud).
    const int nsplits = LISTED ? nact : (nslots + kCols - 1) / kCols; // (local) splits this row looks at
    static_assert(kSlotTargets <= kKnnCap, "the best slot's targets must fit the candidate list");
    int total = 0;
#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 4
    int nf_last = 0, attempts_run = 0;
#endif
    for (int attempt = 0; attempt < 4; ++attempt) {
        const double sq = sqrt(T);
        for (int sp = lane; sp < nsplits && sp < kKnnMaxSplits; sp += 64)
            tau_sp[wave][sp] = T >= 1.0e299 ? kBig : split_tau(px, py, pz, frames[gsplit(sp)], T, sq);
        __builtin_amdgcn_wave_barrier();
        total = 0;
#pragma unroll
        for (int c = 0; c < kOwn; ++c) { // the best slot's own targets (at most kSlotTargets <= kKnnCap)
            const bool keep = dloc[c] <= T;
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const int pos = total + __popcll(km & ((1ull << lane) - 1ull));
                cand_d[wave][pos] = dloc[c];
                cand_j[wave][pos] = oloc[c];
            }
            total += __popcll(km);
        }
        // Which slots to scan.  First filter, in registers: one compare per slot minimum against the
        // bound that holds for every split (all_splits_tau).  The survivors -- a dozen of a row's
        // 1,568 slots on the 100k cloud -- are then checked against their own split's bound and
        // LISTED (compacted through LDS); every lane hands over one survivor per round, so the
        // rounds number what the busiest lane holds (one to three), not the slots per lane.
        const float tall = T >= 1.0e299 ? 3.4028235e38f : all_splits_tau(px, py, pz, *gframe, T, sq);
        const unsigned tall16 = __float_as_uint(tall) >> 16; // a stored (bf16) value is <= tall iff its pattern is <= this one
        unsigned regflags = 0u;
#pragma unroll
        for (int u = 0; u < kPairs; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * lane + 128 * u + h;
                const unsigned v = h ? sv[u] >> 16 : sv[u] & 0xFFFFu;
                regflags |= (e < nloc && e != bslot && v <= tall16) ? (1u << (2 * u + h)) : 0u;
            }
        int nf = 0;
        auto list_flagged = [&](bool cand, int e) { // wave-uniform call; `cand` lanes test slot e against its split
            bool flag = false;
            if (cand) {
                const int sp = e / kCols;
                const float tauf = sp < kKnnMaxSplits ? tau_sp[wave][sp]
                                                      : (T >= 1.0e299 ? kBig : split_tau(px, py, pz, frames[gsplit(sp)], T, sq));
                flag = mine(gslot(e)) <= tauf;
            }
            const unsigned long long fm = __ballot(flag);
            if (flag) {
                const int pos = nf + __popcll(fm & ((1ull << lane) - 1ull));
                if (pos < kKnnFlagCap) flist[wave][pos] = gslot(e);
            }
            nf += __popcll(fm);
        };
        while (__ballot(regflags != 0u)) {
            const bool act = regflags != 0u;
            const int u = act ? __ffs((int)regflags) - 1 : 0;
            regflags &= regflags - 1u; // (0 stays 0)
            list_flagged(act, 2 * lane + 128 * (u >> 1) + (u & 1));
        }
#pragma unroll 1
        for (int e0 = 128 * kPairs; e0 < nloc; e0 += 128) { // slots beyond the register-resident ones (> 106k targets)
            const int e = e0 + 2 * lane;
            const unsigned w = e < nloc ? *reinterpret_cast<const unsigned *>(mine16 + gslot(e)) : 0x7F617F61u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned v = h ? w >> 16 : w & 0xFFFFu;
                const bool cand = e + h < nloc && e + h != bslot && v <= tall16;
                if (__ballot(cand)) list_flagged(cand, e + h);
            }
        }
#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 4
        nf_last = nf, attempts_run = attempt + 1;
#endif
        const bool overflow = nf > kKnnFlagCap; // (only with a bound that rules nothing out)
        const int nfl = overflow ? kKnnFlagCap : nf;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int f0 = 0; f0 < nfl && total <= kKnnCap; f0 += kKnnBatch) { // (once the list is full the bound gets tightened from
            constexpr int kRuns = kSlotTargets / 64;                        //  what it holds: scanning on would only count)
            double d[kKnnBatch][kRuns];
            int jj[kKnnBatch][kRuns];
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q) {
                const int f = f0 + q;
                const int se = flist[wave][f < nfl ? f : nfl - 1];
                const int j0 = (se / kCols) * kSplitTargets + (se % kCols) * kSlotTargets;
#pragma unroll
                for (int o = 0; o < kRuns; ++o) {
                    jj[q][o] = j0 + 64 * o + lane;
                    d[q][o] = kInf;
                    if (f < nfl && jj[q][o] < m)
                        d[q][o] = sqdist(ICPMI_SX(sorted, ms, jj[q][o]), ICPMI_SY(sorted, ms, jj[q][o]), ICPMI_SZ(sorted, ms, jj[q][o]),
                                         px, py, pz);
                }
            }
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q)
#pragma unroll
                for (int o = 0; o < kRuns; ++o) {
                    const bool keep = d[q][o] <= T;
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int pos = total + __popcll(km & ((1ull << lane) - 1ull));
                        if (pos < kKnnCap) {
                            cand_d[wave][pos] = d[q][o];
                            cand_j[wave][pos] = (int)perm[jj[q][o]];
                        }
                    }
                    total += __popcll(km);
                }
        }
        if (overflow && total <= kKnnCap) { // unlisted slots and too few candidates to tighten the bound from
            total = kKnnCap + 1;            // -> the exact kernel takes the row
            break;
        }
        if (total <= kKnnCap) break;
        // New bound from the kKnnCap candidates held (all real targets): each lane takes the smallest
        // of its four, the k-th smallest of those 64 values bounds the k-th nearest distance from above
        // like bound (b) does (64 different targets).  (The exact k-th smallest of all 256 took a
        // 256 x 256 comparison count: ~10 us in a wave that every other wave of the launch then waited
        // for -- on a LiDAR-like frame a dozen rows in ten thousand come here.)
        __builtin_amdgcn_wave_barrier();
        double lm = kInf;
        for (int e = lane; e < kKnnCap; e += 64) {
            const double d = cand_d[wave][e];
            lm = d < lm ? d : lm;
        }
        const double tnew = __shfl(wave_sort_asc(lm, lane), kk - 1, 64);
        __builtin_amdgcn_wave_barrier();
        if (!(tnew < T)) break; // cannot tighten (e.g. hundreds of coincident points)
        T = tnew;
    }
#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 4 /* diagnostic only: per row, the slots listed, the candidates found and T, in place of neighbours */
    if (total >= 0) {
        if (lane < k) knn_idx[(size_t)i * k + lane] = lane == 0 ? nf_last : lane == 1 ? total : lane == 2 ? attempts_run : 0;
        return;
    }
#endif
#if defined(ICPMI_KNN_STOP) && ICPMI_KNN_STOP == 3 /* timing experiment only: stop after collecting candidates */
    if (total >= 0) { if (lane == 0) knn_idx[(size_t)i * k] = total; return; }
#endif
    if (total > kKnnCap) { // still too many targets under the bound: hand the row to the exact kernel
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    knn_rank_write(cand_d[wave], cand_j[wave], cand_r[wave], owner[wave], total, k, lane, knn_idx + (size_t)i * k);
}

// out[perm[i]] = in[i], rows of 3 doubles (normals gathered in sorted order -> point order)
__global__ __launch_bounds__(256) void k_scatter_rows(const double *__restrict__ in, const unsigned *__restrict__ perm,
                                                      int m, double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const size_t o = perm[i];
    out[3 * o] = in[3 * i];
    out[3 * o + 1] = in[3 * i + 1];
    out[3 * o + 2] = in[3 * i + 2];
}

// Exact fp64 k-NN list of ONE row per workgroup (rows the MFMA resolve hands back): every
// thread keeps the k best of its strided share of the targets (sorted, in LDS), then the k
// global best are popped by k rounds of a workgroup-wide argmin over the list heads.
// With `perm` the listed rows are sorted positions (pruned engine): row r is point perm[r] and its
// list is stored at r.
// `qry` (may be null: the rows are rows of `pts`) holds the rows' coordinates when the queries are
// not the targets themselves (icpmi_k_nearest).
__global__ __launch_bounds__(256) void k_knn_exact_rows(const double *__restrict__ pts, int m, int k,
                                                        const int *__restrict__ list,
                                                        const int *__restrict__ list_count,
                                                        int *__restrict__ knn_idx,
                                                        const unsigned *__restrict__ perm,
                                                        const double *__restrict__ qry)
{
    extern __shared__ double knn_smem[];
    constexpr int BLOCK = 256;
    double *ld = knn_smem;
    int *li = reinterpret_cast<int *>(knn_smem + (size_t)k * BLOCK);
    __shared__ double red_d[4];
    __shared__ int red_j[4], red_t[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrows = *list_count;
    for (int lr = blockIdx.x; lr < nrows; lr += gridDim.x) { // block-uniform
        const int i = list[lr];
        const size_t ip = perm ? perm[i] : (unsigned)i;
        const double *qp = qry ? qry : pts;
        const double px = qp[3 * ip], py = qp[3 * ip + 1], pz = qp[3 * ip + 2];
        int cnt = 0;
        double thr = __builtin_inf();
        for (int j = tid; j < m; j += BLOCK) {
            const double d = sqdist(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], px, py, pz);
            if (d < thr) {
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0) {
                    const double prev = ld[(pos - 1) * BLOCK + tid];
                    if (!(prev > d)) break;
                    ld[pos * BLOCK + tid] = prev;
                    li[pos * BLOCK + tid] = li[(pos - 1) * BLOCK + tid];
                    --pos;
                }
                ld[pos * BLOCK + tid] = d;
                li[pos * BLOCK + tid] = j;
                if (cnt < k) ++cnt;
                if (cnt == k) thr = ld[(k - 1) * BLOCK + tid];
            }
        }
        int head = 0;
        const int want = k < m ? k : m;
        for (int out = 0; out < want; ++out) {
            double d = head < cnt ? ld[head * BLOCK + tid] : 1.7976931348623157e308;
            int j = head < cnt ? li[head * BLOCK + tid] : 0x7fffffff;
            int t = tid;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_xor(d, off, 64);
                const int oj = __shfl_xor(j, off, 64), ot = __shfl_xor(t, off, 64);
                if (od < d || (od == d && oj < j)) {
                    d = od;
                    j = oj;
                    t = ot;
                }
            }
            if (lane == 0) {
                red_d[wave] = d;
                red_j[wave] = j;
                red_t[wave] = t;
            }
            __syncthreads();
            d = red_d[0];
            j = red_j[0];
            t = red_t[0];
            for (int w = 1; w < 4; ++w)
                if (red_d[w] < d || (red_d[w] == d && red_j[w] < j)) {
                    d = red_d[w];
                    j = red_j[w];
                    t = red_t[w];
                }
            if (tid == t) ++head;
            if (tid == 0) knn_idx[(size_t)i * k + out] = j == 0x7fffffff ? -1 : j; // (no candidate left: "no neighbour", icp_mi355x.h)
            __syncthreads();
        }
    }
}

// exact fp64 k-NN lists, one row per thread (small clouds, and rows the MFMA resolve hands
// back).  rows: explicit list (list != nullptr, count read from *list_count) or [row0,row1).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_knn_exact_list(const double *__restrict__ pts, int m, int k,
                                                          int row0, int row1,
                                                          const int *__restrict__ list,
                                                          const int *__restrict__ list_count,
                                                          int *__restrict__ knn_idx,
                                                          const double *__restrict__ qry)
{
    extern __shared__ double knn_smem[];
    double *ld = knn_smem;
    int *li = reinterpret_cast<int *>(knn_smem + (size_t)k * BLOCK);
    const int tid = threadIdx.x;
    const double *qp = qry ? qry : pts;
    const int nrows = list ? *list_count : row1 - row0;
    for (int base = blockIdx.x * BLOCK; base < nrows; base += gridDim.x * BLOCK) { // block-uniform
        const int lr = base + tid;
        const bool active = lr < nrows;
        const int i = active ? (list ? list[lr] : row0 + lr) : (list ? list[0] : row0);
        const double px = qp[3 * i], py = qp[3 * i + 1], pz = qp[3 * i + 2];
        int cnt = active ? 0 : k;
        double thr = active ? __builtin_inf() : -1.0;
#pragma unroll 4
        for (int j = 0; j < m; ++j) {
            const double d = sqdist(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], px, py, pz);
            if (d < thr) {
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0) {
                    const double prev = ld[(pos - 1) * BLOCK + tid];
                    if (!(prev > d)) break;
                    ld[pos * BLOCK + tid] = prev;
                    li[pos * BLOCK + tid] = li[(pos - 1) * BLOCK + tid];
                    --pos;
                }
                ld[pos * BLOCK + tid] = d;
                li[pos * BLOCK + tid] = j;
                if (cnt < k) ++cnt;
                if (cnt == k) thr = ld[(k - 1) * BLOCK + tid];
            }
        }
        if (active)
            for (int a = 0; a < cnt; ++a) knn_idx[(size_t)i * k + a] = li[a * BLOCK + tid];
    }
}

// squared distances of finished neighbour lists (entries < 0: no neighbour, distance left untouched)
__global__ __launch_bounds__(256) void k_knn_distances(const double *__restrict__ qry, int nq, const double *__restrict__ tgt,
                                                       int m, int k, const int *__restrict__ knn_idx,
                                                       double *__restrict__ d2)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)nq * k) return;
    const int i = (int)(e / k), j = knn_idx[e];
    if ((unsigned)j < (unsigned)m)
        d2[e] = sqdist(tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2], qry[3 * i], qry[3 * i + 1], qry[3 * i + 2]);
}

// PCA normal from a closest-first neighbour list (icp.hpp:34-63), one row per thread
// With `perm`, rows are sorted positions (pruned engine): the list of row i is that of point
// perm[i]; `scatter` writes its normal where that point lives, normals[perm[i]].
__global__ __launch_bounds__(256) void k_normals_from_knn(const double *__restrict__ pts, int m, int k,
                                                          int row0, int row1,
                                                          const int *__restrict__ knn_idx,
                                                          double *__restrict__ normals,
                                                          const unsigned *__restrict__ perm, int scatter)
{
    const int i = row0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= row1) return;
    const int self = perm ? (int)perm[i] : i;
    const int cnt = k < m ? k : m;
    const int *nb = knn_idx + (size_t)i * k;
    double nx = 0.0, ny = 0.0, nz = 1.0; // icp.hpp:34-37
    if (cnt >= 3) {
        // The neighbours are fetched five at a time -- indices, then coordinates, each batch's loads in flight
        // together -- and summed one after the other in list order as the reference does: as one dependent chain of
        // index load -> coordinate load per neighbour the two passes were 80 round trips to memory (44 us per 100k rows).
        constexpr int kB = 5;
        constexpr int kAll = 20; // the reference's k (icp.hpp:170): the whole list in registers, two round trips in all
        double cx = 0.0, cy = 0.0, cz = 0.0; // icp.hpp:40-44
        double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0; // icp.hpp:47-52
        const double kd = (double)cnt;
        if (cnt <= kAll) {
            int j[kAll];
            double x[kAll], y[kAll], z[kAll];
#pragma unroll
            for (int b = 0; b < kAll; ++b) j[b] = nb[b < cnt ? b : cnt - 1];
            __builtin_amdgcn_sched_barrier(0); // (the scheduler splits the batches otherwise: 18 waits where two do)
#pragma unroll
            for (int b = 0; b < kAll; ++b) {
                const int jj = (unsigned)j[b] < (unsigned)m ? j[b] : self; // rows with NaN coordinates have no list
                x[b] = pts[3 * jj];
                y[b] = pts[3 * jj + 1];
                z[b] = pts[3 * jj + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < kAll; ++b)
                if (b < cnt) {
                    cx += x[b];
                    cy += y[b];
                    cz += z[b];
                }
            cx /= kd;
            cy /= kd;
            cz /= kd;
#pragma unroll
            for (int b = 0; b < kAll; ++b)
                if (b < cnt) {
                    const double dx = x[b] - cx, dy = y[b] - cy, dz = z[b] - cz;
                    c00 += dx * dx;
                    c01 += dx * dy;
                    c02 += dx * dz;
                    c11 += dy * dy;
                    c12 += dy * dz;
                    c22 += dz * dz;
                }
        } else {
        for (int a0 = 0; a0 < cnt; a0 += kB) {
            int j[kB];
            double x[kB], y[kB], z[kB];
#pragma unroll
            for (int b = 0; b < kB; ++b) j[b] = nb[a0 + b < cnt ? a0 + b : cnt - 1];
#pragma unroll
            for (int b = 0; b < kB; ++b) {
                const int jj = (unsigned)j[b] < (unsigned)m ? j[b] : self; // rows with NaN coordinates have no list
                x[b] = pts[3 * jj];
                y[b] = pts[3 * jj + 1];
                z[b] = pts[3 * jj + 2];
            }
#pragma unroll
            for (int b = 0; b < kB; ++b)
                if (a0 + b < cnt) {
                    cx += x[b];
                    cy += y[b];
                    cz += z[b];
                }
        }
        cx /= kd;
        cy /= kd;
        cz /= kd;
        for (int a0 = 0; a0 < cnt; a0 += kB) {
            int j[kB];
            double x[kB], y[kB], z[kB];
#pragma unroll
            for (int b = 0; b < kB; ++b) j[b] = nb[a0 + b < cnt ? a0 + b : cnt - 1];
#pragma unroll
            for (int b = 0; b < kB; ++b) {
                const int jj = (unsigned)j[b] < (unsigned)m ? j[b] : self;
                x[b] = pts[3 * jj];
                y[b] = pts[3 * jj + 1];
                z[b] = pts[3 * jj + 2];
            }
#pragma unroll
            for (int b = 0; b < kB; ++b)
                if (a0 + b < cnt) {
                    const double dx = x[b] - cx, dy = y[b] - cy, dz = z[b] - cz;
                    c00 += dx * dx;
                    c01 += dx * dy;
                    c02 += dx * dz;
                    c11 += dy * dy;
                    c12 += dy * dz;
                    c22 += dz * dz;
                }
        }
        }
        const double cov[6] = {c00 / kd, c01 / kd, c02 / kd, c11 / kd, c12 / kd, c22 / kd};
        double v[3];
        smallest_eigvec_sym3(cov, v); // icp.hpp:55-56
        if (v[2] < 0.0) {             // icp.hpp:59-61
            v[0] = -v[0];
            v[1] = -v[1];
            v[2] = -v[2];
        }
        const double z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; // icp.hpp:63
        if (z > 0.0) {
            const double sn = __dsqrt_rn(z);
            v[0] /= sn;
            v[1] /= sn;
            v[2] /= sn;
        }
        nx = v[0];
        ny = v[1];
        nz = v[2];
    }
    const size_t o = scatter ? (size_t)self : (size_t)i; // by point, or by (sorted) row for the all-gather
    normals[3 * o] = nx;
    normals[3 * o + 1] = ny;
    normals[3 * o + 2] = nz;
}

} // namespace icpmi

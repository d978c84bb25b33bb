// scan_context.h -- Scan Context place-recognition descriptor and its shifted cosine
// distance on the GPU (reference core/scan_context.hpp).
//
//   k_scan_context      scan_context.hpp:44-82   20 rings x 60 sectors, max height per bin
//   k_sc_distances      scan_context.hpp:90-142  one query against a history of descriptors,
//                                                min over the 60 column shifts of 1 - cosine
//
// The max per bin is order independent, so an LDS 64-bit atomic max on an order-preserving
// integer image of the double gives the reference's result bit for bit (up to atan2's last
// ulp at a sector boundary).  The distance sums run in the reference's (ring, sector) order
// with unfused multiply-adds, one thread per (descriptor, shift).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpmi {

constexpr int kScRings = 20, kScSectors = 60, kScCells = kScRings * kScSectors;

__device__ __forceinline__ unsigned long long sc_encode(double v) // order-preserving
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double sc_decode(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(1024) void k_scan_context(const double *__restrict__ cloud, int n,
                                                       double *__restrict__ desc)
{
    __shared__ unsigned long long bins[kScCells];
    const double kMaxRange = 80.0;                         // scan_context.hpp:29
    const double ring_size = kMaxRange / kScRings;         // :47
    const double sector_size = 2.0 * 3.14159265358979323846 / kScSectors; // :48
    const unsigned long long empty = sc_encode(-1.7976931348623157e308); // :45
    for (int e = threadIdx.x; e < kScCells; e += 1024) bins[e] = empty;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double x = cloud[3 * i], y = cloud[3 * i + 1], z = cloud[3 * i + 2];
        const double range = __dsqrt_rn(x * x + y * y);              // :56
        const double angle = atan2(y, x) + 3.14159265358979323846;   // :57
        if (range > kMaxRange || range < 0.1) continue;              // :59
        int ring = (int)(range / ring_size);                         // :62
        int sector = (int)(angle / sector_size);                     // :63
        ring = ring < 0 ? 0 : (ring > kScRings - 1 ? kScRings - 1 : ring);
        sector = sector < 0 ? 0 : (sector > kScSectors - 1 ? kScSectors - 1 : sector);
        if (z == z) atomicMax(&bins[ring * kScSectors + sector], sc_encode(z)); // :69-71 (NaN never wins)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kScCells; e += 1024) {
        const double v = sc_decode(bins[e]);
        desc[e] = v < -1000.0 ? 0.0 : v;                              // :75-81
    }
}

// grid = history size, 64 threads: thread `shift` evaluates column_shifted_distance
__global__ __launch_bounds__(64) void k_sc_distances(const double *__restrict__ query,
                                                     const double *__restrict__ hist, int count,
                                                     double *__restrict__ out)
{
    __shared__ double a[kScCells], b[kScCells];
    const int d = blockIdx.x;
    if (d >= count) return;
    for (int e = threadIdx.x; e < kScCells; e += 64) {
        a[e] = query[e];
        b[e] = hist[(size_t)d * kScCells + e];
    }
    __syncthreads();
    const int shift = threadIdx.x;
    double dist = 1.7976931348623157e308; // scan_context.hpp:91
    if (shift < kScSectors) {
        double sum_ab = 0.0, sum_aa = 0.0, sum_bb = 0.0; // :122-124
        for (int i = 0; i < kScRings; ++i)
            for (int j = 0; j < kScSectors; ++j) {
                int js = j + shift;
                js = js >= kScSectors ? js - kScSectors : js;
                const double va = a[i * kScSectors + j], vb = b[i * kScSectors + js];
                sum_ab += va * vb;
                sum_aa += va * va;
                sum_bb += vb * vb;
            }
        const double norm = __dsqrt_rn(sum_aa) * __dsqrt_rn(sum_bb); // :137
        dist = norm < 1e-10 ? 1.0 : 1.0 - sum_ab / norm;            // :138-141
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(dist, off, 64);
        dist = o < dist ? o : dist;
    }
    if (threadIdx.x == 0) out[d] = dist;
}

} // namespace icpmi

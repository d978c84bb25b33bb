// Micro-benchmark: scheduling variants of the coarse inner loop (tuning aid only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nn_mfma.h"
using namespace icpmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VAR, int QT, int WAVES, int PADKB = 0, int KOFF = 0>
__global__ __launch_bounds__(64 * WAVES) void k_loop(const float4 *__restrict__ Bpack, float *out, int nevertrue)
{
    constexpr int THREADS = 64 * WAVES;
    __shared__ float4 ldsB[32 * 64 + PADKB * 64]; // PADKB KiB of padding limits workgroups per CU
    const int s = blockIdx.y;
    for (int e = 0; e < (32 * 64) / THREADS; ++e) ldsB[threadIdx.x + e * THREADS] = Bpack[(size_t)s * 2048 + threadIdx.x + e * THREADS];
    const int lane = threadIdx.x & 63;
    float a[QT];
    for (int t = 0; t < QT; ++t) a[t] = (float)(lane + t) * 0.01f;
    f32x4 m[QT];
    for (int t = 0; t < QT; ++t) m[t] = (f32x4){kBig, kBig, kBig, kBig};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float keep = 0.f;
    if (VAR == 0) {
#pragma unroll 2
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, zero, 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, zero, 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, zero, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[t][r] = min3f(m[t][r], d0[r], d1[r]);
                    m[t][r] = min3f(m[t][r], d2[r], d3[r]);
                }
            }
        }
    } else if (VAR == 1) { // VALU work independent of the MFMA results
        float x = a[0], y = a[1];
#pragma unroll 2
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, zero, 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, zero, 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, zero, 0, 0, 0);
                asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[t][r] = min3f(m[t][r], x, y);
                    asm volatile("" : "+v"(m[t][r]));
                    m[t][r] = min3f(m[t][r], y, x);
                    asm volatile("" : "+v"(m[t][r]));
                }
            }
        }
    } else if (VAR == 2) { // consume the previous query tile's results (one-stage software pipeline)
        f32x4 p0 = zero, p1 = zero, p2 = zero, p3 = zero;
        int pt = 0;
#pragma unroll 1
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, zero, 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, zero, 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, zero, 0, 0, 0);
                constexpr int dummy = 0;
                (void)dummy;
                const int tp = (t + QT - 1) % QT; // tile whose results p* hold
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[tp][r] = min3f(m[tp][r], p0[r], p1[r]);
                    m[tp][r] = min3f(m[tp][r], p2[r], p3[r]);
                }
                // interleave: MFMA, 2 VALU, MFMA, 2 VALU ...
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                p0 = d0; p1 = d1; p2 = d2; p3 = d3;
            }
        }
        (void)pt;
        // drain (first-iteration garbage min with zero is harmless for timing)
#pragma unroll
        for (int r = 0; r < 4; ++r) m[QT - 1][r] = min3f(m[QT - 1][r], p0[r], p1[r]);
    } else if (VAR == 3) { // min (2-input) instead of min3
#pragma unroll 2
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, zero, 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, zero, 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, zero, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[t][r] = __builtin_fminf(m[t][r], d0[r]);
                    m[t][r] = __builtin_fminf(m[t][r], d1[r]);
                    m[t][r] = __builtin_fminf(m[t][r], d2[r]);
                    m[t][r] = __builtin_fminf(m[t][r], d3[r]);
                }
            }
        }
    } else if (VAR == 4) { // only half the VALU work (1 min3 per MFMA): is the cost per VALU op?
#pragma unroll 2
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, zero, 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, zero, 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, zero, 0, 0, 0);
                asm volatile("" ::"v"(d2), "v"(d3));
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    m[t][r] = min3f(m[t][r], d0[r], d1[r]);
                    m[t][r + 2] = min3f(m[t][r + 2], d0[r + 2], d1[r + 2]);
                }
            }
        }
    }
    else if (VAR == 5) { // integer min on the float bits (C = |P|^2 keeps results >= 0)
        f32x4 cinit[QT];
        for (int t = 0; t < QT; ++t) cinit[t] = (f32x4){5000.f + lane, 5001.f, 5002.f, 5003.f};
        int mi[QT][4];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) mi[t][r] = 0x7f000000;
#pragma unroll 2
        for (int t4 = 0; t4 < 32; ++t4) {
            const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, cinit[t], 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, cinit[t], 0, 0, 0);
                const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, cinit[t], 0, 0, 0);
                const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, cinit[t], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mi[t][r] = min(min(mi[t][r], __float_as_int(d0[r])), __float_as_int(d1[r]));
                    mi[t][r] = min(min(mi[t][r], __float_as_int(d2[r])), __float_as_int(d3[r]));
                }
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) m[t][r] = __int_as_float(mi[t][r]);
    } else if (VAR == 6 || VAR == 7) { // bf16 16x16x32 MFMA (one per 16x16 tile), float (6) or int (7) min
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 ab[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) ab[t][e] = (short)(0x3f80 + lane + t + e);
        f32x4 cinit[QT];
        for (int t = 0; t < QT; ++t) cinit[t] = (f32x4){5000.f + lane, 5001.f, 5002.f, 5003.f};
        int mi[QT][4];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) mi[t][r] = 0x7f000000;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
#pragma unroll 2
        for (int tt = 0; tt < 128; tt += 2) {
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane];
            const bf16x8 b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[t], b0, cinit[t], 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[t], b1, cinit[t], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (VAR == 6) m[t][r] = min3f(m[t][r], d0[r], d1[r]);
                    else mi[t][r] = min(min(mi[t][r], __float_as_int(d0[r])), __float_as_int(d1[r]));
                }
            }
        }
        if (VAR == 7) for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) m[t][r] = __int_as_float(mi[t][r]);
    }
    else if (VAR == 8 || VAR == 9) { // bf16 32x32x16, two chained MFMAs (K = 32) per 32x32 tile; QT = 32-query tiles
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        bf16x8 alo[QT], ahi[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) { alo[t][e] = (short)(0x3f80 + lane + t + e); ahi[t][e] = (short)(0x3f00 + lane + t + e); }
        f32x16 mm[QT];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) mm[t][r] = kBig;
        f32x16 z16;
        for (int r = 0; r < 16; ++r) z16[r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
        f32x16 pa = z16, pb = z16;
        // same pair count as the 16x16 variants: 128 tiles of 16 targets == 64 tiles of 32
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) {
            const bf16x8 b0l = ldsH[((2 * tt) & 31) * 64 + lane], b0h = ldsH[((2 * tt + 1) & 31) * 64 + lane];
            const bf16x8 b1l = ldsH[((2 * tt + 2) & 31) * 64 + lane], b1h = ldsH[((2 * tt + 3) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0l, z16, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], b0h, da, 0, 0, 0);
                f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1l, z16, 0, 0, 0);
                db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], b1h, db, 0, 0, 0);
                if (VAR == 8) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm[t][r] = min3f(mm[t][r], da[r], db[r]);
                } else {
                    const int tp = (t + QT - 1) % QT;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm[tp][r] = min3f(mm[tp][r], pa[r], pb[r]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    pa = da;
                    pb = db;
                }
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) keep += mm[t][r];
    }
    else if (VAR == 10 || VAR == 11) { // bf16 32x32x16, ONE MFMA (K = 16) per 32x32 tile (bf16x2 operands)
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        bf16x8 alo[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) alo[t][e] = (short)(0x3f80 + lane + t + e);
        f32x16 mm[QT];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) mm[t][r] = kBig;
        f32x16 z16;
        for (int r = 0; r < 16; ++r) z16[r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
        f32x16 pa = z16, pb = z16;
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) { // 64 tiles of 32 targets = the same 2048 targets
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane], b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0, z16, 0, 0, 0);
                const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1, z16, 0, 0, 0);
                if (VAR == 10) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm[t][r] = min3f(mm[t][r], da[r], db[r]);
                } else {
                    const int tp = (t + QT - 1) % QT;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm[tp][r] = min3f(mm[tp][r], pa[r], pb[r]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    pa = da;
                    pb = db;
                }
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) keep += mm[t][r];
    }
    else if (VAR == 12) { // bf16 32x32x16, one MFMA per tile; KOFF of the 16 result registers are
                          // min-reduced by the LDS (ds_min_f32, no return) instead of the VALU
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        __shared__ float ldsM[(KOFF > 0 ? KOFF : 1) * QT * WAVES * 64];
        const int wave = threadIdx.x >> 6;
        bf16x8 alo[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) alo[t][e] = (short)(0x3f80 + lane + t + e);
        f32x16 mm[QT];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) mm[t][r] = kBig;
        for (int e = 0; e < KOFF * QT; ++e) ldsM[(e * WAVES + wave) * 64 + lane] = kBig;
        f32x16 z16;
        for (int r = 0; r < 16; ++r) z16[r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) {
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane], b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0, z16, 0, 0, 0);
                const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1, z16, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16 - KOFF; ++r) mm[t][r] = min3f(mm[t][r], da[r], db[r]);
#pragma unroll
                for (int r = 0; r < KOFF; ++r) {
                    float *slot = &ldsM[((t * KOFF + r) * WAVES + wave) * 64 + lane];
                    (void)__hip_atomic_fetch_min(slot, da[16 - KOFF + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_min(slot, db[16 - KOFF + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16 - KOFF; ++r) keep += mm[t][r];
        for (int e = 0; e < KOFF * QT; ++e) keep += ldsM[(e * WAVES + wave) * 64 + lane];
    }
    else if (VAR == 13) { // as 10 with 2-input minima in a tree: t = min(da, db); m = min(m, t)
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        bf16x8 alo[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) alo[t][e] = (short)(0x3f80 + lane + t + e);
        f32x16 mm[QT];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) mm[t][r] = kBig;
        f32x16 z16;
        for (int r = 0; r < 16; ++r) z16[r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) {
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane], b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0, z16, 0, 0, 0);
                const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1, z16, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = __builtin_fminf(da[r], db[r]);
                    asm volatile("" : "+v"(x)); // keep it a 2-input min (no min3 fusion)
                    mm[t][r] = __builtin_fminf(mm[t][r], x);
                }
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) keep += mm[t][r];
    }
    else if (VAR == 14) { // MFMA only: results folded by accumulation in the matrix core (no VALU in the loop)
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        bf16x8 alo[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) alo[t][e] = (short)(0x3f80 + lane + t + e);
        f32x16 mm[QT];
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) mm[t][r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) {
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane], b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                mm[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0, mm[t], 0, 0, 0);
                mm[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1, mm[t], 0, 0, 0);
            }
        }
        for (int t = 0; t < QT; ++t) for (int r = 0; r < 16; ++r) keep += mm[t][r];
    }
    else if (VAR == 15) { // MFMA only, INDEPENDENT results (C = 0, results only kept alive): the pipe's issue rate
        typedef short bf16x8 __attribute__((ext_vector_type(8)));
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        bf16x8 alo[QT];
        for (int t = 0; t < QT; ++t) for (int e = 0; e < 8; ++e) alo[t][e] = (short)(0x3f80 + lane + t + e);
        f32x16 z16;
        for (int r = 0; r < 16; ++r) z16[r] = 0.f;
        const bf16x8 *ldsH = reinterpret_cast<const bf16x8 *>(ldsB);
#pragma unroll 1
        for (int tt = 0; tt < 64; tt += 2) {
            const bf16x8 b0 = ldsH[(tt & 31) * 64 + lane], b1 = ldsH[((tt + 1) & 31) * 64 + lane];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b0, z16, 0, 0, 0);
                const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], b1, z16, 0, 0, 0);
                asm volatile("" ::"v"(da), "v"(db));
            }
        }
    }
    float acc = keep;
    for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) acc += m[t][r];
    if (acc == (float)nevertrue) out[threadIdx.x] = acc;
}

template <typename F>
static float timeit(F f, int reps = 10)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main()
{
    const int n = 100000, m = 100000;
    const int splits = (m + kSplitTargets - 1) / kSplitTargets;
    float4 *bp; float *out;
    CK(hipMalloc(&bp, (size_t)splits * 2048 * 16)); CK(hipMalloc(&out, 4096));
    std::vector<float> h((size_t)splits * 2048 * 4);
    srand(2);
    for (auto &v : h) v = -50 + 100.0f * (float)(rand() % 100000) / 100000.0f;
    CK(hipMemcpy(bp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double ideal_ms = (double)n * m / 256.0 * 32.0 / 1024.0 / 2.4e9 * 1e3;
#define RUN(VAR, QT, W) { const int qpb = 16 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_loop<VAR, QT, W>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("var %d QT=%d W=%d : %.3f ms  (%.1f%%)\n", VAR, QT, W, ms, 100 * ideal_ms / ms); }
#define RUN32(VAR, QT, W) { const int qpb = 32 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_loop<VAR, QT, W>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("var %d QT32=%d W=%d : %.3f ms  (%.1f%%)\n", VAR, QT, W, ms, 100 * ideal_ms / ms); }
#define RUNP(VAR, QT, W, PAD) { const int qpb = 32 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_loop<VAR, QT, W, PAD>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("var %d QT32=%d W=%d padKB=%d : %.3f ms  (%.1f%%)\n", VAR, QT, W, PAD, ms, 100 * ideal_ms / ms); }
#define RUNK(VAR, QT, W, K) { const int qpb = 32 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_loop<VAR, QT, W, 0, K>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("var %d QT32=%d W=%d KOFF=%d : %.3f ms  (%.1f%%)\n", VAR, QT, W, K, ms, 100 * ideal_ms / ms); }
    RUN32(10, 2, 8) RUN32(14, 2, 8) RUN32(15, 2, 8) RUN32(15, 2, 4) RUN32(15, 4, 4) RUN32(14, 4, 4)
    RUNK(12, 2, 8, 0)
    return 0;
}

// Micro-benchmark for k_nn_coarse variants (not part of the product; tuning aid).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lidar_slam_from_scratch_amd/csrc scripts/micro/bench_coarse.hip -o /tmp/bench_coarse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nn_mfma.h"
using namespace icpmi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// loop only: no epilogue, B straight from LDS; keeps m alive with a never-true store
template <int QT, int WAVES, int MINWAVES>
__global__ __launch_bounds__(64 * WAVES, MINWAVES) void k_loop_only(const float4 *__restrict__ Bpack, float *out, int nevertrue)
{
    constexpr int THREADS = 64 * WAVES;
    __shared__ float4 ldsB[32 * 64];
    const int s = blockIdx.y;
    for (int e = 0; e < (32 * 64) / THREADS; ++e) ldsB[threadIdx.x + e * THREADS] = Bpack[(size_t)s * 2048 + threadIdx.x + e * THREADS];
    const int lane = threadIdx.x & 63;
    float a[QT];
    for (int t = 0; t < QT; ++t) a[t] = (float)(lane + t) * 0.01f;
    f32x4 m[QT];
    for (int t = 0; t < QT; ++t) m[t] = (f32x4){kBig, kBig, kBig, kBig};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
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
    float acc = 0;
    for (int t = 0; t < QT; ++t) for (int r = 0; r < 4; ++r) acc += m[t][r];
    if (acc == (float)nevertrue) out[threadIdx.x] = acc;
}

// MFMA only, no VALU tracking at all (accumulate in the MFMA itself)
template <int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_only(const float4 *__restrict__ Bpack, float *out, int nevertrue)
{
    constexpr int THREADS = 64 * WAVES;
    __shared__ float4 ldsB[32 * 64];
    const int s = blockIdx.y;
    for (int e = 0; e < (32 * 64) / THREADS; ++e) ldsB[threadIdx.x + e * THREADS] = Bpack[(size_t)s * 2048 + threadIdx.x + e * THREADS];
    const int lane = threadIdx.x & 63;
    float a[QT];
    for (int t = 0; t < QT; ++t) a[t] = (float)(lane + t) * 0.01f;
    f32x4 m[QT];
    for (int t = 0; t < QT; ++t) m[t] = (f32x4){0, 0, 0, 0};
    __syncthreads();
#pragma unroll 2
    for (int t4 = 0; t4 < 32; ++t4) {
        const float4 b = ldsB[t4 * 64 + lane];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.x, m[t], 0, 0, 0);
            m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.y, m[t], 0, 0, 0);
            m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.z, m[t], 0, 0, 0);
            m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b.w, m[t], 0, 0, 0);
        }
    }
    float acc = 0;
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

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 100000, m = argc > 2 ? atoi(argv[2]) : 100000;
    std::vector<double> hq(3 * (size_t)n), ht(3 * (size_t)m);
    srand(1);
    for (auto &v : hq) v = -50 + 100.0 * rand() / RAND_MAX;
    for (auto &v : ht) v = -50 + 100.0 * rand() / RAND_MAX;
    double *dq, *dt, *part; NnFrame *frame; float4 *bp; float2 *coarse; float *out;
    const int splits = (m + kSplitTargets - 1) / kSplitTargets;
    CK(hipMalloc(&dq, hq.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8)); CK(hipMalloc(&part, 256 * 6 * 8));
    CK(hipMalloc(&frame, sizeof(NnFrame))); CK(hipMalloc(&bp, (size_t)splits * 2048 * 16));
    CK(hipMalloc(&coarse, (size_t)splits * n * 8)); CK(hipMalloc(&out, 4096));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bbox_partial, dim3(256), dim3(256), 0, 0, dt, m, part);
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, 0, part, 256, frame);
    hipLaunchKernelGGL(k_pack_targets, dim3((splits * 2048 + 255) / 256), dim3(256), 0, 0, dt, m, frame, bp, splits);
    CK(hipDeviceSynchronize());
    const double ideal_ms = (double)n * m / 256.0 * 32.0 / 1024.0 / 2.4e9 * 1e3;
    printf("n=%d m=%d splits=%d ideal(2.4GHz) %.3f ms\n", n, m, splits, ideal_ms);
#define RUN_FULL(QT, W) { const int qpb = 16 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_nn_coarse<0, QT, W>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, dq, n, bp, frame, coarse, (float *)nullptr, (const IcpState *)nullptr); }); printf("full      QT=%d W=%d : %.3f ms  (%.1f%% of ideal)\n", QT, W, ms, 100 * ideal_ms / ms); }
#define RUN_LOOP(QT, W, MW) { const int qpb = 16 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_loop_only<QT, W, MW>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("loop-only QT=%d W=%d minw=%d : %.3f ms  (%.1f%%)\n", QT, W, MW, ms, 100 * ideal_ms / ms); }
#define RUN_MFMA(QT, W) { const int qpb = 16 * QT * W; float ms = timeit([&] { hipLaunchKernelGGL((k_mfma_only<QT, W>), dim3((n + qpb - 1) / qpb, splits), dim3(64 * W), 0, 0, bp, out, -12345); }); printf("mfma-only QT=%d W=%d : %.3f ms  (%.1f%%)\n", QT, W, ms, 100 * ideal_ms / ms); }
    RUN_FULL(4, 8) RUN_FULL(4, 4) RUN_FULL(8, 4) RUN_FULL(8, 8) RUN_FULL(8, 2) RUN_FULL(12, 4) RUN_FULL(16, 4) RUN_FULL(16, 2)
    RUN_LOOP(4, 8, 1) RUN_LOOP(4, 4, 1) RUN_LOOP(8, 4, 1) RUN_LOOP(8, 8, 1) RUN_LOOP(16, 4, 1) RUN_LOOP(4, 8, 2) RUN_LOOP(8, 4, 2)
    RUN_MFMA(4, 8) RUN_MFMA(8, 4) RUN_MFMA(16, 4)
    return 0;
}

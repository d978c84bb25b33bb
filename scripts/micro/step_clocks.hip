// Where the single-GPU loop's fused kernel (k_finish_step_transform) spends its ~10 us on a
// filtered scan: a copy of its body with wall_clock64() stamps (100 MHz) between the phases, on
// synthetic partial rows.  Tuning aid, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lidar_slam_from_scratch_amd/csrc \
//         scripts/micro/step_clocks.hip -o /tmp/step_clocks && /tmp/step_clocks [rows] [points]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

using namespace icpmi;

constexpr int kStamps = 8;

__global__ __launch_bounds__(kFinishThreads) void k_probe(const double *__restrict__ partials, int nblocks, int n_local,
                                                          const double *in, double *out, int n, const IcpState *sin,
                                                          IcpState *sout, double *history, long long *stamps)
{
    __shared__ IcpState ls, sums;
    long long t[kStamps];
    t[0] = wall_clock64();
    const int i0 = blockIdx.x * kFinishThreads + threadIdx.x;
    double x = 0.0, y = 0.0, z = 0.0;
    if (i0 < n) x = in[3 * i0], y = in[3 * i0 + 1], z = in[3 * i0 + 2];
    state_copy(&ls, sin);
    __syncthreads();
    t[1] = wall_clock64(); // state in LDS
    finish_sums(partials, nblocks, n_local, &sums);
    __syncthreads();
    t[2] = wall_clock64(); // sums formed
    if (!ls.done && threadIdx.x < kNumSums) ls.sums[threadIdx.x] = sums.sums[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) step_update(&ls, blockIdx.x == 0 ? history : nullptr, 0);
    __syncthreads();
    t[3] = wall_clock64(); // step done
    if (blockIdx.x == 0) state_copy(sout, &ls);
    const double *T = ls.delta;
    const double r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
    const double r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
    const double r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
    for (int i = i0; i < n; i += gridDim.x * kFinishThreads) {
        if (i != i0) x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i] = ((x * r00 + y * r01) + z * r02) + t0;
        out[3 * i + 1] = ((x * r10 + y * r11) + z * r12) + t1;
        out[3 * i + 2] = ((x * r20 + y * r21) + z * r22) + t2;
    }
    t[4] = wall_clock64(); // points stored (issued)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) stamps[k] = t[k];
}


// the pieces of step_update, one thread
__global__ void k_probe_step(const IcpState *sin, IcpState *sout, long long *stamps)
{
    __shared__ IcpState ls;
    state_copy(&ls, sin);
    __syncthreads();
    if (threadIdx.x != 0) return;
    long long t[6];
    t[0] = wall_clock64();
    const double error = __dsqrt_rn(ls.sums[27] / ls.sums[28]);
    ls.last_error = error;
    t[1] = wall_clock64();
    double x[6];
    ldlt6_solve(ls.sums, x);
    ls.delta[0] = x[0]; // (keep the result alive at this point)
    t[2] = wall_clock64();
    twist_to_transform(x, ls.delta);
    t[3] = wall_clock64();
    mul44(ls.delta, ls.total, ls.total);
    t[4] = wall_clock64();
    for (int k = 0; k < 5; ++k) stamps[k] = t[k];
    for (int e = 0; e < 16; ++e) sout->total[e] = ls.total[e];
    sout->last_error = ls.last_error;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 250, n = argc > 2 ? atoi(argv[2]) : 8000;
    std::vector<double> part((size_t)rows * kSumsStride, 0.0), pts(3 * (size_t)n);
    srand(1);
    for (auto &v : pts) v = rand() / (double)RAND_MAX * 40.0 - 20.0;
    // a well-conditioned system: J rows of random points against random unit normals
    for (int r = 0; r < rows; ++r) {
        double acc[28] = {0};
        for (int q = 0; q < 32; ++q) {
            double p[3], nn[3], J[6];
            for (int a = 0; a < 3; ++a) p[a] = rand() / (double)RAND_MAX * 40.0 - 20.0, nn[a] = rand() / (double)RAND_MAX - 0.5;
            const double l = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            for (int a = 0; a < 3; ++a) nn[a] /= l;
            J[0] = p[1] * nn[2] - p[2] * nn[1], J[1] = p[2] * nn[0] - p[0] * nn[2], J[2] = p[0] * nn[1] - p[1] * nn[0];
            J[3] = nn[0], J[4] = nn[1], J[5] = nn[2];
            const double b = 0.01 * (rand() / (double)RAND_MAX - 0.5);
            int o = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) acc[o++] += J[i] * J[j];
            for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * b;
            acc[27] += b * b;
        }
        for (int e = 0; e < 28; ++e) part[(size_t)r * kSumsStride + e] = acc[e];
    }
    IcpState hs = {};
    for (int i = 0; i < 4; ++i) hs.total[5 * i] = 1.0;
    hs.prev_error = 1e300, hs.tolerance = 0.0, hs.min_error = 0.0, hs.max_hist = 64;
    double *d_part, *d_in, *d_out, *d_hist;
    IcpState *d_s;
    long long *d_st;
    CK(hipMalloc(&d_part, part.size() * 8));
    CK(hipMalloc(&d_in, pts.size() * 8));
    CK(hipMalloc(&d_out, pts.size() * 8));
    CK(hipMalloc(&d_hist, 8 * 128));
    CK(hipMalloc(&d_s, 2 * sizeof(IcpState)));
    CK(hipMalloc(&d_st, 8 * kStamps));
    CK(hipMemcpy(d_part, part.data(), part.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
    const int blocks = std::max(1, std::min(32, (n + kFinishThreads - 1) / kFinishThreads));
    std::vector<std::vector<long long>> all;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 30; ++rep) {
        CK(hipMemcpy(d_s, &hs, sizeof(hs), hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(kFinishThreads), 0, 0, (const double *)d_part, rows, rows * 32,
                           (const double *)d_in, d_out, n, (const IcpState *)d_s, d_s + 1, d_hist, d_st);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        std::vector<long long> st(kStamps);
        CK(hipMemcpy(st.data(), d_st, 8 * kStamps, hipMemcpyDeviceToHost));
        all.push_back(st);
    }
    const char *names[] = {"state -> LDS", "partial rows -> sums", "step_update", "store state + move points"};
    for (int k = 0; k < 4; ++k) {
        std::vector<double> v;
        for (auto &st : all) v.push_back((st[k + 1] - st[k]) * 10.0);
        std::sort(v.begin(), v.end());
        printf("%-28s median %7.0f ns\n", names[k], v[v.size() / 2]);
    }
    {   // pieces of the step, with the sums the first kernel left in the second state buffer
        std::vector<std::vector<long long>> al2;
        IcpState h2;
        CK(hipMemcpy(&h2, d_s + 1, sizeof(h2), hipMemcpyDeviceToHost));
        h2.done = 0;
        CK(hipMemcpy(d_s, &h2, sizeof(h2), hipMemcpyHostToDevice));
        for (int rep = 0; rep < 30; ++rep) {
            hipLaunchKernelGGL(k_probe_step, dim3(1), dim3(64), 0, 0, (const IcpState *)d_s, d_s + 1, d_st);
            CK(hipDeviceSynchronize());
            std::vector<long long> st(kStamps);
            CK(hipMemcpy(st.data(), d_st, 8 * kStamps, hipMemcpyDeviceToHost));
            al2.push_back(st);
        }
        const char *n2[] = {"  error = sqrt(sum / count)", "  ldlt6_solve", "  twist_to_transform", "  mul44"};
        for (int k = 0; k < 4; ++k) {
            std::vector<double> w;
            for (auto &st : al2) w.push_back((st[k + 1] - st[k]) * 10.0);
            std::sort(w.begin(), w.end());
            printf("%-28s median %7.0f ns\n", n2[k], w[w.size() / 2]);
        }
    }
    std::vector<double> v;
    for (auto &st : all) v.push_back((st[4] - st[0]) * 10.0);
    std::sort(v.begin(), v.end());
    std::sort(ms.begin(), ms.end());
    printf("inside the kernel            median %7.0f ns; event-to-event %.1f us (rows %d, points %d, %d workgroups)\n",
           v[v.size() / 2], 1e3 * ms[ms.size() / 2], rows, n, blocks);
    return 0;
}

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
    if (threadIdx.x < 64) step_update_wave(&ls, blockIdx.x == 0 ? history : nullptr, 0, threadIdx.x);
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

// the same pieces by the whole wave (step_update_wave's parts)
__global__ void k_probe_step_wave(const IcpState *sin, IcpState *sout, long long *stamps)
{
    __shared__ IcpState ls;
    state_copy(&ls, sin);
    __syncthreads();
    const int lane = threadIdx.x;
    long long t[6];
    t[0] = wall_clock64();
    const double error = __dsqrt_rn(ls.sums[27] / ls.sums[28]);
    if (lane == 0) ls.last_error = error;
    t[1] = wall_clock64();
    double x[6], T[16];
    ldlt6_solve_wave(ls.sums, x, lane);
    if (lane == 0) ls.delta[0] = x[0];
    t[2] = wall_clock64();
    twist_to_transform(x, T);
    if (lane == 0)
        for (int e = 0; e < 16; ++e) ls.delta[e] = T[e];
    __builtin_amdgcn_wave_barrier();
    t[3] = wall_clock64();
    mul44_wave(ls.delta, ls.total, ls.total, lane);
    t[4] = wall_clock64();
    if (lane == 0) {
        for (int k = 0; k < 5; ++k) stamps[k] = t[k];
        for (int e = 0; e < 16; ++e) sout->total[e] = ls.total[e];
        sout->last_error = ls.last_error;
    }
}

// a second copy of twist_to_transform for the one-lane leg of the comparison
__device__ inline void twist_ref(const double *x, double *T)
{
    const double rx = x[0], ry = x[1], rz = x[2];
    const double angle = __dsqrt_rn((rx * rx + ry * ry) + rz * rz);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(angle < 1e-10)) {
        const double ax = rx / angle, ay = ry / angle, az = rz / angle;
        const double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
        double s, c;
        sincos_step(angle, &s, &c);
        const double c1 = 1.0 - c;
        double Kc[9];
        for (int e = 0; e < 9; ++e) Kc[e] = c1 * K[e];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double kk = (Kc[3 * i] * K[j] + Kc[3 * i + 1] * K[3 + j]) + Kc[3 * i + 2] * K[6 + j];
                R[3 * i + j] = (R[3 * i + j] + s * K[3 * i + j]) + kk;
            }
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = x[3 + i];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// sincos_step against the device library's sin and cos: the largest difference in units of the last place
__global__ void k_sincos_ulps(int n, double *out /* per block: max ulps of sin, of cos */)
{
    __shared__ double ms[256], mc[256];
    double es = 0.0, ec = 0.0;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        // angles from 1e-10 to pi/4, logarithmically and linearly spaced halves
        const double u = (q + 0.5) / n;
        const double a = (q & 1) ? 0.78539816339744828 * u : exp(log(1e-10) + u * (log(0.78539816339744828) - log(1e-10)));
        double s, c;
        sincos_step(a, &s, &c);
        const double s0 = sin(a), c0 = cos(a);
        const double us = fabs(s - s0) / (fabs(s0) * 2.220446049250313e-16), uc = fabs(c - c0) / (fabs(c0) * 2.220446049250313e-16);
        es = us > es ? us : es;
        ec = uc > ec ? uc : ec;
    }
    ms[threadIdx.x] = es, mc[threadIdx.x] = ec;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)blockDim.x; ++k) es = ms[k] > es ? ms[k] : es, ec = mc[k] > ec ? mc[k] : ec;
        out[2 * blockIdx.x] = es, out[2 * blockIdx.x + 1] = ec;
    }
}

// one workgroup of 64 per system: the one-lane step (ldlt6_solve, twist with sin / cos, mul44) against the wave's, bit for
// bit; out[b] = number of differing words (0..6+16+16)
__global__ __launch_bounds__(64) void k_compare(const double *systems /* 27 per system */, const double *totals /* 16 */, int *out)
{
    __shared__ double sm[27], tot[16], dl[16], tw[16];
    const int lane = threadIdx.x, b = blockIdx.x;
    if (lane < 27) sm[lane] = systems[(size_t)b * 27 + lane];
    if (lane < 16) tot[lane] = totals[(size_t)b * 16 + lane], tw[lane] = tot[lane];
    __syncthreads();
    double x0[6], x1[6], T0[16], T1[16], C0[16];
    ldlt6_solve(sm, x0);
    ldlt6_solve_wave(sm, x1, lane);
    twist_ref(x0, T0);
    twist_to_transform(x1, T1);
    mul44(T0, tot, C0);
    if (lane == 0)
        for (int e = 0; e < 16; ++e) dl[e] = T1[e];
    __builtin_amdgcn_wave_barrier();
    mul44_wave(dl, tw, tw, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        // same bits, or both NaN (which operand's payload a NaN result carries is the compiler's choice of operand order)
        auto differ = [](double a, double c) { return __double_as_longlong(a) != __double_as_longlong(c) && !(a != a && c != c); };
        int bad = 0;
        for (int e = 0; e < 6; ++e) bad += differ(x0[e], x1[e]);
        for (int e = 0; e < 16; ++e) bad += differ(T0[e], T1[e]);
        for (int e = 0; e < 16; ++e) bad += differ(C0[e], tw[e]);
        out[b] = bad;
    }
}

// lane_xor<X> and the row16_partner reductions of device_math.h against __shfl_xor, every lane of a wave
__global__ __launch_bounds__(64) void k_lane_ops(int *bad)
{
    const int lane = threadIdx.x;
    int nbad = 0;
    for (int rep = 0; rep < 64; ++rep) {
        const int vi = (lane * 2654435761u + rep * 40503u) ^ (rep << 20);
        const double vd = (double)vi * 1.25 + rep;
        nbad += lane_xor<1>(vi) != __shfl_xor(vi, 1, 64);
        nbad += lane_xor<2>(vi) != __shfl_xor(vi, 2, 64);
        nbad += lane_xor<4>(vi) != __shfl_xor(vi, 4, 64);
        nbad += lane_xor<8>(vi) != __shfl_xor(vi, 8, 64);
        nbad += lane_xor<16>(vi) != __shfl_xor(vi, 16, 64);
        nbad += lane_xor<32>(vi) != __shfl_xor(vi, 32, 64);
        nbad += lane_xor<4>(vd) != __shfl_xor(vd, 4, 64);
        nbad += lane_xor<16>(vd) != __shfl_xor(vd, 16, 64);
        nbad += lane_xor<32>(vd) != __shfl_xor(vd, 32, 64);
        // (value, index) minimum over a row of 16 lanes, ties on the value included (vi & 3)
        double d0 = (double)(vi & 3), d1 = d0;
        int j0 = vi >> 8, j1 = j0;
        for (int x = 1; x < 16; x <<= 1) {
            const double od = __shfl_xor(d0, x, 64);
            const int oj = __shfl_xor(j0, x, 64);
            if (od < d0 || (od == d0 && oj < j0)) d0 = od, j0 = oj;
        }
#define STEP(S) { const double od = row16_partner<S>(d1); const int oj = row16_partner<S>(j1); const bool t = (od < d1) | ((od == d1) & (oj < j1)); d1 = t ? od : d1; j1 = t ? oj : j1; }
        STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
        nbad += d0 != d1 || j0 != j1;
        // prefix sums: the whole wave, and every row of 16 lanes
        unsigned inc = (unsigned)vi & 1023u, ref = inc;
        int rinc = vi & 255, rref = rinc;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(ref, off, 64);
            ref += lane >= off ? o : 0u;
        }
        for (int off = 1; off < 16; off <<= 1) {
            const int o = __shfl_up(rref, off, 64);
            rref += (lane & 15) >= off ? o : 0;
        }
        nbad += wave_scan_incl(inc) != ref;
        nbad += row16_scan_incl(rinc) != rref;
    }
    atomicAdd(bad, nbad);
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
    {   // the same pieces by the whole wave
        std::vector<std::vector<long long>> al2;
        for (int rep = 0; rep < 30; ++rep) {
            hipLaunchKernelGGL(k_probe_step_wave, dim3(1), dim3(64), 0, 0, (const IcpState *)d_s, d_s + 1, d_st);
            CK(hipDeviceSynchronize());
            std::vector<long long> st(kStamps);
            CK(hipMemcpy(st.data(), d_st, 8 * kStamps, hipMemcpyDeviceToHost));
            al2.push_back(st);
        }
        const char *n2[] = {"  wave: error", "  wave: ldlt6_solve_wave", "  wave: twist_to_transform", "  wave: mul44_wave"};
        for (int k = 0; k < 4; ++k) {
            std::vector<double> w;
            for (auto &st : al2) w.push_back((st[k + 1] - st[k]) * 10.0);
            std::sort(w.begin(), w.end());
            printf("%-28s median %7.0f ns\n", n2[k], w[w.size() / 2]);
        }
    }
    {   // bit comparison of the one-lane and the wave step on many systems: well-conditioned ones, rank-deficient ones
        // (planes, lines: zero and near-zero pivots), equal diagonals (pivot ties), zeros, huge / tiny scales, NaN
        const int nsys = argc > 3 ? atoi(argv[3]) : 200000;
        std::vector<double> sys((size_t)nsys * 27), tots((size_t)nsys * 16);
        srand(7);
        auto rnd = [] { return rand() / (double)RAND_MAX * 2.0 - 1.0; };
        for (int b = 0; b < nsys; ++b) {
            double acc[27] = {0};
            const int kind = b % 8, rowsn = 3 + rand() % 40;
            const double scale = std::pow(10.0, (rand() % 13) - 6);
            for (int q = 0; q < rowsn; ++q) {
                double pp[3], nn[3], J[6];
                for (int a = 0; a < 3; ++a) pp[a] = rnd() * 20.0 * scale, nn[a] = rnd();
                if (kind == 1) nn[0] = 0, nn[1] = 0, nn[2] = 1;                  // one plane: rank 3
                if (kind == 2) nn[2] = 0;                                        // normals in a plane
                if (kind == 3) pp[0] = pp[1] = pp[2] = 0;                        // rotation unobservable
                if (kind == 4) for (int a = 0; a < 3; ++a) pp[a] = std::round(pp[a]), nn[a] = std::round(nn[a] * 2) / 2; // ties
                const double l = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                if (l > 0) for (int a = 0; a < 3; ++a) nn[a] /= l;
                J[0] = pp[1] * nn[2] - pp[2] * nn[1], J[1] = pp[2] * nn[0] - pp[0] * nn[2], J[2] = pp[0] * nn[1] - pp[1] * nn[0];
                J[3] = nn[0], J[4] = nn[1], J[5] = nn[2];
                const double bb = (kind == 5 ? 3.0 : 0.01) * rnd();               // (kind 5: large angles)
                int o = 0;
                for (int i = 0; i < 6; ++i)
                    for (int j = i; j < 6; ++j) acc[o++] += J[i] * J[j];
                for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * bb;
            }
            if (kind == 6) for (int e = 0; e < 27; ++e) acc[e] = (rand() % 3) - 1.0;      // small integers: indefinite, ties, zeros
            if (kind == 7) { for (int e = 0; e < 27; ++e) acc[e] = 0.0; if (rand() % 2) acc[rand() % 27] = NAN; else acc[rand() % 21] = 1.0; }
            for (int e = 0; e < 27; ++e) sys[(size_t)b * 27 + e] = acc[e];
            for (int e = 0; e < 16; ++e) tots[(size_t)b * 16 + e] = rnd() * 3.0;
        }
        double *d_sys, *d_tot;
        int *d_bad;
        CK(hipMalloc(&d_sys, sys.size() * 8));
        CK(hipMalloc(&d_tot, tots.size() * 8));
        CK(hipMalloc(&d_bad, nsys * 4));
        CK(hipMemcpy(d_sys, sys.data(), sys.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tot, tots.data(), tots.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_compare, dim3(nsys), dim3(64), 0, 0, (const double *)d_sys, (const double *)d_tot, d_bad);
        CK(hipDeviceSynchronize());
        std::vector<int> bad(nsys);
        CK(hipMemcpy(bad.data(), d_bad, nsys * 4, hipMemcpyDeviceToHost));
        long long nbad = 0, per[8] = {0};
        for (int b = 0; b < nsys; ++b) if (bad[b]) ++nbad, ++per[b % 8];
        printf("one-lane step vs wave step, %d systems: %lld differ", nsys, nbad);
        for (int k = 0; k < 8; ++k) printf(" [%d]=%lld", k, per[k]);
        printf("\n");
        if (nbad) return 2;
        int *d_lb, lb = -1;
        CK(hipMalloc(&d_lb, 4));
        CK(hipMemset(d_lb, 0, 4));
        hipLaunchKernelGGL(k_lane_ops, dim3(8), dim3(64), 0, 0, d_lb);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&lb, d_lb, 4, hipMemcpyDeviceToHost));
        printf("lane_xor<1..32>, the row16_partner minimum and the DPP prefix sums against __shfl_xor / __shfl_up: %d differences\n", lb);
        if (lb) return 4;
        double *d_u;
        CK(hipMalloc(&d_u, 2 * 256 * 8));
        hipLaunchKernelGGL(k_sincos_ulps, dim3(256), dim3(256), 0, 0, 1 << 24, d_u);
        CK(hipDeviceSynchronize());
        std::vector<double> u(512);
        CK(hipMemcpy(u.data(), d_u, 512 * 8, hipMemcpyDeviceToHost));
        double us = 0, uc = 0;
        for (int k = 0; k < 256; ++k) us = std::max(us, u[2 * k]), uc = std::max(uc, u[2 * k + 1]);
        printf("sincos_step vs the device library's sin / cos on 16.7M angles in [1e-10, pi/4): max difference %.2f / %.2f ulp\n", us, uc);
        if (us > 2.0 || uc > 2.0) return 3;
    }
    std::vector<double> v;
    for (auto &st : all) v.push_back((st[4] - st[0]) * 10.0);
    std::sort(v.begin(), v.end());
    std::sort(ms.begin(), ms.end());
    printf("inside the kernel            median %7.0f ns; event-to-event %.1f us (rows %d, points %d, %d workgroups)\n",
           v[v.size() / 2], 1e3 * ms[ms.size() / 2], rows, n, blocks);
    return 0;
}

// device_math.h -- small exact-fp64 routines shared by the ICP kernels.
//
// Everything here is written so that the same sequence of IEEE-754 double operations
// (+ - * / sqrt, no FMA: the translation unit is built with -ffp-contract=off) runs on
// the GPU as in a plain host build of the reference, which is what lets the parity
// tests compare normals and nearest-neighbour distances bit for bit.
//
// Reference lines (slam_viz/include/slam_viz/core/):
//   squared distance      kdtree.hpp:124,156
//   PCA normal            icp.hpp:40-63
//   J row, residual       icp.hpp:99-117
//   6x6 LDLT solve        icp.hpp:120 (Eigen 3.4 LDLT, pivoted, restated)
//   Rodrigues             icp.hpp:127-141
//   rigid apply/compose   types.hpp:110-125
#pragma once
#include <hip/hip_runtime.h>

namespace icpmi {

__device__ __forceinline__ double sqdist(double ax, double ay, double az, double bx, double by,
                                         double bz)
{
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// value of `v` in lane `l` (wave-uniform l): two v_readlane_b32, the result lives in scalar registers
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// value of `v` in lane (own lane - N) of the same row of 16 lanes (DPP row_shr: vector-ALU moves, no LDS trip); lanes
// whose source falls off the row get 0
template <int N> __device__ __forceinline__ double row_shr_f64(double v)
{
    static_assert(N >= 1 && N <= 15, "row_shr:1..15");
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Partner values for an all-to-all reduction over a ROW of 16 lanes (a quarter-wave) in four steps, by DPP -- vector-ALU
// moves, where __shfl_xor is an LDS crossbar trip (ds_bpermute, ~100 cycles each on a dependent chain): step 0 reads lane
// ^ 1, step 1 lane ^ 2 (quad permutations), step 2 the mirrored lane of the other quad of its half (row_half_mirror),
// step 3 the mirrored lane of the other half (row_mirror).  After steps 0-1 the four lanes of a quad agree, after step 2
// the eight of a half, so WHICH lane of the other group is read makes no difference -- provided the combining operation is
// a minimum in a total order (no NaN on either side: the callers' distances never are).
template <int S> __device__ __forceinline__ int row16_partner(int v)
{
    static_assert(S >= 0 && S < 4, "four steps");
    constexpr int ctrl = S == 0 ? 0xB1 : S == 1 ? 0x4E : S == 2 ? 0x141 : 0x140;
    return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true); // (every lane has a source: nothing is left undefined)
}
template <int S> __device__ __forceinline__ float row16_partner(float v) { return __int_as_float(row16_partner<S>(__float_as_int(v))); }
template <int S> __device__ __forceinline__ double row16_partner(double v)
{
    return __hiloint2double(row16_partner<S>(__double2hiint(v)), row16_partner<S>(__double2loint(v)));
}

// (d, j) <- the smallest (distance, index) pair among the 16 lanes of a row, in every lane of it: the minimum of d first
// (v_min_f64 on the DPP partners; d is never a NaN here), then the smallest index among the lanes that hold it -- 22
// vector instructions where the pairwise form (three compares and three selects per step) takes 36.
__device__ __forceinline__ double vmin_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); // (no canonicalisation around it: operands are numbers)
    return r;
}
__device__ __forceinline__ void row16_argmin(double &d, int &j)
{
    double m = d;
    m = vmin_f64(m, row16_partner<0>(m));
    m = vmin_f64(m, row16_partner<1>(m));
    m = vmin_f64(m, row16_partner<2>(m));
    m = vmin_f64(m, row16_partner<3>(m));
    int jj = d == m ? j : 0x7fffffff;
    jj = min(jj, row16_partner<0>(jj));
    jj = min(jj, row16_partner<1>(jj));
    jj = min(jj, row16_partner<2>(jj));
    jj = min(jj, row16_partner<3>(jj));
    d = m;
    j = jj;
}

// __shfl_xor(v, X, 64) without the LDS crossbar: the value of lane ^ X by vector-ALU moves.  X = 1, 2: quad permutations;
// 8: a row rotation by 8; 4: rotations by 4 and by 12, chosen by the lane's bit 2; 16 and 32: gfx950's
// v_permlane16_swap / v_permlane32_swap on two copies of the value (odd rows of one exchanged with even rows of the
// other; upper half of one with lower half of the other).  scripts/micro/step_clocks.hip checks every X against
// __shfl_xor on the device.
template <int X> __device__ __forceinline__ int lane_xor(int v)
{
    static_assert(X == 1 || X == 2 || X == 4 || X == 8 || X == 16 || X == 32, "one bit");
    if constexpr (X == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); // (every lane has a source lane)
    else if constexpr (X == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    else if constexpr (X == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);
    else if constexpr (X == 4) {
        const int below = __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, true); // lane - 4 (mod 16)
        const int above = __builtin_amdgcn_mov_dpp(v, 0x12C, 0xf, 0xf, true); // lane + 4 (mod 16)
        return (__lane_id() & 4) ? below : above;
    } else if constexpr (X == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((__lane_id() & 16) ? r[0] : r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((__lane_id() & 32) ? r[0] : r[1]);
    }
}
template <int X> __device__ __forceinline__ unsigned lane_xor(unsigned v) { return (unsigned)lane_xor<X>((int)v); }
template <int X> __device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor<X>(__float_as_int(v))); }
template <int X> __device__ __forceinline__ double lane_xor(double v)
{
    return __hiloint2double(lane_xor<X>(__double2hiint(v)), lane_xor<X>(__double2loint(v)));
}

// lane_xor with the distance as an argument that is a constant after unrolling (the steps of a bitonic sort)
template <typename T> __device__ __forceinline__ T lane_xor_n(T v, int x)
{
    switch (x) {
    case 1: return lane_xor<1>(v);
    case 2: return lane_xor<2>(v);
    case 4: return lane_xor<4>(v);
    case 8: return lane_xor<8>(v);
    case 16: return lane_xor<16>(v);
    default: return lane_xor<32>(v);
    }
}

// inclusive prefix sum over the 64 lanes of a wave, six DPP additions (row shifts by 1, 2, 4, 8 with zero fill, then lane
// 15 of each row to the next row's lanes -- rows 1 and 3 --, then lane 31 to rows 2 and 3): no LDS crossbar trip
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
// the same over each ROW of 16 lanes separately
__device__ __forceinline__ int row16_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    return v;
}

// Unit eigenvector of the smallest eigenvalue of a symmetric 3x3 matrix by cyclic
// Jacobi rotations.  c = {c00,c01,c02,c11,c12,c22}.
__device__ inline void smallest_eigvec_sym3(const double c[6], double out[3])
{
    double a00 = c[0], a01 = c[1], a02 = c[2], a11 = c[3], a12 = c[4], a22 = c[5];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;

// rotate in the (P,Q) plane; R is the remaining index.  APP/AQQ/APQ are the pivot
// entries, ARP/ARQ the two entries coupling R to P and Q, V?P/V?Q eigenvector columns.
#define ICPMI_JACOBI(APP, AQQ, APQ, ARP, ARQ, V0P, V0Q, V1P, V1Q, V2P, V2Q)                  \
    do {                                                                                      \
        const double apq = APQ;                                                               \
        const double g = 100.0 * fabs(apq);                                                   \
        if (sweep > 3 && fabs(APP) + g == fabs(APP) && fabs(AQQ) + g == fabs(AQQ)) {          \
            APQ = 0.0;                                                                        \
        } else if (apq != 0.0) {                                                              \
            const double h = AQQ - APP;                                                       \
            double t;                                                                         \
            if (fabs(h) + g == fabs(h)) {                                                     \
                t = apq / h;                                                                  \
            } else {                                                                          \
                const double theta = 0.5 * h / apq;                                           \
                t = 1.0 / (fabs(theta) + __dsqrt_rn(1.0 + theta * theta));                    \
                if (theta < 0.0) t = -t;                                                      \
            }                                                                                 \
            const double cs = 1.0 / __dsqrt_rn(1.0 + t * t);                                  \
            const double sn = t * cs;                                                         \
            const double arp = ARP, arq = ARQ;                                                \
            APP = APP - t * apq;                                                              \
            AQQ = AQQ + t * apq;                                                              \
            APQ = 0.0;                                                                        \
            ARP = cs * arp - sn * arq;                                                        \
            ARQ = sn * arp + cs * arq;                                                        \
            double vp, vq;                                                                    \
            vp = V0P; vq = V0Q; V0P = cs * vp - sn * vq; V0Q = sn * vp + cs * vq;             \
            vp = V1P; vq = V1Q; V1P = cs * vp - sn * vq; V1Q = sn * vp + cs * vq;             \
            vp = V2P; vq = V2Q; V2P = cs * vp - sn * vq; V2Q = sn * vp + cs * vq;             \
        }                                                                                     \
    } while (0)

    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (off == 0.0) break;
        ICPMI_JACOBI(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21); // (0,1), r=2
        ICPMI_JACOBI(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22); // (0,2), r=1
        ICPMI_JACOBI(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22); // (1,2), r=0
    }
#undef ICPMI_JACOBI
    // first minimum of the diagonal
    double e = a00;
    out[0] = v00; out[1] = v10; out[2] = v20;
    if (a11 < e) { e = a11; out[0] = v01; out[1] = v11; out[2] = v21; }
    if (a22 < e) { out[0] = v02; out[1] = v12; out[2] = v22; }
}

// 6x6 symmetric solve, Eigen-3.4 LDLT semantics: diagonal pivoting on the largest
// |diagonal| (first maximum), unit-lower L, pseudo-inverse of D with |D_i| <= DBL_MIN
// mapped to 0.  Everything stays in registers: the loops are fully unrolled and the run-time
// pivot row is matched against each candidate p, so every array index is a compile-time
// constant (a first version kept the matrix in LDS and indexed it at run time: ~300 dependent
// LDS round trips, most of k_finish_step's time).
__device__ inline void ldlt6_solve(const double *sums27 /* 21 upper-triangle terms + 6 rhs */, double *xout)
{
    double A[6][6];
    {
        int o = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) {
                A[r][c] = sums27[o];
                A[c][r] = sums27[o];
                ++o;
            }
    }
    int perm[6] = {0, 1, 2, 3, 4, 5};
    bool bail = false;
#define ICPMI_SWAP(a, b) do { const double t_ = (a); (a) = (b); (b) = t_; } while (0)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!bail) {
            int piv = k;
            double best = fabs(A[k][k]);
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                const double v = fabs(A[i][i]);
                if (v > best) { best = v; piv = i; }
            }
            perm[k] = piv;
#pragma unroll
            for (int p = k + 1; p < 6; ++p) {
                if (piv == p) {
#pragma unroll
                    for (int j = 0; j < k; ++j) ICPMI_SWAP(A[k][j], A[p][j]);
#pragma unroll
                    for (int i = p + 1; i < 6; ++i) ICPMI_SWAP(A[i][k], A[i][p]);
                    ICPMI_SWAP(A[k][k], A[p][p]);
#pragma unroll
                    for (int i = k + 1; i < p; ++i) ICPMI_SWAP(A[i][k], A[p][i]);
                }
            }
            if (k > 0) {
                double w[6];
                double dot = 0.0;
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    w[j] = A[j][j] * A[k][j];
                    dot += A[k][j] * w[j];
                }
                A[k][k] -= dot;
#pragma unroll
                for (int i = k + 1; i < 6; ++i) {
                    double sacc = 0.0;
#pragma unroll
                    for (int j = 0; j < k; ++j) sacc += A[i][j] * w[j];
                    A[i][k] -= sacc;
                }
            }
            const double d = A[k][k];
            const bool ok = fabs(d) > 0.0;
            if (k == 0 && !ok) {
#pragma unroll
                for (int j = 0; j < 6; ++j) perm[j] = j;
                bail = true;
            } else if (ok) {
#pragma unroll
                for (int i = k + 1; i < 6; ++i) A[i][k] /= d;
            }
        }
    }
    double x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = sums27[21 + i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int p = k + 1; p < 6; ++p)
            if (perm[k] == p) ICPMI_SWAP(x[k], x[p]);
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) sacc += A[i][j] * x[j];
        x[i] -= sacc;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (fabs(A[i][i]) > 2.2250738585072014e-308) x[i] /= A[i][i];
        else x[i] = 0.0;
    }
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        double sacc = 0.0;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) sacc += A[j][i] * x[j];
        x[i] -= sacc;
    }
#pragma unroll
    for (int k = 5; k >= 0; --k)
#pragma unroll
        for (int p = k + 1; p < 6; ++p)
            if (perm[k] == p) ICPMI_SWAP(x[k], x[p]);
#undef ICPMI_SWAP
#pragma unroll
    for (int i = 0; i < 6; ++i) xout[i] = x[i];
}

// The same solve by one whole WAVE (round 4).  ldlt6_solve is ~1,100 instructions of one lane in one dependent chain --
// pivot search, the swaps spelled out for every candidate row, 21 divisions -- 2.3 us on every iteration's critical
// path.  What makes a shorter chain possible: Eigen's LDLT is left-looking, so the diagonal entries it searches for a
// pivot are the INPUT's (column k is reduced only after pivot k has been chosen, the rest of the diagonal never).  The
// whole pivot order therefore follows from the six input diagonal values alone:
//   1. lane 8 i + j holds S[i][j] of the full symmetric matrix (columns 0-5) and the right-hand side (column 6).  One
//      ballot gives the 36 comparisons |S[a][a]| > |S[b][b]|; the selection with Eigen's rule (first maximum of the
//      current arrangement, strict >, so a NaN never wins) is replayed on it with scalar integer operations;
//   2. ONE lane permutation puts rows and columns (and the right-hand side, which rides on the rows) in pivot order:
//      what the k-th swap would have moved is the same values either way, because a row's entries in the finished
//      columns are computed from that row alone and travel with it;
//   3. the factorisation without pivoting, column k by its own lanes: lane (i, k) forms sum_j A[i][j] w[j] over the
//      finished columns in ldlt6_solve's order (A[i][j] from lane (i, j) = own lane - (k - j), a DPP row shift;
//      w[j] = D[j] L[k][j] wave-uniform), one subtraction, ONE division for the column.  No branches: the chain is
//      six times (last product, add, subtract, read the pivot, divide), everything else is scheduled beside it;
//   4. the substitutions on wave-uniform values (L, D as read in step 3), the solution scattered back by one
//      ds_permute.
// Every value goes through the same IEEE operations in the same order as in ldlt6_solve: the result is bit-identical
// (scripts/micro/step_clocks.hip compares the two on 200,000 random, rank-deficient, tie-heavy, zero and NaN systems).
// ALL 64 lanes of the wave must call it; every lane returns the same x.
__device__ __forceinline__ double row_shr_f64(double v, int n)
{
    switch (n) { // (n is a constant after unrolling)
    case 1: return row_shr_f64<1>(v);
    case 2: return row_shr_f64<2>(v);
    case 3: return row_shr_f64<3>(v);
    case 4: return row_shr_f64<4>(v);
    default: return row_shr_f64<5>(v);
    }
}

__device__ inline void ldlt6_solve_wave(const double *sums27, double *xout, int lane)
{
    const int i = lane >> 3, j = lane & 7;
    const bool in = i < 6 && j < 6;
    const int r = i < j ? i : j, c = i < j ? j : i;
    double v = in ? sums27[r * 6 - (r * (r - 1)) / 2 + (c - r)] : (i < 6 && j == 6 ? sums27[21 + i] : 0.0);
    // 1. the pivot order.  P: 4-bit fields, field t = the input row that ends up in place t (fields 6, 7: themselves)
    const int ic = i < 6 ? i : 0, jc = j < 6 ? j : 0;
    const double di = fabs(sums27[ic * 6 - (ic * (ic - 1)) / 2]), dj = fabs(sums27[jc * 6 - (jc * (jc - 1)) / 2]); // |S[i][i]|, |S[j][j]|
    const unsigned long long G = __ballot(in && di > dj);           // bit 8 a + b: |S[a][a]| > |S[b][b]|
    const unsigned long long Z = __ballot(in && j == 0 && di > 0.0); // bit 8 a: |S[a][a]| > 0 (false for a NaN)
    unsigned P = 0;
    bool sorted;
    {   // six different magnitudes (and no NaN): the selection is the descending order whatever it swapped on the way.
        // Lane t < 6: how many diagonal entries are greater than S[t][t] = its place; a place nobody takes means equal
        // values or a NaN somewhere, and the selection is replayed below
        const unsigned long long col = G & (0x0000010101010101ull << (lane & 7));
        const int place = lane < 6 ? __popcll(col) : 8;
        unsigned long long all = ~0ull;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const unsigned long long who = __ballot(place == t);
            all = who ? all : 0ull;
            P |= (unsigned)(__ffsll((long long)who) - 1) << (4 * t);
        }
        sorted = all != 0ull;
        P |= 0x76000000u;
    }
    if (!sorted) {
        P = 0x76543210u;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const unsigned at_k = (P >> (4 * k)) & 15u;
            unsigned best = at_k, piv = k;
#pragma unroll
            for (int ii = k + 1; ii < 6; ++ii) {
                const unsigned ci = (P >> (4 * ii)) & 15u;
                if ((G >> (8 * ci + best)) & 1ull) { best = ci; piv = ii; }
            }
            const unsigned xo = at_k ^ best;
            P ^= (xo << (4 * k)) | (xo << (4 * piv));
            if (k == 0 && !((Z >> (8 * best)) & 1ull)) break; // |pivot 0| is 0 or a NaN: Eigen stops here, the first swap stays
        }
    }
    const bool bail = !((Z >> (8 * (P & 15u))) & 1ull);
    // 2. rows and columns into pivot order
    {
        const unsigned ci = (P >> (4 * i)) & 15u, cj = (P >> (4 * j)) & 15u;
        v = __shfl(v, (int)(8 * ci + cj), 64);
    }
    unsigned Px = P;
    if (bail) { // (rare) ldlt6_solve resets its permutation: the right-hand side stays as it came
        if (i < 6 && j == 6) v = sums27[21 + i];
        Px = 0x76543210u;
    }
    // 3. L and D, no pivoting
    double d[6], L[6][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k > 0) {
            double acc = 0.0;
#pragma unroll
            for (int jj = 0; jj < k; ++jj) {
                L[k][jj] = readlane_f64(v, 8 * k + jj);
                const double w = d[jj] * L[k][jj];
                acc += row_shr_f64(v, k - jj) * w;
            }
            v = (!bail && in && j == k && i >= k) ? v - acc : v;
        }
        d[k] = readlane_f64(v, 9 * k);
        const bool ok = fabs(d[k]) > 0.0;
        v = (!bail && ok && in && j == k && i > k) ? v / d[k] : v;
    }
    // 4. substitutions (wave-uniform), ldlt6_solve's order
    double x[6];
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) x[ii] = readlane_f64(v, 8 * ii + 6);
#pragma unroll
    for (int ii = 1; ii < 6; ++ii) {
        double sacc = 0.0;
#pragma unroll
        for (int jj = 0; jj < ii; ++jj) sacc += L[ii][jj] * x[jj];
        x[ii] -= sacc;
    }
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) x[ii] = fabs(d[ii]) > 2.2250738585072014e-308 ? x[ii] / d[ii] : 0.0;
#pragma unroll
    for (int ii = 4; ii >= 0; --ii) {
        double sacc = 0.0;
#pragma unroll
        for (int jj = ii + 1; jj < 6; ++jj) sacc += L[jj][ii] * x[jj];
        x[ii] -= sacc;
    }
    // place t of the pivot order is input variable field t of Px: scatter
    double xl = x[0];
#pragma unroll
    for (int ii = 1; ii < 6; ++ii) xl = lane == ii ? x[ii] : xl;
    const int dest = lane < 8 ? (int)((Px >> (4 * lane)) & 15u) : lane;
    const int lo = __builtin_amdgcn_ds_permute(dest * 4, __double2loint(xl)), hi = __builtin_amdgcn_ds_permute(dest * 4, __double2hiint(xl));
    const double xs = __hiloint2double(hi, lo);
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) xout[ii] = readlane_f64(xs, ii);
}

// sin and cos of an ICP step's rotation angle (>= 0; icp.hpp:133-134 calls std::sin / std::cos).  Below pi/4 -- every
// step of a registration that is not lost -- no argument reduction is needed and the two kernel polynomials of fdlibm /
// msun (k_sin.c, k_cos.c: errors below 1 ulp on this interval) are evaluated directly: two short dependent chains
// instead of the device library's reduction + polynomials (~150 instructions in front of the 4x4 product on every
// iteration's critical path).  From pi/4 on, the library's sincos.  The reference's libm and the device library already
// differ in the last place here; the parity tolerance on poses (1e-9) is nine orders above that.
__device__ __forceinline__ void sincos_step(double a, double *s, double *c)
{
#ifdef ICPMI_SINCOS_LIBRARY /* A/B: the device library's sincos at every angle */
    if (false) {
#else
    if (a < 0.78539816339744828) {
#endif
        const double z = a * a;
        {
            const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                         S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
            const double w = z * z, r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6), v = z * a;
            *s = a + v * (S1 + z * r);
        }
        {
            const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                         C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
            const double w = z * z, r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
            const double hz = 0.5 * z, t = 1.0 - hz;
            *c = t + (((1.0 - t) - hz) + z * r);
        }
    } else {
        sincos(a, s, c);
    }
}

// x = [rx ry rz tx ty tz] -> row-major 4x4 (icp.hpp:123-143)
__device__ inline void twist_to_transform(const double *x, double *T)
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

// C = A * B, 4x4 row-major (types.hpp:118-120); C may alias B
__device__ inline void mul44(const double *A, const double *B, double *C)
{
    double tmp[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            tmp[4 * i + j] = s;
        }
    for (int e = 0; e < 16; ++e) C[e] = tmp[e];
}

// The same product by 16 lanes of one wave, lane 4 i + j forming C[i][j] with mul44's order of additions; A, B, C in
// LDS or global memory, C may alias B (every lane has read its operands before the first store is issued: one wave).
__device__ __forceinline__ void mul44_wave(const double *A, const double *B, double *C, int lane)
{
    const int i = (lane >> 2) & 3, j = lane & 3;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) C[lane] = s;
}

} // namespace icpmi

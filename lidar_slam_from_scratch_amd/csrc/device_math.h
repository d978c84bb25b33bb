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

// x = [rx ry rz tx ty tz] -> row-major 4x4 (icp.hpp:123-143)
__device__ inline void twist_to_transform(const double *x, double *T)
{
    const double rx = x[0], ry = x[1], rz = x[2];
    const double angle = __dsqrt_rn((rx * rx + ry * ry) + rz * rz);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(angle < 1e-10)) {
        const double ax = rx / angle, ay = ry / angle, az = rz / angle;
        const double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
        const double s = sin(angle), c1 = 1.0 - cos(angle);
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

} // namespace icpmi

// nn_mfma.h -- nearest-neighbour search, engine 2: fp32 MFMA coarse pass + certified fp64
// resolve.  Returns exactly what k_nn_f64 returns (the fp64 nearest neighbour in the
// reference's arithmetic, kdtree.hpp:112-142) at the FP32 matrix rate instead of the
// FP64 vector rate.
//
// Coarse pass (k_nn_coarse).  With both clouds centred on c and rounded to fp32
// (P = fl32(p - c), Q = fl32(q - c)),
//     |P - Q|^2 = |P|^2 + [Px Py Pz 1] . [-2Qx -2Qy -2Qz |Q|^2]
// is a K = 4 contraction: one v_mfma_f32_16x16x4_f32 evaluates 16 queries x 16 targets.
// A wave keeps 4 query tiles (64 queries) as A operands in registers and streams target
// tiles (B operands) from LDS; a 2048-target "split" (32 KiB, already in MFMA operand
// order in HBM) is staged once per workgroup and shared by its 8 waves.  The only VALU
// work per MFMA is two v_min3_f32: lane l, register r keeps the running minimum of query
// row (l>>4)*4+r against the targets of column l&15 -- a "slot" = 128 targets that are
// CONTIGUOUS in the caller's array (tile t, column c of split s is target s*2048+c*128+t).
// The epilogue adds |P|^2, tags each slot minimum with its column in the 4 low mantissa
// bits and reduces the 16 columns to (smallest, second smallest): 8 bytes per query per
// split.  No index is tracked in the loop.
//
// Resolve (k_nn_resolve), all fp64 in the reference's operation order.  Per query: take the
// split/column with the smallest coarse value, evaluate its 128 targets exactly -> D.  Every
// target with exact distance <= D has a coarse value <= tau = D + E, where E bounds the
// fp32 error (derivation below).  Every other slot whose recorded minimum is <= tau is
// evaluated exactly too (whole split when its second-smallest column is <= tau).  The
// result is the exact minimum, ties to the lowest index: bit-identical to k_nn_f64.
//
// Error bound.  u = 2^-24.  Let a >= |P| + |Q|.
//   coordinate rounding: |P-(p-c)| <= u|p-c|, same for Q, so
//        | |P-Q|^2 - |p-q|^2 | <= eps (2 d + eps),  eps = u a (1 + 1e-6),  d = |p-q|
//   arithmetic: fl32(|Q|^2) (1 rounding of an exact fp64 value), 4 chained FMAs in the MFMA
//        (guide: bitwise a k-ordered fmaf chain), fl32(|P|^2) formed with 3 roundings, 1
//        final add: every intermediate is bounded by a^2, so <= 9 u a^2; 16 u a^2 is used.
//   column tag: < 2^-20 relative on the stored minimum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace icpmi {

constexpr int kSplitTiles = 128;                  // target tiles (16 targets each) per split
constexpr int kSplitTargets = kSplitTiles * 16;   // 2048
constexpr int kSlotTargets = kSplitTiles;         // targets per (split, column) slot
constexpr int kCoarseQT = 4;                      // query tiles per wave
constexpr int kCoarseWaves = 8;                   // waves per workgroup
constexpr int kCoarseThreads = 64 * kCoarseWaves;
constexpr int kCoarseQueries = 16 * kCoarseQT * kCoarseWaves; // queries per workgroup
constexpr float kBig = 3.0e38f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NnFrame {
    double c[3];     // centre both clouds are expressed about
    double rq;       // upper bound of |Q| over all targets (inflated)
    double lo[3], hi[3];
};

// ---- bounding box of the target -> frame -------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox_partial(const double *__restrict__ pts, int m,
                                                      double *__restrict__ part /*[grid][6]*/)
{
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256)
        for (int a = 0; a < 3; ++a) {
            const double v = pts[3 * i + a];
            lo[a] = v < lo[a] ? v : lo[a];
            hi[a] = v > hi[a] ? v : hi[a];
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

__global__ void k_bbox_final(const double *__restrict__ part, int nblocks, NnFrame *frame)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308};
    for (int b = 0; b < nblocks; ++b)
        for (int a = 0; a < 3; ++a) {
            lo[a] = part[b * 6 + a] < lo[a] ? part[b * 6 + a] : lo[a];
            hi[a] = part[b * 6 + 3 + a] > hi[a] ? part[b * 6 + 3 + a] : hi[a];
        }
    double h2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        frame->lo[a] = lo[a];
        frame->hi[a] = hi[a];
        frame->c[a] = 0.5 * (lo[a] + hi[a]);
        const double h = 0.5 * (hi[a] - lo[a]);
        h2 += h * h;
    }
    frame->rq = sqrt(h2) * (1.0 + 1e-6) + 1e-300;
}

// ---- targets -> MFMA B operands, split-major, 4 tiles per float4 --------------------------
// Bpack[(s*32 + t4)*64 + lane].{x,y,z,w}: tile t = 4*t4 + e, k = lane>>4, column = lane&15,
// target j = s*2048 + column*128 + t.  Padding targets get (0,0,0,kBig): never the minimum.
__global__ __launch_bounds__(256) void k_pack_targets(const double *__restrict__ tgt, int m,
                                                      const NnFrame *__restrict__ frame,
                                                      float4 *__restrict__ Bpack, int splits)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= splits * 32 * 64) return;
    const int lane = g & 63, t4 = (g >> 6) & 31, s = g >> 11;
    const int col = lane & 15, k = lane >> 4;
    float v[4];
    for (int e = 0; e < 4; ++e) {
        const int t = 4 * t4 + e;
        const long j = (long)s * kSplitTargets + col * kSlotTargets + t;
        if (j >= m) {
            v[e] = k < 3 ? 0.0f : kBig;
        } else if (k < 3) {
            v[e] = -2.0f * (float)(tgt[3 * j + k] - frame->c[k]);
        } else {
            const double qx = (double)(float)(tgt[3 * j] - frame->c[0]);
            const double qy = (double)(float)(tgt[3 * j + 1] - frame->c[1]);
            const double qz = (double)(float)(tgt[3 * j + 2] - frame->c[2]);
            v[e] = (float)((qx * qx + qy * qy) + qz * qz);
        }
    }
    Bpack[g] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- coarse pass ---------------------------------------------------------------------------
__device__ __forceinline__ float min3f(float a, float b, float c)
{
    return __builtin_fminf(__builtin_fminf(a, b), c); // -> v_min3_f32
}

// MODE 0: 1-NN epilogue -> coarse[split][n] = (tagged min, second min over columns)
// MODE 1: k-NN epilogue  -> slotmin[query][split*16 + column], every column minimum kept
// QT = query tiles (16 queries) per wave, a multiple of 4; WAVES = waves per workgroup.
template <int MODE, int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_coarse(
    const double *__restrict__ qry, int n, const float4 *__restrict__ Bpack,
    const NnFrame *__restrict__ frame, float2 *__restrict__ coarse /*[split][n]*/,
    float *__restrict__ slotmin /*[n][splits*16]*/, const IcpState *__restrict__ st)
{
    static_assert(QT % 4 == 0, "epilogue works on groups of 64 queries");
    constexpr int THREADS = 64 * WAVES;
    constexpr int SCRATCH4 = WAVES * 64 * 20 / 4 > 32 * 64 ? WAVES * 64 * 20 / 4 : 32 * 64;
    if (st && st->done) return;
    // 32 KiB of B operands; reused (plus a tail) for the epilogue's transpose
    __shared__ float4 ldsB[SCRATCH4];
    const int s = blockIdx.y;
    {
        const float4 *src = Bpack + (size_t)s * (32 * 64);
#pragma unroll
        for (int e = 0; e < (32 * 64) / THREADS; ++e) ldsB[threadIdx.x + e * THREADS] = src[threadIdx.x + e * THREADS];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * WAVES + wave) * (16 * QT);
    const double c0 = frame->c[0], c1 = frame->c[1], c2 = frame->c[2];

    // A operands: lane l holds component k = l>>4 of query row l&15; k == 3 is the constant 1
    float a[QT];
    {
        const int row = lane & 15, k = lane >> 4;
        const double ck = k == 0 ? c0 : (k == 1 ? c1 : c2);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            int i = q0 + t * 16 + row;
            i = i < n ? i : n - 1;
            a[t] = k < 3 ? (float)(qry[3 * i + k] - ck) : 1.0f;
        }
    }
    // |P|^2 of the queries this lane owns in the epilogue (lane-per-query there)
    float pn[QT / 4];
#pragma unroll
    for (int gq = 0; gq < QT / 4; ++gq) {
        const int iq = q0 + gq * 64 + lane < n ? q0 + gq * 64 + lane : n - 1;
        const float px = (float)(qry[3 * iq] - c0), py = (float)(qry[3 * iq + 1] - c1),
                    pz = (float)(qry[3 * iq + 2] - c2);
        pn[gq] = (px * px + py * py) + pz * pz;
    }
    f32x4 m[QT];
#pragma unroll
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

    // epilogue.  Transpose through LDS, 64 queries at a time, so that each lane owns ONE
    // query and its 16 column minima (row stride 20 floats: conflict-free ds_read_b128),
    // then + |P|^2 and either the tagged (min, second min) pair or the raw 16 values go
    // out, coalesced.
    __syncthreads(); // every wave is done with the B operands
    float *sc = reinterpret_cast<float *>(ldsB) + wave * (64 * 20);
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int gq = 0; gq < QT / 4; ++gq) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[(t * 16 + g * 4 + r) * 20 + col] = m[gq * 4 + t][r];
        __builtin_amdgcn_wave_barrier();
        float v[16];
        {
            const float4 *rowp = reinterpret_cast<const float4 *>(sc + lane * 20);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 x = rowp[e];
                v[4 * e] = x.x + pn[gq];
                v[4 * e + 1] = x.y + pn[gq];
                v[4 * e + 2] = x.z + pn[gq];
                v[4 * e + 3] = x.w + pn[gq];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int iq = q0 + gq * 64 + lane;
        if (MODE == 1) {
            if (iq < n) {
                float4 *dst = reinterpret_cast<float4 *>(slotmin + (size_t)iq * (gridDim.y * 16) + s * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
            }
        } else {
            float v1 = kBig, v2 = kBig;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float x = __uint_as_float((__float_as_uint(v[c]) & 0xFFFFFFF0u) | (unsigned)c);
                const float hi = __builtin_fmaxf(v1, x);
                v1 = __builtin_fminf(v1, x);
                v2 = __builtin_fminf(v2, hi);
            }
            if (iq < n) coarse[(size_t)s * n + iq] = make_float2(v1, v2);
        }
    }
}

// ---- resolve ---------------------------------------------------------------------------------
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

// exact scan of targets [j0, j0+len) for the query (px,py,pz), all lanes cooperate
__device__ __forceinline__ void scan_range(const double *__restrict__ tgt, int m, int j0, int len,
                                           double px, double py, double pz, int lane, double &bd,
                                           int &bj)
{
    double d = 1.7976931348623157e308;
    int j = 0x7fffffff;
    for (int o = lane; o < len; o += 64) {
        const int jj = j0 + o;
        if (jj < m) {
            const double dd = sqdist(tgt[3 * jj], tgt[3 * jj + 1], tgt[3 * jj + 2], px, py, pz);
            if (dd < d) { // ascending jj: strict keeps the lowest index
                d = dd;
                j = jj;
            }
        }
    }
    wave_argmin(d, j);
    if (d < bd || (d == bd && j < bj)) {
        bd = d;
        bj = j;
    }
}

// One wave resolves 16 queries.  Lane = (query ql = lane&15, quarter = lane>>4): the four
// quarters share the bookkeeping of a query (each scans a quarter of the splits) and each
// quarter-wave scans one winning slot at a time, 8 consecutive targets (192 contiguous
// bytes) per lane, so 4 slots are in flight per wave and ~6 waves per SIMD hide the latency.
constexpr int kResolveQ = 16;

__global__ __launch_bounds__(256) void k_nn_resolve(const double *__restrict__ qry, int n,
                                                    const double *__restrict__ tgt, int m,
                                                    const float2 *__restrict__ coarse, int splits,
                                                    const NnFrame *__restrict__ frame,
                                                    int *__restrict__ idx, double *__restrict__ d2out,
                                                    unsigned long long *__restrict__ counters,
                                                    const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    const int lane = threadIdx.x & 63;
    const int ql = lane & 15, quarter = lane >> 4;
    const int qbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kResolveQ;
    if (qbase >= n) return; // wave-uniform
    const int i = qbase + ql;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = qry[3 * ic], py = qry[3 * ic + 1], pz = qry[3 * ic + 2];

    // phase 1: smallest coarse value over the splits (each quarter takes every 4th split)
    float best = kBig;
    int bs = 0;
    for (int s = quarter; s < splits; s += 4) {
        const float v = coarse[(size_t)s * n + ic].x;
        if (v < best) {
            best = v;
            bs = s;
        }
    }
#pragma unroll
    for (int x = 16; x < 64; x <<= 1) {
        const float ov = __shfl_xor(best, x, 64);
        const int os = __shfl_xor(bs, x, 64);
        if (ov < best || (ov == best && os < bs)) {
            best = ov;
            bs = os;
        }
    }
    const int bcol = (int)(__float_as_uint(best) & 15u);

    // phase 2: exact evaluation of the winning slots, one query per quarter-wave and round
    double bd = 1.7976931348623157e308;
    int bj = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int src = quarter * 4 + r; // the query this quarter scans now (a lane of quarter 0)
        const double qx = __shfl(px, src, 64), qy = __shfl(py, src, 64), qz = __shfl(pz, src, 64);
        const int s = __shfl(bs, src, 64), c = __shfl(bcol, src, 64);
        const int j0 = s * kSplitTargets + c * kSlotTargets + ql * 8;
        double d = 1.7976931348623157e308;
        int j = 0x7fffffff;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int jj = j0 + o;
            const int jc = jj < m ? jj : m - 1;
            const double dd = sqdist(tgt[3 * jc], tgt[3 * jc + 1], tgt[3 * jc + 2], qx, qy, qz);
            if (jj < m && dd < d) {
                d = dd;
                j = jj;
            }
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
            const double od = __shfl_xor(d, x, 64);
            const int oj = __shfl_xor(j, x, 64);
            if (od < d || (od == d && oj < j)) {
                d = od;
                j = oj;
            }
        }
        // query ql was scanned by quarter ql>>2 in round ql&3
        const double rd = __shfl(d, (ql >> 2) * 16, 64);
        const int rj = __shfl(j, (ql >> 2) * 16, 64);
        if ((ql & 3) == r) {
            bd = rd;
            bj = rj;
        }
    }

    // phase 3: certificate.  tau bounds the coarse value of any target at distance <= bd
    const double dx = px - frame->c[0], dy = py - frame->c[1], dz = pz - frame->c[2];
    const double a = sqrt((dx * dx + dy * dy) + dz * dz) * (1.0 + 1e-6) + frame->rq;
    const double u = 5.9604644775390625e-08; // 2^-24
    const double eps = u * a * (1.0 + 1e-6);
    double tau = bd + eps * (2.0 * sqrt(bd) + eps) + 16.0 * u * a * a;
    tau = tau * (1.0 + 4e-6) + 1e-300; // column tag (2^-20) + slack for this fp64 evaluation
    // round up: next float above the nearest-rounded value (tau > 0)
    const float tauf = __uint_as_float(__float_as_uint((float)tau) + 1u);

    unsigned extra_slots = 0, extra_splits = 0;
    for (int s0 = 0; s0 < splits; s0 += 4) {
        const int s = s0 + quarter;
        float2 v = make_float2(kBig, kBig);
        if (s < splits) v = coarse[(size_t)s * n + ic];
        const bool whole = valid && v.y <= tauf;            // a second column is inside the bound
        const bool slot = valid && !whole && s != bs && v.x <= tauf;
        unsigned long long pend = __ballot(whole || slot);
        while (pend) {                                      // rare; wave-uniform loop
            const int L = __ffsll((long long)pend) - 1;
            pend &= pend - 1;
            const double qx = __shfl(px, L, 64), qy = __shfl(py, L, 64), qz = __shfl(pz, L, 64);
            const int w = __shfl((int)whole, L, 64);
            const int c = __shfl((int)(__float_as_uint(v.x) & 15u), L, 64);
            const int sL = s0 + (L >> 4);
            double d = 1.7976931348623157e308;
            int j = 0x7fffffff;
            if (w) scan_range(tgt, m, sL * kSplitTargets, kSplitTargets, qx, qy, qz, lane, d, j);
            else scan_range(tgt, m, sL * kSplitTargets + c * kSlotTargets, kSlotTargets, qx, qy, qz, lane, d, j);
            if (ql == (L & 15)) { // every replica of that query takes the result
                if (d < bd || (d == bd && j < bj)) {
                    bd = d;
                    bj = j;
                }
            }
            if (lane == L) {
                if (w) ++extra_splits;
                else ++extra_slots;
            }
        }
    }
    if (valid && quarter == 0) {
        idx[i] = bj;
        if (d2out) d2out[i] = bd;
    }
    if (counters) {
        unsigned es = extra_slots, ef = extra_splits;
        for (int off = 32; off > 0; off >>= 1) {
            es += __shfl_down(es, off, 64);
            ef += __shfl_down(ef, off, 64);
        }
        if (lane == 0 && (es | ef)) {
            atomicAdd(&counters[0], (unsigned long long)es);
            atomicAdd(&counters[1], (unsigned long long)ef);
        }
    }
}

// ---- k-NN on the same coarse pass ---------------------------------------------------------------
// Resolve for the k nearest neighbours of target row i among all targets (icp.hpp:32,
// kdtree.hpp:65-78), one wave per row.  Each slot minimum belongs to a distinct target, so
// the k-th smallest slot minimum tS bounds the k-th neighbour: at least k targets have coarse
// value <= tS, hence true squared distance <= dmax (solved from d <= tS + E(d)).  Every true
// k-neighbour then has coarse value <= dmax + E(dmax): the slots under that bound are scanned
// exactly (fp64, reference operation order) and every target with exact distance <= dmax is
// collected; the k smallest by (distance, index) are written closest first -- the order
// kdtree.hpp:72-76 returns and icp.hpp:41-51 sums in.  A looser tS (the k-th smallest of the
// 64 per-lane minima instead of all slot minima) is still a valid bound and is what is used.
constexpr int kKnnCap = 256; // candidates per row held in LDS; overflow -> exact fallback list

__global__ __launch_bounds__(256) void k_knn_resolve(const double *__restrict__ pts, int m, int k,
                                                     int row0, int nrows,
                                                     const float *__restrict__ slotmin, int nslots,
                                                     const NnFrame *__restrict__ frame,
                                                     int *__restrict__ knn_idx /*[m][k]*/,
                                                     int *__restrict__ fb_list, int *__restrict__ fb_count)
{
    __shared__ double cand_d[4][kKnnCap];
    __shared__ int cand_j[4][kKnnCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = blockIdx.x * 4 + wave;
    if (local >= nrows) return; // wave-uniform
    const int i = row0 + local;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const float *mine = slotmin + (size_t)local * nslots;

    // per-lane minimum of the slot minima
    float lmin = kBig;
    for (int e = lane; e < nslots; e += 64) lmin = __builtin_fminf(lmin, mine[e]);
    // k-th smallest of the 64 lane minima (ties ordered by lane)
    int rank = 0;
#pragma unroll 8
    for (int L = 0; L < 64; ++L) {
        const float v = __shfl(lmin, L, 64);
        rank += (v < lmin || (v == lmin && L < lane)) ? 1 : 0;
    }
    const int kk = k < 64 ? k : 64;
    const unsigned long long who = __ballot(rank == kk - 1);
    const float tS = __shfl(lmin, __ffsll((long long)who) - 1, 64);

    // bounds (see the header comment of this file for E)
    const double dx = px - frame->c[0], dy = py - frame->c[1], dz = pz - frame->c[2];
    const double a = sqrt((dx * dx + dy * dy) + dz * dz) * (1.0 + 1e-6) + frame->rq;
    const double u = 5.9604644775390625e-08;
    const double eps = u * a * (1.0 + 1e-6);
    const double A = 16.0 * u * a * a;
    const double ts = tS > 0.f ? (double)tS : 0.0;
    const double xr = eps + sqrt(eps * eps + (ts + eps * eps + A)); // sqrt(dmax)
    const double dmax = xr * xr * (1.0 + 1e-9);
    double tau = dmax + eps * (2.0 * xr + eps) + A;
    tau = tau * (1.0 + 4e-6) + 1e-300;
    const float tauf = tS >= kBig ? kBig : __uint_as_float(__float_as_uint((float)tau) + 1u);

    // exact scan of every slot under the bound; keep targets with exact distance <= dmax
    int total = 0;
    for (int e0 = 0; e0 < nslots; e0 += 64) {
        const int e = e0 + lane;
        const bool flag = e < nslots && mine[e] <= tauf;
        unsigned long long pend = __ballot(flag);
        while (pend) {
            const int L = __ffsll((long long)pend) - 1;
            pend &= pend - 1;
            const int se = e0 + L;
            const int j0 = (se >> 4) * kSplitTargets + (se & 15) * kSlotTargets;
#pragma unroll
            for (int o = 0; o < kSlotTargets; o += 64) {
                const int jj = j0 + o + lane;
                double d = 1.7976931348623157e308;
                if (jj < m) d = sqdist(pts[3 * jj], pts[3 * jj + 1], pts[3 * jj + 2], px, py, pz);
                const bool keep = d <= dmax;
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const int pos = total + __popcll(km & ((1ull << lane) - 1ull));
                    if (pos < kKnnCap) {
                        cand_d[wave][pos] = d;
                        cand_j[wave][pos] = jj;
                    }
                }
                total += __popcll(km);
            }
        }
    }
    if (total > kKnnCap) { // too many near-equidistant targets: hand the row to the exact kernel
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // rank by (distance, index); the k smallest go out closest first
    for (int e = lane; e < total; e += 64) {
        const double d = cand_d[wave][e];
        const int j = cand_j[wave][e];
        int r = 0;
        for (int f = 0; f < total; ++f) {
            const double df = cand_d[wave][f];
            const int jf = cand_j[wave][f];
            r += (df < d || (df == d && jf < j)) ? 1 : 0;
        }
        if (r < k) knn_idx[(size_t)i * k + r] = j;
    }
}

// exact fp64 k-NN lists, one row per thread (small clouds, and rows the MFMA resolve hands
// back).  rows: explicit list (list != nullptr, count read from *list_count) or [row0,row1).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_knn_exact_list(const double *__restrict__ pts, int m, int k,
                                                          int row0, int row1,
                                                          const int *__restrict__ list,
                                                          const int *__restrict__ list_count,
                                                          int *__restrict__ knn_idx)
{
    extern __shared__ double knn_smem[];
    double *ld = knn_smem;
    int *li = reinterpret_cast<int *>(knn_smem + (size_t)k * BLOCK);
    const int tid = threadIdx.x;
    const int nrows = list ? *list_count : row1 - row0;
    for (int base = blockIdx.x * BLOCK; base < nrows; base += gridDim.x * BLOCK) { // block-uniform
        const int lr = base + tid;
        const bool active = lr < nrows;
        const int i = active ? (list ? list[lr] : row0 + lr) : (list ? list[0] : row0);
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
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

// PCA normal from a closest-first neighbour list (icp.hpp:34-63), one row per thread
__global__ __launch_bounds__(256) void k_normals_from_knn(const double *__restrict__ pts, int m, int k,
                                                          int row0, int row1,
                                                          const int *__restrict__ knn_idx,
                                                          double *__restrict__ normals)
{
    const int i = row0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= row1) return;
    const int cnt = k < m ? k : m;
    const int *nb = knn_idx + (size_t)i * k;
    double nx = 0.0, ny = 0.0, nz = 1.0; // icp.hpp:34-37
    if (cnt >= 3) {
        double cx = 0.0, cy = 0.0, cz = 0.0; // icp.hpp:40-44
        for (int a = 0; a < cnt; ++a) {
            const int j = nb[a];
            cx += pts[3 * j];
            cy += pts[3 * j + 1];
            cz += pts[3 * j + 2];
        }
        const double kd = (double)cnt;
        cx /= kd;
        cy /= kd;
        cz /= kd;
        double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0; // icp.hpp:47-52
        for (int a = 0; a < cnt; ++a) {
            const int j = nb[a];
            const double dx = pts[3 * j] - cx, dy = pts[3 * j + 1] - cy, dz = pts[3 * j + 2] - cz;
            c00 += dx * dx;
            c01 += dx * dy;
            c02 += dx * dz;
            c11 += dy * dy;
            c12 += dy * dz;
            c22 += dz * dz;
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
    normals[3 * i] = nx;
    normals[3 * i + 1] = ny;
    normals[3 * i + 2] = nz;
}

} // namespace icpmi

// icp_small.h -- one ICP iteration's search + residual work for SMALL clouds in ONE kernel.
//
// The reference's real caller registers voxel-filtered scans of 5-20k points against each other
// (slam_node.cpp:134-138, loop_closure.hpp:105-109): the target is a handful of 2048-target splits.
// There the general path -- k_nn_coarse (minima of every (query, split) to memory), k_nn_resolve4
// (reads them back, exact scans, normal-equation terms) and the pose update of the points -- is
// three dependent launches of a few microseconds each for ~2 us of matrix work, and the
// intermediate is written and read again for nothing.  k_icp_small does an iteration's row work
// in one launch and keeps the intermediate in LDS.  (Measured and dropped, scripts/small_clock.py:
// four waves per workgroup with whole splits per wave -- fewer instructions in all, but one wave per
// SIMD leaves every MFMA -> minimum dependency and every load exposed: 30k cycles against 21k; slot
// scans by the whole wave with DPP reductions instead of quarter-waves: 6.4k cycles against 4.4k; the step of
// the PREVIOUS pass -- k_finish_step's work -- repeated by every workgroup at the head of this kernel, so that
// an iteration is one launch: 21.0 us per iteration against 20.3, the serial step does not hide under the
// operand stream but in front of an issue-bound remainder, and 250 workgroups sitting through it keep a
// second stream's kernels off the chip: 0.37 ms per frame of the file stream against 0.32.)
//
//   workgroup = 32 rows (one MFMA tile of queries) x ALL splits, 8 waves = two per SIMD.
//   1. every wave loads the 32 rows and moves them by the pending pose update (icp.hpp:174-176
//      for the first pass, :225-226 afterwards; same operation order as k_transform), wave 0
//      stores the moved rows;
//   2. the target's half-splits (32 tiles = 1024 targets) are dealt to the waves; a wave builds its
//      A operands about the split's centre (coarse_build_a's arithmetic), streams its 32 B tiles
//      from memory straight into registers (no other wave wants them: no LDS staging, no
//      workgroup barrier) and runs the coarse loop and the 1-NN epilogue of nn_mfma.h: the
//      (column-tagged minimum, second minimum) of every row against its half-split -> LDS;
//   3. the resolve of nn_mfma.h on those records, one row per quarter-wave as k_nn_resolve4:
//      per-split records merged from the two halves, smallest split, exact fp64 scan of the
//      winning slot, certificate (split_tau; slots / whole splits under the bound rescanned).  The scan also fetches each
//      candidate's NORMAL from a Morton-ordered copy, so the winner's matched point and normal
//      are at hand when it is known: no dependent gather;
//   4. J row and b of every row (icp.hpp:99-117), summed per workgroup in k_nn_resolve4<8>'s
//      order: the partial rows -- hence error history and pose -- are bit-identical to the general
//      path's (tests/test_gpu_parity.py::test_small_cloud_kernel_gives_the_general_path_bits).
//
// The certificate argument is nn_mfma.h's: the records are exactly the values k_nn_coarse would have
// written (same operands, same MFMA, same epilogue; the second minimum kept in fp32 instead of
// bf16 rounded down, which only tightens it), so the result is the exact fp64 nearest neighbour,
// ties to the lowest original index.
#pragma once
#include "nn_mfma.h"

namespace icpmi {

constexpr int kSmallWaves = 8;
constexpr int kSmallThreads = 64 * kSmallWaves;
constexpr int kSmallQ = 32;                                  // rows (queries) per workgroup
constexpr int kSmallUnitTiles = 32;                          // target tiles per unit of a wave's work
constexpr int kSmallUnitsPerSplit = kSplitTiles / kSmallUnitTiles;
constexpr int kSmallMaxSplits = 16;                          // one lane of a quarter-wave per split
static_assert(kSmallQ == 4 * kSmallWaves, "one row per quarter-wave in the resolve part");
static_assert(kSplitTiles % kSmallUnitTiles == 0 && kSmallUnitTiles % 2 == 0, "whole units, tiles in pairs");
static_assert(kSlotTargets == 64, "the slot scan takes four targets per lane of a quarter-wave");

// (v1, v2) <- the two smallest column minima of the union of two records over the SAME columns
// (two halves of a split).  If both minima sit in the same column the other half's minimum is not
// a second column.
__device__ __forceinline__ void small_merge(float &v1, float &v2, const float w1, const float w2)
{
    const bool same = ((__float_as_uint(v1) ^ __float_as_uint(w1)) & 31u) == 0u;
    const float lo = w1 < v1 ? w1 : v1, hi = w1 < v1 ? v1 : w1;
    float n2 = w2 < v2 ? w2 : v2;
    if (!same) n2 = hi < n2 ? hi : n2;
    v1 = lo;
    v2 = n2;
}

// Diagnostic build only (-DICPMI_SMALL_CLOCKS, scripts/small_clock.py): s_memtime stamps of the first and the last
// wave of every workgroup at the kernel's phases, everything outstanding drained first (which perturbs what it measures).
#ifdef ICPMI_SMALL_CLOCKS
#define ICPMI_SMALL_STAMP(k)                                                   \
    do {                                                                       \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            \
        stamp[k] = __builtin_amdgcn_s_memtime();                               \
    } while (0)
#else
#define ICPMI_SMALL_STAMP(k) do { } while (0)
#endif
constexpr int kSmallStamps = 12;

__global__ __launch_bounds__(kSmallThreads) void k_icp_small(
    const double *in, double *cur, int n, const IcpState *__restrict__ st, int which,
    const uint4 *__restrict__ Bpack, const SplitFrame *__restrict__ frames, int splits,
    const double *__restrict__ sorted, const double *__restrict__ nrm_sorted, const unsigned *__restrict__ perm,
    int m, int ms, const double *__restrict__ tgt_orig, const double *__restrict__ nrm,
    double *__restrict__ partials, unsigned long long *__restrict__ counters, unsigned long long *__restrict__ clocks)
{
#ifdef ICPMI_SMALL_CLOCKS
    unsigned long long stamp[kSmallStamps];
    for (int k = 0; k < kSmallStamps; ++k) stamp[k] = 0;
    stamp[0] = __builtin_amdgcn_s_memtime();
#endif
    __shared__ uint4 scratch[kSmallWaves][32 * 36 / 4]; // per wave: A rows, then the epilogue's transpose
    __shared__ float2 rec[kSmallMaxSplits * kSmallUnitsPerSplit][kSmallQ];
    __shared__ double jrow[kSmallQ][29];
    __shared__ double red[kSmallWaves][28];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * kSmallQ;
    const int nunits = splits * kSmallUnitsPerSplit;

    // Everything the kernel needs first is requested at once, oldest first what is needed first (vector loads
    // return in order): the state (is the loop over? the pending pose update), the rows, then the B operands
    // of the wave's first unit -- ONE trip to memory, the loop-ended test behind the requests.
    const int done = st->done;
    const double *T = which ? st->total : st->delta;
    const double r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
    const double r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
    const double r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
    const int iq = q0 + (lane & 31) < n ? q0 + (lane & 31) : n - 1;
    double x = in[3 * iq], y = in[3 * iq + 1], z = in[3 * iq + 2];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z)); // (the row loads stay in front of the operand loads and of the test below)
    int u = wave;
    uint4 b[kSmallUnitTiles];
    if (u < nunits) {
        const uint4 *tiles = Bpack + (size_t)u * (kSmallUnitTiles * 64) + lane; // unit u = tiles [32 u, 32 u + 32) of the packed array
#pragma unroll
        for (int tt = 0; tt < kSmallUnitTiles; ++tt) b[tt] = tiles[tt * 64];
    }
    if (done) return; // the loop has ended: the source stays where it is (icp.hpp:210-217)
    // 1. the workgroup's 32 rows, lanes l and l + 32 of every wave holding row l & 31, moved by the pending
    // update (icp.hpp:174-176 / :225-226, k_transform's operation order)
    const double px = ((x * r00 + y * r01) + z * r02) + t0;
    const double py = ((x * r10 + y * r11) + z * r12) + t1;
    const double pz = ((x * r20 + y * r21) + z * r22) + t2;

    // 2. coarse records of this wave's units
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll 1
    while (u < nunits) {
        const int s = u / kSmallUnitsPerSplit;
        bf16x8 afrag;
        float pn;
        {
            const float fx = (float)(px - frames[s].c[0]), fy = (float)(py - frames[s].c[1]), fz = (float)(pz - frames[s].c[2]);
            unsigned xh, xm, yh, ym, zh, zm;
            split2(fx, xh, xm);
            split2(fy, yh, ym);
            split2(fz, zh, zm);
            const float tx = __uint_as_float(xh << 16) + __uint_as_float(xm << 16);
            const float ty = __uint_as_float(yh << 16) + __uint_as_float(ym << 16);
            const float tz = __uint_as_float(zh << 16) + __uint_as_float(zm << 16);
            pn = (tx * tx + ty * ty) + tz * tz; // |P|^2 of the represented point (coarse_build_a)
            const unsigned pnh = __float_as_uint(pn) >> 16;
            pn -= __uint_as_float(pnh << 16);
            const unsigned one = 0x3f80u;
            // (lanes l and l + 32 formed the same row: the lower keeps its first half, the upper its second -- the MFMA's A
            // fragment, without the trip through LDS it took until round 4)
            const uint4 u0 = make_uint4(xh | (xh << 16), xm | (xm << 16), yh | (yh << 16), ym | (ym << 16));
            const uint4 u1 = make_uint4(zh | (zh << 16), zm | (zm << 16), one | (one << 16), one | (pnh << 16));
            afrag = __builtin_bit_cast(bf16x8, lane < 32 ? u0 : u1);
        }
        ICPMI_SMALL_STAMP(1); // state, rows and the unit's operands here, A built
        f32x16 mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) mn[r] = kBig;
#pragma unroll
        for (int tt = 0; tt < kSmallUnitTiles; tt += 2) {
            const f32x16 da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, b[tt]), zero, 0, 0, 0);
            const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, b[tt + 1]), zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) mn[r] = min3f(mn[r], da[r], db[r]);
        }
        // epilogue (coarse_epilogue, MODE 0): transpose, tag, top two of the 32 columns, + the rest of |P|^2
        {
            float *sc = reinterpret_cast<float *>(scratch[wave]);
            const int ql = lane & 31, half = lane >> 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + ql] = mn[r];
            __builtin_amdgcn_wave_barrier();
            float v[16];
            const float4 *rowp = reinterpret_cast<const float4 *>(sc + ql * 36 + half * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 xx = rowp[e];
                v[4 * e] = xx.x, v[4 * e + 1] = xx.y, v[4 * e + 2] = xx.z, v[4 * e + 3] = xx.w;
            }
            __builtin_amdgcn_wave_barrier();
            float v1 = kBig, v2 = kBig;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float xx = tag_low5(v[c], (unsigned)c);
                v2 = med3_raw(v1, v2, xx);
                v1 = min_raw(v1, xx);
            }
            const unsigned col = (__float_as_uint(v1) & 15u) | ((unsigned)half << 4);
            v1 = tag_low5(min_raw(__uint_as_float(__float_as_uint(v1) & 0xFFFFFFE0u) + pn, kBig), col);
            v2 = min_raw(__uint_as_float(__float_as_uint(v2) & 0xFFFFFFE0u) + pn, kBig);
            const float o1 = lane_xor<32>(v1), o2 = lane_xor<32>(v2);
            const float hi = __builtin_fmaxf(v1, o1);
            v1 = __builtin_fminf(v1, o1);
            v2 = min3f(hi, v2, o2);
            if (half == 0) rec[u][ql] = make_float2(v1, v2);
        }
        u += kSmallWaves;
        if (u < nunits) { // (targets of more than four splits: the next unit's operands)
            const uint4 *tiles = Bpack + (size_t)u * (kSmallUnitTiles * 64) + lane;
#pragma unroll
            for (int tt = 0; tt < kSmallUnitTiles; ++tt) b[tt] = tiles[tt * 64];
        }
    }
    ICPMI_SMALL_STAMP(2); // this wave's records written
    __syncthreads(); // records complete; every wave has read the old rows of `in`
    ICPMI_SMALL_STAMP(3);
    if (wave == 0 && lane < 32 && q0 + lane < n) {
        cur[3 * (q0 + lane)] = px;
        cur[3 * (q0 + lane) + 1] = py;
        cur[3 * (q0 + lane) + 2] = pz;
    }

    // 3. resolve, one row per quarter-wave (k_nn_resolve4): lane ql of the quarter holds split ql's record.
    // (Measured and dropped: the steps in leaner layouts -- selection and terms one lane per row in wave 0, the
    // certificate one lane per (row, split) -- with the rows' state handed on through LDS: fewer instructions, but
    // three more workgroup barriers, each waiting for the wave whose operands arrived last: 23.3 us per
    // iteration against 20.3.)
    const int quarter = lane >> 4, ql = lane & 15;
    const int qi = wave * 4 + quarter;
    const bool valid = q0 + qi < n;
    const double qx = __shfl(px, qi, 64), qy = __shfl(py, qi, 64), qz = __shfl(pz, qi, 64);
    float v1 = kBig, v2 = kBig;
    if (ql < splits) {
        const float2 a = rec[ql * kSmallUnitsPerSplit][qi];
        v1 = a.x, v2 = a.y;
#pragma unroll
        for (int h = 1; h < kSmallUnitsPerSplit; ++h) {
            const float2 w = rec[ql * kSmallUnitsPerSplit + h][qi];
            small_merge(v1, v2, w.x, w.y);
        }
    }
    int bs = ql < splits ? ql : 0x7fffffff;
    int bcol;
    {
        float best = v1;
#define ICPMI_STEP(S)                                                                     \
        {                                                                                 \
            const float ov = row16_partner<S>(best);                                      \
            const int os = row16_partner<S>(bs);                                          \
            const bool take = (ov < best) | ((ov == best) & (os < bs));                   \
            best = take ? ov : best;                                                      \
            bs = take ? os : bs;                                                          \
        }
        ICPMI_STEP(0) ICPMI_STEP(1) ICPMI_STEP(2) ICPMI_STEP(3) // (device_math.h: DPP partners, no LDS trip)
#undef ICPMI_STEP
        bcol = (int)(__float_as_uint(best) & 31u);
    }

    // exact scan of the winning slot: lane ql takes sorted positions ql, ql + 16, ... of it, each candidate's
    // NORMAL fetched next to its coordinates; the lane that ends up holding the winner has the matched point
    // and its normal in registers
    double ld = 1.7976931348623157e308, lq0 = 0.0, lq1 = 0.0, lq2 = 0.0, ln0 = 0.0, ln1 = 0.0, ln2 = 0.0;
    int lj = 0x7fffffff;
    {
        const int j0 = bs * kSplitTargets + bcol * kSlotTargets + ql;
        double cx[4], cy[4], cz[4], a0[4], a1[4], a2[4];
        int oj[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int jj = j0 + 16 * o;
            const int jc = jj < m ? jj : m - 1;
            cx[o] = ICPMI_SX(sorted, ms, jc), cy[o] = ICPMI_SY(sorted, ms, jc), cz[o] = ICPMI_SZ(sorted, ms, jc);
            a0[o] = ICPMI_SX(nrm_sorted, ms, jc), a1[o] = ICPMI_SY(nrm_sorted, ms, jc), a2[o] = ICPMI_SZ(nrm_sorted, ms, jc);
            oj[o] = (int)perm[jc];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int jj = j0 + 16 * o;
            const double dd = sqdist(cx[o], cy[o], cz[o], qx, qy, qz);
            if (jj < m && (dd < ld || (dd == ld && oj[o] < lj))) {
                ld = dd, lj = oj[o];
                lq0 = cx[o], lq1 = cy[o], lq2 = cz[o];
                ln0 = a0[o], ln1 = a1[o], ln2 = a2[o];
            }
        }
    }
    double bd = ld;
    int bj = lj;
    row16_argmin(bd, bj);
    ICPMI_SMALL_STAMP(4); // slots scanned

    // certificate: every split's record against its bound (resolve_certify without the first filter:
    // there are at most 16 splits, one per lane of the quarter)
    {
        const double sq = sqrt(bd);
        const bool look = valid && finite3(qx, qy, qz);
        bool whole = false, slot = false;
        if (ql < splits && look) {
            const float tauf = split_tau(qx, qy, qz, frames[ql], bd, sq);
            whole = v2 <= tauf;
            slot = !whole && ql != bs && v1 <= tauf;
        }
        unsigned long long pend = __ballot(whole || slot);
        while (pend) { // rare; wave-uniform loop
            const int L = __ffsll((long long)pend) - 1;
            pend &= pend - 1;
            const double sx = __shfl(qx, L, 64), sy = __shfl(qy, L, 64), sz = __shfl(qz, L, 64);
            const int w = __shfl((int)whole, L, 64);
            const int c = __shfl((int)(__float_as_uint(v1) & 31u), L, 64);
            const int sL = L & 15;
            double d = 1.7976931348623157e308;
            int j = 0x7fffffff;
            if (w)
                scan_split<2>(sorted, perm, m, ms, reinterpret_cast<const double *>(frames + splits), sL, sx, sy, sz,
                              __shfl(bd, L, 64), lane, d, j);
            else scan_range<kSlotTargets, 1>(sorted, perm, m, ms, sL * kSplitTargets + c * kSlotTargets, sx, sy, sz, lane, d, j);
            if ((lane >> 4) == (L >> 4) && (d < bd || (d == bd && j < bj))) {
                bd = d;
                bj = j;
            }
            if (counters && lane == L) atomicAdd(&counters[w ? 1 : 0], 1ull);
        }
    }
    ICPMI_SMALL_STAMP(5); // certificate done

    // 4. J row and b (icp.hpp:99-117) by the lane that holds the winner's point and normal; if the certificate
    // moved the winner out of the scanned slot (a few rows in a thousand) or the row has no neighbour
    // (non-finite: index 0 stands in, as in the general path), lane 0 of the quarter gathers by original index
    {
        bool have = valid && lj == bj && bj != 0x7fffffff;
        const unsigned long long anyhave = __ballot(have) & (0xFFFFull << (16 * quarter));
        if (valid && !anyhave && ql == 0) {
            const int j = (unsigned)bj < (unsigned)m ? bj : 0;
            lq0 = tgt_orig[3 * j], lq1 = tgt_orig[3 * j + 1], lq2 = tgt_orig[3 * j + 2];
            ln0 = nrm[3 * j], ln1 = nrm[3 * j + 1], ln2 = nrm[3 * j + 2];
            have = true;
        }
        if (have) {
            double J[6];
            J[0] = qy * ln2 - qz * ln1; // p x n, icp.hpp:105
            J[1] = qz * ln0 - qx * ln2;
            J[2] = qx * ln1 - qy * ln0;
            J[3] = ln0;
            J[4] = ln1;
            J[5] = ln2;
            const double e0 = lq0 - qx, e1 = lq1 - qy, e2 = lq2 - qz;
            const double bb = (e0 * ln0 + e1 * ln1) + e2 * ln2; // icp.hpp:116
            double *row = jrow[qi];
            int o = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) row[o++] = J[r] * J[c];
#pragma unroll
            for (int r = 0; r < 6; ++r) row[21 + r] = J[r] * bb;
            row[27] = bb * bb;
        } else if (!valid && ql == 0) {
#pragma unroll
            for (int e = 0; e < 28; ++e) jrow[qi][e] = 0.0;
        }
    }
    ICPMI_SMALL_STAMP(6); // rows of terms in LDS
    __builtin_amdgcn_wave_barrier();
    // k_nn_resolve4<8>'s order: a wave's four rows, then the waves in order
    if (lane < 28)
        red[wave][lane] = ((jrow[wave * 4][lane] + jrow[wave * 4 + 1][lane]) + jrow[wave * 4 + 2][lane]) + jrow[wave * 4 + 3][lane];
    __syncthreads();
    if (threadIdx.x < 28) {
        const int e = threadIdx.x;
        double v = red[0][e];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) v += red[w][e];
        partials[(size_t)blockIdx.x * kSumsStride + e] = v;
    }
#ifdef ICPMI_SMALL_CLOCKS
    ICPMI_SMALL_STAMP(7);
    if (clocks && lane == 0 && (wave == 0 || wave == kSmallWaves - 1)) {
        unsigned long long *o = clocks + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * kSmallStamps;
        for (int k = 0; k < kSmallStamps; ++k) o[k] = stamp[k];
        o[11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

} // namespace icpmi

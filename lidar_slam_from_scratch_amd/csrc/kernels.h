// kernels.h -- HIP kernels of the point-to-plane ICP path for gfx950 (wave64).
//
//   k_nn_f64          exact fp64 exhaustive nearest neighbour (kdtree.hpp:43-59,112-142)
//   k_nn_merge        min over target splits
//   (k-NN + PCA normals live in nn_mfma.h: k_knn_resolve / k_knn_exact_list / k_normals_from_knn)
//   k_reduce          residuals + 6x6 normal-equation partial sums (icp.hpp:99-120,198-206)
//   k_finish / k_step fixed-order final sum, error, convergence test, LDLT solve,
//                     Rodrigues, pose accumulation (icp.hpp:207-231)
//   k_transform       cloud * R^T + t^T (icp.hpp:174-176,225-226)
//   k_finish_step_transform / k_step_transform
//                     the loop's fused forms: final sum + step + pose update of the rows in one
//                     launch (single GPU) / step + pose update behind the all-reduce (sharded)
//
// All arithmetic that decides an index or a flag is fp64 in the reference's operation
// order (no FMA contraction in this TU).  Reductions use wave shuffles + LDS and a fixed
// partial order: results are run-to-run bit-stable (no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "device_math.h"

namespace icpmi {

constexpr int kWave = 64;
constexpr int kNumSums = 29;       // 21 JtJ (upper, row by row) + 6 Jtb + sum b^2 + count
constexpr int kSumsStride = 32;
// Sharded runs all-reduce one more word: sums[kDoneSlot] = 1 on a rank whose loop has ended.
// Small integers add exactly in any order, so every rank reads the same count whatever the
// collective's algorithm: the hosts stop queueing iterations on that agreed count alone.
constexpr int kDoneSlot = 29;
constexpr int kNumExchanged = 30;

struct IcpState {
    double total[16];       // accumulated source->target transform (icp.hpp:178,229)
    double delta[16];       // last update (icp.hpp:220)
    double sums[kSumsStride];
    double prev_error;      // icp.hpp:179,231
    double last_error;
    double final_error;
    double tolerance;
    double min_error;
    int32_t hist_len;
    int32_t done;           // loop left (break or exhausted): later kernels are no-ops
    int32_t converged;
    int32_t loops;
    int32_t max_hist;
    int32_t error;          // sharded runs: the ranks disagreed on `done` (k_step)
    int32_t finalized;      // the post-loop entry (icp.hpp:235-252) is already in the history (see step_update)
    int32_t pad;
};

// ------------------------------------------------------------------------------------
// exact fp64 exhaustive 1-NN.  grid = (query blocks, target splits), 256 threads, QPT
// queries per thread held in registers.  Every lane needs the same target point at the
// same time, so targets come through the scalar unit (wave-uniform address -> s_load into
// SGPRs, broadcast to the VALU for free) instead of LDS.
// ------------------------------------------------------------------------------------
template <int QPT>
__global__ __launch_bounds__(256) void k_nn_f64(const double *__restrict__ qry, int n,
                                                const double *__restrict__ tgt, int m,
                                                int tgt_per_split,
                                                double *__restrict__ out_d2,
                                                int *__restrict__ out_idx,
                                                const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    const int base = blockIdx.x * (256 * QPT) + threadIdx.x;
    double px[QPT], py[QPT], pz[QPT], best[QPT];
    int bidx[QPT];
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int i = base + r * 256;
        const int ic = i < n ? i : n - 1;
        px[r] = qry[3 * ic];
        py[r] = qry[3 * ic + 1];
        pz[r] = qry[3 * ic + 2];
        best[r] = 1.7976931348623157e308; // DBL_MAX, kdtree.hpp:54
        bidx[r] = -1;
    }
    const int j0 = blockIdx.y * tgt_per_split;
    const int j1 = min(m, j0 + tgt_per_split);
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const double tx = tgt[3 * j], ty = tgt[3 * j + 1], tz = tgt[3 * j + 2];
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const double d = sqdist(tx, ty, tz, px[r], py[r], pz[r]);
            if (d < best[r]) { // strict: first (lowest-index) minimum wins
                best[r] = d;
                bidx[r] = j;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int i = base + r * 256;
        if (i < n) {
            out_d2[(size_t)blockIdx.y * n + i] = best[r];
            out_idx[(size_t)blockIdx.y * n + i] = bidx[r];
        }
    }
}

// min over splits; ties -> lowest index (splits are in increasing index order)
__global__ __launch_bounds__(256) void k_nn_merge(const double *__restrict__ part_d2,
                                                  const int *__restrict__ part_idx, int n,
                                                  int splits, int *__restrict__ idx,
                                                  double *__restrict__ d2,
                                                  const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double best = part_d2[i];
    int bi = part_idx[i];
    for (int s = 1; s < splits; ++s) {
        const double d = part_d2[(size_t)s * n + i];
        if (d < best) {
            best = d;
            bi = part_idx[(size_t)s * n + i];
        }
    }
    idx[i] = bi;
    if (d2) d2[i] = best;
}

// ------------------------------------------------------------------------------------
// residuals + normal equations.  Each thread accumulates the 28 sums over a grid-stride
// slice, then wave shuffle reduction -> LDS across the 4 waves -> one partial row per
// block.  HBM-bound gather: 24 B (p) + 4 B (idx) + 24 B (q) + 24 B (n) per point.
// When `matched` is non-null, q and n come from row i of (tgt, nrm) (icp.hpp:89-93).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__global__ __launch_bounds__(256) void k_reduce(const double *__restrict__ cur, int n,
                                                const double *__restrict__ tgt, int m_tgt,
                                                const double *__restrict__ nrm,
                                                const int *__restrict__ idx,
                                                double *__restrict__ partials,
                                                const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    double acc[28];
#pragma unroll
    for (int e = 0; e < 28; ++e) acc[e] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int j = idx ? idx[i] : i;
        // a query with NaN/Inf coordinates has no nearest neighbour (index -1, like
        // kdtree.hpp:53); the sums are garbage then anyway, but the gather must stay in bounds
        j = (unsigned)j < (unsigned)m_tgt ? j : 0;
        const double p0 = cur[3 * i], p1 = cur[3 * i + 1], p2 = cur[3 * i + 2];
        const double q0 = tgt[3 * j], q1 = tgt[3 * j + 1], q2 = tgt[3 * j + 2];
        const double n0 = nrm[3 * j], n1 = nrm[3 * j + 1], n2 = nrm[3 * j + 2];
        double J[6];
        J[0] = p1 * n2 - p2 * n1; // p x n, icp.hpp:105
        J[1] = p2 * n0 - p0 * n2;
        J[2] = p0 * n1 - p1 * n0;
        J[3] = n0;
        J[4] = n1;
        J[5] = n2;
        const double d0 = q0 - p0, d1 = q1 - p1, d2 = q2 - p2;
        const double b = (d0 * n0 + d1 * n1) + d2 * n2; // icp.hpp:116
        int o = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) acc[o++] += J[r] * J[c];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] += J[r] * b;
        acc[27] += b * b;
    }
    __shared__ double red[4][28];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 28; ++e) {
        const double s = wave_sum(acc[e]);
        if (lane == 0) red[wave][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        const int e = threadIdx.x;
        partials[(size_t)blockIdx.x * kSumsStride + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
}

// sum the per-block partial rows in a fixed order -> st->sums[0..27]; sums[28] = count.
// kFinishThreads threads = G row groups x 32 columns: every load instruction reads whole
// 256-byte rows, thread (g, e) adds rows g, g+G, ... of column e with sixteen loads in flight,
// then the G groups are added in order.  Must be called by the whole workgroup.
constexpr int kFinishThreads = 1024;
constexpr int kFinishGroups = kFinishThreads / 32;

__device__ inline void finish_sums(const double *__restrict__ partials, int nblocks, int n_local,
                                   IcpState *st)
{
    __shared__ double fs[kFinishGroups][32];
    constexpr int G = kFinishGroups;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    // U partial sums per thread = U loads in flight: the kernel is a chain of memory round trips
    // (1,563 rows at 100k points: 49 per thread, 4 rounds instead of 13 with U = 4)
    constexpr int U = 16;
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.0;
    // Every round's loads are UNCONDITIONAL (a clamped row, its value dropped by a select: acc + 0.0 is acc, no slot ever
    // holds -0.0) so that they are all in flight together in the last, partial round too.  Written with a guard per load that
    // round was a chain of dependent round trips -- and below 481 rows it is the only round: eight trips for the 250 rows
    // of a filtered scan, four for the 98 group sums of C3.  Only as many slots as the launch's rows fill are loaded (4, 8
    // or 16: a load instruction costs the CU's address unit 16 cycles whether its lanes' rows are clamped or not -- with
    // all sixteen slots always loaded, C3's step kernel lost 0.5 us where the filtered scan's gained 0.5).
    const int last = nblocks - 1;
    auto rounds = [&](auto slots) {
        constexpr int S = decltype(slots)::value;
        for (int b = g; b < nblocks; b += U * G) {
            double v[S];
#pragma unroll
            for (int u = 0; u < S; ++u) {
                const int bb = b + u * G;
                v[u] = partials[(size_t)(bb < nblocks ? bb : last) * kSumsStride + e];
            }
#pragma unroll
            for (int u = 0; u < S; ++u) acc[u] += (b + u * G < nblocks) ? v[u] : 0.0;
        }
    };
    if (nblocks <= 4 * G) rounds(std::integral_constant<int, 4>());
    else if (nblocks <= 8 * G) rounds(std::integral_constant<int, 8>());
    else rounds(std::integral_constant<int, U>());
#pragma unroll
    for (int w = U / 2; w > 0; w >>= 1) // fixed pairwise order
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    fs[g][e] = acc[0];
    __syncthreads();
    if (threadIdx.x < 28) {
        double s = fs[0][e];
        for (int k = 1; k < G; ++k) s += fs[k][e];
        st->sums[e] = s;
    }
    if (threadIdx.x == 0) st->sums[28] = (double)n_local;
}

// A second level for the sums of very many partial rows (round 4): a resolve workgroup leaves one row per 64 queries --
// 15,625 at 1M rows -- and every workgroup of the step kernel summed them all (4 MB through each of its CUs).  From
// kSumTreeFrom rows on, this small kernel first adds the rows in groups of kSumGroup, in index order: the step kernels sum
// ceil(nblocks / kSumGroup) rows.  A kernel of its own, not a hand-off inside the resolve kernel: with the last workgroup
// of a group adding the group's rows behind agent-scope fences the resolve took 170 us instead of 27 at C3 (a release
// writes the XCD's L2 back, an acquire drops the CU's L1, once per workgroup and seven workgroups to a CU), and the
// fence-free sc1 form is measured for one workgroup per CU only (MI355X_MICROARCH.md).  A launch boundary costs ~3 us and
// the step kernel's sum of C3's 1,563 rows 6.7: measured at C3, default engine, 30 iterations, same box: 9,614 -> 9,810
// iterations/s with the threshold at 1,024 rows (ICPMI_SUM_TREE_FROM moves it; the sums' order of additions differs on
// either side of it, inside every tolerance of the tests).
constexpr int kSumGroup = 16;
constexpr int kSumTreeFrom = 1024;
__global__ __launch_bounds__(256) void k_sum_groups(const double *__restrict__ rows, int nblocks, double *__restrict__ rows2,
                                                    const IcpState *__restrict__ st)
{
    if (st && st->done) return;
    const int e = threadIdx.x & 31, grp = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r0 = grp * kSumGroup;
    if (r0 >= nblocks || e >= kNumSums) return;
    const int members = nblocks - r0 < kSumGroup ? nblocks - r0 : kSumGroup;
    double v[kSumGroup];
#pragma unroll
    for (int r = 0; r < kSumGroup; ++r) v[r] = r < members ? rows[(size_t)(r0 + r) * kSumsStride + e] : 0.0; // (all in flight)
    double s = v[0];
#pragma unroll
    for (int r = 1; r < kSumGroup; ++r) s += v[r];
    rows2[(size_t)grp * kSumsStride + e] = s;
}

// error, convergence tests, solve, accumulate (icp.hpp:206-231 / 251-255)
__device__ inline void step_update(IcpState *st, double *history, int final_pass)
{
    if (st->done) {
        if (final_pass && !st->finalized) {
            // the loop broke on convergence: the source has not moved since the last
            // evaluation, so the reference's post-loop pass (icp.hpp:235-252) recomputes
            // exactly last_error
            st->final_error = st->last_error;
            if (history && st->hist_len < st->max_hist) history[st->hist_len] = st->last_error;
            st->hist_len += 1;
            st->finalized = 1;
        }
        return;
    }
    const double error = __dsqrt_rn(st->sums[27] / st->sums[28]);
    if (history && st->hist_len < st->max_hist) history[st->hist_len] = error;
    st->hist_len += 1;
    st->last_error = error;
    if (final_pass) {
        st->final_error = error;
        st->done = 1;
        return;
    }
    st->loops += 1;
    if (error < st->min_error || fabs(st->prev_error - error) < st->tolerance) { // icp.hpp:210-213, 214-217
        st->converged = 1;
        st->done = 1;
        // The post-loop pass of a loop that broke here restates this very error (the source has not moved:
        // icp.hpp:235-252 on unchanged data).  It is entered into the history NOW, so that a host that has seen the
        // loop end need not queue a post-loop pass at all.
        st->final_error = error;
        if (history && st->hist_len < st->max_hist) history[st->hist_len] = error;
        st->hist_len += 1;
        st->finalized = 1;
        return;
    }
    double x[6];
    ldlt6_solve(st->sums, x);       // icp.hpp:120
    twist_to_transform(x, st->delta);    // icp.hpp:123-143
    mul44(st->delta, st->total, st->total); // icp.hpp:229
    st->prev_error = error;              // icp.hpp:231
}

// step_update by the workgroup's first WAVE (all 64 lanes call it, `st` in LDS): the bookkeeping is lane 0's, the solve
// and the 4x4 product are spread over the lanes (ldlt6_solve_wave, mul44_wave), Rodrigues is computed by every lane alike.
// Same operations on every value as step_update -> same bits; ~2.6 instead of 4.7 us between the sums and the moved rows
// of every iteration (scripts/micro/step_clocks.hip).
__device__ inline void step_update_wave(IcpState *st, double *history, int final_pass, int lane)
{
    const bool l0 = lane == 0;
    if (st->done) {
        if (l0 && final_pass && !st->finalized) {
            st->final_error = st->last_error;
            if (history && st->hist_len < st->max_hist) history[st->hist_len] = st->last_error;
            st->hist_len += 1;
            st->finalized = 1;
        }
        return;
    }
    const double error = __dsqrt_rn(st->sums[27] / st->sums[28]);
    const double prev_error = st->prev_error;
    const bool stop = error < st->min_error || fabs(prev_error - error) < st->tolerance; // icp.hpp:210-213, 214-217
    const int hist_len = st->hist_len, max_hist = st->max_hist;
    __builtin_amdgcn_wave_barrier();
    if (l0) {
        if (history && hist_len < max_hist) history[hist_len] = error;
        st->hist_len = hist_len + 1;
        st->last_error = error;
        if (final_pass) {
            st->final_error = error;
            st->done = 1;
        } else {
            st->loops += 1;
            if (stop) { // (the post-loop entry is made now: see step_update)
                st->converged = 1;
                st->done = 1;
                st->final_error = error;
                if (history && hist_len + 1 < max_hist) history[hist_len + 1] = error;
                st->hist_len = hist_len + 2;
                st->finalized = 1;
            } else {
                st->prev_error = error; // icp.hpp:231
            }
        }
    }
    if (final_pass || stop) return;
    double x[6], T[16];
    ldlt6_solve_wave(st->sums, x, lane); // icp.hpp:120
    twist_to_transform(x, T);            // icp.hpp:123-143
    if (l0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st->delta[e] = T[e];
    }
    __builtin_amdgcn_wave_barrier();
    mul44_wave(st->delta, st->total, st->total, lane); // icp.hpp:229
}

// single GPU: final sum + step in one launch
// `progress` (may be null) is one word of host-mapped memory per iteration slot: the device
// publishes (iteration + 1) * 2 + done there, so the host can stop queueing iterations
// without any copy or event in the stream.
__device__ __forceinline__ void publish_progress(int *progress, int ticket, int done)
{
    // relaxed: the host reads this one word only (results are read after a stream synchronisation);
    // a release here would write the L2 back first, microseconds on every iteration's critical path
    if (progress) __hip_atomic_store(progress, ticket * 2 + (done ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The state is staged in LDS for the serial part: step_update touches ~100 of its words one
// after the other, and each would otherwise be a round trip to device memory by a lone thread.
static_assert(sizeof(IcpState) % 8 == 0, "IcpState is copied as 64-bit words");
__device__ __forceinline__ void state_copy(void *dst, const void *src)
{
    const unsigned long long *s = static_cast<const unsigned long long *>(src);
    unsigned long long *d = static_cast<unsigned long long *>(dst);
    for (unsigned w = threadIdx.x; w < sizeof(IcpState) / 8; w += blockDim.x) d[w] = s[w];
}

__global__ __launch_bounds__(kFinishThreads) void k_finish_step(const double *__restrict__ partials,
                                                     int nblocks, int n_local, IcpState *st,
                                                     double *history, int final_pass, int *progress,
                                                     int ticket)
{
    __shared__ IcpState ls, sums; // `sums`: only its sums[] are used
    // the state and the partial rows are requested together (one round trip instead of two); the
    // sums are taken over only if the loop has not ended
    state_copy(&ls, st);
    finish_sums(partials, nblocks, n_local, &sums);
    __syncthreads();
    if (!ls.done && threadIdx.x < kNumSums) ls.sums[threadIdx.x] = sums.sums[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 64) {
        step_update_wave(&ls, history, final_pass, threadIdx.x);
        if (threadIdx.x == 0) publish_progress(progress, ticket, ls.done);
    }
    __syncthreads();
    state_copy(st, &ls);
}

// multi GPU: k_finish -> ncclAllReduce(st->sums, kNumExchanged) -> k_step
__global__ __launch_bounds__(kFinishThreads) void k_finish(const double *__restrict__ partials, int nblocks,
                                                int n_local, IcpState *st)
{
    if (st->done) {
        // keep the all-reduce operands finite and identical on every rank
        if (threadIdx.x < kNumSums) st->sums[threadIdx.x] = 0.0;
        if (threadIdx.x == 0) st->sums[kDoneSlot] = 1.0;
        return;
    }
    finish_sums(partials, nblocks, n_local, st);
    if (threadIdx.x == 0) st->sums[kDoneSlot] = 0.0;
}

// The word published to the host carries the AGREED end of the loop: the number of ranks that
// entered this iteration with `done` set (decided one step earlier) came through the exchange
// as an exact integer.  All of them: the loop has ended everywhere, the hosts stop queueing at
// the same iteration and therefore queue the same number of collectives.  Some but not all
// (differing configs, or an all-reduce that did not hand every rank the same bits): every rank
// sees the same count, flags the error and ends its loop at this same iteration.
__global__ __launch_bounds__(64) void k_step(IcpState *st, double *history, int final_pass, int *progress, int ticket,
                                             int n_ranks)
{
    __shared__ IcpState ls;
    state_copy(&ls, st);
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave (step_update_wave)
        const double ndone = ls.sums[kDoneSlot];
        const bool all = ndone == (double)n_ranks, some = ndone > 0.0 && !all;
        if (some && threadIdx.x == 0) {
            ls.error = 1;
            ls.done = 1;
        }
        __builtin_amdgcn_wave_barrier();
        step_update_wave(&ls, history, final_pass, threadIdx.x);
        if (threadIdx.x == 0) publish_progress(progress, ticket, all || some);
    }
    __syncthreads();
    state_copy(st, &ls);
}
// ---- moving the rows (icp.hpp:174-176, 225-226), with the next pass's bounds ---------------------------------------
// What the bounded correspondence search (nn_bounded.h) wants to know about a moved row before it starts: the exact squared
// distance to the target the row was matched with one pass ago, and its fp32 images rounded up.  tgt == nullptr: nothing.
struct RowBounds {
    const double *tgt; // the target, caller's order
    const int *idx;    // the rows' matches of the pass just finished
    int m;
    double *ub;        // [n] |moved row - tgt[idx]|^2, +Inf without a match
    float *ubf, *sqf;  // [n] (float)ub and its square root, rounded up; NaN / 0 for a row with a non-finite coordinate
    int *cnt;          // [n] the row's list length, cleared here for the coming coarse pass
};
// B rows of one thread (i0, i0 + stride, ...): every load that does not depend on the pose -- the rows, their previous
// matches' indices, then those targets -- is requested by load(), which the fused kernels call BEFORE their serial step, so
// that the round trips run under it; finish() moves the rows and stores the bounds.
template <int B>
struct RowBatch {
    double x[B], y[B], z[B], tx[B], ty[B], tz[B];
    bool have[B];
    int j[B];
    // the rows and their previous matches' indices ...
    __device__ __forceinline__ void load_rows(const double *in, const RowBounds &rb, int i0, int stride, int n,
                                              const unsigned *__restrict__ perm = nullptr)
    {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * stride;
            x[b] = y[b] = z[b] = 0.0;
            j[b] = -1;
            if (i < n) {
                const size_t ii = perm ? perm[i] : (unsigned)i;
                x[b] = in[3 * ii], y[b] = in[3 * ii + 1], z[b] = in[3 * ii + 2];
                if (rb.tgt) j[b] = rb.idx[i];
            }
        }
    }
    // ... and the matched targets, a round trip that DEPENDS on the indices: a kernel with other loads to issue (the state,
    // the partial rows) calls this after them -- waves issue in order, so behind this gather's wait for the indices those
    // loads would start one round trip late
    __device__ __forceinline__ void load_matches(const RowBounds &rb)
    {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            have[b] = rb.tgt && (unsigned)j[b] < (unsigned)rb.m;
            tx[b] = ty[b] = tz[b] = 0.0;
            if (have[b]) tx[b] = rb.tgt[3 * (size_t)j[b]], ty[b] = rb.tgt[3 * (size_t)j[b] + 1], tz[b] = rb.tgt[3 * (size_t)j[b] + 2];
        }
    }
    __device__ __forceinline__ void load(const double *in, const RowBounds &rb, int i0, int stride, int n,
                                         const unsigned *__restrict__ perm = nullptr)
    {
        load_rows(in, rb, i0, stride, n, perm);
        load_matches(rb);
    }
    __device__ __forceinline__ void finish(double *out, const RowBounds &rb, int i0, int stride, int n, const double *T) const
    {
        const double r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
        const double r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
        const double r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * stride;
            if (i >= n) continue;
            const double px = ((x[b] * r00 + y[b] * r01) + z[b] * r02) + t0;
            const double py = ((x[b] * r10 + y[b] * r11) + z[b] * r12) + t1;
            const double pz = ((x[b] * r20 + y[b] * r21) + z[b] * r22) + t2;
            out[3 * i] = px;
            out[3 * i + 1] = py;
            out[3 * i + 2] = pz;
            if (rb.tgt) {
                // A row with a NaN or infinite coordinate has no neighbour (kdtree.hpp:125): NaN, under which the coarse pass
                // lists nothing.  No previous match (every target non-finite ...): +Inf, everything is listed.
                double ub = __builtin_inf();
                float ubf = __builtin_nanf(""), sqf = 0.f;
                if (__builtin_isfinite(px) && __builtin_isfinite(py) && __builtin_isfinite(pz)) {
                    if (have[b]) ub = sqdist(tx[b], ty[b], tz[b], px, py, pz);
                    ubf = (float)ub;
                    ubf = (double)ubf < ub ? __uint_as_float(__float_as_uint(ubf) + 1u) : ubf; // (ub >= 0; Inf stays Inf)
                    sqf = __builtin_amdgcn_sqrtf(ubf);
                    sqf = sqf < 3.0e38f ? __uint_as_float(__float_as_uint(sqf) + 2u) : sqf;   // (1 ulp of v_sqrt_f32 and one more)
                }
                rb.ub[i] = ub;
                rb.ubf[i] = ubf;
                rb.sqf[i] = sqf;
                rb.cnt[i] = 0;
            }
        }
    }
};
constexpr int kRowBatch = 4;


// multi GPU, one launch fewer per iteration: k_step and k_transform in one kernel.  Every workgroup
// repeats the (deterministic) step from the all-reduced sums on its own LDS copy of the state --
// the same bits in, the same bits out -- and moves its points with the update it has just formed
// (icp.hpp:220-226); workgroup 0 alone records the history entry, publishes the progress word and
// stores the new state.  It stores it into the OTHER of two state buffers: a workgroup that starts
// late must still read the state of before the step, so the loop's kernels alternate between the
// two (`sin` of one iteration is `sout` of the previous one).
__global__ __launch_bounds__(256) void k_step_transform(const double *in, double *out, int n, const IcpState *sin,
                                                        IcpState *sout, double *history, int *progress, int ticket,
                                                        int n_ranks, const RowBounds rb)
{
    __shared__ IcpState ls;
    const int i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    RowBatch<kRowBatch> rows;
    rows.load(in, rb, i0, stride, n);
    state_copy(&ls, sin);
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave (step_update_wave)
        const double ndone = ls.sums[kDoneSlot];
        const bool all = ndone == (double)n_ranks, some = ndone > 0.0 && !all;
        if (some && threadIdx.x == 0) {
            ls.error = 1;
            ls.done = 1;
        }
        __builtin_amdgcn_wave_barrier();
        step_update_wave(&ls, blockIdx.x == 0 ? history : nullptr, 0, threadIdx.x);
        if (threadIdx.x == 0 && blockIdx.x == 0) publish_progress(progress, ticket, all || some);
    }
    __syncthreads();
    if (blockIdx.x == 0) state_copy(sout, &ls);
    if (ls.done) return; // the loop ended before or in this step: the source stays where it is (icp.hpp:210-217)
    for (int base = i0; base < n; base += kRowBatch * stride) {
        if (base != i0) rows.load(in, rb, base, stride, n);
        rows.finish(out, rb, base, stride, n, ls.delta);
    }
}

// single GPU, two launches fewer per iteration: k_finish_step and k_transform in one kernel, on the
// same footing as k_step_transform.  EVERY workgroup sums all partial rows itself (same rows, same
// order, same bits: a 400 KB read out of the L2 at 100k points, 50 KB at the 7k points of a filtered
// scan), repeats the step on its own LDS copy of the state and moves its share of the points;
// workgroup 0 alone records the history entry, publishes the progress word and stores the new state
// into the other state buffer.  No workgroup waits for another one.
__global__ __launch_bounds__(kFinishThreads) void k_finish_step_transform(
    const double *__restrict__ partials, int nblocks, int n_local, const double *in, double *out, int n,
    const IcpState *sin, IcpState *sout, double *history, int *progress, int ticket, const RowBounds rb)
{
    __shared__ IcpState ls, sums; // `sums`: only its sums[] are used
    // The kernel is a chain of memory round trips, so everything that does not depend on the state
    // is requested first: this thread's first rows (with their previous matches), and the partial rows (summed whether
    // or not the loop has ended; the sums are taken over only if it has not, like k_finish_step).
    const int i0 = blockIdx.x * kFinishThreads + threadIdx.x, stride = gridDim.x * kFinishThreads;
    RowBatch<kRowBatch> rows;
    rows.load_rows(in, rb, i0, stride, n);
    state_copy(&ls, sin);
    finish_sums(partials, nblocks, n_local, &sums);
    __syncthreads();
    rows.load_matches(rb); // (in flight under the step)
    if (!ls.done && threadIdx.x < kNumSums) ls.sums[threadIdx.x] = sums.sums[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave (step_update_wave)
        step_update_wave(&ls, blockIdx.x == 0 ? history : nullptr, 0, threadIdx.x);
        if (threadIdx.x == 0 && blockIdx.x == 0) publish_progress(progress, ticket, ls.done);
    }
    __syncthreads();
    if (blockIdx.x == 0) state_copy(sout, &ls);
    if (ls.done) return; // the loop ended before or in this step: the source stays where it is (icp.hpp:210-217)
    for (int base = i0; base < n; base += kRowBatch * stride) {
        if (base != i0) rows.load(in, rb, base, stride, n);
        rows.finish(out, rb, base, stride, n, ls.delta);
    }
}

// one-shot solve for icpmi_solve_point_to_plane (icp.hpp:89-144)
__global__ __launch_bounds__(kFinishThreads) void k_finish_solve(const double *__restrict__ partials,
                                                      int nblocks, int n_local, IcpState *st)
{
    finish_sums(partials, nblocks, n_local, st);
    __syncthreads();
    if (threadIdx.x == 0) {
        double x[6];
            ldlt6_solve(st->sums, x);
        twist_to_transform(x, st->delta);
    }
}

// out = in * R^T + t^T with T read from device memory (st->delta or a staged matrix).
// which: 0 = st->delta, 1 = st->total
// perm (may be null): out row i is made from in row perm[i] (the ICP loop takes its source rows in Morton order)
__global__ __launch_bounds__(256) void k_transform(const double *in, double *out, int n,
                                                   const IcpState *__restrict__ st, int which,
                                                   int honour_done, const unsigned *__restrict__ perm = nullptr,
                                                   const RowBounds rb = RowBounds{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr})
{
    if (honour_done && st->done) return;
    const double *T = which ? st->total : st->delta;
    const int i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    RowBatch<kRowBatch> rows;
    for (int base = i0; base < n; base += kRowBatch * stride) {
        rows.load(in, rb, base, stride, n, perm);
        rows.finish(out, rb, base, stride, n, T);
    }
}

} // namespace icpmi

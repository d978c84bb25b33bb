// nn_culled.h -- the culled correspondence search (ICPMI_SEARCH_MFMA_PRUNED, what AUTO runs on large targets), round 4:
// the cull at WAVE granularity.
//
// The reference's search is a culling search: KDTree::nearest_recursive (kdtree.hpp:112-142) leaves a subtree as soon as
// the splitting plane is farther away than the best distance so far.  Here the same rule is applied to whole groups of
// rows against whole splits of the target, before any pair is evaluated:
//   * the rows of the moved source are kept in Morton order (one sort per call), so 64 consecutive rows -- what ONE
//     wave of the coarse pass works on -- are a compact blob: their bounding box is a few nearest-neighbour distances wide;
//   * every row comes with a bound ub(row) on its nearest-neighbour distance before the pass starts: the exact distance
//     to the target it was matched with one pass ago (RowBounds, kernels.h) or, in a call's first pass, to the nearest
//     of the sorted targets around the row's own place in the target's Morton order (k_nn_prebound1);
//   * a split s of 2048 sorted targets can hold a target within ub(row) of a row of group g only if
//     gap^2(box(g), box(s)) <= max over the group's rows of ub: every other (group, split) pair is culled -- strictly
//     greater, with a margin for the test's own roundings, so not even a target at EQUAL distance is lost and the
//     lowest-index tie rule survives;
//   * the surviving pairs are appended to one list per SPLIT (block_cull), and k_nn_coarse_groups runs the MFMA unit of
//     nn_mfma.h on them: a workgroup takes eight groups of ONE split's list -- any eight, they need not be neighbours --
//     stages the split's operands through LDS once and gives each wave one group.  The epilogue is MODE 2's: the slots
//     under the row's per-split threshold tau_s(ub(row)) are listed per row, and k_nn_resolve_bounded (nn_bounded.h) scans
//     the listed slots exactly, the bound's target as the incumbent.
// Round 3 culled (512-row block, split) pairs: 84 % of them at C3, and a surviving pair cost all eight waves of a unit
// their 128 MFMAs although most of the eight had nothing within reach.  At 64 rows the box is half as wide per axis.
// Exactness is the argument of nn_bounded.h unchanged: a target within the row's final distance <= ub(row) lies in a
// split whose box is within sqrt(ub(row)) <= sqrt(max ub) of the row, hence of the group's box; that pair is evaluated,
// the target's slot is listed for the row and scanned.  Rows whose list does not fit take the resolve's exhaustive
// search behind the split and slot boxes, which needs no list at all.
#pragma once
#include "nn_mfma.h"

namespace icpmi {

constexpr int kGroupRows = 32;          // rows per group: one 32-row MFMA tile (a wave of the coarse pass takes two, any two of a split's list)
constexpr int kGroupStamps = 12;        // diagnostic build (-DICPMI_GROUPS_CLOCKS): words per workgroup of k_nn_coarse_groups
constexpr int kCullMaxSplits = 3072;    // splits whose running sums fit the coarse kernel's LDS beside its 36 KiB of staging (2 x 3072 + 1 words of the 64 KiB a workgroup may have): 6.3M targets; beyond: all pairs
static_assert(kGroupRows == kTile && kCoarseQT == 2, "a group is one tile; a wave of the coarse unit takes two");
static_assert(sizeof(uint4) * CoarseLds<kCoarseWaves>::SCRATCH16 + sizeof(unsigned) * (2 * kCullMaxSplits + 1) + 64 <= 65536,
              "k_nn_coarse_groups: staging + the running sums of kCullMaxSplits splits must fit a workgroup's 64 KiB of LDS");

// One list of row groups per target split (the pairs that survived the box test), for the coming coarse pass.
struct GroupLists {
    unsigned *cnt;    // [nsplits] groups appended so far (nullptr: no culling wanted)
    unsigned *items;  // [nsplits][cap]
    int cap;          // groups a list can hold = groups of the pass (a group enters a split's list at most once)
};

// The box test for the groups of a workgroup.  Per wave: (lo, hi, ub) are the lanes' own rows of group g (a lane without a
// row: lo = +big, hi = -big, ub = 0), reduced here over the wave.  A bound of +Inf (a row without any) is replaced by the one
// that needs no match at all: every split holds a target, and that target is no farther from any row of the group than
// the largest distance between the two boxes.
// The same test for all the groups of a workgroup at once (16 in a 1024-thread one), with ONE global atomic per (workgroup, split that
// any of its groups reaches) instead of one per (group, split): the workgroup counts its survivors per split in LDS,
// reserves a run in each split's list, and its waves place their groups inside the runs.  (With an atomic per pair the
// ~7,400 pairs of a C3 pass queued up on 49 addresses: 150 returning atomics per address, one after the other in the L2 --
// the kernel took 29.6 us against the 16.9 of the form without lists.)  Must be called by every wave of the workgroup
// (four barriers); a wave without rows passes lo = +big, hi = -big.
constexpr int kCullMaxRows = 1 << 25;                      // rows of a source (k_nn_coarse_groups keeps a list's length in 20 bits: tiles of 32 rows)
constexpr int kCullLdsBoxes = 128; // targets of up to this many splits (262k points): the splits' boxes are staged in LDS
struct CullLds {
    unsigned cnt[kCullMaxSplits];   // survivors of this workgroup per split, then the cursor inside its run
    double box[kCullLdsBoxes][6];   // lo[3], hi[3] of every split (cull_stage_boxes), when they fit
};
// Requested early by the kernels that have a serial step in front of their box test (the boxes do not depend on the
// pose): block_cull's first barrier publishes them.  As six strided 8-byte loads per (lane, split) -- twice, for the
// count and for the placement -- the boxes were most of what the test added to its kernel.
__device__ __forceinline__ void cull_stage_boxes(CullLds &lds, const SplitFrame *__restrict__ frames, const int nsplits)
{
    if (nsplits > kCullLdsBoxes) return;
    for (int t = threadIdx.x; t < nsplits * 6; t += blockDim.x) {
        const int s = t / 6, a = t - 6 * s;
        lds.box[s][a] = a < 3 ? frames[s].lo[a] : frames[s].hi[a - 3];
    }
}
// `g2` = the wave's first tile (its 64 rows are tiles g2 and g2 + 1: lanes 0-31 and 32-63).
__device__ __forceinline__ void block_cull(CullLds &lds, const int g2, double (&lo)[3], double (&hi)[3], double ub,
                                           const SplitFrame *__restrict__ frames, const int nsplits, const GroupLists &gl,
                                           const int lane)
{
    __syncthreads(); // (a workgroup with several rounds of rows: the previous round's cursors are no longer in use)
    for (int s = threadIdx.x; s < nsplits; s += blockDim.x) lds.cnt[s] = 0u;
    // box and bound of each HALF of the wave (one tile each): five exchange steps inside the halves
    // (lane_xor: DPP / v_permlane16_swap moves, device_math.h -- as __shfl_xor these were 70 LDS crossbar trips)
#define ICPMI_STEP(X)                                                                     \
    {                                                                                     \
        _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                   \
            const double l2 = lane_xor<X>(lo[a]), h2 = lane_xor<X>(hi[a]);                \
            lo[a] = l2 < lo[a] ? l2 : lo[a];                                              \
            hi[a] = h2 > hi[a] ? h2 : hi[a];                                              \
        }                                                                                 \
        const double u2 = lane_xor<X>(ub);                                                \
        ub = u2 > ub ? u2 : ub;                                                           \
    }
    ICPMI_STEP(16) ICPMI_STEP(8) ICPMI_STEP(4) ICPMI_STEP(2) ICPMI_STEP(1)
#undef ICPMI_STEP
    const bool any = lo[0] <= hi[0]; // a finite row in this half's tile
    if (__ballot(any && !(ub < 1.0e300)) != 0ull) { // (rare: a tile without any bound)
        double far2 = __builtin_inf();
        for (int s = lane & 31; s < nsplits; s += 32) {
            double f2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double f1 = hi[a] - frames[s].lo[a], f3 = frames[s].hi[a] - lo[a];
                const double f = f1 > f3 ? f1 : f3;
                f2 += f * f;
            }
            far2 = f2 < far2 ? f2 : far2;
        }
#define ICPMI_STEP(X) { const double o = lane_xor<X>(far2); far2 = o < far2 ? o : far2; }
        ICPMI_STEP(16) ICPMI_STEP(8) ICPMI_STEP(4) ICPMI_STEP(2) ICPMI_STEP(1)
#undef ICPMI_STEP
        if (!(ub < 1.0e300)) ub = far2 * (1.0 + 1e-12);
    }
    // every lane gets both tiles' boxes: A = lanes 0-31's, B = lanes 32-63's
    double alo[3], ahi[3], blo[3], bhi[3], aub, bub;
    bool aany, bany;
    {
        const bool upper = lane >= 32;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double ol = lane_xor<32>(lo[a]), oh = lane_xor<32>(hi[a]);
            alo[a] = upper ? ol : lo[a], ahi[a] = upper ? oh : hi[a];
            blo[a] = upper ? lo[a] : ol, bhi[a] = upper ? hi[a] : oh;
        }
        const double ou = lane_xor<32>(ub);
        aub = upper ? ou : ub, bub = upper ? ub : ou;
        const int oany = lane_xor<32>((int)any);
        aany = upper ? (oany != 0) : any, bany = upper ? any : (oany != 0);
    }
    const bool staged = nsplits <= kCullLdsBoxes;
    // both tiles against split s -> bit 0: tile A within reach, bit 1: tile B
    auto reach2 = [&](const int s) -> unsigned {
        double sl[3], sh[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            sl[a] = staged ? lds.box[s][a] : frames[s].lo[a];
            sh[a] = staged ? lds.box[s][3 + a] : frames[s].hi[a];
        }
        double ga = 0.0, gb = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double a1 = alo[a] - sh[a], a3 = sl[a] - ahi[a], b1 = blo[a] - sh[a], b3 = sl[a] - bhi[a];
            const double xa = a1 > a3 ? a1 : a3, xb = b1 > b3 ? b1 : b3;
            ga += xa > 0.0 ? xa * xa : 0.0;
            gb += xb > 0.0 ? xb * xb : 0.0;
        }
        const bool ra = aany & !(ga * (1.0 - 1e-12) > aub * (1.0 + 1e-12)), rb = bany & !(gb * (1.0 - 1e-12) > bub * (1.0 + 1e-12));
        return (unsigned)ra | ((unsigned)rb << 1);
    };
    __syncthreads(); // counters cleared, boxes staged
    unsigned first_bits = 0u; // the lane's own split of the first round, kept for the placement below
    for (int s = lane; s < nsplits; s += 64) {
        const unsigned bits = reach2(s);
        if (s == lane) first_bits = bits;
        if (bits) atomicAdd(&lds.cnt[s], (bits & 1u) + (bits >> 1));
    }
    __syncthreads(); // counted
    for (int s = threadIdx.x; s < nsplits; s += blockDim.x) {
        const unsigned c = lds.cnt[s];
        if (c) lds.cnt[s] = atomicAdd(gl.cnt + s, c); // the run [base, base + c) of split s's list is this workgroup's
    }
    __syncthreads(); // runs reserved: cnt[] now holds each run's cursor
    for (int s = lane; s < nsplits; s += 64) {
        const unsigned bits = s == lane ? first_bits : reach2(s);
        if (bits) {
            const unsigned pos = atomicAdd(&lds.cnt[s], (bits & 1u) + (bits >> 1));
            unsigned *dst = gl.items + (size_t)s * gl.cap + pos;
            if (bits & 1u) dst[0] = (unsigned)g2;
            if (bits & 2u) dst[bits & 1u] = (unsigned)g2 + 1u;
        }
    }
}

// What a kernel that moves rows leaves for the bounded pass that follows (RowBatch::finish's images, kernels.h), for one
// row per lane; returns the row's contribution to its group's box and bound.
__device__ __forceinline__ void row_bound_store(const RowBounds &rb, const int i, const bool valid, const double px, const double py,
                                                const double pz, const bool have, const double tx, const double ty, const double tz,
                                                double (&lo)[3], double (&hi)[3], double &ubg)
{
    lo[0] = lo[1] = lo[2] = 1.7e308;
    hi[0] = hi[1] = hi[2] = -1.7e308;
    ubg = 0.0;
    if (!valid) return;
    double ub = __builtin_inf();
    float ubf = __builtin_nanf(""), sqf = 0.f;
    const bool fin = __builtin_isfinite(px) && __builtin_isfinite(py) && __builtin_isfinite(pz);
    if (fin) {
        if (have) ub = sqdist(tx, ty, tz, px, py, pz);
        ubf = (float)ub;
        ubf = (double)ubf < ub ? __uint_as_float(__float_as_uint(ubf) + 1u) : ubf; // (ub >= 0; Inf stays Inf)
        sqf = __builtin_amdgcn_sqrtf(ubf);
        sqf = sqf < 3.0e38f ? __uint_as_float(__float_as_uint(sqf) + 2u) : sqf;   // (1 ulp of v_sqrt_f32 and one more)
        lo[0] = hi[0] = px, lo[1] = hi[1] = py, lo[2] = hi[2] = pz;
        ubg = ub == ub ? ub : __builtin_inf(); // (a NaN distance -- a non-finite target behind the match: cannot happen, kept safe)
    }
    rb.ub[i] = ub;
    rb.ubf[i] = ubf;
    rb.sqf[i] = sqf;
    rb.cnt[i] = 0;
}

// ---- a call's FIRST pass: the bound from the row's place in the target's Morton order ---------------------------------
// One row per lane.  The row's Morton key in the target's frame (k_morton_keys' arithmetic) is looked up in the sorted
// keys; the kPreWindow sorted positions around that place are neighbours of the row in space wherever the curve is
// continuous, and the nearest of them, exactly measured, is a real target: its distance bounds the row's nearest-
// neighbour distance like a previous match does, and it is written to idx[] AS the previous match.  Rows of a wave are
// neighbours too (the source is in Morton order), so their windows overlap and the loads mostly hit the same lines.
// With group lists the wave then culls the splits for its 64 rows (pruned engine).
constexpr int kPreWindow = 128;
constexpr int kPreThreads = 512;
__global__ __launch_bounds__(kPreThreads) void k_nn_prebound1(const double *__restrict__ cur, int n, const double *__restrict__ sorted,
                                                      const unsigned *__restrict__ keys_sorted, const unsigned *__restrict__ perm,
                                                      int m, int ms, const NnFrame *__restrict__ frame, int *__restrict__ idx,
                                                      const RowBounds rb, const SplitFrame *__restrict__ frames, int nsplits,
                                                      const GroupLists gl)
{
    __shared__ CullLds cl;
    if (gl.cnt) cull_stage_boxes(cl, frames, nsplits);
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kPreThreads + threadIdx.x;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = cur[3 * ic], py = cur[3 * ic + 1], pz = cur[3 * ic + 2];
    unsigned key;
    {
        double ext = 0.0;
        for (int a = 0; a < 3; ++a) ext = frame->hi[a] - frame->lo[a] > ext ? frame->hi[a] - frame->lo[a] : ext;
        unsigned q[3];
        for (int a = 0; a < 3; ++a) {
            double f = ext > 0.0 ? ((a == 0 ? px : a == 1 ? py : pz) - frame->lo[a]) / ext : 0.0;
            f = !(f >= 0.0) ? 0.0 : (f > 1.0 ? 1.0 : f);
            const int qi = (int)(f * 1023.0);
            q[a] = (unsigned)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
        }
        key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    int lo_p = 0, hi_p = m;
    while (lo_p < hi_p) {
        const int mid = (lo_p + hi_p) >> 1;
        if (keys_sorted[mid] < key) lo_p = mid + 1;
        else hi_p = mid;
    }
    int w0 = lo_p - kPreWindow / 2;
    w0 = w0 + kPreWindow > m ? m - kPreWindow : w0;
    w0 = w0 < 0 ? 0 : w0;
    const int w1 = w0 + kPreWindow < m ? w0 + kPreWindow : m;
    double bd = 1.7976931348623157e308;
    int bj = -1;
#pragma unroll 8
    for (int j = w0; j < w1; ++j) {
        const double d = sqdist(ICPMI_SX(sorted, ms, j), ICPMI_SY(sorted, ms, j), ICPMI_SZ(sorted, ms, j), px, py, pz);
        if (d < bd) { // (a NaN or infinite distance -- a non-finite target or row -- never enters)
            bd = d;
            bj = j;
        }
    }
    const bool have = bj >= 0;
    const int jo = have ? (int)perm[bj] : -1;
    if (valid) idx[i] = jo;
    double lo[3], hi[3], ubg;
    // (the bound is restated from the target's coordinates exactly as RowBatch::finish forms it from the caller's array:
    // the sorted copy holds the same doubles)
    const int bjc = have ? bj : 0;
    row_bound_store(rb, i, valid, px, py, pz, have, ICPMI_SX(sorted, ms, bjc), ICPMI_SY(sorted, ms, bjc), ICPMI_SZ(sorted, ms, bjc), lo, hi,
                    ubg);
    if (gl.cnt) block_cull(cl, (i - lane) / kGroupRows, lo, hi, ubg, frames, nsplits, gl, lane);
}

// ---- the pose update of the rows with the next pass's bounds and group lists -------------------------------------------
// k_transform's work (icp.hpp:174-176, 225-226) for one row per lane, 64 consecutive rows per wave = one group; the rows'
// bounds like RowBatch::finish, then the box test of the group.  Used where the step is a kernel of its own (sharded runs:
// the all-reduce sits between the sums and the step).
__global__ __launch_bounds__(256) void k_transform_cull(const double *in, const unsigned *__restrict__ perm, double *out, int n,
                                                        const IcpState *__restrict__ st, int which, int honour_done,
                                                        const RowBounds rb, const SplitFrame *__restrict__ frames, int nsplits,
                                                        const GroupLists gl)
{
    __shared__ CullLds cl;
    if (honour_done && st->done) return;
    if (gl.cnt && rb.ub) cull_stage_boxes(cl, frames, nsplits);
    const double *T = which ? st->total : st->delta;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    double p[3] = {0.0, 0.0, 0.0}, t[3] = {0.0, 0.0, 0.0};
    bool have = false;
    if (valid) {
        const size_t si = perm ? perm[i] : (unsigned)i;
        const double x = in[3 * si], y = in[3 * si + 1], z = in[3 * si + 2];
        const int j = rb.tgt && rb.idx ? rb.idx[i] : -1;
        have = (unsigned)j < (unsigned)rb.m;
        if (have) t[0] = rb.tgt[3 * (size_t)j], t[1] = rb.tgt[3 * (size_t)j + 1], t[2] = rb.tgt[3 * (size_t)j + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] = ((x * T[4 * r] + y * T[4 * r + 1]) + z * T[4 * r + 2]) + T[4 * r + 3];
        out[3 * i] = p[0];
        out[3 * i + 1] = p[1];
        out[3 * i + 2] = p[2];
    }
    if (!rb.ub) return;
    double lo[3], hi[3], ubg;
    row_bound_store(rb, i, valid, p[0], p[1], p[2], have, t[0], t[1], t[2], lo, hi, ubg);
    if (gl.cnt) block_cull(cl, (i - lane) / kGroupRows, lo, hi, ubg, frames, nsplits, gl, lane);
}

// Single GPU: final sum + step + pose update of the rows (k_finish_step_transform, kernels.h) + the rows' bounds and the
// group lists of the coming pass, in one launch.  Every workgroup sums all partial rows itself and repeats the step on
// its own copy of the state; a wave takes whole groups (64 consecutive rows), a row per lane.
__global__ __launch_bounds__(kFinishThreads) void k_finish_step_transform_cull(
    const double *__restrict__ partials, int nblocks, int n_local, const double *in, double *out, int n, const IcpState *sin,
    IcpState *sout, double *history, int *progress, int ticket, const RowBounds rb, const SplitFrame *__restrict__ frames,
    int nsplits, const GroupLists gl)
{
    __shared__ IcpState ls, sums; // `sums`: only its sums[] are used
    __shared__ CullLds cl;
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * kFinishThreads + threadIdx.x, stride = gridDim.x * kFinishThreads;
    // everything that does not depend on the state is requested first: this thread's first row, its previous match and
    // that target; the partial rows (k_finish_step_transform's order of business)
    double x = 0.0, y = 0.0, z = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
    bool have = false;
    int jprev = -1;
    auto load_row = [&](const int i) { // the row and its previous match's index ...
        x = y = z = 0.0;
        jprev = -1;
        if (i < n) {
            x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
            jprev = rb.idx[i];
        }
    };
    auto load_match = [&]() { // ... and the matched target: a round trip that depends on the index (RowBatch::load_matches)
        tx = ty = tz = 0.0;
        have = (unsigned)jprev < (unsigned)rb.m;
        if (have) tx = rb.tgt[3 * (size_t)jprev], ty = rb.tgt[3 * (size_t)jprev + 1], tz = rb.tgt[3 * (size_t)jprev + 2];
    };
    load_row(i0);
    cull_stage_boxes(cl, frames, nsplits);
    state_copy(&ls, sin);
    finish_sums(partials, nblocks, n_local, &sums);
    __syncthreads();
    load_match(); // (in flight under the step.  The splits' boxes staged here too, instead of in front: +2 % on the kernel)
    if (!ls.done && threadIdx.x < kNumSums) ls.sums[threadIdx.x] = sums.sums[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave (step_update_wave)
        step_update_wave(&ls, blockIdx.x == 0 ? history : nullptr, 0, threadIdx.x);
        if (threadIdx.x == 0 && blockIdx.x == 0) publish_progress(progress, ticket, ls.done);
    }
    __syncthreads();
    if (blockIdx.x == 0) state_copy(sout, &ls);
    if (ls.done) return; // the loop ended before or in this step: the source stays where it is (icp.hpp:210-217)
    const double *T = ls.delta;
    const double r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
    const double r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
    const double r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
    // (workgroup-uniform trip count -- block_cull has barriers: the workgroup's first row decides; a wave past the end
    // of the rows brings an empty box)
    for (int base = i0; base - (int)threadIdx.x < n; base += stride) {
        if (base != i0) {
            load_row(base);
            load_match();
        }
        const bool valid = base < n;
        const double px = ((x * r00 + y * r01) + z * r02) + t0;
        const double py = ((x * r10 + y * r11) + z * r12) + t1;
        const double pz = ((x * r20 + y * r21) + z * r22) + t2;
        if (valid) {
            out[3 * base] = px;
            out[3 * base + 1] = py;
            out[3 * base + 2] = pz;
        }
        double lo[3], hi[3], ubg;
        row_bound_store(rb, base, valid, px, py, pz, have, tx, ty, tz, lo, hi, ubg);
        block_cull(cl, (base - lane) / kGroupRows, lo, hi, ubg, frames, nsplits, gl, lane);
    }
}

// Sharded runs: k_step_transform (kernels.h) with the group lists -- the step from the all-reduced sums repeated by every
// workgroup, the pose update of this rank's rows, their bounds, the box test of their groups.
__global__ __launch_bounds__(256) void k_step_transform_cull(const double *in, double *out, int n, const IcpState *sin, IcpState *sout,
                                                             double *history, int *progress, int ticket, int n_ranks, const RowBounds rb,
                                                             const SplitFrame *__restrict__ frames, int nsplits, const GroupLists gl)
{
    __shared__ IcpState ls;
    __shared__ CullLds cl;
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    double x = 0.0, y = 0.0, z = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
    bool have = false;
    auto load_row = [&](const int i) {
        x = y = z = tx = ty = tz = 0.0;
        have = false;
        if (i < n) {
            x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
            const int j = rb.idx[i];
            have = (unsigned)j < (unsigned)rb.m;
            if (have) tx = rb.tgt[3 * (size_t)j], ty = rb.tgt[3 * (size_t)j + 1], tz = rb.tgt[3 * (size_t)j + 2];
        }
    };
    load_row(i0);
    cull_stage_boxes(cl, frames, nsplits);
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
    if (ls.done) return;
    const double *T = ls.delta;
    const double r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
    const double r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
    const double r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
    for (int base = i0; base - (int)threadIdx.x < n; base += stride) { // (workgroup-uniform trip count: block_cull has barriers)
        if (base != i0) load_row(base);
        const bool valid = base < n;
        const double px = ((x * r00 + y * r01) + z * r02) + t0;
        const double py = ((x * r10 + y * r11) + z * r12) + t1;
        const double pz = ((x * r20 + y * r21) + z * r22) + t2;
        if (valid) {
            out[3 * base] = px;
            out[3 * base + 1] = py;
            out[3 * base + 2] = pz;
        }
        double lo[3], hi[3], ubg;
        row_bound_store(rb, base, valid, px, py, pz, have, tx, ty, tz, lo, hi, ubg);
        block_cull(cl, (base - lane) / kGroupRows, lo, hi, ubg, frames, nsplits, gl, lane);
    }
}

// ---- normal estimation: the groups are runs of 32 sorted target rows, their bound the largest of the rows' own ----------
// (k_knn_prebound's T: the k-th neighbour of every row lies within it).  One wave per pair of groups; rows = sorted
// positions row0 .. row0 + nrows, groups and bounds numbered from the launch's first row (row0 a multiple of 64).
__global__ __launch_bounds__(1024) void k_knn_group_cull(const double *__restrict__ sorted, int m, int ms, int row0, int nrows,
                                                        const double *__restrict__ t_row, const SplitFrame *__restrict__ frames,
                                                        int nsplits, const GroupLists gl)
{
    __shared__ CullLds cl;
    cull_stage_boxes(cl, frames, nsplits);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 6); // the wave's 64 rows: groups 2 w and 2 w + 1
    const int local = w * 64 + lane;
    const int j = row0 + local;
    const bool valid = local < nrows && j < m;
    double lo[3] = {1.7e308, 1.7e308, 1.7e308}, hi[3] = {-1.7e308, -1.7e308, -1.7e308}, ub = 0.0;
    if (valid) {
        const double x = ICPMI_SX(sorted, ms, j), y = ICPMI_SY(sorted, ms, j), z = ICPMI_SZ(sorted, ms, j);
        if (finite3(x, y, z)) { // (a non-finite row has no neighbours: its list stays empty and the exact kernel answers)
            lo[0] = hi[0] = x, lo[1] = hi[1] = y, lo[2] = hi[2] = z;
            const double t = t_row[local];
            ub = t == t ? t : __builtin_inf();
        }
    }
    block_cull(cl, 2 * w, lo, hi, ub, frames, nsplits, gl, lane); // (a wave past the last group brings empty boxes)
}

// ---- the coarse pass over the group lists ------------------------------------------------------------------------------
// Two (row group, split) pairs per wave -- its two 32-row tiles, any two of the list -- 2 WAVES pairs of the SAME split per workgroup: the unit of nn_mfma.h (operands of the
// split staged through LDS in two chunks, 128 MFMAs per wave, MODE 2 epilogue) with each wave's rows taken from the
// split's list.  A fixed grid strides over the chunks of all lists: chunk c belongs to the split s with
// pre[s] <= c < pre[s + 1], pre = the running sum of ceil(cnt[s] / WAVES) -- formed by every workgroup for itself (a few
// dozen loads and one wave scan), so no kernel sits between the box test and this one.  `cnt_next`: the counters the NEXT
// pass's lists will be appended to (by the kernel that moves the rows after this pass); nobody reads or writes them while
// this kernel runs, so they are cleared here.
// The chunks are handed out DYNAMICALLY: a workgroup starts on chunk blockIdx and takes its next one from a counter
// (`work`: this pass's; `work_next`: the next pass's, cleared here).  With a static stride the two workgroups of a CU do not
// share it evenly -- the SIMDs' arbiter serves the older workgroup's waves first (in-kernel stamps, scripts/groups_clock.py:
// first chunk 14.8 us in the workgroup that arrived first, 26 us in the other) -- so the 84 late workgroups that also
// had a second chunk were a tail of 12 us behind everybody else: 42 us where 30 do.
// Dynamic LDS: (2 nsplits + 1) words.
template <bool QSOA, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nn_coarse_groups(
    const double *__restrict__ qry, int n, size_t qstride, const uint4 *__restrict__ Bpack, const SplitFrame *__restrict__ frames,
    int nsplits, const unsigned *__restrict__ items, int cap, const unsigned *__restrict__ cnt, unsigned *__restrict__ cnt_next,
    unsigned long long *__restrict__ stats /* [0] += pairs run, [1] += pairs of the pass (may be null) */, unsigned long long pairs_total,
    const IcpState *__restrict__ st, const KnnLists kl, unsigned *__restrict__ work, unsigned *__restrict__ work_next,
    unsigned long long *__restrict__ clocks = nullptr /* diagnostic build only */)
{
#ifdef ICPMI_GROUPS_CLOCKS /* scripts/groups_clock.py: 100 MHz stamps per workgroup -- entry, lists known, each chunk's ends, exit */
    unsigned long long stamp[kGroupStamps];
    for (int k = 0; k < kGroupStamps; ++k) stamp[k] = 0;
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    int nchunk = 0;
#endif
    __shared__ uint4 lds[CoarseLds<WAVES>::SCRATCH16];
    extern __shared__ unsigned dyn_lds[];
    // pre[s]: FULL chunks (2 WAVES entries) of the splits before s; word2[s]: bits 0-19 the length of split s's list, bits
    // 20-31 the split whose PARTIAL chunk is the s-th of the partial chunks (two arrays in one)
    unsigned *pre = dyn_lds, *word2 = dyn_lds + nsplits + 1;
    constexpr unsigned PER = 2 * WAVES;
    static_assert(kCullMaxSplits <= 4096 && (PER & (PER - 1)) == 0, "12 bits of split, chunks of a power of two");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int done = st ? st->done : 0;
    __shared__ unsigned s_full, s_total;
    if (wave == 0) {
        unsigned run = 0, listed = 0;
        for (int s0 = 0; s0 < nsplits; s0 += 64) {
            const int s = s0 + lane;
            const unsigned c = s < nsplits ? cnt[s] : 0u;
            const unsigned ch = c / PER;
            const unsigned inc = wave_scan_incl(ch);
            const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)wave_scan_incl(c), 63);
            if (s < nsplits) {
                pre[s] = run + inc - ch;
                word2[s] = c; // (< 2^20: at most n / 32 entries, n < 2^25 -- engine_for)
            }
            run += (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
            listed += tot;
        }
        // the partial chunks (the last, short chunk of a list) behind all the full ones, the longer ones first: the chunks
        // are handed out in index order, so the kernel's tail -- the last chunks, each alone on its CU -- is made of
        // the cheapest ones (a chunk of two tiles keeps one wave busy, a full one eight)
        unsigned parts = 0;
#pragma unroll 1
        for (int cls = 3; cls >= 0; --cls) {
            for (int s0 = 0; s0 < nsplits; s0 += 64) {
                const int s = s0 + lane;
                const unsigned rem = s < nsplits ? (word2[s] & 0xFFFFFu) % PER : 0u;
                const bool in = rem != 0u && (int)(rem * 4u / PER) == cls;
                const unsigned long long mask = __ballot(in);
                if (in) atomicOr(&word2[parts + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))], (unsigned)s << 20);
                parts += (unsigned)__popcll(mask);
            }
        }
        if (lane == 0) {
            pre[nsplits] = run;
            s_full = run;
            s_total = run + parts;
            if (blockIdx.x == 0 && stats && !done) {
                atomicAdd(stats, (unsigned long long)listed);
                atomicAdd(stats + 1, pairs_total);
            }
        }
    }
    if (done) return; // (the whole grid alike: nothing was cleared, nothing will be appended)
    if (blockIdx.x == 0 && cnt_next)
        for (int s = threadIdx.x; s < nsplits; s += 64 * WAVES) cnt_next[s] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && work_next) *work_next = 0u;
    __shared__ unsigned next_chunk;
    __syncthreads();
#ifdef ICPMI_GROUPS_CLOCKS
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned nfull = s_full, total = s_total;
    // chunk c -> its split, and this wave's two list entries (requested, not waited for)
    auto lookup = [&](const unsigned c, int &s, bool &active, bool &active_b, uint2 &gv) {
        unsigned item, len;
        if (c < nfull) { // (workgroup-uniform)
            int slo = 0, shi = nsplits; // pre[slo] <= c < pre[shi]
            while (shi - slo > 1) {
                const int mid = (slo + shi) >> 1;
                if (pre[mid] <= c) slo = mid;
                else shi = mid;
            }
            s = __builtin_amdgcn_readfirstlane(slo);
            len = word2[s] & 0xFFFFFu;
            item = (c - pre[s]) * PER + 2u * (unsigned)wave; // (even: cap is even, the pair is 8-byte aligned)
        } else {
            s = __builtin_amdgcn_readfirstlane((int)(word2[c - nfull] >> 20));
            len = word2[s] & 0xFFFFFu;
            item = (len & ~(PER - 1u)) + 2u * (unsigned)wave;
        }
        active = item < len;
        active_b = item + 1u < len;
        gv = active ? *reinterpret_cast<const uint2 *>(items + (size_t)s * cap + item) : make_uint2(0u, 0u);
    };
    unsigned c = blockIdx.x;
#pragma unroll 1
    while (c < total) {
        uint2 gv = make_uint2(0u, 0u);
        int s = 0;
        bool active = false, active_b = false;
        lookup(c, s, active, active_b, gv);
        const int g = __builtin_amdgcn_readfirstlane((int)gv.x);
        // (a wave with one tile only -- the odd last entry of a list: the second tile's rows lie past the end, nothing of it is listed)
        const int gb = active_b ? __builtin_amdgcn_readfirstlane((int)gv.y) * kGroupRows : n;
#ifdef ICPMI_GROUPS_CLOCKS
        if (nchunk < 3) stamp[2 + 2 * nchunk] = __builtin_amdgcn_s_memrealtime();
        if (nchunk == 0) stamp[8] = (unsigned long long)s | ((unsigned long long)__popcll(__ballot(active)) << 32);
#endif
        coarse_unit_rows<2, kCoarseQT, WAVES, QSOA>(lds, g * kGroupRows, active, s, nsplits, qry, n, qstride, Bpack, frames, nullptr, nullptr,
                                                    kl, 0, gb, false /* operands behind the A build: measured, nn_mfma.h */);
        __syncthreads(); // the epilogue's LDS is the next chunk's operand buffer; everybody has read next_chunk
#ifdef ICPMI_GROUPS_CLOCKS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (nchunk < 3) stamp[3 + 2 * nchunk] = __builtin_amdgcn_s_memrealtime();
        ++nchunk;
#endif
        // the next chunk is claimed only now, when this one is done: a workgroup that claimed ahead would sit on a chunk
        // while its faster neighbours have run out (the other workgroup of the CU fills the claim's round trip)
        if (threadIdx.x == 0) next_chunk = gridDim.x + atomicAdd(work, 1u);
        __syncthreads();
        c = next_chunk;
    }
#ifdef ICPMI_GROUPS_CLOCKS
    if (clocks && threadIdx.x == 0) {
        stamp[10] = __builtin_amdgcn_s_memrealtime();
        stamp[11] = (unsigned long long)nchunk;
        stamp[9] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); // HW_ID | XCC_ID
        for (int k = 0; k < kGroupStamps; ++k) clocks[(size_t)blockIdx.x * kGroupStamps + k] = stamp[k];
    }
#endif
}

} // namespace icpmi

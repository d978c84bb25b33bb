// nn_bounded.h -- the correspondence search of an ICP pass that has a previous pass behind it (icp.hpp:181-196 from
// the second iteration on), all-pairs MFMA engine, round 3.
//
// Every row of the moved source still knows the target it was matched with one pass ago.  That target is a real
// target, so its exact distance to the moved row, ub, bounds the new nearest-neighbour distance from above BEFORE
// anything is searched -- and with the bound in hand the coarse pass no longer has to keep its minima:
//   RowBounds (kernels.h) ub(row) = |row - target[previous match]|^2 exactly (and its fp32 images, rounded up), left by
//                        the kernel that moves the rows after a pass (k_finish_step_transform, k_step_transform,
//                        k_transform): the previous matches' coordinates are fetched under that kernel's serial step
//   k_nn_coarse_bounded  the all-pairs pass with the MODE 2 epilogue: the columns (slots of 64 sorted targets) whose
//                        minimum is <= tau_s(ub(row)), the bound on the coarse value of any target of split s within
//                        ub (nn_mfma.h), are listed per row, 16 columns to a word -- one word per row, rarely two,
//                        instead of 6 B per (row, split): 29 MB written and read back per C3 pass before, 2.9 GB at
//                        1M x 1M
//   k_nn_resolve_bounded exact fp64 scan of the listed slots (the reference's operation order), smallest distance,
//                        ties to the lowest ORIGINAL index, against the previous match as the incumbent
// Every target at exact distance <= the final minimum <= ub has a coarse value <= its split's tau_s(ub) (the error
// bound of nn_mfma.h), so its slot is listed and it is scanned: the result is the exact nearest neighbour with the reference's
// tie rule -- what k_nn_resolve returns, bit for bit -- and there is no certificate left to run, because the bound was
// applied before the search instead of after it.  A row whose list does not fit (more than kNnEntCap words or
// kNnSlotCap slots: a pose update so large that ub spans many slots) is searched exhaustively instead, split by
// split behind the split's bounding box and slot by slot behind the slot's (scan_split) with the incumbent's distance
// as the radius: slower for that row (a few dozen slot scans instead of one or two), never wrong.  Since round 4 the first
// pass of a call is bounded too (k_nn_prebound1, nn_culled.h: the nearest sorted target around the row's Morton place).
// The normal-equation terms are formed and summed exactly as in k_nn_resolve<16> (resolve_finish), so the partial rows,
// hence history and pose, are bit-identical to the unbounded pass's.
#pragma once
#include "nn_mfma.h"

namespace icpmi {

constexpr int kNnSlotCap = 16; // listed slots scanned per row

// Q = 16 queries per wave, the lane layout, workgroup shape and sums of k_nn_resolve<16>.
#ifndef ICPMI_BOUNDED_OCC
#define ICPMI_BOUNDED_OCC 5 /* waves per SIMD the register allocation must allow.  With a slot's loads requested as a batch (SlotBatch, nn_mfma.h) and all four rounds' state live the kernel wanted ~107 registers: at 5 (96, nothing spilled) 19.8 us per C3 pass, at 6 (40 dwords spilled) 24.2, at 7 (72, more spilled) 30.4; before the batch -- every candidate's loads next to their use, sixteen dependent trips per wave -- 7 was the best (22.1).  Round by round (one round's query, distances and indices live: 75 registers) it fits 7 without spills, and measures 19.3 at 5 or 6, 19.9 at 7, 22.0 at 8: 5 stays (scripts/ab_kernels.sh, same box) */
#endif
__global__ __launch_bounds__(64 * kResolveWW) __attribute__((amdgpu_waves_per_eu(ICPMI_BOUNDED_OCC, 8))) void k_nn_resolve_bounded(
    const double *__restrict__ qry, int n, const double *__restrict__ sorted, const unsigned *__restrict__ perm, int m, int ms,
    int splits, const SplitFrame *__restrict__ frames, const double *__restrict__ ub_row, const int *__restrict__ cnt_row,
    const unsigned *__restrict__ ent_row, int *__restrict__ idx /* in: previous match, out: this pass's */,
    unsigned long long *__restrict__ counters, const double *__restrict__ tgt_orig, const double *__restrict__ nrm,
    double *__restrict__ partials, const IcpState *__restrict__ st)
{
    constexpr int Q = 16, ROUNDS = Q / 4;
    static_assert(kNnEntCap == 8, "two words per sub-lane");
    __shared__ int flist[kResolveWW][Q][kNnSlotCap];
    if (st && st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & (Q - 1), sub = lane / Q;
    const int l16 = lane & 15, quarter = lane >> 4;
    const int qbase = (blockIdx.x * kResolveWW + wave) * Q;
    const int i = qbase + ql; // waves past the end run on a clamped query and write nothing
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = qry[3 * ic], py = qry[3 * ic + 1], pz = qry[3 * ic + 2];
    const bool look = valid && finite3(px, py, pz);
    const double kMax = 1.7976931348623157e308;

    // the incumbent: the previous match at its exact distance
    const int jprev = idx[ic];
    double bd = ub_row[ic];
    int bj = (unsigned)jprev < (unsigned)m ? jprev : 0x7fffffff;
    if (!look) bj = 0x7fffffff;
    if (bj == 0x7fffffff) bd = kMax;

    // the row's list: sub-lane `sub` of query ql holds words 2 sub and 2 sub + 1.  The FIRST listed slot is the lowest bit
    // of word 0 (a word is only ever appended with a bit set); all but a hundredth of the rows list nothing else.
    const int cnt = cnt_row[ic];
    const uint2 w = *reinterpret_cast<const uint2 *>(ent_row + (size_t)ic * kNnEntCap + 2 * sub);
    const unsigned w0 = (look && 2 * sub < cnt) ? w.x : 0u, w1 = (look && 2 * sub + 1 < cnt) ? w.y : 0u;
    const int mine = __popc(w0 & 0xFFFFu) + __popc(w1 & 0xFFFFu);
    // prefix over the query's four sub-lanes (one in each row of 16 lanes) and the total, by two row exchanges
    const int pair = mine + lane_xor<16>(mine);                   // this sub-lane's pair of rows
    const int other = lane_xor<32>(pair);                          // the other pair
    const int incl = ((sub & 1) ? pair : mine) + ((sub & 2) ? other : 0);
    const int nsl = pair + other; // listed slots of query ql (every sub-lane of the query knows it)
    auto word_base = [](unsigned word) -> int { return (int)(word >> 17) * kCols + (int)((word >> 16) & 1u) * 16; };
    const int first_slot = word_base(w0) + ((w0 & 0xFFFFu) ? __ffs((int)(w0 & 0xFFFFu)) - 1 : 0); // (valid in sub-lane 0)
    // More words or slots than fit -> the exhaustive search below.  So does a finite row that lists NOTHING: with a previous
    // match it lists at least that match's slot and without one everything, unless a coordinate is beyond fp32's range
    // (a singular step can throw the cloud 1e47 away: fp64 still tells the targets apart, the coarse pass sees NaN and
    // lists nothing -- found by scripts/fuzz_bounded.py, seed 7368).
    const bool over = look && (cnt <= 0 || cnt > kNnEntCap || nsl > kNnSlotCap);
    const int nsl_eff = over ? 0 : nsl;
#if defined(ICPMI_NNB_STOP) && ICPMI_NNB_STOP == 1 /* timing experiments only (WRONG results): where the kernel's time goes */
    if (nsl_eff >= 0) { if (valid && sub == 0) idx[i] = first_slot + (int)bd; return; }
#endif

    // exact evaluation of the listed slots, one query per quarter-wave and round (lane takes l16, l16+16, ...: coalesced),
    // the original indices beside the coordinates.  Round by round -- query, first slot, further slots, the quarter's
    // minimum, the owner's update -- so that only one round's distances, indices and query are live at a time.
    // Query ql is scanned by quarter ql / ROUNDS in round ql % ROUNDS, and every lane of that quarter ends up with the
    // result: the query's sub-lane IN that quarter is its OWNER from here on (incumbent, exhaustive search, terms) -- no
    // exchange between the rows of 16 lanes (as a gather from the scanning quarter this was twelve LDS crossbar trips).
    const bool own = sub == ql / ROUNDS;
    const bool more = __ballot(nsl_eff > 1) != 0ull; // further slots: through LDS, sub-lane `sub` expands its two words
    if (more) {
        int pos = incl - mine;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned word = h ? w1 : w0;
            unsigned msk = word & 0xFFFFu;
            const int base = word_base(word);
            while (msk) {
                if (pos < kNnSlotCap) flist[wave][ql][pos] = base + __ffs((int)msk) - 1;
                ++pos;
                msk &= msk - 1u;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int src = quarter * ROUNDS + r; // the query this quarter scans in round r (a lane with sub == 0)
        // (measured and dropped: the coordinates by a load of their own instead of six LDS crossbar trips -- twelve more
        // vector-memory instructions per lane in a kernel that waits on memory: 22.6 -> 23.8 us at C3, same box)
        const double qx = __shfl(px, src, 64), qy = __shfl(py, src, 64), qz = __shfl(pz, src, 64);
        const int nsr = __shfl(nsl_eff, src, 64);
        double d = kMax;
        int jo = 0x7fffffff;
        auto scan_slot = [&](const int slot, const bool act) {
            const int j0 = (slot / kCols) * kSplitTargets + (slot % kCols) * kSlotTargets + l16;
            SlotBatch b;
            b.load(sorted, perm, m, ms, j0);
            b.eval(m, j0, act, qx, qy, qz, d, jo);
        };
        scan_slot(__shfl(first_slot, src, 64), nsr > 0);
        if (more)
            for (int t = 1; __ballot(t < nsr) != 0ull; ++t) scan_slot(t < nsr ? flist[wave][src][t] : 0, t < nsr);
        row16_argmin(d, jo);
        const bool take = own & ((ql % ROUNDS) == r) & look & ((d < bd) | ((d == bd) & (jo < bj)));
        bd = take ? d : bd;
        bj = take ? jo : bj;
    }

#if defined(ICPMI_NNB_STOP) && ICPMI_NNB_STOP == 3
    if (nsl_eff >= 0) { if (valid && own) idx[i] = bj + (int)bd; return; }
#endif
    // rows whose list did not fit: every split whose bounding box is within the incumbent's distance, its slots culled
    // by their boxes (scan_split), nearest-first is not needed for correctness -- the radius only shrinks
    unsigned extra_splits = 0;
    unsigned long long pend = __ballot(over && own);
    while (pend) { // rare; wave-uniform loop
        const int L = __ffsll((long long)pend) - 1; // (the query's owner)
        pend &= pend - 1;
        const double qx = __shfl(px, L, 64), qy = __shfl(py, L, 64), qz = __shfl(pz, L, 64);
        double qd = __shfl(bd, L, 64);
        int qj = __shfl(bj, L, 64);
        for (int s0 = 0; s0 < splits; s0 += 64) {
            bool keep = false;
            if (s0 + lane < splits) {
                const SplitFrame &f = frames[s0 + lane];
                double lb = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double q = px_sel(a, qx, qy, qz);
                    const double g1 = f.lo[a] - q, g2 = q - f.hi[a];
                    const double g = g1 > g2 ? (g1 > 0.0 ? g1 : 0.0) : (g2 > 0.0 ? g2 : 0.0);
                    lb += g * g;
                }
                keep = !(lb * (1.0 - 1e-9) > qd); // (an empty split's box is inverted: infinitely far)
            }
            unsigned long long smask = __ballot(keep);
            while (smask) {
                const int sL = s0 + __ffsll((long long)smask) - 1;
                smask &= smask - 1;
                scan_split<ICPMI_RESOLVE_SCANBATCH>(sorted, perm, m, ms, reinterpret_cast<const double *>(frames + splits), sL, qx, qy,
                                                    qz, qd, lane, qd, qj);
            }
        }
        if ((lane & (Q - 1)) == (L & (Q - 1))) {
            bd = qd;
            bj = qj;
        }
        if (lane == L) ++extra_splits;
    }

#if defined(ICPMI_NNB_STOP) && ICPMI_NNB_STOP == 4
    if (nsl_eff >= 0) { if (valid && own) idx[i] = bj + (int)bd; return; }
#endif
    const unsigned extra_slots = (own && look && !over && nsl > 1) ? (unsigned)(nsl - 1) : 0u;
    resolve_finish<Q>(lane, wave, ql, own, i, valid, bd, bj, px, py, pz, m, idx, nullptr, counters, extra_slots, extra_splits,
                      tgt_orig, nrm, partials, -1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);
}

// One query per QUARTER-wave: the lane layout, workgroup shape and sums of k_nn_resolve4<WAVES> (sources of at most
// 32,768 rows: a wave's chain of round trips is what counts there, and this one has a quarter of the rounds).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_resolve4_bounded(
    const double *__restrict__ qry, int n, const double *__restrict__ sorted, const unsigned *__restrict__ perm, int m, int ms,
    int splits, const SplitFrame *__restrict__ frames, const double *__restrict__ ub_row, const int *__restrict__ cnt_row,
    const unsigned *__restrict__ ent_row, int *__restrict__ idx /* in: previous match, out: this pass's */,
    unsigned long long *__restrict__ counters, const double *__restrict__ tgt_orig, const double *__restrict__ nrm,
    double *__restrict__ partials, const IcpState *__restrict__ st)
{
    __shared__ int flist[WAVES][4][kNnSlotCap];
    if (st && st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, quarter = lane >> 4;
    const int qbase = (blockIdx.x * WAVES + wave) * 4;
    const int i = qbase + quarter; // waves past the end run on a clamped query and write nothing
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const double px = qry[3 * ic], py = qry[3 * ic + 1], pz = qry[3 * ic + 2];
    const bool look = valid && finite3(px, py, pz);
    const double kMax = 1.7976931348623157e308;

    // the incumbent: the previous match at its exact distance
    const int jprev = idx[ic];
    double bd = ub_row[ic];
    int bj = (unsigned)jprev < (unsigned)m ? jprev : 0x7fffffff;
    if (!look) bj = 0x7fffffff;
    if (bj == 0x7fffffff) bd = kMax;

    // the row's list: lane ql < 8 of the quarter holds word ql; the first listed slot is the lowest bit of word 0
    const int cnt = cnt_row[ic];
    const unsigned w = (look && ql < cnt && ql < kNnEntCap) ? ent_row[(size_t)ic * kNnEntCap + ql] : 0u;
    const int mine = __popc(w & 0xFFFFu);
    const int incl = row16_scan_incl(mine); // (lanes 8-15 of the quarter hold no word: their count is 0)
    int nsl = mine;
    nsl += row16_partner<0>(nsl), nsl += row16_partner<1>(nsl), nsl += row16_partner<2>(nsl), nsl += row16_partner<3>(nsl);
    auto word_base = [](unsigned word) -> int { return (int)(word >> 17) * kCols + (int)((word >> 16) & 1u) * 16; };
    const unsigned w0 = (unsigned)__shfl((int)w, quarter * 16, 64);
    const int first_slot = word_base(w0) + ((w0 & 0xFFFFu) ? __ffs((int)(w0 & 0xFFFFu)) - 1 : 0);
    const bool over = look && (cnt <= 0 || cnt > kNnEntCap || nsl > kNnSlotCap); // (cnt == 0: see k_nn_resolve_bounded)
    const int ns = over ? 0 : nsl;

    double d = kMax;
    int jo = 0x7fffffff;
    auto scan_slot = [&](const int slot, const bool act) {
        const int j0 = (slot / kCols) * kSplitTargets + (slot % kCols) * kSlotTargets + ql;
        // (load-then-use per candidate, not SlotBatch: one slot per quarter here, and the compiler's own order -- which
        // keeps the four candidates' loads in flight in this kernel -- measured 4 % faster than the forced batch)
        const uint4 *rec = sorted_records(sorted, ms);
#pragma unroll
        for (int o = 0; o < kSlotTargets / 16; ++o) {
            const int jj = j0 + 16 * o, jc = jj < m ? jj : m - 1; // (clamped: see SlotBatch)
            const uint4 a = rec[2 * (size_t)jc], b = rec[2 * (size_t)jc + 1]; // (the record: two loads, not four)
            const int oj = (int)b.z;
            const double dd = sqdist(__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z), __hiloint2double((int)b.y, (int)b.x), px, py, pz);
            const bool take = act & (jj < m) & ((dd < d) | ((dd == d) & (oj < jo))); // (selects, not branches: SlotBatch)
            d = take ? dd : d;
            jo = take ? oj : jo;
        }
    };
    scan_slot(first_slot, ns > 0);
    if (__ballot(ns > 1) != 0ull) { // further slots: through LDS, lane ql expands its word
        int pos = incl - mine;
        unsigned msk = w & 0xFFFFu;
        const int base = word_base(w);
        while (msk) {
            if (pos < kNnSlotCap) flist[wave][quarter][pos] = base + __ffs((int)msk) - 1;
            ++pos;
            msk &= msk - 1u;
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = 1; __ballot(t < ns) != 0ull; ++t) scan_slot(t < ns ? flist[wave][quarter][t] : 0, t < ns);
    }
    row16_argmin(d, jo);
    {
        const bool take = look & ((d < bd) | ((d == bd) & (jo < bj)));
        bd = take ? d : bd;
        bj = take ? jo : bj;
    }

    // rows whose list did not fit: exhaustive search behind the split and slot boxes (see k_nn_resolve_bounded)
    unsigned extra_splits = 0;
    unsigned long long pend = __ballot(over && ql == 0);
    while (pend) { // rare; wave-uniform loop
        const int L = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const double qx = __shfl(px, L, 64), qy = __shfl(py, L, 64), qz = __shfl(pz, L, 64);
        double qd = __shfl(bd, L, 64);
        int qj = __shfl(bj, L, 64);
        for (int s0 = 0; s0 < splits; s0 += 64) {
            bool keep = false;
            if (s0 + lane < splits) {
                const SplitFrame &f = frames[s0 + lane];
                double lb = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double q = px_sel(a, qx, qy, qz);
                    const double g1 = f.lo[a] - q, g2 = q - f.hi[a];
                    const double g = g1 > g2 ? (g1 > 0.0 ? g1 : 0.0) : (g2 > 0.0 ? g2 : 0.0);
                    lb += g * g;
                }
                keep = !(lb * (1.0 - 1e-9) > qd);
            }
            unsigned long long smask = __ballot(keep);
            while (smask) {
                const int sL = s0 + __ffsll((long long)smask) - 1;
                smask &= smask - 1;
                scan_split<ICPMI_RESOLVE4_SCANBATCH>(sorted, perm, m, ms, reinterpret_cast<const double *>(frames + splits), sL, qx, qy,
                                                     qz, qd, lane, qd, qj);
            }
        }
        if ((lane >> 4) == (L >> 4)) {
            bd = qd;
            bj = qj;
        }
        if (lane == L) ++extra_splits;
    }

    const unsigned extra_slots = (ql == 0 && look && !over && nsl > 1) ? (unsigned)(nsl - 1) : 0u;
    resolve4_finish<WAVES>(lane, wave, ql, quarter, i, valid, bd, bj, px, py, pz, m, idx, nullptr, counters, extra_slots, extra_splits,
                           tgt_orig, nrm, partials, -1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);
}

} // namespace icpmi

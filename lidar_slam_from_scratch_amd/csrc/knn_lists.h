// knn_lists.h -- k nearest neighbours of every point of the target among the target (icp.hpp:32,
// kdtree.hpp:65-78), all-pairs MFMA engine, round 3: the rows come WITH a bound, so the coarse pass writes
// lists instead of minima.
//
// Round 2's form kept every slot minimum of every row (2 B x 32 x splits per row: 314 MB per 100k rows,
// rows in chunks from 300k targets) and found the bound from them afterwards; its resolve was ~1,400
// instructions per row, most of them loading, bounding and filtering those minima.  Here the rows are
// taken in the target's MORTON order, where a row's neighbours in the array are neighbours in space:
//   k_knn_prebound      T(row) = the k-th smallest of 32 disjoint groups of exact distances to the 256 sorted
//                       positions around the row: k different targets within T, hence d_k <= T
//   k_nn_coarse_rows    the all-pairs pass, MODE 2 epilogue: per (row, split, half) one word with the columns
//                       whose minimum is <= tau_s(T(row)), the bound on the coarse value of any target of split s
//                       within T (nn_mfma.h), appended to the row's list -- a few words per row (148 B per row with
//                       bound and count, whatever the target's size)
//   k_knn_resolve_lists exact fp64 scan of the listed slots, candidates with distance <= T ranked by
//                       (distance, original index), the k smallest written closest first
// Every true k-neighbour t of a row has exact distance <= d_k <= T, so its coarse value, and with it its
// slot's minimum, is <= tau_s(T): its slot is listed, t is scanned exactly and ranked.  The result is the
// exact list whatever the bound was -- bit-identical to round 2's and to the oracle's.  Rows whose list
// does not fit (kKnnEntCap words, kKnnSlotCap slots, kKnnCap candidates after tightening: hundreds of
// coincident points, NaN coordinates) go to k_knn_exact_rows as before.
#pragma once
#include "nn_mfma.h"

namespace icpmi {

#ifndef ICPMI_KNN_LIST_CAP
#define ICPMI_KNN_LIST_CAP 128 /* 256 -> 128: 12 KB of LDS per workgroup instead of 23.5, eight waves per SIMD instead of six: 114.7 -> 108.2 us per 100k rows; a row with more candidates within its bound tightens the bound from those it holds and goes round again */
#endif
constexpr int kKnnListCap = ICPMI_KNN_LIST_CAP; // candidates per row held in LDS by k_knn_resolve_lists (more: the bound is tightened and the row redone)
constexpr int kKnnSlotCap = 192; // listed slots scanned per row
constexpr int kKnnWindow = 256;  // sorted positions around a row that bound its k-th neighbour

// Ascending bitonic sort of one value per lane within each HALF of the wave (15 compare-exchange steps).
__device__ __forceinline__ float half_sort_asc(float v, int hl)
{
#pragma unroll
    for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = lane_xor_n(v, j);
            const bool up = (hl & k) == 0, lower = (hl & j) == 0;
            const float mn = o < v ? o : v, mx = o < v ? v : o;
            v = (lower == up) ? mn : mx;
        }
    }
    return v;
}

// rows = sorted positions row0 .. row0 + nrows; outputs indexed from the launch's first row.  A workgroup takes 64
// consecutive rows: their windows overlap almost entirely, so the union (at most 320 positions) is staged in LDS once
// -- as one load per (row, position) the kernel pulled 600 MB through the L2 per 100k rows.  One row per HALF-wave
// (k <= 32): lane l of a half takes positions w0 + l + 32 c -- different targets in different lanes -- and the k-th
// smallest of the 32 lane minima comes out of a 15-step sort shared by two rows; the coarse-value bounds of the 64
// rows are formed at the end, one row per lane.
constexpr int kPreRows = 64;
__global__ __launch_bounds__(256) void k_knn_prebound(const double *__restrict__ sorted, int m, int ms, int k, int row0,
                                                      int nrows, double *__restrict__ t_row, float *__restrict__ tf_row,
                                                      float *__restrict__ sqf_row, int *__restrict__ cnt_row)
{
    __shared__ double wx[kPreRows + kKnnWindow], wy[kPreRows + kKnnWindow], wz[kPreRows + kKnnWindow];
    __shared__ double tl[kPreRows];
    const int lane = threadIdx.x & 63, hl = lane & 31, wave = threadIdx.x >> 6;
    const int first = row0 + blockIdx.x * kPreRows;                 // the workgroup's first row (sorted position)
    const int last = min(first + kPreRows, row0 + nrows) - 1;       // ... and its last
    auto window = [&](int r) -> int { // first position of row r's window
        int w0 = r - kKnnWindow / 2;
        w0 = w0 + kKnnWindow > m ? m - kKnnWindow : w0;
        return w0 < 0 ? 0 : w0;
    };
    const int lo = window(first), hi = min(window(last) + kKnnWindow, m); // (monotone in r: the union is [lo, hi))
    for (int t = threadIdx.x; t < hi - lo; t += 256) {
        wx[t] = ICPMI_SX(sorted, ms, lo + t);
        wy[t] = ICPMI_SY(sorted, ms, lo + t);
        wz[t] = ICPMI_SZ(sorted, ms, lo + t);
    }
    __syncthreads();
    const double kInf = 1.7976931348623157e308;
    const int kk = k < 32 ? k : 32;
#pragma unroll 1
    for (int it = 0; it < kPreRows / 8; ++it) {
        const int rl = wave * (kPreRows / 4) + it * 2 + (lane >> 5); // row within the workgroup
        const int r = first + rl <= last ? first + rl : last;
        const double px = wx[r - lo], py = wy[r - lo], pz = wz[r - lo]; // (a row lies inside its own window)
        const int off = window(r) - lo + hl;
        double lbest = kInf;
#pragma unroll
        for (int c = 0; c < kKnnWindow / 32; ++c) {
            const int j = off + 32 * c;
            const bool in = lo + j < hi; // (only a target of fewer than 256 points has a shorter window)
            const double d = sqdist(wx[in ? j : 0], wy[in ? j : 0], wz[in ? j : 0], px, py, pz);
            lbest = (in && d < lbest) ? d : lbest; // (a NaN distance never enters)
        }
        // k-th smallest of the 32 lane minima, as fp32 rounded UP (an upper bound stays one)
        float lbf = (float)lbest;
        lbf = (double)lbf < lbest ? __uint_as_float(__float_as_uint(lbf) + 1u) : lbf;
        const float tk = __shfl(half_sort_asc(lbf, hl), (lane & 32) + kk - 1, 64);
        if (hl == 0) tl[rl] = (double)tk;
    }
    __syncthreads();
    if (threadIdx.x < kPreRows && first + (int)threadIdx.x <= last) {
        const int r = first + threadIdx.x, local = r - row0;
        const double T = tl[threadIdx.x];
        // fewer than k finite lane minima (tiny or mostly non-finite targets, a NaN row): everything is listed
        const bool open = !(T < 1.0e299);
        t_row[local] = open ? __builtin_inf() : T;
        // fp32 images for the coarse pass's per-split thresholds (MODE 2, nn_mfma.h), rounded up; +Inf lists everything
        const float tf = open ? __builtin_inff() : (float)T; // (T came from an fp32 value: exact)
        float sq = __builtin_amdgcn_sqrtf(tf);
        sq = sq < 3.0e38f ? __uint_as_float(__float_as_uint(sq) + 2u) : sq; // (1 ulp of v_sqrt_f32 and one more)
        tf_row[local] = tf;
        sqf_row[local] = sq;
        cnt_row[local] = 0;
    }
}

// The bound for ARBITRARY query points (icpmi_k_nearest): a query has no place in the sorted target, so it is given one --
// its Morton key in the target's frame, looked up in the sorted keys -- and bounded by the 256 sorted positions around it
// like a row of the target.  One query per half-wave (k <= 32), the window read from memory (neighbouring queries need
// not be neighbours in space).
__global__ __launch_bounds__(256) void k_knn_prebound_q(const double *__restrict__ qry, int q0, int nq,
                                                        const double *__restrict__ sorted, const unsigned *__restrict__ keys_sorted,
                                                        int m, int ms, int k, const NnFrame *__restrict__ frame,
                                                        double *__restrict__ t_row, float *__restrict__ tf_row,
                                                        float *__restrict__ sqf_row, int *__restrict__ cnt_row)
{
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const int local = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool active = local < nq;
    const size_t i = (size_t)q0 + (active ? local : nq - 1);
    const double px = qry[3 * i], py = qry[3 * i + 1], pz = qry[3 * i + 2];
    // the query's Morton key (k_morton_keys' arithmetic) and its place among the sorted keys (first key >= it)
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
    while (lo_p < hi_p) { // (the same trips in every lane of the half: broadcast loads)
        const int mid = (lo_p + hi_p) >> 1;
        if (keys_sorted[mid] < key) lo_p = mid + 1;
        else hi_p = mid;
    }
    int w0 = lo_p - kKnnWindow / 2;
    w0 = w0 + kKnnWindow > m ? m - kKnnWindow : w0;
    w0 = w0 < 0 ? 0 : w0;
    const double kInf = 1.7976931348623157e308;
    double lbest = kInf;
#pragma unroll
    for (int c = 0; c < kKnnWindow / 32; ++c) {
        const int j = w0 + 32 * c + hl;
        const int jc = j < m ? j : m - 1;
        const double d = sqdist(ICPMI_SX(sorted, ms, jc), ICPMI_SY(sorted, ms, jc), ICPMI_SZ(sorted, ms, jc), px, py, pz);
        lbest = (j < m && d < lbest) ? d : lbest; // (a NaN distance never enters)
    }
    float lbf = (float)lbest;
    lbf = (double)lbf < lbest ? __uint_as_float(__float_as_uint(lbf) + 1u) : lbf;
    const int kk = k < 32 ? k : 32;
    const float tk = __shfl(half_sort_asc(lbf, hl), (lane & 32) + kk - 1, 64);
    if (hl == 0 && active) {
        const double T = (double)tk;
        const bool open = !(T < 1.0e299); // fewer than k finite lane minima, a NaN query: everything is listed
        t_row[local] = open ? __builtin_inf() : T;
        const float tf = open ? __builtin_inff() : tk;
        float sq = __builtin_amdgcn_sqrtf(tf);
        sq = sq < 3.0e38f ? __uint_as_float(__float_as_uint(sq) + 2u) : sq;
        tf_row[local] = tf;
        sqf_row[local] = sq;
        cnt_row[local] = 0;
    }
}

// QROWS: the rows are arbitrary query points (`qry`, N x 3; icpmi_k_nearest) instead of sorted positions of the target
// itself; list i then belongs to query i.
template <bool QROWS>
__global__ __launch_bounds__(256) void k_knn_resolve_lists(const double *__restrict__ qry, const double *__restrict__ sorted,
                                                           const unsigned *__restrict__ perm, int m, int ms, int k,
                                                           int row0, int nrows, const double *__restrict__ t_row,
                                                           const int *__restrict__ cnt_row,
                                                           const unsigned *__restrict__ ent_row,
                                                           int *__restrict__ knn_idx /*[m][k], by sorted position*/,
                                                           int *__restrict__ fb_list, int *__restrict__ fb_count)
{
    __shared__ double cand_d[4][kKnnListCap];
    __shared__ int cand_j[4][kKnnListCap];
    __shared__ int cand_r[4][kKnnListCap], owner[4][kKnnListCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave); // scalar: the row's words are read by scalar loads
    if (local >= nrows) return;
    const int i = row0 + local;
    const int cnt = cnt_row[local];
    double T = t_row[local];
    const unsigned e = lane < cnt && lane < kKnnEntCap ? ent_row[(size_t)local * kKnnEntCap + lane] : 0u;
    const double px = QROWS ? qry[3 * (size_t)i] : ICPMI_SX(sorted, ms, i), py = QROWS ? qry[3 * (size_t)i + 1] : ICPMI_SY(sorted, ms, i),
                 pz = QROWS ? qry[3 * (size_t)i + 2] : ICPMI_SZ(sorted, ms, i);
    // The listed slots are walked word by word, bit by bit, with SCALAR instructions (a word is read out of its lane;
    // which slot comes next is the same for the whole wave).  Any order: the ranking below is exact.
    int nf = 0;
    for (int w = 0; w < cnt && w < kKnnEntCap; ++w) nf += __popc((unsigned)__builtin_amdgcn_readlane((int)e, w) & 0xFFFFu);
    // no word at all (a sane row lists its own slot: NaN coordinates), more words or slots than fit: exact kernel
    if (cnt <= 0 || cnt > kKnnEntCap || nf > kKnnSlotCap) {
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    const int kk = k < 64 ? k : 64;
    const double kInf = 1.7976931348623157e308;
    int total = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        total = 0;
        int w = 0, base = 0;
        unsigned msk = 0u;
        auto next_slot = [&]() -> int { // -1: none left
            while (msk == 0u && w < cnt) {
                const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)e, w++);
                msk = word & 0xFFFFu;
                base = (int)(word >> 17) * kCols + (int)((word >> 16) & 1u) * 16;
            }
            if (msk == 0u) return -1;
            const int b = __ffs((int)msk) - 1;
            msk &= msk - 1u;
            return base + b;
        };
#pragma unroll 1
        while (total <= kKnnListCap) {
            constexpr int kRuns = kSlotTargets / 64;
            int se[kKnnBatch];
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q) se[q] = next_slot();
            if (se[0] < 0) break;
            double d[kKnnBatch][kRuns];
            int jj[kKnnBatch][kRuns];
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q) {
                const int sq = se[q] < 0 ? se[0] : se[q];
                const int j0 = (sq / kCols) * kSplitTargets + (sq % kCols) * kSlotTargets;
#pragma unroll
                for (int o = 0; o < kRuns; ++o) {
                    // (clamped: a row that lists everything -- no finite bound -- lists the padding slots behind the target's
                    // last point too, whose positions lie outside the sorted copy)
                    jj[q][o] = j0 + 64 * o + lane;
                    const int jc = jj[q][o] < m ? jj[q][o] : m - 1;
                    d[q][o] = sqdist(ICPMI_SX(sorted, ms, jc), ICPMI_SY(sorted, ms, jc), ICPMI_SZ(sorted, ms, jc), px, py, pz);
                    if (!(se[q] >= 0 && jj[q][o] < m)) d[q][o] = __builtin_nan(""); // never kept
                }
            }
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q)
#pragma unroll
                for (int o = 0; o < kRuns; ++o) {
                    const bool keep = d[q][o] <= T;
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int pos = total + __popcll(km & ((1ull << lane) - 1ull));
                        if (pos < kKnnListCap) {
                            cand_d[wave][pos] = d[q][o];
                            cand_j[wave][pos] = jj[q][o]; // sorted position (knn_rank_write translates the winners)
                        }
                    }
                    total += __popcll(km);
                }
        }
        if (total <= kKnnListCap) break;
        // more than fit: the k-th smallest of the 64 lanes' minima over the candidates held (64 groups of different
        // targets) is a tighter valid bound; collect again with it
        __builtin_amdgcn_wave_barrier();
        double lm = kInf;
        for (int c = lane; c < kKnnListCap; c += 64) {
            const double dd = cand_d[wave][c];
            lm = dd < lm ? dd : lm;
        }
        const double tnew = __shfl(wave_sort_asc(lm, lane), kk - 1, 64);
        __builtin_amdgcn_wave_barrier();
        if (!(tnew < T)) break; // cannot tighten (e.g. hundreds of coincident points)
        T = tnew;
    }
    if (total > kKnnListCap) {
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    knn_rank_write(cand_d[wave], cand_j[wave], cand_r[wave], owner[wave], total, k, lane, knn_idx + (size_t)i * k, perm);
}

} // namespace icpmi

// knn_lists.h -- k nearest neighbours of every point of the target among the target (icp.hpp:32,
// kdtree.hpp:65-78), all-pairs MFMA engine, round 3: the rows come WITH a bound, so the coarse pass writes
// lists instead of minima.
//
// Round 2's form kept every slot minimum of every row (2 B x 32 x splits per row: 314 MB per 100k rows,
// rows in chunks from 300k targets) and found the bound from them afterwards; its resolve was ~1,400
// instructions per row, most of them loading, bounding and filtering those minima.  Here the rows are
// taken in the target's MORTON order, where a row's neighbours in the array are neighbours in space:
//   k_knn_prebound      T(row) = the k-th smallest of 64 disjoint groups of exact distances to the 256 sorted
//                       positions around the row: k different targets within T, hence d_k <= T; and
//                       thr(row) = the bound on the coarse value of any target within T that holds for EVERY
//                       split (all_splits_tau, nn_mfma.h)
//   k_nn_coarse_rows    the all-pairs pass, MODE 2 epilogue: per (row, split, half) one word with the columns
//                       whose minimum is <= thr(row), appended to the row's list -- a few words per row
//                       (144 B per row with bound and count, whatever the target's size)
//   k_knn_resolve_lists exact fp64 scan of the listed slots, candidates with distance <= T ranked by
//                       (distance, original index), the k smallest written closest first
// Every true k-neighbour t of a row has exact distance <= d_k <= T, so its coarse value, and with it its
// slot's minimum, is <= thr(row): its slot is listed, t is scanned exactly and ranked.  The result is the
// exact list whatever the bound was -- bit-identical to round 2's and to the oracle's.  Rows whose list
// does not fit (kKnnEntCap words, kKnnSlotCap slots, kKnnCap candidates after tightening: hundreds of
// coincident points, NaN coordinates) go to k_knn_exact_rows as before.
#pragma once
#include "nn_mfma.h"

namespace icpmi {

constexpr int kKnnSlotCap = 192; // listed slots scanned per row
constexpr int kKnnWindow = 256;  // sorted positions around a row that bound its k-th neighbour

// rows = sorted positions row0 .. row0 + nrows; outputs indexed from the launch's first row
__global__ __launch_bounds__(256) void k_knn_prebound(const double *__restrict__ sorted, int m, int ms, int k, int row0,
                                                      int nrows, const NnFrame *__restrict__ gframe,
                                                      double *__restrict__ t_row, float *__restrict__ thr_row,
                                                      int *__restrict__ cnt_row)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = blockIdx.x * 4 + wave;
    if (local >= nrows) return; // wave-uniform
    const int r = row0 + local;
    const double px = ICPMI_SX(sorted, ms, r), py = ICPMI_SY(sorted, ms, r), pz = ICPMI_SZ(sorted, ms, r);
    int w0 = r - kKnnWindow / 2;
    w0 = w0 + kKnnWindow > m ? m - kKnnWindow : w0;
    w0 = w0 < 0 ? 0 : w0;
    const double kInf = 1.7976931348623157e308;
    double lbest = kInf;
#pragma unroll
    for (int c = 0; c < kKnnWindow / 64; ++c) { // lane l: positions w0 + l + 64 c -- different targets in different lanes
        const int j = w0 + 64 * c + lane;
        const int jc = j < m ? j : m - 1;
        const double d = sqdist(ICPMI_SX(sorted, ms, jc), ICPMI_SY(sorted, ms, jc), ICPMI_SZ(sorted, ms, jc), px, py, pz);
        lbest = (j < m && d < lbest) ? d : lbest; // (a NaN distance never enters)
    }
    // k-th smallest of the 64 lane minima, as fp32 rounded UP (an upper bound stays one)
    float lbf = (float)lbest;
    lbf = (double)lbf < lbest ? __uint_as_float(__float_as_uint(lbf) + 1u) : lbf;
    const int kk = k < 64 ? k : 64;
    const double T = (double)__shfl(wave_sort_asc(lbf, lane), kk - 1, 64);
    if (lane == 0) {
        // fewer than k finite lane minima (tiny or mostly non-finite targets, a NaN row): everything is listed
        const bool open = !(T < 1.0e299);
        t_row[local] = open ? __builtin_inf() : T;
        thr_row[local] = open ? 3.4028235e38f : all_splits_tau(px, py, pz, *gframe, T, sqrt(T));
        cnt_row[local] = 0;
    }
}

__global__ __launch_bounds__(256) void k_knn_resolve_lists(const double *__restrict__ sorted,
                                                           const unsigned *__restrict__ perm, int m, int ms, int k,
                                                           int row0, int nrows, const double *__restrict__ t_row,
                                                           const int *__restrict__ cnt_row,
                                                           const unsigned *__restrict__ ent_row,
                                                           int *__restrict__ knn_idx /*[m][k], by sorted position*/,
                                                           int *__restrict__ fb_list, int *__restrict__ fb_count)
{
    __shared__ double cand_d[4][kKnnCap];
    __shared__ int cand_j[4][kKnnCap];
    __shared__ int cand_r[4][kKnnCap], owner[4][kKnnCap];
    __shared__ int flist[4][kKnnSlotCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = blockIdx.x * 4 + wave;
    if (local >= nrows) return; // wave-uniform
    const int i = row0 + local;
    const int cnt = cnt_row[local];
    double T = t_row[local];
    const unsigned e = lane < cnt && lane < kKnnEntCap ? ent_row[(size_t)local * kKnnEntCap + lane] : 0u;
    const double px = ICPMI_SX(sorted, ms, i), py = ICPMI_SY(sorted, ms, i), pz = ICPMI_SZ(sorted, ms, i);
    // the listed slots, one word's columns after the other (any order: the ranking below is exact)
    unsigned mask = e & 0xFFFFu;
    const int mine = __popc(mask);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < kKnnEntCap; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        incl += lane >= off ? o : 0;
    }
    const int nf = __shfl(incl, kKnnEntCap - 1, 64);
    // no word at all (a sane row lists its own slot: NaN coordinates), more words or slots than fit: exact kernel
    if (cnt <= 0 || cnt > kKnnEntCap || nf > kKnnSlotCap) {
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    {
        int pos = incl - mine;
        const int base = (int)(e >> 17) * kCols + (int)((e >> 16) & 1u) * 16;
        while (mask) {
            flist[wave][pos++] = base + __ffs((int)mask) - 1;
            mask &= mask - 1u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int kk = k < 64 ? k : 64;
    const double kInf = 1.7976931348623157e308;
    int total = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        total = 0;
#pragma unroll 1
        for (int f0 = 0; f0 < nf && total <= kKnnCap; f0 += kKnnBatch) {
            constexpr int kRuns = kSlotTargets / 64;
            double d[kKnnBatch][kRuns];
            int jj[kKnnBatch][kRuns];
#pragma unroll
            for (int q = 0; q < kKnnBatch; ++q) {
                const int f = f0 + q;
                const int se = flist[wave][f < nf ? f : nf - 1];
                const int j0 = (se / kCols) * kSplitTargets + (se % kCols) * kSlotTargets;
#pragma unroll
                for (int o = 0; o < kRuns; ++o) {
                    jj[q][o] = j0 + 64 * o + lane;
                    const int jc = jj[q][o] < m ? jj[q][o] : m - 1;
                    d[q][o] = sqdist(ICPMI_SX(sorted, ms, jc), ICPMI_SY(sorted, ms, jc), ICPMI_SZ(sorted, ms, jc), px, py, pz);
                    if (!(f < nf && jj[q][o] < m)) d[q][o] = __builtin_nan(""); // never kept
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
                        if (pos < kKnnCap) {
                            cand_d[wave][pos] = d[q][o];
                            cand_j[wave][pos] = (int)perm[jj[q][o]];
                        }
                    }
                    total += __popcll(km);
                }
        }
        if (total <= kKnnCap) break;
        // more than fit: the k-th smallest of the 64 lanes' minima over the candidates held (64 groups of different
        // targets) is a tighter valid bound; collect again with it
        __builtin_amdgcn_wave_barrier();
        double lm = kInf;
        for (int c = lane; c < kKnnCap; c += 64) {
            const double dd = cand_d[wave][c];
            lm = dd < lm ? dd : lm;
        }
        const double tnew = __shfl(wave_sort_asc(lm, lane), kk - 1, 64);
        __builtin_amdgcn_wave_barrier();
        if (!(tnew < T)) break; // cannot tighten (e.g. hundreds of coincident points)
        T = tnew;
    }
    if (total > kKnnCap) {
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = i;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    knn_rank_write(cand_d[wave], cand_j[wave], cand_r[wave], owner[wave], total, k, lane, knn_idx + (size_t)i * k);
}

} // namespace icpmi

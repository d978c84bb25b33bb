/*
 * icp_oracle.c -- CPU restatement of the reference's point-to-plane ICP path.
 *
 * TEST INFRASTRUCTURE ONLY (see icp_oracle.h).  PARITY UNPINNED: no reference
 * fixture exists for this path and the reference cannot be built here.
 *
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/slam_viz/include/slam_viz/core/.  Where the reference hands
 * the arithmetic to Eigen 3.4.0 (unvendored dependency, CMakeLists.txt:25) the
 * published algorithm of that Eigen routine is restated and named.
 */
#define _POSIX_C_SOURCE 200809L
#include "icp_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* KD-tree (kdtree.hpp:18-186)                                               */
/* ------------------------------------------------------------------------- */

typedef struct {
    int index; /* kdtree.hpp:82 */
    int left;  /* kdtree.hpp:83 */
    int right; /* kdtree.hpp:84 */
} orc_node;

struct orc_kdtree {
    double *points; /* private copy, kdtree.hpp:20,182 */
    int *indices;   /* kdtree.hpp:183 */
    orc_node *nodes;/* kdtree.hpp:184, pushed post-order */
    int n_nodes;
    int n;
    int root;       /* kdtree.hpp:185 */
};

/* Quickselect standing in for std::nth_element (kdtree.hpp:94-101): afterwards
 * idx[mid] holds the element a full sort would put there, everything before it
 * compares <= and everything after >= on `axis`.  The tree's shape may differ
 * from libstdc++'s introselect; the searches below are exact for any valid
 * median split, so results only differ on exact fp64 distance ties. */
static void select_nth(const double *pts, int *idx, int lo, int hi, int nth, int axis)
{
    while (hi - lo > 1) {
        int a = idx[lo], b = idx[lo + (hi - lo) / 2], c = idx[hi - 1];
        double va = pts[3 * a + axis], vb = pts[3 * b + axis], vc = pts[3 * c + axis];
        double pivot = va < vb ? (vb < vc ? vb : (va < vc ? vc : va))
                               : (va < vc ? va : (vb < vc ? vc : vb));
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (pts[3 * idx[i] + axis] < pivot) ++i;
            while (pts[3 * idx[j] + axis] > pivot) --j;
            if (i <= j) {
                int t = idx[i];
                idx[i] = idx[j];
                idx[j] = t;
                ++i;
                --j;
            }
        }
        /* [lo, j] <= pivot, [i, hi) >= pivot, (j, i) == pivot */
        if (nth <= j)
            hi = j + 1;
        else if (nth >= i)
            lo = i;
        else
            return;
    }
}

/* kdtree.hpp:87-110 */
static int build_rec(orc_kdtree *t, int start, int end, int depth)
{
    if (start >= end) return -1;
    int axis = depth % 3;
    int mid = (start + end) / 2;
    select_nth(t->points, t->indices, start, end, mid, axis);
    orc_node node;
    node.index = t->indices[mid];
    node.left = build_rec(t, start, mid, depth + 1);
    node.right = build_rec(t, mid + 1, end, depth + 1);
    t->nodes[t->n_nodes] = node;
    return t->n_nodes++;
}

orc_kdtree *orc_kdtree_build(const double *points_xyz, int n)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(*t));
    if (!t) return NULL;
    t->n = n;
    t->points = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    t->indices = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    t->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)(n > 0 ? n : 1));
    memcpy(t->points, points_xyz, sizeof(double) * 3 * (size_t)n);
    for (int i = 0; i < n; ++i) t->indices[i] = i; /* kdtree.hpp:21-24 */
    t->n_nodes = 0;
    t->root = build_rec(t, 0, n, 0);               /* kdtree.hpp:25 */
    return t;
}

void orc_kdtree_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->points);
    free(t->indices);
    free(t->nodes);
    free(t);
}

/* (point - query).squaredNorm(), kdtree.hpp:124,156: (x^2 + y^2) + z^2. */
static inline double sqdist3(const double *p, const double *q)
{
    double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return (dx * dx + dy * dy) + dz * dz;
}

/* kdtree.hpp:112-142 */
static void search_nearest(const orc_kdtree *t, int node_idx, const double *query, int depth,
                           int *best_idx, double *best_dist_sq)
{
    if (node_idx < 0) return;
    const orc_node *node = &t->nodes[node_idx];
    const double *point = &t->points[3 * node->index];
    double dist_sq = sqdist3(point, query);
    if (dist_sq < *best_dist_sq) { /* strict, kdtree.hpp:125 */
        *best_dist_sq = dist_sq;
        *best_idx = node->index;
    }
    int axis = depth % 3;
    double diff = query[axis] - point[axis];
    int first = diff < 0 ? node->left : node->right;
    int second = diff < 0 ? node->right : node->left;
    search_nearest(t, first, query, depth + 1, best_idx, best_dist_sq);
    if (diff * diff < *best_dist_sq) /* kdtree.hpp:139 */
        search_nearest(t, second, query, depth + 1, best_idx, best_dist_sq);
}

typedef struct {
    const orc_kdtree *t;
    const double *q;
    int begin, end;
    int *idx;
    double *d2;
} nn_job;

static void *nn_worker(void *arg)
{
    nn_job *j = (nn_job *)arg;
    for (int i = j->begin; i < j->end; ++i) { /* kdtree.hpp:51-58 */
        int best_idx = -1;
        double best = DBL_MAX;
        search_nearest(j->t, j->t->root, &j->q[3 * i], 0, &best_idx, &best);
        j->idx[i] = best_idx;
        if (j->d2) j->d2[i] = best;
    }
    return NULL;
}

#define ORC_MAX_THREADS 256

void orc_nearest_batch(const orc_kdtree *t, const double *queries_xyz, int nq, int *indices,
                       double *dist_sq, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
    if (nthreads == 1 || nq < 4 * nthreads) {
        nn_job j = {t, queries_xyz, 0, nq, indices, dist_sq};
        nn_worker(&j);
        return;
    }
    pthread_t th[ORC_MAX_THREADS];
    nn_job jobs[ORC_MAX_THREADS];
    for (int k = 0; k < nthreads; ++k) {
        jobs[k].t = t;
        jobs[k].q = queries_xyz;
        jobs[k].begin = (int)((long long)nq * k / nthreads);
        jobs[k].end = (int)((long long)nq * (k + 1) / nthreads);
        jobs[k].idx = indices;
        jobs[k].d2 = dist_sq;
        pthread_create(&th[k], NULL, nn_worker, &jobs[k]);
    }
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
}

/* std::priority_queue<std::pair<double,int>> (kdtree.hpp:67,149): a max-heap
 * ordered by (distance, index) lexicographically. */
typedef struct {
    double d;
    int i;
} heap_ent;

static inline int ent_less(heap_ent a, heap_ent b)
{
    return a.d < b.d || (!(b.d < a.d) && a.i < b.i);
}

static void heap_push(heap_ent *h, int *n, heap_ent e)
{
    int c = (*n)++;
    while (c > 0) {
        int p = (c - 1) / 2;
        if (!ent_less(h[p], e)) break;
        h[c] = h[p];
        c = p;
    }
    h[c] = e;
}

static void heap_pop(heap_ent *h, int *n)
{
    heap_ent e = h[--(*n)];
    int p = 0, size = *n;
    for (;;) {
        int c = 2 * p + 1;
        if (c >= size) break;
        if (c + 1 < size && ent_less(h[c], h[c + 1])) ++c;
        if (!ent_less(e, h[c])) break;
        h[p] = h[c];
        p = c;
    }
    if (size > 0) h[p] = e;
}

/* kdtree.hpp:144-180 */
static void search_k_nearest(const orc_kdtree *t, int node_idx, const double *query, int depth,
                             int k, heap_ent *heap, int *hn)
{
    if (node_idx < 0) return;
    const orc_node *node = &t->nodes[node_idx];
    const double *point = &t->points[3 * node->index];
    double dist_sq = sqdist3(point, query);
    if (*hn < k) {
        heap_ent e = {dist_sq, node->index};
        heap_push(heap, hn, e);
    } else if (dist_sq < heap[0].d) { /* strict, kdtree.hpp:160 */
        heap_pop(heap, hn);
        heap_ent e = {dist_sq, node->index};
        heap_push(heap, hn, e);
    }
    int axis = depth % 3;
    double diff = query[axis] - point[axis];
    int first = diff < 0 ? node->left : node->right;
    int second = diff < 0 ? node->right : node->left;
    search_k_nearest(t, first, query, depth + 1, k, heap, hn);
    double threshold = *hn < k ? DBL_MAX : heap[0].d; /* kdtree.hpp:173-175 */
    if (diff * diff < threshold) search_k_nearest(t, second, query, depth + 1, k, heap, hn);
}

/* kdtree.hpp:65-78: pop everything (largest first) then reverse. */
static int drain_heap(heap_ent *heap, int hn, int *out_idx)
{
    int count = hn;
    for (int pos = count - 1; pos >= 0; --pos) {
        out_idx[pos] = heap[0].i;
        heap_pop(heap, &hn);
    }
    return count;
}

int orc_k_nearest(const orc_kdtree *t, const double query[3], int k, int *out_idx)
{
    if (k <= 0) return 0;
    heap_ent *heap = (heap_ent *)malloc(sizeof(heap_ent) * (size_t)(k + 1));
    int hn = 0;
    search_k_nearest(t, t->root, query, 0, k, heap, &hn);
    int count = drain_heap(heap, hn, out_idx);
    free(heap);
    return count;
}

void orc_nearest_batch_brute(const double *targets_xyz, int m, const double *queries_xyz,
                             int nq, int *indices, double *dist_sq)
{
    for (int i = 0; i < nq; ++i) {
        int best_idx = -1;
        double best = DBL_MAX;
        for (int j = 0; j < m; ++j) {
            double d = sqdist3(&targets_xyz[3 * j], &queries_xyz[3 * i]);
            if (d < best) {
                best = d;
                best_idx = j;
            }
        }
        indices[i] = best_idx;
        if (dist_sq) dist_sq[i] = best;
    }
}

int orc_k_nearest_brute(const double *targets_xyz, int m, const double query[3], int k,
                        int *out_idx)
{
    if (k <= 0) return 0;
    heap_ent *heap = (heap_ent *)malloc(sizeof(heap_ent) * (size_t)(k + 1));
    int hn = 0;
    for (int j = 0; j < m; ++j) {
        heap_ent e = {sqdist3(&targets_xyz[3 * j], query), j};
        if (hn < k) {
            heap_push(heap, &hn, e);
        } else if (ent_less(e, heap[0])) {
            heap_pop(heap, &hn);
            heap_push(heap, &hn, e);
        }
    }
    int count = drain_heap(heap, hn, out_idx);
    free(heap);
    return count;
}

/* ------------------------------------------------------------------------- */
/* 3x3 symmetric eigenvector (icp.hpp:55-56)                                 */
/* ------------------------------------------------------------------------- */

/* The reference calls Eigen::SelfAdjointEigenSolver<Matrix3d>(cov) and takes
 * eigenvectors().col(0): the unit eigenvector of the smallest eigenvalue.
 * Eigen 3.4.0 reaches it by tridiagonalisation + implicit symmetric QR; this
 * restatement uses the cyclic Jacobi method instead (same mathematical result,
 * agreement to rounding, sign free -- the caller fixes the sign).  Only + - * /
 * sqrt and comparisons are used so the HIP kernel can be bit-identical. */
void orc_smallest_eigenvector(const double cov[9], double v[3])
{
    double a00 = cov[0], a01 = cov[1], a02 = cov[2];
    double a11 = cov[4], a12 = cov[5], a22 = cov[8];
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};

    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p) {
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[p][q];
                double g = 100.0 * fabs(apq);
                if (sweep > 3 && fabs(A[p][p]) + g == fabs(A[p][p]) &&
                    fabs(A[q][q]) + g == fabs(A[q][q])) {
                    A[p][q] = 0.0;
                    A[q][p] = 0.0;
                    continue;
                }
                if (apq == 0.0) continue;
                double h = A[q][q] - A[p][p];
                double t;
                if (fabs(h) + g == fabs(h)) {
                    t = apq / h;
                } else {
                    double theta = 0.5 * h / apq;
                    t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                    if (theta < 0.0) t = -t;
                }
                double c = 1.0 / sqrt(1.0 + t * t);
                double s = t * c;
                int r = 3 - p - q; /* the third index */
                double arp = A[r][p], arq = A[r][q];
                A[p][p] = A[p][p] - t * apq;
                A[q][q] = A[q][q] + t * apq;
                A[p][q] = 0.0;
                A[q][p] = 0.0;
                A[r][p] = c * arp - s * arq;
                A[p][r] = A[r][p];
                A[r][q] = s * arp + c * arq;
                A[q][r] = A[r][q];
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    v[0] = V[0][m];
    v[1] = V[1][m];
    v[2] = V[2][m];
}

/* ------------------------------------------------------------------------- */
/* Normal estimation (icp.hpp:23-67)                                         */
/* ------------------------------------------------------------------------- */

static void normal_from_neighbors(const double *points, const int *nb, int count, double *out)
{
    if (count < 3) { /* icp.hpp:34-37 */
        out[0] = 0;
        out[1] = 0;
        out[2] = 1;
        return;
    }
    double cx = 0, cy = 0, cz = 0; /* icp.hpp:40-44 */
    for (int a = 0; a < count; ++a) {
        const double *p = &points[3 * nb[a]];
        cx += p[0];
        cy += p[1];
        cz += p[2];
    }
    double kd = (double)count;
    cx /= kd;
    cy /= kd;
    cz /= kd;
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* icp.hpp:47-52 */
    for (int a = 0; a < count; ++a) {
        const double *p = &points[3 * nb[a]];
        double d[3] = {p[0] - cx, p[1] - cy, p[2] - cz};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[3 * r + c] += d[r] * d[c];
    }
    for (int e = 0; e < 9; ++e) cov[e] /= kd;
    double nrm[3];
    orc_smallest_eigenvector(cov, nrm); /* icp.hpp:55-56 */
    if (nrm[2] < 0) {                   /* icp.hpp:59-61 */
        nrm[0] = -nrm[0];
        nrm[1] = -nrm[1];
        nrm[2] = -nrm[2];
    }
    /* icp.hpp:63, Eigen normalized(): v / sqrt(v.squaredNorm()) when > 0 */
    double z = (nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2];
    if (z > 0) {
        double s = sqrt(z);
        nrm[0] /= s;
        nrm[1] /= s;
        nrm[2] /= s;
    }
    out[0] = nrm[0];
    out[1] = nrm[1];
    out[2] = nrm[2];
}

typedef struct {
    const double *points;
    const orc_kdtree *t;
    int k, begin, end;
    double *normals;
} normal_job;

static void *normal_worker(void *arg)
{
    normal_job *j = (normal_job *)arg;
    int k = j->k;
    heap_ent *heap = (heap_ent *)malloc(sizeof(heap_ent) * (size_t)(k + 1));
    int *nb = (int *)malloc(sizeof(int) * (size_t)(k + 1));
    for (int i = j->begin; i < j->end; ++i) { /* icp.hpp:30-64 */
        int hn = 0;
        search_k_nearest(j->t, j->t->root, &j->points[3 * i], 0, k, heap, &hn);
        int count = drain_heap(heap, hn, nb);
        normal_from_neighbors(j->points, nb, count, &j->normals[3 * i]);
    }
    free(heap);
    free(nb);
    return NULL;
}

void orc_estimate_normals(const double *points_xyz, int m, const orc_kdtree *t, int k,
                          double *normals_xyz, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
    if (k < 1) k = 1;
    if (nthreads == 1 || m < 4 * nthreads) {
        normal_job j = {points_xyz, t, k, 0, m, normals_xyz};
        normal_worker(&j);
        return;
    }
    pthread_t th[ORC_MAX_THREADS];
    normal_job jobs[ORC_MAX_THREADS];
    for (int w = 0; w < nthreads; ++w) {
        jobs[w].points = points_xyz;
        jobs[w].t = t;
        jobs[w].k = k;
        jobs[w].begin = (int)((long long)m * w / nthreads);
        jobs[w].end = (int)((long long)m * (w + 1) / nthreads);
        jobs[w].normals = normals_xyz;
        pthread_create(&th[w], NULL, normal_worker, &jobs[w]);
    }
    for (int w = 0; w < nthreads; ++w) pthread_join(th[w], NULL);
}

/* the same for rows [row0, row1) only (icp.hpp:30-64 is a loop over independent rows): out_xyz
 * receives row1 - row0 rows.  What a test of a row slice needs without paying for the whole cloud. */
void orc_estimate_normals_rows(const double *points_xyz, int m, const orc_kdtree *t, int k, int row0,
                               int row1, double *out_xyz, int nthreads)
{
    if (row0 < 0) row0 = 0;
    if (row1 > m) row1 = m;
    const int rows = row1 - row0;
    if (rows <= 0) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
    if (k < 1) k = 1;
    if (rows < 4 * nthreads) nthreads = 1;
    pthread_t th[ORC_MAX_THREADS];
    normal_job jobs[ORC_MAX_THREADS];
    for (int w = 0; w < nthreads; ++w) {
        jobs[w].points = points_xyz;
        jobs[w].t = t;
        jobs[w].k = k;
        jobs[w].begin = row0 + (int)((long long)rows * w / nthreads);
        jobs[w].end = row0 + (int)((long long)rows * (w + 1) / nthreads);
        jobs[w].normals = out_xyz - 3 * (ptrdiff_t)row0; /* the worker writes row i at normals + 3 i */
        if (nthreads == 1) normal_worker(&jobs[w]);
        else pthread_create(&th[w], NULL, normal_worker, &jobs[w]);
    }
    if (nthreads > 1)
        for (int w = 0; w < nthreads; ++w) pthread_join(th[w], NULL);
}

/* ------------------------------------------------------------------------- */
/* Point-to-plane solve (icp.hpp:89-144)                                     */
/* ------------------------------------------------------------------------- */

void orc_normal_equations(const double *src, const double *tgt, const double *nrm, int n,
                          double out[28])
{
    double A[6][6];
    double g[6] = {0, 0, 0, 0, 0, 0};
    double bb = 0;
    memset(A, 0, sizeof(A));
    for (int i = 0; i < n; ++i) { /* icp.hpp:99-117 */
        const double *p = &src[3 * i], *q = &tgt[3 * i], *nn = &nrm[3 * i];
        double J[6];
        J[0] = p[1] * nn[2] - p[2] * nn[1]; /* p x n, icp.hpp:105 */
        J[1] = p[2] * nn[0] - p[0] * nn[2];
        J[2] = p[0] * nn[1] - p[1] * nn[0];
        J[3] = nn[0];
        J[4] = nn[1];
        J[5] = nn[2];
        double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        double b = (dx * nn[0] + dy * nn[1]) + dz * nn[2]; /* icp.hpp:116 */
        for (int r = 0; r < 6; ++r) {
            for (int c = r; c < 6; ++c) A[r][c] += J[r] * J[c];
            g[r] += J[r] * b;
        }
        bb += b * b;
    }
    int o = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) out[o++] = A[r][c];
    for (int r = 0; r < 6; ++r) out[21 + r] = g[r];
    out[27] = bb;
}

/* Eigen 3.4.0 LDLT<MatrixXd, Lower> (icp.hpp:120 `.ldlt().solve()`), restated from
 * the published algorithm: ldlt_inplace<Lower>::unblocked -- at step k pivot on the
 * largest |diagonal| of the trailing block (first maximum wins), symmetric swap,
 * A_kk -= A10 . (D .* A10), A21 -= A20 (D .* A10), A21 /= A_kk when A_kk != 0 --
 * and LDLT::_solve_impl -- P b, L solve, D pseudo-inverse with |D_i| <= DBL_MIN
 * treated as zero, L^T solve, P^T.  Inner sums run in index order; Eigen's
 * packetised order may differ in the last bit (unpinned). */
static void ldlt6_solve(double M[6][6], const double rhs[6], double x[6])
{
    const int n = 6;
    int tr[6];
    double temp[6];
    for (int k = 0; k < n; ++k) {
        int big = k;
        double bigv = fabs(M[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(M[i][i]) > bigv) {
                bigv = fabs(M[i][i]);
                big = i;
            }
        tr[k] = big;
        if (big != k) {
            for (int j = 0; j < k; ++j) {
                double t = M[k][j];
                M[k][j] = M[big][j];
                M[big][j] = t;
            }
            for (int i = big + 1; i < n; ++i) {
                double t = M[i][k];
                M[i][k] = M[i][big];
                M[i][big] = t;
            }
            {
                double t = M[k][k];
                M[k][k] = M[big][big];
                M[big][big] = t;
            }
            for (int i = k + 1; i < big; ++i) {
                double t = M[i][k];
                M[i][k] = M[big][i];
                M[big][i] = t;
            }
        }
        if (k > 0) {
            double acc = 0;
            for (int j = 0; j < k; ++j) {
                temp[j] = M[j][j] * M[k][j];
                acc += M[k][j] * temp[j];
            }
            M[k][k] -= acc;
            for (int i = k + 1; i < n; ++i) {
                double s = 0;
                for (int j = 0; j < k; ++j) s += M[i][j] * temp[j];
                M[i][k] -= s;
            }
        }
        double akk = M[k][k];
        int valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) {
            for (int j = 0; j < n; ++j) tr[j] = j;
            break;
        }
        if (valid)
            for (int i = k + 1; i < n; ++i) M[i][k] /= akk;
    }
    for (int i = 0; i < n; ++i) x[i] = rhs[i];
    for (int k = 0; k < n; ++k) {
        double t = x[k];
        x[k] = x[tr[k]];
        x[tr[k]] = t;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < i; ++j) s += M[i][j] * x[j];
        x[i] -= s;
    }
    for (int i = 0; i < n; ++i) {
        if (fabs(M[i][i]) > DBL_MIN)
            x[i] /= M[i][i];
        else
            x[i] = 0;
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = 0;
        for (int j = i + 1; j < n; ++j) s += M[j][i] * x[j];
        x[i] -= s;
    }
    for (int k = n - 1; k >= 0; --k) {
        double t = x[k];
        x[k] = x[tr[k]];
        x[tr[k]] = t;
    }
}

static void identity16(double T[16])
{
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

void orc_solve_from_sums(const double sums[28], double T[16])
{
    double M[6][6], rhs[6], x[6];
    int o = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
            M[r][c] = sums[o];
            M[c][r] = sums[o];
            ++o;
        }
    for (int r = 0; r < 6; ++r) rhs[r] = sums[21 + r];
    ldlt6_solve(M, rhs, x); /* icp.hpp:120 */

    /* icp.hpp:123-141 */
    double rx = x[0], ry = x[1], rz = x[2];
    double angle = sqrt((rx * rx + ry * ry) + rz * rz);
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (!(angle < 1e-10)) {
        double ax = rx / angle, ay = ry / angle, az = rz / angle;
        double K[3][3] = {{0, -az, ay}, {az, 0, -ax}, {-ay, ax, 0}};
        double s = sin(angle), c1 = 1 - cos(angle);
        /* I + sin*K + (1-cos)*K*K parses as (I + sin*K) + ((1-cos)*K)*K */
        double Mk[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Mk[i][j] = c1 * K[i][j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = (Mk[i][0] * K[0][j] + Mk[i][1] * K[1][j]) + Mk[i][2] * K[2][j];
                R[i][j] = (R[i][j] + s * K[i][j]) + kk;
            }
    }
    identity16(T); /* types.hpp:84-88 */
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[i][j];
        T[4 * i + 3] = x[3 + i];
    }
}

void orc_solve_point_to_plane(const double *src, const double *tgt, const double *nrm, int n,
                              double T[16])
{
    double sums[28];
    orc_normal_equations(src, tgt, nrm, n, sums);
    orc_solve_from_sums(sums, T);
}

/* ------------------------------------------------------------------------- */
/* Voxel downsampling (src/core/file_utils.cpp:148-196)                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    long long x, y, z; /* file_utils.cpp:156 */
    int index;
} voxel_ent;

static int voxel_cmp(const void *a, const void *b)
{
    const voxel_ent *p = (const voxel_ent *)a, *q = (const voxel_ent *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->z != q->z) return p->z < q->z ? -1 : 1;
    return p->index < q->index ? -1 : (p->index > q->index ? 1 : 0); /* input order inside a voxel */
}

int orc_voxel_downsample(const double *pts, int n, double voxel_size, double *out)
{
    if (voxel_size <= 0) { /* file_utils.cpp:152 */
        memcpy(out, pts, sizeof(double) * 3 * (size_t)n);
        return n;
    }
    voxel_ent *e = (voxel_ent *)malloc(sizeof(voxel_ent) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { /* file_utils.cpp:175-181 */
        e[i].x = (long long)floor(pts[3 * i] / voxel_size);
        e[i].y = (long long)floor(pts[3 * i + 1] / voxel_size);
        e[i].z = (long long)floor(pts[3 * i + 2] / voxel_size);
        e[i].index = i;
    }
    qsort(e, (size_t)n, sizeof(voxel_ent), voxel_cmp);
    int count = 0;
    for (int a = 0; a < n;) { /* file_utils.cpp:186-193 */
        int b = a;
        double cx = 0, cy = 0, cz = 0;
        while (b < n && e[b].x == e[a].x && e[b].y == e[a].y && e[b].z == e[a].z) {
            cx += pts[3 * e[b].index];
            cy += pts[3 * e[b].index + 1];
            cz += pts[3 * e[b].index + 2];
            ++b;
        }
        const double k = (double)(b - a);
        out[3 * count] = cx / k;
        out[3 * count + 1] = cy / k;
        out[3 * count + 2] = cz / k;
        ++count;
        a = b;
    }
    free(e);
    return count;
}

/* ------------------------------------------------------------------------- */
/* Occupancy grid insert (src/ros/slam_node.cpp:211-221, config slam_node.hpp:35-40) */
/* ------------------------------------------------------------------------- */

/* SlamNode::update_occupancy_grid(cloud, sensor): the cell every world point marks, or none.
   keep[i] = 0 where the reference `continue`s (height, range) -- and, beyond the reference,
   where its static_cast<int>(std::floor(x / resolution)) is undefined (non-finite quotient or
   one outside int): such a point marks nothing here.  The reference's container is an
   unordered_set: callers compare as sets. */
void orc_occupancy_cells(const double *world_xyz, int n, const double sensor_xyz[3], double resolution,
                         double height_min, double height_max, double max_range, int *cells_xy,
                         unsigned char *keep)
{
    for (int i = 0; i < n; ++i) {
        const double x = world_xyz[3 * i], y = world_xyz[3 * i + 1], z = world_xyz[3 * i + 2];
        keep[i] = 0;
        cells_xy[2 * i] = cells_xy[2 * i + 1] = 0;
        if (z < height_min || z > height_max) continue;                                   /* :214 */
        const double dx = x - sensor_xyz[0], dy = y - sensor_xyz[1];
        const double r = sqrt(dx * dx + dy * dy);                                         /* :215 */
        if (r > max_range || r < 0.5) continue;                                           /* :216 */
        const double cx = floor(x / resolution), cy = floor(y / resolution);              /* :217-218 */
        if (!(fabs(cx) <= 2147483646.0) || !(fabs(cy) <= 2147483646.0)) continue;         /* UB in the reference */
        cells_xy[2 * i] = (int)cx;
        cells_xy[2 * i + 1] = (int)cy;
        keep[i] = 1;
    }
}

/* ------------------------------------------------------------------------- */
/* Scan Context (scan_context.hpp)                                           */
/* ------------------------------------------------------------------------- */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_scan_context(const double *cloud, int n, double *desc)
{
    const int R = ORC_SC_RINGS, S = ORC_SC_SECTORS;
    const double max_range = 80.0; /* scan_context.hpp:29 */
    for (int e = 0; e < R * S; ++e) desc[e] = -DBL_MAX; /* :45 */
    const double ring_size = max_range / R;         /* :47 */
    const double sector_size = 2.0 * M_PI / S;      /* :48 */
    for (int i = 0; i < n; ++i) {
        const double x = cloud[3 * i], y = cloud[3 * i + 1], z = cloud[3 * i + 2];
        const double range = sqrt(x * x + y * y);   /* :56 */
        const double angle = atan2(y, x) + M_PI;    /* :57 */
        if (range > max_range || range < 0.1) continue; /* :59 */
        int ring = (int)(range / ring_size);        /* :62 */
        int sector = (int)(angle / sector_size);    /* :63 */
        ring = ring < 0 ? 0 : (ring > R - 1 ? R - 1 : ring);       /* :65 */
        sector = sector < 0 ? 0 : (sector > S - 1 ? S - 1 : sector); /* :66 */
        if (z > desc[ring * S + sector]) desc[ring * S + sector] = z; /* :69-71 */
    }
    for (int e = 0; e < R * S; ++e)
        if (desc[e] < -1000) desc[e] = 0; /* :75-81 */
}

/* scan_context.hpp:121-142 */
static double sc_shifted_distance(const double *a, const double *b, int shift)
{
    const int R = ORC_SC_RINGS, S = ORC_SC_SECTORS;
    double sum_ab = 0, sum_aa = 0, sum_bb = 0;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < S; ++j) {
            const double va = a[i * S + j], vb = b[i * S + (j + shift) % S];
            sum_ab += va * vb;
            sum_aa += va * va;
            sum_bb += vb * vb;
        }
    const double norm = sqrt(sum_aa) * sqrt(sum_bb);
    if (norm < 1e-10) return 1.0;
    return 1.0 - sum_ab / norm;
}

double orc_scan_context_distance(const double *a, const double *b)
{
    double best = DBL_MAX; /* scan_context.hpp:91 */
    for (int shift = 0; shift < ORC_SC_SECTORS; ++shift) {
        const double d = sc_shifted_distance(a, b, shift);
        if (d < best) best = d;
    }
    return best;
}

/* ------------------------------------------------------------------------- */
/* Driver (icp.hpp:157-258)                                                  */
/* ------------------------------------------------------------------------- */

void orc_icp_config_default(orc_icp_config *c)
{
    c->max_iterations = 50; /* types.hpp:144 */
    c->tolerance = 1e-6;    /* types.hpp:145 */
    c->min_error = 1e-9;    /* types.hpp:146 */
    identity16(c->initial_transform);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* cloud * R^T + t^T row-wise (types.hpp:110-115, icp.hpp:174-176,225-226) */
static void apply_rt(const double T[16], const double *in, double *out, int n)
{
    for (int i = 0; i < n; ++i) {
        double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        for (int r = 0; r < 3; ++r)
            out[3 * i + r] = ((x * T[4 * r] + y * T[4 * r + 1]) + z * T[4 * r + 2]) + T[4 * r + 3];
    }
}

/* this * other, types.hpp:118-120 */
static void mul44(const double A[16], const double B[16], double C[16])
{
    double tmp[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            tmp[4 * i + j] = s;
        }
    memcpy(C, tmp, sizeof(tmp));
}

/* icp.hpp:198-206 / 243-251: sqrt(sum(((q-p).n)^2) / N) */
static double plane_rms(const double *cur, const double *tgt, const double *normals,
                        const int *idx, int n)
{
    double error = 0;
    for (int i = 0; i < n; ++i) {
        const double *p = &cur[3 * i], *q = &tgt[3 * idx[i]], *nn = &normals[3 * idx[i]];
        double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        double pd = (dx * nn[0] + dy * nn[1]) + dz * nn[2];
        error += pd * pd;
    }
    return sqrt(error / (double)n);
}

int orc_icp_point_to_plane(const double *source_xyz, int n_src, const double *target_xyz,
                           int n_tgt, const orc_icp_config *cfg, int normal_k, int flags,
                           int nthreads, orc_icp_result *res, double *error_history,
                           int history_cap)
{
    if (!source_xyz || !target_xyz || !cfg || !res || n_src <= 0 || n_tgt <= 0) return -1;
    int faithful = flags & 1;
    int hist = 0;
    memset(res, 0, sizeof(*res));

    double t0 = now_s();
    orc_kdtree *tree = orc_kdtree_build(target_xyz, n_tgt); /* icp.hpp:166 */
    double *normals = (double *)malloc(sizeof(double) * 3 * (size_t)n_tgt);
    orc_estimate_normals(target_xyz, n_tgt, tree, normal_k, normals, nthreads); /* :169-171 */
    double t1 = now_s();

    double *cur = (double *)malloc(sizeof(double) * 3 * (size_t)n_src);
    double *nxt = (double *)malloc(sizeof(double) * 3 * (size_t)n_src);
    double *mq = (double *)malloc(sizeof(double) * 3 * (size_t)n_src);
    double *mn = (double *)malloc(sizeof(double) * 3 * (size_t)n_src);
    int *idx = (int *)malloc(sizeof(int) * (size_t)n_src);
    int *idx2 = (int *)malloc(sizeof(int) * (size_t)n_src);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)n_src);

    apply_rt(cfg->initial_transform, source_xyz, cur, n_src); /* icp.hpp:174-176 */
    double total[16];
    memcpy(total, cfg->initial_transform, sizeof(total)); /* icp.hpp:178 */
    double prev_error = DBL_MAX;                           /* icp.hpp:179 */
    int converged = 0, loops = 0;

    for (int iter = 0; iter < cfg->max_iterations; ++iter) { /* icp.hpp:181 */
        ++loops;
        orc_nearest_batch(tree, cur, n_src, idx, d2, nthreads); /* icp.hpp:185 */
        if (faithful) {
            /* find_correspondences also takes sqrt of every distance (kdtree.hpp:212) */
            for (int i = 0; i < n_src; ++i) d2[i] = sqrt(d2[i]);
            orc_nearest_batch(tree, cur, n_src, idx2, d2, nthreads); /* icp.hpp:190 */
        }
        double error = plane_rms(cur, target_xyz, normals, idx, n_src); /* icp.hpp:198-206 */
        if (hist < history_cap && error_history) error_history[hist] = error;
        ++hist; /* icp.hpp:207 */
        if (error < cfg->min_error) { /* icp.hpp:210-213 */
            converged = 1;
            break;
        }
        if (fabs(prev_error - error) < cfg->tolerance) { /* icp.hpp:214-217 */
            converged = 1;
            break;
        }
        for (int i = 0; i < n_src; ++i) { /* gathers: kdtree.hpp:210-211, icp.hpp:192-195 */
            memcpy(&mq[3 * i], &target_xyz[3 * idx[i]], 3 * sizeof(double));
            memcpy(&mn[3 * i], &normals[3 * idx[i]], 3 * sizeof(double));
        }
        double delta[16];
        orc_solve_point_to_plane(cur, mq, mn, n_src, delta); /* icp.hpp:220 */
        apply_rt(delta, cur, nxt, n_src);                     /* icp.hpp:225-226 */
        double *sw = cur;
        cur = nxt;
        nxt = sw;
        mul44(delta, total, total); /* icp.hpp:229 */
        prev_error = error;         /* icp.hpp:231 */
    }
    double t2 = now_s();

    orc_nearest_batch(tree, cur, n_src, idx, d2, nthreads); /* icp.hpp:237 */
    if (faithful) {
        for (int i = 0; i < n_src; ++i) d2[i] = sqrt(d2[i]);
        orc_nearest_batch(tree, cur, n_src, idx2, d2, nthreads); /* icp.hpp:241 */
    }
    double final_error = plane_rms(cur, target_xyz, normals, idx, n_src); /* icp.hpp:243-251 */
    if (hist < history_cap && error_history) error_history[hist] = final_error;
    ++hist; /* icp.hpp:252 */
    double t3 = now_s();

    memcpy(res->transformation, total, sizeof(total)); /* icp.hpp:254 */
    res->converged = converged;
    res->num_iterations = hist - 1; /* icp.hpp:255 */
    res->final_error = final_error;
    res->history_len = hist;
    res->setup_seconds = t1 - t0;
    res->loop_seconds = t2 - t1;
    res->final_seconds = t3 - t2;
    res->loop_iterations = loops;

    free(cur);
    free(nxt);
    free(mq);
    free(mn);
    free(idx);
    free(idx2);
    free(d2);
    free(normals);
    orc_kdtree_free(tree);
    return 0;
}

/*
 * icp_oracle.h -- CPU restatement of the reference's point-to-plane ICP path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (kaushik884/LiDAR-SLAM-from-scratch) ships no
 * tests, golden vectors or fixtures for this path, and it cannot be compiled in
 * this image (it needs Eigen3 >= 3.3, absent here).  This file
 * restates the reference's own algorithm (file:line cited per function, paths
 * relative to slam_viz/include/slam_viz/core/) and the published algorithms of
 * the Eigen 3.4.0 routines it calls.  It is cross-checked in tests/ against
 * independent NumPy/SciPy primitives, not against the reference itself.
 *
 * Plain C99, fp64 throughout, built with -ffp-contract=off so that no FMA is
 * formed (the reference is built -O3 for baseline x86-64: no FMA either).
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kdtree orc_kdtree;

/* kdtree.hpp:20-26,87-110 -- median-split tree, axis = depth % 3. */
orc_kdtree *orc_kdtree_build(const double *points_xyz, int n);
void orc_kdtree_free(orc_kdtree *t);

/* kdtree.hpp:43-59 (nearest_batch) -> indices + squared distances.
 * nthreads > 1 partitions the queries statically over pthreads ("not the
 * reference": the reference is single-threaded). */
void orc_nearest_batch(const orc_kdtree *t, const double *queries_xyz, int nq,
                       int *indices, double *dist_sq, int nthreads);

/* kdtree.hpp:65-78,144-180 (k_nearest) -> indices closest first; returns count. */
int orc_k_nearest(const orc_kdtree *t, const double query[3], int k, int *out_idx);

/* Exhaustive search with the same fp64 arithmetic; ties -> lowest index.
 * Used to show kd-tree == brute force on the fixtures. */
void orc_nearest_batch_brute(const double *targets_xyz, int m, const double *queries_xyz,
                             int nq, int *indices, double *dist_sq);
int orc_k_nearest_brute(const double *targets_xyz, int m, const double query[3], int k,
                        int *out_idx);

/* icp.hpp:23-67 (estimate_normals). */
void orc_estimate_normals(const double *points_xyz, int m, const orc_kdtree *t, int k,
                          double *normals_xyz, int nthreads);
/* rows [row0, row1) only; out_xyz receives row1 - row0 rows */
void orc_estimate_normals_rows(const double *points_xyz, int m, const orc_kdtree *t, int k, int row0,
                               int row1, double *out_xyz, int nthreads);

/* icp.hpp:89-144 (solve_point_to_plane) -> row-major 4x4. */
void orc_solve_point_to_plane(const double *source_xyz, const double *target_xyz,
                              const double *normals_xyz, int n, double T_rowmajor[16]);

/* The 27 sums of the normal equations + sum(b^2), serial order (icp.hpp:99-120,198-206).
 * out[0..20] = upper triangle of J^T J row by row, out[21..26] = J^T b, out[27] = sum b^2. */
void orc_normal_equations(const double *source_xyz, const double *target_xyz,
                          const double *normals_xyz, int n, double out[28]);

/* 6x6 pivoted LDLT solve + Rodrigues on given sums (icp.hpp:120-143). */
void orc_solve_from_sums(const double sums[28], double T_rowmajor[16]);

/* 3x3 symmetric eigen: eigenvector of the smallest eigenvalue (icp.hpp:55-56). */
void orc_smallest_eigenvector(const double cov_rowmajor[9], double v[3]);

/* src/core/file_utils.cpp:148-196 (voxel_downsample): centroid of the points of every
 * occupied voxel, key = floor(coord / voxel_size) per axis, points summed in input order.
 * The reference emits voxels in std::unordered_map iteration order (implementation-defined);
 * here they come out sorted by (kx, ky, kz).  Returns the number of voxels (<= n); out_xyz
 * must hold 3*n doubles.  voxel_size <= 0 copies the input (file_utils.cpp:152). */
int orc_voxel_downsample(const double *points_xyz, int n, double voxel_size, double *out_xyz);

/* scan_context.hpp:44-82 (ScanContext::compute): 20 rings x 60 sectors max-height descriptor,
 * row-major desc[ring*60 + sector]; empty bins 0. */
#define ORC_SC_RINGS 20
#define ORC_SC_SECTORS 60
/* update_occupancy_grid (src/ros/slam_node.cpp:211-221): per point the grid cell it marks (keep = 1) or none */
void orc_occupancy_cells(const double *world_xyz, int n, const double sensor_xyz[3], double resolution,
                         double height_min, double height_max, double max_range, int *cells_xy,
                         unsigned char *keep);

void orc_scan_context(const double *cloud_xyz, int n, double *desc /* 20*60 */);
/* scan_context.hpp:90-101,121-142 (distance): min over the 60 column shifts of 1 - cosine. */
double orc_scan_context_distance(const double *a, const double *b);

typedef struct {
    int max_iterations;          /* types.hpp:144 */
    double tolerance;            /* types.hpp:145 */
    double min_error;            /* types.hpp:146 */
    double initial_transform[16];/* types.hpp:147, row-major here */
} orc_icp_config;

typedef struct {
    double transformation[16];   /* types.hpp:156, row-major here */
    int converged;               /* types.hpp:157 */
    int num_iterations;          /* types.hpp:158 */
    double final_error;          /* types.hpp:160 */
    int history_len;             /* error_history.size(), types.hpp:159 */
    /* timing, not part of the reference's result */
    double setup_seconds;        /* tree build + normals (icp.hpp:166-171) */
    double loop_seconds;         /* icp.hpp:181-232 */
    double final_seconds;        /* icp.hpp:235-252 */
    int loop_iterations;         /* loop bodies entered */
} orc_icp_result;

void orc_icp_config_default(orc_icp_config *c);

/* icp.hpp:157-258.  flags bit0: 1 = faithful (two NN passes per iteration and two
 * after the loop, icp.hpp:185,190,237,241), 0 = deduplicated (identical results).
 * normal_k is 20 in the reference (icp.hpp:170).  Returns 0, or -1 on bad input. */
int orc_icp_point_to_plane(const double *source_xyz, int n_src, const double *target_xyz,
                           int n_tgt, const orc_icp_config *cfg, int normal_k, int flags,
                           int nthreads, orc_icp_result *res, double *error_history,
                           int history_cap);

#ifdef __cplusplus
}
#endif
#endif

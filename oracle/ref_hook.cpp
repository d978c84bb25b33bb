// ref_hook.cpp -- OPTIONAL direct-oracle hook (last row).  TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper that calls the REFERENCE's own header-only code
// (slam_viz/core/{types,kdtree,icp}.hpp, included from where it lies -- nothing is copied)
// so that, in an environment that has Eigen3 >= 3.3 and a checkout of the reference, the
// oracle (icp_oracle.c) and the GPU path can be diffed against the reference itself and the
// "parity unpinned" status lifted.  In this image Eigen3 is absent: `make ref` reports that and
// builds nothing, and tests/test_reference_hook.py skips.
//
// Build (oracle/Makefile, target `ref`):
//   g++ -O3 -DNDEBUG -std=c++17 -fPIC -shared -I$(REFERENCE_INCLUDE_DIR) -I$(EIGEN_INCLUDE_DIR) \
//       ref_hook.cpp -o _ref/libslam_ref.so
#if !__has_include(<Eigen/Dense>)
#error "Eigen3 headers not found: the reference cannot be built here (see oracle/Makefile, DESIGN.md section 2)"
#endif
#include <cstring>
#include <vector>

#include "slam_viz/core/icp.hpp"
#include "slam_viz/core/kdtree.hpp"
#include "slam_viz/core/types.hpp"

namespace {
using Mat = slam::PointCloud::Matrix; // N x 3 fp64 row-major (types.hpp:17)

Mat to_mat(const double *xyz, int n)
{
    Mat m(n, 3);
    if (n > 0) std::memcpy(m.data(), xyz, sizeof(double) * 3 * static_cast<size_t>(n));
    return m;
}
void put_rowmajor(const Eigen::Matrix4d &M, double *out16)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out16[4 * r + c] = M(r, c); // Eigen is column-major: element-wise
}
} // namespace

extern "C" {

// slam::icp_point_to_plane (icp.hpp:157-258).  Returns error_history.size().
int ref_icp_point_to_plane(const double *src_xyz, int n, const double *tgt_xyz, int m, int max_iterations,
                           double tolerance, double min_error, const double *initial_rowmajor16,
                           double *transformation_rowmajor16, int *converged, int *num_iterations,
                           double *final_error, double *error_history, int history_cap)
{
    slam::ICPConfig cfg;
    cfg.max_iterations = max_iterations;
    cfg.tolerance = tolerance;
    cfg.min_error = min_error;
    if (initial_rowmajor16) {
        Eigen::Matrix4d T0;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) T0(r, c) = initial_rowmajor16[4 * r + c];
        cfg.initial_transform = slam::Transformation(T0);
    }
    const slam::PointCloud source(to_mat(src_xyz, n)), target(to_mat(tgt_xyz, m));
    const slam::ICPResult r = slam::icp_point_to_plane(source, target, cfg);
    put_rowmajor(r.transformation.matrix(), transformation_rowmajor16);
    *converged = r.converged ? 1 : 0;
    *num_iterations = r.num_iterations;
    *final_error = r.final_error;
    const int len = static_cast<int>(r.error_history.size());
    for (int i = 0; i < len && i < history_cap; ++i) error_history[i] = r.error_history[i];
    return len;
}

// KDTree(points) + nearest_batch (kdtree.hpp:20-26,43-59)
void ref_nearest_batch(const double *tgt_xyz, int m, const double *qry_xyz, int nq, int *indices, double *dist_sq)
{
    const slam::KDTree tree(to_mat(tgt_xyz, m));
    std::vector<int> idx;
    std::vector<double> d2;
    tree.nearest_batch(to_mat(qry_xyz, nq), idx, d2);
    for (int i = 0; i < nq; ++i) {
        indices[i] = idx[i];
        dist_sq[i] = d2[i];
    }
}

// estimate_normals (icp.hpp:23-67)
void ref_estimate_normals(const double *pts_xyz, int m, int k, double *normals_xyz)
{
    const Mat pts = to_mat(pts_xyz, m);
    const slam::KDTree tree(pts);
    const Mat nrm = slam::estimate_normals(pts, tree, k);
    std::memcpy(normals_xyz, nrm.data(), sizeof(double) * 3 * static_cast<size_t>(m));
}

// solve_point_to_plane (icp.hpp:89-144)
void ref_solve_point_to_plane(const double *src_xyz, const double *tgt_xyz, const double *nrm_xyz, int n,
                              double *transformation_rowmajor16)
{
    const slam::Transformation T = slam::solve_point_to_plane(to_mat(src_xyz, n), to_mat(tgt_xyz, n), to_mat(nrm_xyz, n));
    put_rowmajor(T.matrix(), transformation_rowmajor16);
}

} // extern "C"

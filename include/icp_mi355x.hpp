// icp_mi355x.hpp -- header-only C++17 host mirror of the reference's registration API on
// top of the C ABI (icp_mi355x.h).  Eigen-free: the reference's types are restated with
// plain storage so a caller without Eigen can use the same names and call shapes:
//
//   reference (slam_viz/include/slam_viz/core/)          here (namespace icp_mi355x)
//   slam::PointCloud            types.hpp:15-61          PointCloud   (row-major N x 3 fp64)
//   slam::Transformation        types.hpp:74-136         Transformation (row-major 4x4)
//   slam::ICPConfig             types.hpp:143-148        ICPConfig
//   slam::ICPResult             types.hpp:155-164        ICPResult
//   slam::icp_point_to_plane    icp.hpp:157-161          icp_point_to_plane
//   slam::KDTree                kdtree.hpp:18-186        KDTree (nearest, nearest_batch)
//   slam::NearestNeighborSearch kdtree.hpp:193-221       NearestNeighborSearch (find_correspondences)
//   slam::estimate_normals      icp.hpp:23-67            estimate_normals
//   slam::solve_point_to_plane  icp.hpp:89-144           solve_point_to_plane
//   (north_star wording)                                 ICP::align
//   slam::ScanContext           scan_context.hpp:44-142      ScanContext (compute, distance), scan_context_distances
//   slam::LoopClosureConfig / LoopClosureResult / LoopClosureDetector
//                               loop_closure.hpp:14-148      LoopClosureConfig, LoopClosureResult, LoopClosureDetector
//   SlamNode::process_frame, registration and map side   OdometryStream (push, map_update)
//     (slam_viz/src/ros/slam_node.cpp:118-157)
//   OccupancyGridConfig / GridCell / update_occupancy_grid / cells_to_occupancy_grid_msg
//     (slam_viz/include/slam_viz/ros/slam_node.hpp:35-58, slam_node.cpp:211-221,279-297)
//                                                        OccupancyGridConfig, GridCell, OccupancyGrid
//
// A caller that already has Eigen and the reference's own types uses
// slam_icp_adapter.hpp instead, which keeps slam::icp_point_to_plane's exact signature.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "icp_mi355x.h"

namespace icp_mi355x {

class PointCloud { // types.hpp:15-61
public:
    PointCloud() = default;
    explicit PointCloud(std::vector<double> xyz) : xyz_(std::move(xyz))
    {
        if (xyz_.size() % 3) throw std::invalid_argument("PointCloud: size not a multiple of 3");
    }
    PointCloud(const double *xyz, std::size_t n) : xyz_(xyz, xyz + 3 * n) {}
    const double *data() const { return xyz_.data(); }
    double *data() { return xyz_.data(); }
    std::size_t size() const { return xyz_.size() / 3; }
    bool empty() const { return xyz_.empty(); }
    const double *row(std::size_t i) const { return &xyz_[3 * i]; }
    std::array<double, 3> centroid() const
    {
        std::array<double, 3> c{0, 0, 0};
        for (std::size_t i = 0; i < size(); ++i)
            for (int a = 0; a < 3; ++a) c[a] += xyz_[3 * i + a];
        if (size())
            for (int a = 0; a < 3; ++a) c[a] /= static_cast<double>(size());
        return c;
    }
    PointCloud copy() const { return PointCloud(xyz_); }

private:
    std::vector<double> xyz_;
};

class Transformation { // types.hpp:74-136, row-major storage
public:
    Transformation() { m_ = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
    explicit Transformation(const std::array<double, 16> &row_major) : m_(row_major) {}
    static Transformation identity() { return Transformation(); }
    static Transformation from_rt(const std::array<double, 9> &R, const std::array<double, 3> &t)
    {
        Transformation T;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T.m_[4 * i + j] = R[3 * i + j];
            T.m_[4 * i + 3] = t[i];
        }
        return T;
    }
    const std::array<double, 16> &matrix() const { return m_; }
    double operator()(int r, int c) const { return m_[4 * r + c]; }
    std::array<double, 3> t() const { return {m_[3], m_[7], m_[11]}; }
    std::array<double, 3> apply(const std::array<double, 3> &p) const
    {
        std::array<double, 3> o;
        for (int r = 0; r < 3; ++r)
            o[r] = ((p[0] * m_[4 * r] + p[1] * m_[4 * r + 1]) + p[2] * m_[4 * r + 2]) + m_[4 * r + 3];
        return o;
    }
    PointCloud apply(const PointCloud &cloud) const // cloud * R^T + t^T, types.hpp:110-115
    {
        std::vector<double> out(3 * cloud.size());
        for (std::size_t i = 0; i < cloud.size(); ++i) {
            const double *p = cloud.row(i);
            auto o = apply({p[0], p[1], p[2]});
            out[3 * i] = o[0];
            out[3 * i + 1] = o[1];
            out[3 * i + 2] = o[2];
        }
        return PointCloud(std::move(out));
    }
    Transformation compose(const Transformation &other) const // this applied after other
    {
        std::array<double, 16> c{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += m_[4 * i + k] * other.m_[4 * k + j];
                c[4 * i + j] = s;
            }
        return Transformation(c);
    }
    Transformation operator*(const Transformation &other) const { return compose(other); }
    Transformation inverse() const // (R^T, -R^T t), types.hpp:128-132
    {
        std::array<double, 9> Rt;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[3 * i + j] = m_[4 * j + i];
        std::array<double, 3> ti;
        for (int i = 0; i < 3; ++i) ti[i] = -(Rt[3 * i] * m_[3] + Rt[3 * i + 1] * m_[7] + Rt[3 * i + 2] * m_[11]);
        return from_rt(Rt, ti);
    }

private:
    std::array<double, 16> m_;
};

struct ICPConfig { // types.hpp:143-148
    int max_iterations = 50;
    double tolerance = 1e-6;
    double min_error = 1e-9;
    Transformation initial_transform = Transformation::identity();
};

struct ICPResult { // types.hpp:155-164
    Transformation transformation;
    bool converged = false;
    int num_iterations = 0;
    std::vector<double> error_history;
    double final_error = 0.0;
    bool success() const { return converged && final_error < 0.1; }
};

class IcpError : public std::runtime_error {
public:
    IcpError(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

// RAII owner of an icpmi_ctx (device buffers, stream, optional RCCL communicator).
class Context {
public:
    explicit Context(int device = 0, int normal_k = 20, int search = ICPMI_SEARCH_AUTO)
    {
        icpmi_options o;
        icpmi_options_default(&o);
        o.device = device;
        o.normal_k = normal_k;
        o.search = search;
        int rc = icpmi_create(&o, &ctx_);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(nullptr));
    }
    ~Context() { icpmi_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    icpmi_ctx *get() const { return ctx_; }

private:
    icpmi_ctx *ctx_ = nullptr;
};

// One context per thread, created on first use (the reference's function is stateless;
// the context only caches device allocations between calls).
inline Context &default_context()
{
    thread_local Context ctx;
    return ctx;
}

namespace detail {
inline icpmi_config to_c(const ICPConfig &c)
{
    icpmi_config k;
    icpmi_config_default(&k);
    k.max_iterations = c.max_iterations;
    k.tolerance = c.tolerance;
    k.min_error = c.min_error;
    for (int i = 0; i < 16; ++i) k.initial_transform[i] = c.initial_transform.matrix()[i];
    return k;
}
} // namespace detail

// Raw-pointer form: row-major N x 3 fp64 clouds (what PointCloud::points().data() is in
// the reference, types.hpp:17).
inline ICPResult icp_point_to_plane(Context &ctx, const double *source_xyz, std::size_t n_src,
                                    const double *target_xyz, std::size_t n_tgt,
                                    const ICPConfig &config = ICPConfig())
{
    icpmi_config k = detail::to_c(config);
    std::vector<double> hist(static_cast<std::size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1);
    icpmi_result r;
    int rc = icpmi_align(ctx.get(), source_xyz, static_cast<int64_t>(n_src), target_xyz,
                         static_cast<int64_t>(n_tgt), &k, &r, hist.data(), static_cast<int32_t>(hist.size()));
    if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx.get()));
    ICPResult out;
    std::array<double, 16> m;
    for (int i = 0; i < 16; ++i) m[i] = r.transformation[i];
    out.transformation = Transformation(m);
    out.converged = r.converged != 0;
    out.num_iterations = r.num_iterations;
    out.final_error = r.final_error;
    hist.resize(static_cast<std::size_t>(r.history_len));
    out.error_history = std::move(hist);
    return out;
}

// Several independent registrations of one source at once (icpmi_align_batch): what the verifications of one
// LoopClosureDetector::detect() are (loop_closure.hpp:94-123).  Each result is that of icp_point_to_plane alone.
inline std::vector<ICPResult> icp_point_to_plane_batch(Context &ctx, const PointCloud &source,
                                                       const std::vector<const PointCloud *> &targets, const ICPConfig &config)
{
    const std::size_t k = targets.size();
    std::vector<ICPResult> out(k);
    if (k == 0) return out;
    std::vector<const double *> sp(k, source.data()), tp(k);
    std::vector<int64_t> ns(k, static_cast<int64_t>(source.size())), nt(k);
    for (std::size_t i = 0; i < k; ++i) {
        tp[i] = targets[i]->data();
        nt[i] = static_cast<int64_t>(targets[i]->size());
    }
    std::vector<icpmi_config> cfgs(k, detail::to_c(config));
    const std::size_t stride = static_cast<std::size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1;
    std::vector<double> hist(k * stride);
    std::vector<icpmi_result> res(k);
    std::vector<int32_t> status(k);
    const int rc = icpmi_align_batch(ctx.get(), static_cast<int32_t>(k), sp.data(), ns.data(), tp.data(), nt.data(), cfgs.data(),
                                     res.data(), hist.data(), static_cast<int32_t>(stride), status.data());
    if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx.get()));
    for (std::size_t i = 0; i < k; ++i) {
        std::array<double, 16> m;
        for (int e = 0; e < 16; ++e) m[e] = res[i].transformation[e];
        out[i].transformation = Transformation(m);
        out[i].converged = res[i].converged != 0;
        out[i].num_iterations = res[i].num_iterations;
        out[i].final_error = res[i].final_error;
        out[i].error_history.assign(hist.begin() + static_cast<std::ptrdiff_t>(i * stride),
                                    hist.begin() + static_cast<std::ptrdiff_t>(i * stride) + res[i].history_len);
    }
    return out;
}

// Same call shape as slam::icp_point_to_plane (icp.hpp:157-161).
inline ICPResult icp_point_to_plane(const PointCloud &source, const PointCloud &target,
                                    const ICPConfig &config = ICPConfig())
{
    return icp_point_to_plane(default_context(), source.data(), source.size(), target.data(),
                              target.size(), config);
}

// Same call shape as slam::voxel_downsample (src/core/file_utils.cpp:148-196); voxels come
// out sorted by key (the reference's order is std::unordered_map iteration order).
inline PointCloud voxel_downsample(Context &ctx, const PointCloud &points, double voxel_size)
{
    std::vector<double> out(3 * points.size());
    int64_t rows = 0;
    int rc = icpmi_voxel_downsample(ctx.get(), points.data(), static_cast<int64_t>(points.size()), voxel_size,
                                    out.data(), static_cast<int64_t>(points.size()), &rows);
    if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx.get()));
    out.resize(3 * static_cast<std::size_t>(rows));
    return PointCloud(std::move(out));
}
inline PointCloud voxel_downsample(const PointCloud &points, double voxel_size)
{
    return voxel_downsample(default_context(), points, voxel_size);
}

// ---- stage-level mirrors (the pieces slam::icp_point_to_plane is made of) ------------------------
// slam::KDTree (kdtree.hpp:18-186).  The device search needs no tree: the object keeps the target
// rows (as the reference's constructor copies them, kdtree.hpp:20) and searches through the C ABI.
class KDTree {
public:
    explicit KDTree(const PointCloud &points, Context *ctx = nullptr) : pts_(points.copy()), ctx_(ctx) {}
    std::size_t size() const { return pts_.size(); }
    const PointCloud &points() const { return pts_; }
    // kdtree.hpp:43-59: nearest target row and squared distance for every query row
    void nearest_batch(const PointCloud &queries, std::vector<int> &indices, std::vector<double> &distances_sq) const
    {
        indices.assign(queries.size(), -1);
        distances_sq.assign(queries.size(), 0.0);
        if (queries.empty()) return;
        Context &c = ctx_ ? *ctx_ : default_context();
        std::vector<int32_t> idx(queries.size());
        int rc = icpmi_nearest_batch(c.get(), pts_.data(), static_cast<int64_t>(pts_.size()), queries.data(),
                                     static_cast<int64_t>(queries.size()), idx.data(), distances_sq.data());
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
        for (std::size_t i = 0; i < idx.size(); ++i) indices[i] = idx[i];
    }
    // kdtree.hpp:28-38: (index, squared distance) of the nearest row to one point
    std::pair<int, double> nearest(const std::array<double, 3> &query) const
    {
        std::vector<int> i;
        std::vector<double> d;
        nearest_batch(PointCloud(query.data(), 1), i, d);
        return {i[0], d[0]};
    }
    // kdtree.hpp:65-78: indices of the k nearest rows to one point, closest first
    std::vector<int> k_nearest(const std::array<double, 3> &query, int k) const
    {
        std::vector<int32_t> idx(static_cast<std::size_t>(k > 0 ? k : 0), -1);
        Context &c = ctx_ ? *ctx_ : default_context();
        int rc = icpmi_k_nearest(c.get(), pts_.data(), static_cast<int64_t>(pts_.size()), query.data(), 1, k, idx.data(), nullptr);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
        std::vector<int> out;
        for (int32_t j : idx)
            if (j >= 0) out.push_back(j);
        return out;
    }

private:
    PointCloud pts_;
    Context *ctx_;
};

// slam::NearestNeighborSearch (kdtree.hpp:193-221)
class NearestNeighborSearch {
public:
    explicit NearestNeighborSearch(const PointCloud &target, Context *ctx = nullptr) : tree_(target, ctx) {}
    const KDTree &tree() const { return tree_; }
    // kdtree.hpp:198-214: matched_target.row(i) = target.row(nearest(i)), distances = sqrt(d^2)
    void find_correspondences(const PointCloud &source, PointCloud &matched_target, std::vector<double> &distances) const
    {
        std::vector<int> idx;
        std::vector<double> d2;
        tree_.nearest_batch(source, idx, d2);
        std::vector<double> rows(3 * source.size());
        distances.resize(source.size());
        for (std::size_t i = 0; i < source.size(); ++i) {
            const double *q = tree_.points().row(static_cast<std::size_t>(idx[i]));
            rows[3 * i] = q[0];
            rows[3 * i + 1] = q[1];
            rows[3 * i + 2] = q[2];
            distances[i] = std::sqrt(d2[i]);
        }
        matched_target = PointCloud(std::move(rows));
    }

private:
    KDTree tree_;
};

// slam::estimate_normals (icp.hpp:23-67): unit normals of `points` from their k nearest neighbours
inline PointCloud estimate_normals(const PointCloud &points, int k = 20, Context *ctx = nullptr)
{
    Context &c = ctx ? *ctx : default_context();
    std::vector<double> out(3 * points.size());
    int rc = icpmi_estimate_normals(c.get(), points.data(), static_cast<int64_t>(points.size()), k, out.data());
    if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
    return PointCloud(std::move(out));
}

// slam::solve_point_to_plane (icp.hpp:89-144): one linearised step for given correspondences
inline Transformation solve_point_to_plane(const PointCloud &source, const PointCloud &target, const PointCloud &normals,
                                           Context *ctx = nullptr)
{
    if (source.size() != target.size() || source.size() != normals.size())
        throw std::invalid_argument("solve_point_to_plane: row counts differ");
    Context &c = ctx ? *ctx : default_context();
    std::array<double, 16> T{};
    int rc = icpmi_solve_point_to_plane(c.get(), source.data(), target.data(), normals.data(),
                                        static_cast<int64_t>(source.size()), T.data());
    if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
    return Transformation(T);
}

// ---- the caller's frame step (slam_viz/src/ros/slam_node.cpp:118-157) ---------------------------------
struct OccupancyGridConfig { // slam_node.hpp:35-40
    double resolution = 0.2;
    double height_min = 0.3;
    double height_max = 2.0;
    double max_range = 40.0;
};
struct GridCell { // slam_node.hpp:45-50
    int x, y;
    bool operator==(const GridCell &o) const { return x == o.x && y == o.y; }
};
namespace detail {
inline icpmi_grid_config to_c(const OccupancyGridConfig &g)
{
    icpmi_grid_config c;
    c.resolution = g.resolution;
    c.height_min = g.height_min;
    c.height_max = g.height_max;
    c.max_range = g.max_range;
    return c;
}
} // namespace detail

// occupied_cells_ and the functions around it (slam_node.cpp:211-226,279-297); the set lives in the
// context's device memory, sorted by (x, y).
class OccupancyGrid {
public:
    explicit OccupancyGrid(OccupancyGridConfig config = OccupancyGridConfig(), Context *ctx = nullptr)
        : config_(config), ctx_(ctx ? ctx : &default_context())
    {
    }
    // update_occupancy_grid(cloud, sensor) (slam_node.cpp:211-221) -> cells in the set afterwards
    std::size_t update(const PointCloud &world, const std::array<double, 3> &sensor)
    {
        const icpmi_grid_config g = detail::to_c(config_);
        int64_t n = 0;
        const int rc = icpmi_occupancy_update(ctx_->get(), world.data(), static_cast<int64_t>(world.size()), sensor.data(), &g, &n);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        return static_cast<std::size_t>(n);
    }
    void clear() { icpmi_occupancy_clear(ctx_->get()); } // slam_node.cpp:224
    std::vector<GridCell> cells() const
    {
        int64_t n = 0;
        int rc = icpmi_occupancy_cells(ctx_->get(), nullptr, 0, &n);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        std::vector<GridCell> out(static_cast<std::size_t>(n));
        static_assert(sizeof(GridCell) == 2 * sizeof(int32_t), "cells are read as int32 pairs");
        if (n > 0) {
            rc = icpmi_occupancy_cells(ctx_->get(), reinterpret_cast<int32_t *>(out.data()), n, &n);
            if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        }
        return out;
    }
    const OccupancyGridConfig &config() const { return config_; }

private:
    OccupancyGridConfig config_;
    Context *ctx_;
};

// process_frame as two calls per frame with the scans resident in device memory: push() is
// lines 122-138 (voxel filter, guards, registration against the previous filtered scan), the caller
// applies its gate and pose update (139-145), map_update() is lines 147-153 (world points of the
// scan just pushed, occupancy insert with the new pose's translation as the sensor position).
class OdometryStream {
public:
    explicit OdometryStream(Context *ctx = nullptr) : ctx_(ctx ? ctx : &default_context()) {}
    struct Step {
        ICPResult result;            // identity / converged = false unless `registered`
        bool registered = false;     // an ICP ran (source = this scan, target = the previous one)
        bool first_frame = false;    // slam_node.cpp:69-72
        bool too_few_points = false; // slam_node.cpp:125-130
        std::size_t filtered_points = 0;
    };
    Step push(const PointCloud &raw, double voxel_size, long long min_points, const ICPConfig &config = ICPConfig())
    {
        icpmi_config k = detail::to_c(config);
        std::vector<double> hist(static_cast<std::size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1);
        icpmi_result out;
        icpmi_stream_info info;
        const int rc = icpmi_stream_push_host(ctx_->get(), raw.data(), static_cast<int64_t>(raw.size()), voxel_size, min_points, &k,
                                              &out, hist.data(), static_cast<int32_t>(hist.size()), &info);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        return finish(out, info, std::move(hist));
    }
    // the scan as a file (load_ply / load_bin, slam_node.cpp:121)
    Step push_file(const std::string &path, double voxel_size, long long min_points, const ICPConfig &config = ICPConfig())
    {
        icpmi_config k = detail::to_c(config);
        std::vector<double> hist(static_cast<std::size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1);
        icpmi_result out;
        icpmi_stream_info info;
        const int rc = icpmi_stream_push_file(ctx_->get(), path.c_str(), voxel_size, min_points, &k, &out, hist.data(),
                                              static_cast<int32_t>(hist.size()), &info);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        return finish(out, info, std::move(hist));
    }
    // start reading the NEXT frame's file on the context's worker thread; call before pushing the current frame
    void prefetch_file(const std::string &path) { icpmi_stream_prefetch_file(ctx_->get(), path.c_str()); }
    // world = curr * R^T + t^T (:147) of the scan just pushed; with `grid`, update_occupancy_grid(world, t) (:153)
    // into the context's cell set (read it through OccupancyGrid::cells / raster on the same context)
    PointCloud map_update(const Transformation &new_pose, const OccupancyGridConfig *grid = nullptr, std::size_t *n_cells = nullptr)
    {
        int64_t nw = static_cast<int64_t>(last_filtered_), nc = 0;
        std::vector<double> world(3 * last_filtered_);
        icpmi_grid_config g;
        if (grid) g = detail::to_c(*grid);
        const int rc = icpmi_stream_map_update(ctx_->get(), new_pose.matrix().data(), grid ? &g : nullptr, world.data(), nw, &nw, &nc);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        if (n_cells) *n_cells = static_cast<std::size_t>(nc);
        return PointCloud(std::move(world));
    }
    // `curr` (slam_node.cpp:122) of the frame just pushed, for what the node does with it on the host (:160)
    PointCloud current_scan() const
    {
        int64_t n = static_cast<int64_t>(last_filtered_);
        std::vector<double> xyz(3 * last_filtered_);
        const int rc = icpmi_stream_current_scan(ctx_->get(), xyz.data(), n, &n);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        return PointCloud(std::move(xyz));
    }
    void reset() { icpmi_stream_reset(ctx_->get()); }

private:
    Step finish(const icpmi_result &out, const icpmi_stream_info &info, std::vector<double> hist)
    {
        Step step;
        step.filtered_points = last_filtered_ = static_cast<std::size_t>(info.n_filtered);
        step.first_frame = info.status == ICPMI_STREAM_FIRST_FRAME;
        step.too_few_points = info.status == ICPMI_STREAM_TOO_FEW_POINTS;
        step.registered = info.status == ICPMI_STREAM_REGISTERED;
        if (step.registered) {
            std::array<double, 16> m;
            for (int i = 0; i < 16; ++i) m[i] = out.transformation[i];
            step.result.transformation = Transformation(m);
            step.result.converged = out.converged != 0;
            step.result.num_iterations = out.num_iterations;
            step.result.final_error = out.final_error;
            hist.resize(static_cast<std::size_t>(out.history_len));
            step.result.error_history = std::move(hist);
        }
        return step;
    }
    Context *ctx_;
    std::size_t last_filtered_ = 0;
};

// ---- loop closure: Scan Context candidates + ICP verification (core/scan_context.hpp, core/loop_closure.hpp) ----
// slam::ScanContext: the 20 x 60 max-height descriptor and its column-shift cosine distance.
class ScanContext {
public:
    static constexpr int kRings = ICPMI_SC_RINGS, kSectors = ICPMI_SC_SECTORS;
    ScanContext() : d_(static_cast<std::size_t>(kRings) * kSectors, 0.0) {}
    static ScanContext compute(const PointCloud &cloud, Context *ctx = nullptr) // scan_context.hpp:44-82
    {
        Context &c = ctx ? *ctx : default_context();
        ScanContext sc;
        const int rc = icpmi_scan_context(c.get(), cloud.data(), static_cast<int64_t>(cloud.size()), sc.d_.data());
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
        return sc;
    }
    double distance(const ScanContext &other, Context *ctx = nullptr) const // scan_context.hpp:90-101
    {
        Context &c = ctx ? *ctx : default_context();
        double d = 0.0;
        const int rc = icpmi_scan_context_distances(c.get(), d_.data(), other.d_.data(), 1, &d);
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(c.get()));
        return d;
    }
    const std::vector<double> &descriptor() const { return d_; } // row-major [ring][sector]

private:
    std::vector<double> d_;
};

struct LoopClosureConfig { // loop_closure.hpp:14-19
    int frame_gap = 50;
    double sc_distance_threshold = 0.25;
    double icp_fitness_threshold = 0.3;
    int max_candidates = 3;
};
struct LoopClosureResult { // loop_closure.hpp:25-31
    int query_frame = 0, match_frame = 0;
    Transformation transform;
    double scan_context_distance = 0.0, icp_fitness = 0.0;
};

// slam::LoopClosureDetector (loop_closure.hpp:41-148): keeps every frame's cloud and descriptor, and
// detect() looks for closures of the most recently added frame -- the distances of its descriptor to the
// whole history in ONE device call, the candidate filter and the sort on the host as in the reference,
// up to max_candidates ICP verifications (30 iterations, tolerance 1e-6, :102-109) through the C ABI.
class LoopClosureDetector {
public:
    explicit LoopClosureDetector(LoopClosureConfig config = LoopClosureConfig(), Context *ctx = nullptr)
        : config_(config), ctx_(ctx ? ctx : &default_context())
    {
    }
    void addFrame(const PointCloud &cloud, int frame_idx) // loop_closure.hpp:53-60
    {
        const ScanContext sc = ScanContext::compute(cloud, ctx_);
        descriptors_.insert(descriptors_.end(), sc.descriptor().begin(), sc.descriptor().end());
        clouds_.push_back(cloud.copy());
        frame_indices_.push_back(frame_idx);
    }
    std::size_t size() const { return frame_indices_.size(); }
    void clear()
    {
        descriptors_.clear();
        clouds_.clear();
        frame_indices_.clear();
    }
    std::vector<LoopClosureResult> detect() // loop_closure.hpp:66-126
    {
        std::vector<LoopClosureResult> results;
        if (frame_indices_.size() < 2) return results; // :69
        constexpr std::size_t kDesc = static_cast<std::size_t>(ScanContext::kRings) * ScanContext::kSectors;
        const std::size_t q = frame_indices_.size() - 1;
        std::vector<double> dist(q);
        const int rc = icpmi_scan_context_distances(ctx_->get(), descriptors_.data() + q * kDesc, descriptors_.data(),
                                                    static_cast<int64_t>(q), dist.data()); // :84 for every i
        if (rc != ICPMI_OK) throw IcpError(rc, icpmi_last_error(ctx_->get()));
        std::vector<std::pair<double, int>> candidates;
        for (std::size_t i = 0; i < q; ++i) {
            if (frame_indices_[q] - frame_indices_[i] < config_.frame_gap) continue;                       // :80-82
            if (dist[i] < config_.sc_distance_threshold) candidates.emplace_back(dist[i], static_cast<int>(i)); // :86-89
        }
        std::sort(candidates.begin(), candidates.end()); // :93
        // The reference verifies the candidates one after the other until max_candidates are ACCEPTED (:96-123).
        // The registrations are independent: the next (max_candidates - accepted) of them, all of which the
        // sequential loop would reach, run side by side (icpmi_align_batch); outcomes in the reference's order.
        int verified = 0;
        std::size_t pos = 0;
        ICPConfig icp;                                     // :102-105
        icp.max_iterations = 30;
        icp.tolerance = 1e-6;
        while (pos < candidates.size() && verified < config_.max_candidates) { // :97
            const std::size_t take = std::min<std::size_t>(candidates.size() - pos,
                                                           std::min<std::size_t>(static_cast<std::size_t>(config_.max_candidates - verified), ICPMI_MAX_BATCH));
            std::vector<const PointCloud *> tg;
            for (std::size_t i = 0; i < take; ++i) tg.push_back(&clouds_[static_cast<std::size_t>(candidates[pos + i].second)]);
            const std::vector<ICPResult> rs = icp_point_to_plane_batch(*ctx_, clouds_[q], tg, icp); // :109
            for (std::size_t i = 0; i < take; ++i) {
                const ICPResult &r = rs[i];
                const auto &cand = candidates[pos + i];
                if (r.converged && r.final_error < config_.icp_fitness_threshold) {              // :112
                    LoopClosureResult out;
                    out.query_frame = frame_indices_[q];
                    out.match_frame = frame_indices_[cand.second];
                    out.transform = r.transformation;
                    out.scan_context_distance = cand.first;
                    out.icp_fitness = r.final_error;
                    results.push_back(out);
                    ++verified;
                }
            }
            pos += take;
        }
        return results;
    }
    const LoopClosureConfig &config() const { return config_; }

private:
    LoopClosureConfig config_;
    Context *ctx_;
    std::vector<double> descriptors_; // 1200 per frame, frame-major
    std::vector<PointCloud> clouds_;
    std::vector<int> frame_indices_;
};

// `ICP(config).align(source, target)`: the facade BASELINE.json's north_star names.
class ICP {
public:
    explicit ICP(ICPConfig config = ICPConfig()) : config_(std::move(config)) {}
    ICPResult align(const PointCloud &source, const PointCloud &target) const
    {
        return icp_point_to_plane(source, target, config_);
    }
    ICPConfig &config() { return config_; }

private:
    ICPConfig config_;
};

} // namespace icp_mi355x

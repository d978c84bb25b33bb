// slam_icp_adapter.hpp -- drop-in for a caller that HAS Eigen and the reference's own
// types (slam::PointCloud / Transformation / ICPConfig / ICPResult, core/types.hpp).
//
// Include this INSTEAD of "slam_viz/core/icp.hpp" (it includes types.hpp itself) in BOTH places that
// include it -- src/ros/slam_node.cpp:3 and core/loop_closure.hpp:5 (slam_node.cpp reaches the
// latter through slam_node.hpp:18) -- the two call sites, slam_node.cpp:138 and
// loop_closure.hpp:109, stay as they are:
//
//     auto result = slam::icp_point_to_plane(source, target, icp_cfg);
//
// Leaving one of the two includes in place puts the reference's and this definition of
// slam::icp_point_to_plane into one translation unit: the compiler then stops with a
// redefinition error naming both headers (there is no silent mix of the two).
//
// Differences from the reference implementation, by design:
//   - any error from the library (empty cloud, HIP failure) yields
//     ICPResult{converged = false}, so slam_node.cpp:139's gate falls back to identity
//     instead of the reference's undefined behaviour on empty input;
//   - Eigen::Matrix4d is column-major, the C ABI is row-major: converted element-wise.
//
// This header cannot be compiled in the authoring image (no Eigen); it is kept free of
// anything but the reference's public accessors: points().data(), points().rows(),
// matrix()(r,c), Transformation(Matrix4).  tests/cpp/adapter_check.cpp compiles it wherever
// <Eigen/Dense> and a reference checkout exist (tests/test_boundary.py).
#pragma once

#include <utility>
#include <vector>

#include "slam_viz/core/types.hpp"

#include "icp_mi355x.h"

namespace slam {

namespace icp_mi355x_detail {
inline icpmi_ctx *context()
{
    struct Holder {
        icpmi_ctx *ctx = nullptr;
        Holder()
        {
            icpmi_options o;
            icpmi_options_default(&o);
            if (icpmi_create(&o, &ctx) != ICPMI_OK) ctx = nullptr;
        }
        ~Holder() { icpmi_destroy(ctx); }
    };
    thread_local Holder h;
    return h.ctx;
}
} // namespace icp_mi355x_detail

// Same signature as core/icp.hpp:157-161.
inline ICPResult icp_point_to_plane(const PointCloud &source, const PointCloud &target,
                                    const ICPConfig &config = ICPConfig())
{
    ICPResult result; // converged = false, identity transform (types.hpp:155-160)
    icpmi_ctx *ctx = icp_mi355x_detail::context();
    if (!ctx) return result;

    icpmi_config k;
    icpmi_config_default(&k);
    k.max_iterations = config.max_iterations;
    k.tolerance = config.tolerance;
    k.min_error = config.min_error;
    const auto &M0 = config.initial_transform.matrix();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) k.initial_transform[4 * r + c] = M0(r, c);

    std::vector<double> hist(static_cast<size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1);
    icpmi_result out;
    // PointCloud::Matrix is Eigen row-major N x 3 (types.hpp:17): data() is xyzxyz...
    const int rc = icpmi_align(ctx, source.points().data(), static_cast<int64_t>(source.points().rows()),
                               target.points().data(), static_cast<int64_t>(target.points().rows()), &k,
                               &out, hist.data(), static_cast<int32_t>(hist.size()));
    if (rc != ICPMI_OK) return result;

    Transformation::Matrix4 M;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) M(r, c) = out.transformation[4 * r + c];
    result.transformation = Transformation(M);
    result.converged = out.converged != 0;
    result.num_iterations = out.num_iterations;
    result.final_error = out.final_error;
    result.error_history.assign(hist.begin(), hist.begin() + out.history_len);
    return result;
}

// Drop-in for slam::voxel_downsample (src/core/file_utils.cpp:148-196; callers
// slam_node.cpp:69,122).  Named differently so that it can coexist with the reference's own
// definition in file_utils.cpp; on any library error the input is returned unchanged.
inline PointCloud::Matrix voxel_downsample_mi355x(const PointCloud::Matrix &points, double voxel_size)
{
    icpmi_ctx *ctx = icp_mi355x_detail::context();
    if (!ctx || points.rows() == 0) return points;
    PointCloud::Matrix out(points.rows(), 3);
    int64_t rows = 0;
    if (icpmi_voxel_downsample(ctx, points.data(), static_cast<int64_t>(points.rows()), voxel_size, out.data(),
                               static_cast<int64_t>(points.rows()), &rows) != ICPMI_OK)
        return points;
    out.conservativeResize(rows, 3);
    return out;
}

// The registration part of SlamNode::process_frame (slam_node.cpp:122-152) as one call per frame,
// for a node that hands over its RAW scan instead of calling voxel_downsample and
// icp_point_to_plane itself: the filtered scan of frame t stays in device memory and is the
// target of frame t+1 (slam_node.cpp:132-133,152), so nothing is uploaded twice.  In
// process_frame:
//     auto step = stream_.push(raw, config_.voxel_size, config_.min_points, icp_cfg);
//     if (step.first_frame) { ... } else if (step.too_few_points) { repeat the last pose } else { use step.result }
struct OdometryStream {
    struct Step {
        ICPResult result;            // identity / converged = false unless `registered`
        bool registered = false;     // an ICP ran (source = this scan, target = the previous one)
        bool first_frame = false;    // slam_node.cpp:69-72
        bool too_few_points = false; // slam_node.cpp:125-130
        long long filtered_points = 0;
    };
    Step push(const PointCloud::Matrix &raw, double voxel_size, long long min_points, const ICPConfig &config = ICPConfig())
    {
        Step step;
        icpmi_ctx *ctx = icp_mi355x_detail::context();
        if (!ctx || raw.rows() == 0) return step;
        icpmi_config k;
        icpmi_config_default(&k);
        k.max_iterations = config.max_iterations;
        k.tolerance = config.tolerance;
        k.min_error = config.min_error;
        const auto &M0 = config.initial_transform.matrix();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) k.initial_transform[4 * r + c] = M0(r, c);
        std::vector<double> hist(static_cast<size_t>(config.max_iterations > 0 ? config.max_iterations : 0) + 1);
        icpmi_result out;
        icpmi_stream_info info;
        if (icpmi_stream_push_host(ctx, raw.data(), static_cast<int64_t>(raw.rows()), voxel_size, min_points, &k, &out,
                                   hist.data(), static_cast<int32_t>(hist.size()), &info) != ICPMI_OK)
            return step;
        step.filtered_points = last_filtered_ = info.n_filtered;
        step.first_frame = info.status == ICPMI_STREAM_FIRST_FRAME;
        step.too_few_points = info.status == ICPMI_STREAM_TOO_FEW_POINTS;
        step.registered = info.status == ICPMI_STREAM_REGISTERED;
        if (step.registered) {
            Transformation::Matrix4 M;
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) M(r, c) = out.transformation[4 * r + c];
            step.result.transformation = Transformation(M);
            step.result.converged = out.converged != 0;
            step.result.num_iterations = out.num_iterations;
            step.result.final_error = out.final_error;
            step.result.error_history.assign(hist.begin(), hist.begin() + out.history_len);
        }
        return step;
    }
    // The map side of process_frame (slam_node.cpp:147-153) for the scan just pushed, which never came to the
    // host: returns world = curr * new_pose.R()^T + new_pose.t()^T (what :147 computes and :155 publishes) and,
    // with `grid` (the node's grid_config_ fields), does update_occupancy_grid(world, new_pose.t()) on the
    // device-resident cell set.  In process_frame, in place of lines 147 and 153:
    //     auto world = stream_.map_update(new_pose, &grid_);   // grid_ = {resolution, height_min, height_max, max_range}
    // and cells_to_occupancy_grid_msg (:279-297) iterates stream_.occupied_cells() instead of occupied_cells_.
    PointCloud::Matrix map_update(const Transformation &new_pose, const icpmi_grid_config *grid = nullptr)
    {
        icpmi_ctx *ctx = icp_mi355x_detail::context();
        if (!ctx || last_filtered_ <= 0) return PointCloud::Matrix(0, 3);
        double T[16];
        const auto &M = new_pose.matrix();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) T[4 * r + c] = M(r, c);
        PointCloud::Matrix world(last_filtered_, 3);
        int64_t nw = 0, nc = 0;
        if (icpmi_stream_map_update(ctx, T, grid, world.data(), static_cast<int64_t>(world.rows()), &nw, &nc) != ICPMI_OK)
            return PointCloud::Matrix(0, 3);
        return world;
    }
    // `curr` of the frame just pushed (slam_node.cpp:122), for loop_detector_.addFrame(curr, frame_idx) and
    // downsampled_clouds_ (:160): the one thing of the stream the node still wants on the host
    PointCloud::Matrix current_scan() const
    {
        icpmi_ctx *ctx = icp_mi355x_detail::context();
        PointCloud::Matrix out(last_filtered_ > 0 ? last_filtered_ : 0, 3);
        int64_t n = 0;
        if (!ctx || last_filtered_ <= 0 || icpmi_stream_current_scan(ctx, out.data(), static_cast<int64_t>(out.rows()), &n) != ICPMI_OK)
            return PointCloud::Matrix(0, 3);
        return out;
    }
    // the occupied cells as (x, y) pairs, sorted by x then y (occupied_cells_, slam_node.hpp:151)
    std::vector<std::pair<int, int>> occupied_cells() const
    {
        std::vector<std::pair<int, int>> out;
        icpmi_ctx *ctx = icp_mi355x_detail::context();
        int64_t n = 0;
        if (!ctx || icpmi_occupancy_cells(ctx, nullptr, 0, &n) != ICPMI_OK || n <= 0) return out;
        std::vector<int32_t> xy(2 * static_cast<size_t>(n));
        if (icpmi_occupancy_cells(ctx, xy.data(), n, &n) != ICPMI_OK) return out;
        out.reserve(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; ++i) out.emplace_back(xy[2 * i], xy[2 * i + 1]);
        return out;
    }
    void reset()
    {
        if (icpmi_ctx *ctx = icp_mi355x_detail::context()) {
            icpmi_stream_reset(ctx);
            icpmi_occupancy_clear(ctx);
        }
        last_filtered_ = 0;
    }

private:
    long long last_filtered_ = 0;
};

// The facade named in BASELINE.json's north_star.
struct ICP {
    ICPConfig config;
    ICPResult align(const PointCloud &source, const PointCloud &target) const
    {
        return icp_point_to_plane(source, target, config);
    }
};

} // namespace slam

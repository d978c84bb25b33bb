// Exercises the frame-step part of the C++17 host mirror (include/icp_mi355x.hpp) on the GPU:
//   stream_demo <out.f64> <frame0.bin> <frame1.bin> ...
// runs SlamNode::process_frame's loop (slam_viz/src/ros/slam_node.cpp:118-157) -- OdometryStream::push_file,
// the gate of lines 139-140, new_pose = poses.back() * delta, OdometryStream::map_update with the
// occupancy grid -- and writes, as fp64:
//   [frames, per frame: status(0 registered / 1 first / 2 too few), iterations, filtered points, cells after the
//    frame, pose(16), checksum of the world points (sum of all coordinates)]
//   [cells, x y per cell] [raster: width, height, origin_x, origin_y, occupied count]
// tests/test_gpu_occupancy.py::test_cpp_stream_mirror compares every number with the Python harness.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "icp_mi355x.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s out.f64 frame.bin...\n", argv[0]);
        return 2;
    }
    try {
        namespace im = icp_mi355x;
        im::OdometryStream stream;
        const im::OccupancyGridConfig grid_config;                       // slam_node.hpp:35-40 defaults
        im::OccupancyGrid grid(grid_config);
        im::ICPConfig cfg;                                               // slam_node.cpp:134-136: 50 iterations, 1e-6
        im::Transformation pose;                                         // poses_[0] = identity (slam_node.cpp:73)
        std::vector<double> out{static_cast<double>(argc - 2)};
        for (int k = 2; k < argc; ++k) {
            if (k + 1 < argc) stream.prefetch_file(argv[k + 1]);          // read beside this frame's work
            const im::OdometryStream::Step step = stream.push_file(argv[k], 0.5, 1000, cfg);
            if (step.registered) {
                const bool gated = !step.result.converged || step.result.final_error > 1.0;      // :139-140
                pose = pose * (gated ? im::Transformation::identity() : step.result.transformation); // :142
            }
            std::size_t n_cells = 0;
            const im::PointCloud world = stream.map_update(pose, &grid_config, &n_cells);        // :147-153
            double sum = 0.0;
            for (std::size_t i = 0; i < 3 * world.size(); ++i) sum += world.data()[i];
            out.push_back(step.registered ? 0.0 : (step.first_frame ? 1.0 : 2.0));
            out.push_back(step.registered ? step.result.num_iterations : 0);
            out.push_back(static_cast<double>(step.filtered_points));
            out.push_back(static_cast<double>(n_cells));
            out.insert(out.end(), pose.matrix().begin(), pose.matrix().end());
            out.push_back(sum);
        }
        const std::vector<im::GridCell> cells = grid.cells();
        out.push_back(static_cast<double>(cells.size()));
        for (const im::GridCell &c : cells) {
            out.push_back(c.x);
            out.push_back(c.y);
        }
        std::ofstream f(argv[1], std::ios::binary);
        f.write(reinterpret_cast<const char *>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(double)));
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "stream_demo: %s\n", e.what());
        return 1;
    }
}

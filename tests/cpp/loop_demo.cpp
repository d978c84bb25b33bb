// Exercises the loop-closure part of the C++17 host mirror (include/icp_mi355x.hpp) on the GPU:
//   loop_demo <out.f64> <frame_gap> <sc_distance_threshold> <icp_fitness_threshold> <cloud0.f64> <cloud1.f64> ...
// adds the clouds (row-major N x 3 fp64 files) as frames 0, 1, ... to a LoopClosureDetector
// (core/loop_closure.hpp:41-148), calls detect() after every frame like SlamNode does, and writes, as fp64:
//   [results, per result: query_frame, match_frame, scan_context_distance, icp_fitness, transform(16)]
//   [distance of the last frame's descriptor to the first one's, through ScanContext::distance]
// tests/test_gpu_parity.py::test_cpp_loop_closure_mirror compares every number with the Python mirror.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "icp_mi355x.hpp"

static std::vector<double> read_f64(const char *path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<double> v(static_cast<std::size_t>(bytes) / sizeof(double));
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s out.f64 frame_gap sc_threshold icp_threshold cloud.f64...\n", argv[0]);
        return 2;
    }
    try {
        namespace im = icp_mi355x;
        im::LoopClosureConfig cfg;                                   // loop_closure.hpp:14-19
        cfg.frame_gap = std::atoi(argv[2]);
        cfg.sc_distance_threshold = std::atof(argv[3]);
        cfg.icp_fitness_threshold = std::atof(argv[4]);
        im::LoopClosureDetector detector(cfg);
        std::vector<im::LoopClosureResult> found;
        im::ScanContext first, last;
        for (int k = 5; k < argc; ++k) {
            const im::PointCloud cloud(read_f64(argv[k]));
            detector.addFrame(cloud, k - 5);                         // slam_node.cpp:160
            for (const im::LoopClosureResult &r : detector.detect()) found.push_back(r);   // slam_node.cpp:162
            last = im::ScanContext::compute(cloud);
            if (k == 5) first = last;
        }
        std::vector<double> out{static_cast<double>(found.size())};
        for (const im::LoopClosureResult &r : found) {
            out.push_back(r.query_frame);
            out.push_back(r.match_frame);
            out.push_back(r.scan_context_distance);
            out.push_back(r.icp_fitness);
            out.insert(out.end(), r.transform.matrix().begin(), r.transform.matrix().end());
        }
        out.push_back(last.distance(first));
        out.push_back(static_cast<double>(detector.size()));
        std::ofstream f(argv[1], std::ios::binary);
        f.write(reinterpret_cast<const char *>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(double)));
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "loop_demo: %s\n", e.what());
        return 1;
    }
}

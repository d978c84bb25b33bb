// Exercises the C++17 host mirror (include/icp_mi355x.hpp) end to end on the GPU:
//   mirror_demo <source.f64> <target.f64> <out.f64>
// reads two row-major N x 3 fp64 clouds, runs NearestNeighborSearch::find_correspondences,
// estimate_normals, solve_point_to_plane and ICP(config).align through the mirror, and writes
//   [n, m, idx(n), dist(n), normals(3m), T_solve(16), T_icp(16), converged, num_iterations, final_error,
//    history_len, history...]                                                       as fp64.
// tests/test_gpu_parity.py::test_cpp_mirror_end_to_end compares every number with the oracle.
#include <cstdio>
#include <fstream>
#include <iostream>
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
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s source.f64 target.f64 out.f64\n", argv[0]);
        return 2;
    }
    try {
        namespace im = icp_mi355x;
        const im::PointCloud source(read_f64(argv[1])), target(read_f64(argv[2]));
        std::vector<double> out{static_cast<double>(source.size()), static_cast<double>(target.size())};

        im::NearestNeighborSearch nn(target);                           // kdtree.hpp:193-221
        im::PointCloud matched;
        std::vector<double> dist;
        nn.find_correspondences(source, matched, dist);
        std::vector<int> idx;
        std::vector<double> d2;
        nn.tree().nearest_batch(source, idx, d2);
        for (int i : idx) out.push_back(i);
        for (double d : dist) out.push_back(d);

        const im::PointCloud normals = im::estimate_normals(target, 20); // icp.hpp:23-67
        out.insert(out.end(), normals.data(), normals.data() + 3 * normals.size());

        std::vector<double> mn(3 * source.size());                      // normals of the matched rows
        for (std::size_t i = 0; i < source.size(); ++i)
            for (int a = 0; a < 3; ++a) mn[3 * i + a] = normals.row(static_cast<std::size_t>(idx[i]))[a];
        const im::Transformation Ts = im::solve_point_to_plane(source, matched, im::PointCloud(std::move(mn)));
        out.insert(out.end(), Ts.matrix().begin(), Ts.matrix().end());

        im::ICPConfig cfg;                                              // types.hpp:143-148 defaults
        const im::ICPResult r = im::ICP(cfg).align(source, target);     // icp.hpp:157-258
        out.insert(out.end(), r.transformation.matrix().begin(), r.transformation.matrix().end());
        out.push_back(r.converged ? 1.0 : 0.0);
        out.push_back(r.num_iterations);
        out.push_back(r.final_error);
        out.push_back(static_cast<double>(r.error_history.size()));
        out.insert(out.end(), r.error_history.begin(), r.error_history.end());

        std::ofstream f(argv[3], std::ios::binary);
        f.write(reinterpret_cast<const char *>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(double)));
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "mirror_demo: %s\n", e.what());
        return 1;
    }
}

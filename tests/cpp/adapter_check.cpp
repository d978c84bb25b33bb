// Compile-only check of include/slam_icp_adapter.hpp: instantiates every entry point of the
// adapter against the reference's OWN types.  Needs Eigen3 and a reference checkout on the include
// path (-I<reference>/slam_viz/include); where either is missing (this image has no Eigen) the
// translation unit is empty and says so.
#if __has_include(<Eigen/Dense>) && __has_include("slam_viz/core/types.hpp")
#include "slam_icp_adapter.hpp"

int main()
{
    slam::PointCloud a, b;
    slam::ICPConfig cfg;
    slam::ICPResult r = slam::icp_point_to_plane(a, b, cfg);
    slam::ICP icp;
    r = icp.align(a, b);
    slam::OdometryStream stream;
    slam::OdometryStream::Step s = stream.push(a.points(), 0.5, 1000, cfg);
    slam::PointCloud::Matrix v = slam::voxel_downsample_mi355x(a.points(), 0.5);
    icpmi_grid_config grid;
    icpmi_grid_config_default(&grid);
    slam::PointCloud::Matrix world = stream.map_update(slam::Transformation::identity(), &grid);
    std::vector<std::pair<int, int>> cells = stream.occupied_cells();
    slam::PointCloud::Matrix curr = stream.current_scan();
    stream.reset();
    return (r.converged || s.registered || v.rows() > 0 || world.rows() > 0 || !cells.empty() || curr.rows() > 0) ? 1 : 0;
}
#else
#error "adapter_check: <Eigen/Dense> or slam_viz/core/types.hpp not found -- nothing to check here"
#endif

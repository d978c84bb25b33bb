"""MI355X-native point-to-plane ICP registration (gfx950 HIP kernels behind a C ABI).

Scope: the one hot path of kaushik884/LiDAR-SLAM-from-scratch that BASELINE.json names --
slam::icp_point_to_plane (icp.hpp) with its nearest-neighbour search (kdtree.hpp) and
types (types.hpp).  See DESIGN.md.
"""
from .icp import (ICP, ICPConfig, ICPResult, KDTree, NearestNeighborSearch, PointCloud, Transformation,
                  estimate_normals, icp_point_to_plane, solve_point_to_plane)

__all__ = ["ICP", "ICPConfig", "ICPResult", "KDTree", "NearestNeighborSearch", "PointCloud", "Transformation",
           "estimate_normals", "icp_point_to_plane", "solve_point_to_plane"]

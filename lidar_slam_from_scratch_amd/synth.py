"""Seeded synthetic scan pairs.  No dataset ships with the
reference (README.md:155-167 asks the user to download KITTI) and there is no
network here, so every workload is generated: numpy Generator(PCG64(seed)), fp64.

C1  room corner, 5k pts, known SE(3) offset          (BASELINE.json configs[0])
C2  LiDAR-like ray cast of a street scene, ~20k pts  (stand-in for the KITTI pair)
C3  100k -> 100k uniform-random scan                 (roofline bench, configs[2])
C4  1M -> 1M, same recipe as C3                      (configs[3])
C5  synthetic drive, frame stream                    (stand-in for configs[4])
"""
import numpy as np


def rotvec_to_matrix(rv):
    """Rodrigues formula (same form as icp.hpp:127-141)."""
    rv = np.asarray(rv, dtype=np.float64)
    angle = np.linalg.norm(rv)
    if angle < 1e-10:
        return np.eye(3)
    a = rv / angle
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_transform(rotvec, t):
    T = np.eye(4)
    T[:3, :3] = rotvec_to_matrix(rotvec)
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


def apply_transform(T, pts):
    """cloud * R^T + t^T row-wise (types.hpp:110-115)."""
    return pts @ T[:3, :3].T + T[:3, 3]


def invert_transform(T):
    """types.hpp:128-132"""
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


C1_ROTVEC = (0.01, -0.02, 0.03)
C1_TRANS = (0.10, -0.05, 0.02)


def _room_corner(n, seed, noise=0.005, extent=20.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    plane = rng.integers(0, 3, size=n)
    uv = rng.uniform(0.0, extent, size=(n, 2))
    w = rng.normal(0.0, noise, size=n)
    pts = np.empty((n, 3))
    for ax in range(3):
        m = plane == ax
        others = [a for a in range(3) if a != ax]
        pts[m, ax] = w[m]
        pts[m, others[0]] = uv[m, 0]
        pts[m, others[1]] = uv[m, 1]
    return pts


def c1_room_corner(n=5000):
    """Target on three orthogonal 20 m planes; source = an independent sample of the
    same surfaces moved by T^-1.  ICP should return ~T (source -> target)."""
    T = make_transform(C1_ROTVEC, C1_TRANS)
    target = _room_corner(n, seed=1)
    source = apply_transform(invert_transform(T), _room_corner(n, seed=2))
    return source, target, T


def kat1_exact_pair(n=3000, seed=11):
    """KAT-1: noise-free planes, source = T^-1(target) point for point."""
    T = make_transform((0.004, -0.006, 0.008), (0.03, -0.02, 0.01))
    target = _room_corner(n, seed=seed, noise=0.0)
    source = apply_transform(invert_transform(T), target)
    return source, target, T


def c3_uniform(n=100_000, seed=4, perm_seed=5, jitter=0.01):
    """Literal "random scan": target uniform in [-50,50]x[-50,50]x[-3,5] m; source =
    permuted target + N(0, jitter) moved by the C1 transform's inverse."""
    rng = np.random.Generator(np.random.PCG64(seed))
    target = np.empty((n, 3))
    target[:, 0] = rng.uniform(-50, 50, n)
    target[:, 1] = rng.uniform(-50, 50, n)
    target[:, 2] = rng.uniform(-3, 5, n)
    prng = np.random.Generator(np.random.PCG64(perm_seed))
    perm = prng.permutation(n)
    src = target[perm] + prng.normal(0.0, jitter, size=(n, 3))
    T = make_transform(C1_ROTVEC, C1_TRANS)
    source = apply_transform(invert_transform(T), src)
    return source, target, T


def c4_uniform(n=1_000_000):
    return c3_uniform(n=n, seed=6, perm_seed=8)


# --------------------------------------------------------------------------- #
# LiDAR-like ray cast (C2 / C5 stand-in)                                       #
# --------------------------------------------------------------------------- #

def _scene(seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    boxes = []
    for _ in range(12):
        c = np.array([rng.uniform(-60, 60), rng.uniform(-25, 25), 0.0])
        if abs(c[1]) < 4.0:
            c[1] = np.sign(c[1] + 1e-9) * (4.0 + abs(c[1]))
        half = np.array([rng.uniform(1.0, 4.0), rng.uniform(1.0, 4.0), rng.uniform(1.0, 3.0)])
        lo = c - half
        hi = c + half
        lo[2] = -1.73
        hi[2] = -1.73 + 2 * half[2]
        boxes.append((lo, hi))
    walls = [(np.array([-90.0, 30.0, -1.73]), np.array([90.0, 31.0, 6.0])),
             (np.array([-90.0, -31.0, -1.73]), np.array([90.0, -30.0, 6.0])),
             (np.array([85.0, -31.0, -1.73]), np.array([86.0, 31.0, 6.0])),
             (np.array([-86.0, -31.0, -1.73]), np.array([-85.0, 31.0, 6.0]))]
    return boxes + walls


def _raycast(origin, R, scene, beams=64, azimuths=1800, max_range=80.0):
    el = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d_local = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :],
                        np.broadcast_to(se, (beams, azimuths))], axis=-1).reshape(-1, 3)
    d = d_local @ R.T
    t_hit = np.full(d.shape[0], np.inf)
    # ground plane z = -1.73 (sensor height of the KITTI rig) in world frame
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (-1.73 - origin[2]) / d[:, 2]
    ok = (tg > 0) & np.isfinite(tg)
    t_hit[ok] = tg[ok]
    for lo, hi in scene:
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - origin) / d
            t2 = (hi - origin) / d
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmax > 0) & (tmin > 0)
        t_hit = np.where(hit & (tmin < t_hit), tmin, t_hit)
    keep = np.isfinite(t_hit) & (t_hit <= max_range) & (t_hit > 0.5)
    world = origin + d[keep] * t_hit[keep, None]
    return (world - origin) @ R  # sensor frame


def voxel_centroids(points, voxel):
    """Centroid per occupied voxel, key = floor(coord / voxel) (file_utils.cpp:148-196).
    Output is sorted by key: the reference's order is std::unordered_map iteration
    order (implementation-defined); ICP results do not depend on source order beyond
    fp64 summation order."""
    keys = np.floor(points / voxel).astype(np.int64)
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    keys = keys[order]
    pts = points[order]
    change = np.any(np.diff(keys, axis=0) != 0, axis=1)
    starts = np.concatenate([[0], np.nonzero(change)[0] + 1])
    counts = np.diff(np.concatenate([starts, [len(pts)]]))
    sums = np.add.reduceat(pts, starts, axis=0)
    return sums / counts[:, None]


def lidar_pose(frame, step=1.0, yaw_step_deg=1.0, x0=-40.0):
    """Pose of the sensor at `frame`: `step` metres forward per frame from x0, a slow weave and a
    slowly growing yaw.  The defaults are the 12-frame drive of round 1; the 200-frame drive of
    SURVEY 8d (DRIVE_200) starts further back and moves 0.6 m per frame so that it stays inside
    the scene's walls (x in [-85, 85])."""
    yaw = np.deg2rad(yaw_step_deg) * frame * 0.25
    R = rotvec_to_matrix((0.0, 0.0, yaw))
    origin = np.array([x0 + step * frame, 0.5 * np.sin(0.05 * frame), 0.0])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = origin
    return T


DRIVE_200 = dict(step=0.6, yaw_step_deg=0.5, x0=-60.0)


def lidar_frame(frame, voxel=0.5, seed=3, beams=64, azimuths=1800, range_noise=0.01, **drive):
    """One voxel-downsampled scan of the synthetic street, in the sensor frame."""
    scene = _scene(seed)
    T = lidar_pose(frame, **drive)
    pts = _raycast(T[:3, 3], T[:3, :3], scene, beams=beams, azimuths=azimuths)
    rng = np.random.Generator(np.random.PCG64(1000 + frame))
    r = np.linalg.norm(pts, axis=1, keepdims=True)
    pts = pts * (1.0 + rng.normal(0.0, range_noise, size=r.shape) / np.maximum(r, 1e-9))
    return voxel_centroids(pts, voxel) if voxel else pts


def c2_lidar_pair(voxel=0.5, beams=64, azimuths=1800):
    """Frames 0 and 1 of the synthetic drive: source = frame 1, target = frame 0, as
    slam_node.cpp:132-133 does (source = current, target = previous).  Ground truth
    T maps the frame-1 sensor frame into frame 0's."""
    tgt = lidar_frame(0, voxel, beams=beams, azimuths=azimuths)
    src = lidar_frame(1, voxel, beams=beams, azimuths=azimuths)
    T = invert_transform(lidar_pose(0)) @ lidar_pose(1)
    return src, tgt, T


def pose_delta(Ta, Tb):
    """(translation distance [m], rotation angle [rad]) between two 4x4 poses."""
    dt = float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    Rd = Ta[:3, :3].T @ Tb[:3, :3]
    c = (np.trace(Rd) - 1.0) / 2.0
    s = np.linalg.norm([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) / 2.0
    return dt, float(np.arctan2(s, c))

"""The oracle's restatement of SlamNode::update_occupancy_grid (slam_viz/src/ros/slam_node.cpp:211-221)
against a direct Python reading of those lines, on random clouds and on the boundary values of each test.
CPU only; the GPU side is compared with this oracle in tests/test_gpu_occupancy.py."""
import math

import numpy as np


def python_reading(world, sensor, resolution=0.2, height_min=0.3, height_max=2.0, max_range=40.0):
    cells = set()
    np.seterr(over="ignore")
    for x, y, z in np.asarray(world, dtype=np.float64):
        if z < height_min or z > height_max:                     # :214
            continue
        r = math.sqrt((x - sensor[0]) * (x - sensor[0]) + (y - sensor[1]) * (y - sensor[1]))   # :215
        if r > max_range or r < 0.5:                             # :216
            continue
        cx, cy = np.floor(x / resolution), np.floor(y / resolution)     # :217-218
        if not (abs(cx) <= 2147483646.0 and abs(cy) <= 2147483646.0):   # undefined cast in the reference: marks nothing
            continue
        cells.add((int(cx), int(cy)))
    return cells


def test_random_clouds(oracle):
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 3000))
        w = rng.normal(0, 25, (n, 3))
        w[:, 2] = rng.uniform(-1, 3, n)
        sensor = rng.normal(0, 5, 3)
        grid = dict(resolution=float(rng.choice([0.05, 0.2, 0.5, 1.0])), height_min=0.3, height_max=2.0,
                    max_range=float(rng.choice([10.0, 40.0])))
        got = oracle.occupancy_update(set(), w, sensor, **grid)
        assert got == python_reading(w, sensor, **grid)


def test_boundaries_and_specials(oracle):
    res = 0.2
    s = np.array([1.0, -2.0, 0.0])
    w = np.array([
        [5.0, 5.0, 0.3], [5.0, 5.2, 2.0],                       # z on both bounds: kept (strict tests, :214)
        [5.0, 5.4, np.nextafter(0.3, 0)], [5.0, 5.6, np.nextafter(2.0, 3)],   # just outside
        [1.5, -2.0, 1.0], [np.nextafter(1.5, 0), -2.0, 1.0],            # r == 0.5 kept, just under dropped
        [41.0, -2.0, 1.0], [np.nextafter(41.0, 50), -2.0, 1.0],               # r == max_range kept, just over dropped
        [-0.2, -0.2, 1.0], [-0.2000001, 3.0, 1.0], [0.6, 0.6000000000000001, 1.0],   # cell edges, negative side
        [np.nan, 1.0, 1.0], [3.0, np.inf, 1.0], [3.0, 3.0, np.nan],           # NaN z passes :214; NaN / inf quotients mark nothing
        [1e300, 0.0, 1.0], [4e8, 4e8, 1.0],
        [5.0, 5.0, 1.0], [5.01, 5.01, 1.5],                      # duplicates of the first cell
    ])
    cells, keep = oracle.occupancy_cells(w, s, resolution=res)
    want = python_reading(w, s, resolution=res)
    assert set(map(tuple, cells[keep].tolist())) == want
    assert keep.tolist() == [True, True, False, False, True, False, True, False, True, True, True, False, False, True,
                             False, False, True, True]
    assert (25, 25) in want and (-1, -1) in want and (-2, 15) in want
    assert (2, 3) in want           # 0.6 / 0.2 = 2.9999999999999996 in fp64: the cell below, as in the reference
    # z = NaN is not rejected by `z < min || z > max` (both false): the point marks its cell
    assert (15, 15) in want

"""Latency of small filters with the BeamSensorModel: full update cycles."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, OccupancyGrid, se2_from_xytheta

steps = 20
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
for (n, beams) in [(2000, 60), (2000, 180), (2000, 1080), (10000, 180), (15000, 1080), (20000, 1080), (50000, 1080)]:
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), BeamModelParam(beam_max_range=30.0), AmclParams(min_particles=n, max_particles=n), seed=42)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    sub = max(1, 1080 // beams)
    ms = []
    for c in range(steps):
        pts = np.ascontiguousarray(scans[c][::sub])
        f.sync()
        t0 = time.perf_counter()
        assert f.update(controls[c], pts) is not None
        f.sync()
        ms.append((time.perf_counter() - t0) * 1e3)
    print(f"beam model: particles {n} beams {len(pts)}: median {np.median(ms[5:]):.3f} ms per update (min {min(ms[5:]):.3f})", flush=True)
    f.close()

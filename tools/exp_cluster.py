"""Update cycles with beluga_ros::Amcl's estimate (cluster_based_estimate) against the core filter's (estimate): wall time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = 24
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
for n in (2000, 100_000, 1_000_000):
    for kind in (0, 1):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_estimate_kind(kind)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        ms = []
        for c in range(steps):
            f.sync()
            t0 = time.perf_counter()
            f.update(controls[c], scans[c])
            f.sync()
            ms.append((time.perf_counter() - t0) * 1e3)
        print(f"n {n:8d} estimate kind {kind}: median {np.median(ms[6:]):.3f} ms per update (last {ms[-1]:.3f})", flush=True)
        f.close()

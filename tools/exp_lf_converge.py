"""LF kernel time, share of beam groups through an LDS patch and the set's spread, cycle by cycle from the bench's start."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = int(os.environ.get("CYCLES", 60))
cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
f.profile_enable(2)
p0 = t0 = 0
for c in range(cycles):
    f.profile_read(reset=True)
    est = f.update(se2_from_xytheta(*odoms[c]), scans[c])
    f.sync()
    p = f.profile_read(reset=True)
    planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
    cov = np.asarray(est[1])
    print(c, "lf_us", round(1e3 * p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1), 1), "through", round((through - t0) / max(planned - p0, 1), 4),
          "sigma", [round(float(np.sqrt(cov[i, i])), 4) for i in range(3)], flush=True)
    p0, t0 = planned, through
f.close()

"""Per-cycle series of the bench workload: LF kernel time (HIP events, every cycle), patch fraction of that launch, cloud sigmas."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = 40
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
n = 1_000_000
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
if os.environ.get("LOOSE"):
    f.set_option("lf_loose_below", int(os.environ["LOOSE"]))
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
f.profile_enable(2)
p0 = t0 = 0
for c in range(steps):
    f.profile_read(reset=True)
    est = f.update(controls[c], scans[c])
    f.sync()
    p = f.profile_read(reset=True)
    planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
    frac = (through - t0) / max(planned - p0, 1)
    p0, t0 = planned, through
    cov = est[1]
    print(f"cycle {c:2d} lf_ms {p['sensor_kernel'][0]:.4f} patch_frac {frac:.4f} sigma_x {np.sqrt(cov[0,0]):.4f} sigma_y {np.sqrt(cov[1,1]):.4f} sigma_t {np.sqrt(cov[2,2]):.4f}", flush=True)
f.close()

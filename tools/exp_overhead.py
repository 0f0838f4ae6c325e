"""Per-cycle host/launch overhead: the default bench workload with tiny particle counts."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells, truth, odoms, scans, _poses = bench.make_workload(60)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
for n, beams in [(20000, 1080), (20000, 8), (1_000_000, 8)]:
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=1)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    ctrl = [se2_from_xytheta(*o) for o in odoms]
    for c in range(5):
        f.update(ctrl[c], scans[c][:beams])
    f.profile_enable(True); f.profile_read(reset=True)
    t0 = time.perf_counter()
    for c in range(5, 55):
        f.update(ctrl[c], scans[c][:beams])
    dt = (time.perf_counter() - t0) / 50
    prof = f.profile_read()
    print(n, beams, "ms/cycle", round(dt * 1e3, 4), {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()})
    f.close()

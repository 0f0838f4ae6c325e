"""Where the host spends a fixed-size cycle (mcl_get_counter host_ns_*), next to the wall time per cycle, for the bench workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = 60
n = int(os.environ.get("N", 1_000_000))
cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [np.ascontiguousarray(se2_from_xytheta(*o)) for o in odoms]
scans = [np.ascontiguousarray(s, dtype=np.float64) for s in scans]
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
for c in range(20):
    f.update(controls[c], scans[c])
names = ["host_ns_to_first_launch", "host_ns_other_launches", "host_ns_wait", "host_ns_after_wait", "host_cycles"]
before = [f.counter(k) for k in names]
t0 = time.perf_counter()
for c in range(20, cycles):
    f.update(controls[c], scans[c])
wall = time.perf_counter() - t0
after = [f.counter(k) for k in names]
k = after[4] - before[4]
print(f"{k} cycles, wall {wall / k * 1e6:.1f} us per cycle; inside mcl_update: " +
      ", ".join(f"{nm[8:]} {(a - b) / k / 1e3:.1f} us" for nm, a, b in zip(names[:4], after, before)) +
      f"; outside the library {wall / k * 1e6 - sum(a - b for a, b in zip(after[:4], before[:4])) / k / 1e3:.1f} us")
f.close()

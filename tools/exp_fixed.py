"""Fixed-size filter of N particles on the bench workload (multinomial resample every cycle): a few cycles, for kernel traces.
Usage: N=10000000 python tools/exp_fixed.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(os.environ.get("N", 10_000_000))
cells, truth, odoms, scans, _poses = bench.make_workload(steps + 1)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
ms = []
for c in range(steps):
    f.sync()
    t0 = time.perf_counter()
    assert f.update(controls[c], scans[c]) is not None
    f.sync()
    ms.append((time.perf_counter() - t0) * 1e3)
print("N", n, "ms per cycle", [round(x, 3) for x in ms], flush=True)
f.close()

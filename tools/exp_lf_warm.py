"""Is the LF kernel's slow first window a matter of cold caches?  At chosen cycles of the bench's run the same reweight (same cloud, same scan) is
launched four times in a row on a copy of the filter's state: the first finds the caches as the cycle leaves them, the others warm."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = 46
at = [8, 12, 16, 24, 45]
n = 1_000_000
cells, truth, odoms, scans, _poses = bench.make_workload(cycles + 1)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
mk = lambda: Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f = mk()
g = mk()
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
g.initialize(truth, np.diag([0.25, 0.25, 0.04]))
g.profile_enable(2)
f.profile_enable(2)
for c in range(cycles):
    if c in at:
        # the cloud as this cycle's LF kernel sees it: propagate a copy, then reweight it repeatedly (weights do not matter for the time)
        states, w = f.particles()
        g.set_particles(states, w)
        g.propagate(se2_from_xytheta(*odoms[c]), se2_from_xytheta(*odoms[c - 1]), c + 1)
        times = []
        for rep in range(4):
            g.profile_read(reset=True)
            g.reweight(scans[c])
            g.sync()
            p = g.profile_read(reset=True)
            times.append(1e3 * p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
        f.profile_read(reset=True)
    est = f.update(se2_from_xytheta(*odoms[c]), scans[c])
    if c in at:
        f.sync()
        p = f.profile_read(reset=True)
        print(f"cycle {c}: in the cycle {1e3 * p['sensor_kernel'][0] / max(p['sensor_kernel'][1], 1):6.1f} us; the same reweight four times in a row: " + " ".join(f"{t:6.1f}" for t in times), flush=True)
f.close(); g.close()

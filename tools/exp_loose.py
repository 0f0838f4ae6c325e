"""LF kernel time over cycles 5 .. 24 of the bench workload (the driver's timed window) and once settled (cycles 40 .. 59), by the
share of fitting groups below which a workgroup gathers everything (option lf_loose_below, in 256ths)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = 60
cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
for loose in [int(v) for v in sys.argv[1:]] or [176, 192, 208, 224]:
    rows = []
    for rep in range(2):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_option("lf_loose_below", loose)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        f.profile_enable(2)
        ms = []
        for c in range(cycles):
            f.profile_read(reset=True)
            f.update(se2_from_xytheta(*odoms[c]), scans[c])
            f.sync()
            p = f.profile_read(reset=True)
            ms.append(p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
        f.close()
        rows.append((np.mean(ms[5:25]), np.mean(ms[40:60])))
    print(f"lf_loose_below {loose}: LF ms cycles 5-24 {[round(r[0], 4) for r in rows]}, cycles 40-59 {[round(r[1], 4) for r in rows]}", flush=True)

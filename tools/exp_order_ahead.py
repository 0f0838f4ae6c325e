"""Cycle rate with / without the order computed a cycle ahead, with the bench's stage profiling mode (events around the sensor kernel of every
4th cycle) and without; counters of how often the order / the normals drawn ahead were used."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells, truth, odoms, scans, _ = bench.make_workload(70)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
n = 1_000_000
for rep in range(2):
    for prof in (0, 1):
        for ahead in (1, 0):
            f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
            f.set_option("order_ahead", ahead)
            f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
            for c in range(5):
                f.update(controls[c], scans[c])
            f.profile_enable(prof)
            rates = []
            for w0 in (5, 25, 45):
                f.sync()
                t = time.perf_counter()
                for c in range(w0, w0 + 20):
                    f.update(controls[c], scans[c])
                f.sync()
                rates.append(20.0 / (time.perf_counter() - t))
            print(f"rep {rep} profile {prof} order_ahead {ahead}: {' '.join(f'{r:.1f}' for r in rates)}  order used {f.counter('order_ahead_used')} missed {f.counter('order_ahead_missed')} normals used {f.counter('noise_ahead_used')}", flush=True)
            f.close()

"""Beam model, wave-per-particle kernel over the grid maps vs the ordered kernel with scan segments: update time by set size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, OccupancyGrid, se2_from_xytheta

steps = 16
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
for beams in (180, 1080):
    for n in (4000, 8000, 16000, 32000, 64000):
        row = []
        for name, thr in (("wave", 1 << 30), ("ordered", 0)):
            f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), BeamModelParam(beam_max_range=30.0), AmclParams(min_particles=n, max_particles=n), seed=42)
            f.set_option("beam_sort_min_particles", thr)
            f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
            sub = max(1, 1080 // beams)
            ms = []
            for c in range(steps):
                pts = np.ascontiguousarray(scans[c][::sub])
                f.sync()
                t0 = time.perf_counter()
                f.update(controls[c], pts)
                f.sync()
                ms.append((time.perf_counter() - t0) * 1e3)
            row.append(f"{name} {np.median(ms[4:]):.3f}")
            f.close()
        print(f"beams {beams} particles {n}: " + "  ".join(row), flush=True)

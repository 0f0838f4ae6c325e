"""Mid-size filters: LF kernel time by variant (default = ordered lanes + LDS patches with segments, lf_patch=0 = ordered gather,
lf_variant=3 = wave per particle with the lanes over the beams, no ordering) after a few cycles of the bench workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = 24
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
for n in (20_000, 50_000, 100_000, 200_000, 400_000):
    for name, opts in (("default", {}), ("gather", {"lf_patch": 0}), ("beams", {"lf_variant": 3})):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
        for k, v in opts.items():
            f.set_option(k, v)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        ms = []
        for c in range(steps):
            f.sync()
            t0 = time.perf_counter()
            f.update(controls[c], scans[c])
            f.sync()
            ms.append((time.perf_counter() - t0) * 1e3)
        f.profile_enable(2)
        f.profile_read(reset=True)
        for c in range(8):
            f.update(controls[c], scans[c])
        f.sync()
        prof = f.profile_read(reset=True)
        planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
        print(f"n {n:7d} {name:8s} update {np.median(ms[8:]):.3f} ms  sensor_kernel {prof['sensor_kernel'][0] / max(prof['sensor_kernel'][1], 1):.4f}  reweight {prof['reweight'][0] / max(prof['reweight'][1], 1):.4f}  patch frac {through / max(planned, 1):.3f}", flush=True)
        f.close()

"""Latency of small filters (the reference's usual sizes): full update cycles, KLD-adaptive and fixed, per-stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = 30
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
for (lo, hi, beams) in [(500, 2000, 180), (2000, 2000, 180), (2000, 2000, 1080), (10000, 10000, 1080), (50000, 50000, 1080), (100000, 100000, 1080)]:
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=lo, max_particles=hi), seed=42)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    sub = max(1, 1080 // beams)
    ms = []
    for c in range(steps):
        pts = np.ascontiguousarray(scans[c][::sub])
        f.sync()
        t0 = time.perf_counter()
        assert f.update(controls[c], pts) is not None
        f.sync()
        ms.append((time.perf_counter() - t0) * 1e3)
    f.profile_enable(2)
    f.profile_read(reset=True)
    for c in range(8):
        f.update(controls[c], np.ascontiguousarray(scans[c][::sub]))
    f.sync()
    prof = f.profile_read(reset=True)
    print(f"particles {lo}..{hi} beams {len(pts)}: median {np.median(ms[5:]):.3f} ms per update (min {min(ms[5:]):.3f}); now {f.last_info['num_particles']} particles; stages",
          {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}, flush=True)
    f.close()

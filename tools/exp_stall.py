"""Are there cycles that take milliseconds longer than their neighbours?  One filter of the bench's configuration, CYCLES cycles timed one by
one (the trajectory repeated), with the bench's stage events (profile 1: HIP events around the sensor kernel of every 4th cycle), without
them (0) and with the per-stage events (2); every cycle beyond three times the median is listed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = 60
cells, truth, odoms, scans, _ = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
# (the odometry of a repeated trajectory: increments replayed on a running pose)
incs = []
prev = se2_from_xytheta(*odoms[0])
for o in odoms:
    incs.append(o)
for prof in (1, 0, 1, 0, 2):
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    f.profile_enable(prof)
    ms = []
    c = 0
    forward = True
    for k in range(cycles):
        # walk the recorded trajectory forwards and backwards (the scans of the same poses: the filter stays localised)
        idx = c
        t0 = time.perf_counter()
        f.update(se2_from_xytheta(*odoms[idx]), scans[idx])
        ms.append((time.perf_counter() - t0) * 1e3)
        if forward:
            c += 1
            if c == steps - 1:
                forward = False
        else:
            c -= 1
            if c == 0:
                forward = True
        if prof and k % 64 == 63:
            f.profile_read(reset=True)
    ms = np.asarray(ms[30:])
    med = float(np.median(ms))
    slow = [(int(i) + 30, round(float(v), 2)) for i, v in enumerate(ms) if v > 3 * med]
    print(f"profile {prof}: {len(ms)} cycles, median {med:.3f} ms, p99 {np.percentile(ms, 99):.3f}, max {ms.max():.2f}; beyond 3 x median: {len(slow)} {slow[:12]}", flush=True)
    f.close()

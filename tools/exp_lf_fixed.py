"""LF reweight on a FIXED synthetic cloud (normal around the bench's true pose), timed by the library's HIP events: the timing
does not depend on what the kernel computes, so timing-only what-if builds can be compared.  Usage: exp_lf_fixed.py [sx sy st]..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cells, truth, odoms, scans, _poses = bench.make_workload(2)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = int(os.environ.get("N", 1_000_000))
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
for sig in [(0.33, 0.14, 0.11), (0.1, 0.2, 0.107), (0.05, 0.05, 0.02)]:
    states = synth.normal_particles(n, truth, sig, seed=9)
    w0 = np.ones(n)
    ms = []
    for rep in range(6):
        f.set_particles(states, w0)
        f.profile_enable(2)
        f.profile_read(reset=True)
        f.reweight(scans[0])
        f.sync()
        p = f.profile_read(reset=True)
        ms.append(p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
    planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
    print("sigma", sig, "lf_ms", [round(x, 4) for x in ms[1:]], "median", round(float(np.median(ms[1:])), 4), "patch frac (running)", round(through / max(planned, 1), 4), flush=True)
f.close()

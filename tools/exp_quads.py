"""Does the LF kernel's time follow the number of lines a quad of lanes touches?  Same kernel, same particle count, but
every pose repeated 1 / 2 / 4 / 8 times (repeats are adjacent after the spatial ordering)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells, truth, odoms, scans, _poses = bench.make_workload(4)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=1)
for rep in (1, 2, 4, 8):
    base = synth.normal_particles(n // rep, truth, (0.5, 0.5, 0.2), seed=5)
    states = np.repeat(base, rep, axis=0)
    f.set_particles(states, np.ones(n))
    f.reweight(scans[0])
    f.profile_enable(True); f.profile_read(reset=True)
    for _ in range(5):
        f.reweight(scans[0])
    f.sync()
    p = f.profile_read()
    print("repeat", rep, "LF kernel ms", round(p["sensor_kernel"][0] / p["sensor_kernel"][1], 4), "reweight ms", round(p["reweight"][0] / p["reweight"][1], 4))

"""Beam model on a dispersed 1M-particle set (initialize_from_map on the bench map): the ordered kernel under the heading-major
and the position-major ordering key (option key_layout), sensor kernel time by the library's HIP events."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, OccupancyGrid, se2_from_xytheta

cells, truth, odoms, scans, _poses = bench.make_workload(2)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = int(os.environ.get("N", 1_000_000))
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), BeamModelParam(beam_max_range=bench.MAX_RANGE),
         AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize_from_map()
states, w0 = f.particles()
ref = None
for layout in (0, 1, 0, 1):
    f.set_option("key_layout", layout)
    ms = []
    for rep in range(2):
        f.set_particles(states, w0)
        f.profile_enable(2)
        f.profile_read(reset=True)
        f.reweight(scans[0])
        f.sync()
        p = f.profile_read(reset=True)
        ms.append(p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
    w = f.particles()[1]
    if ref is None:
        ref = w
    print("key_layout", layout, "sensor_kernel_ms", [round(x, 2) for x in ms], "identical to first", bool(np.array_equal(w, ref)),
          "cells visited", f.beam_cells_visited(), flush=True)
f.close()

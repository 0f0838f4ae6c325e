"""N update cycles of the bench workload (BASELINE configs[1]) on one filter - something for rocprofv3 to trace:
    rocprofv3 --kernel-trace --stats -d out -o t -- python tools/exp_cycles.py [--cycles 30] [--particles 1000000] [opt=val ...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=30)
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("options", nargs="*")
args = ap.parse_args()
cells, truth, odoms, scans, _poses = bench.make_workload(args.cycles)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = args.particles
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
for kv in args.options:
    k, _, v = kv.partition("=")
    f.set_option(k, int(v))
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
for c in range(args.cycles):
    assert f.update(se2_from_xytheta(*odoms[c]), scans[c]) is not None
f.sync()
f.close()

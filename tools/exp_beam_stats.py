"""Measurement build (-DMCL_BEAM_STATS): per wave-beam frequencies of the stages of the beam kernel's walk on config 5."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd import capi
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, OccupancyGrid, se2_from_xytheta

cells, truth, odoms, scans, _poses = bench.make_workload(4)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
n = 1_000_000
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), BeamModelParam(beam_max_range=30.0), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
lib = capi.load()
out = (C.c_ulonglong * 32)()
for c in range(1):
    f.update(controls[c], scans[c])
    f.sync()
    assert lib.mcl_debug_beam_stats(out, 1) == 0
    v = list(out)
    beams_w = max(v[0], 1)
    names = ["beams", "walks, shared axis", "walks, mixed axes", "block columns", "columns examined", "columns with a hit", "tail groups",
             "window walks", "grid walks", "skips", "certified starts", "skips behind a skip", "skips of 15+ columns", "-", "head cells examined", "tail cells examined"]
    print(f"cycle {c}:")
    for i, name in enumerate(names):
        if v[2 * i] == 0:
            continue
        print(f"  {name:22s} waves {v[2*i]:14d} ({v[2*i]/beams_w:8.3f} per wave-beam)   lanes {v[2*i+1]:16d} ({v[2*i+1]/max(v[2*i],1):6.2f} lanes per wave event)")
f.close()

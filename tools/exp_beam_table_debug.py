import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, OccupancyGrid, se2_from_xytheta
W, H, res = 1500, 1200, 0.05
origin = se2_from_xytheta(-37.5, -30.0, 0.0)
cells = np.zeros((H, W), dtype=np.int8)
cells[:] = -1
cells[150:-150, 200:-200] = 0
cells[600, 300:900] = 100
grid = OccupancyGrid(cells=cells, resolution=res, origin=origin)
centre = (0.0, 0.0, 0.3)
angles = synth.lidar_angles(61, 360.0)
ranges = synth.cast_scan(cells, res, (-37.5, -30.0), centre, angles, 40.0, 0.01, 1)
pts = synth.scan_points(ranges, angles)
pts[::7] *= 3.0
n = 20000
states = synth.normal_particles(n, centre, (8.0, 8.0, 1.5), seed=5)
states[:5, 2] += 500.0
out = []
for table in (1, 0):  # noqa
    f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), BeamModelParam(beam_max_range=40.0), AmclParams(min_particles=n, max_particles=n), seed=11)
    f.set_option("beam_table", table)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    out.append(f.particles()[1].copy())
    f.close()
from oracle import binding as orc
want = orc.beam_weights(grid.cells, res, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 40.0), states, pts, threads=orc.max_threads())
print("table vs oracle mismatches", int((np.abs(out[0] - want) > 1e-10 * np.abs(want)).sum()), " inline vs oracle", int((np.abs(out[1] - want) > 1e-10 * np.abs(want)).sum()))
print("ranges", np.round(np.hypot(pts[:, 0], pts[:, 1]), 3))
bad = np.nonzero(np.abs(out[0] - out[1]) > 1e-10 * np.abs(out[1]))[0]
print(len(bad), "mismatches")
for i in bad[:12]:
    cx, cy = (states[i, 2] + 37.5) / res, (states[i, 3] + 30.0) / res
    print(i, "cell", round(cx, 1), round(cy, 1), "occupancy", cells[int(cy), int(cx)] if 0 <= cy < H and 0 <= cx < W else None, "table", out[0][i], "inline", out[1][i])

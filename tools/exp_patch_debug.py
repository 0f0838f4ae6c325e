"""Patch kernel vs gather kernel, bit for bit, under the planner's options (debugging aid)."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
angles = synth.lidar_angles(1080, 270.0)
ranges = synth.cast_scan(cells, 0.05, (-10.0, -10.0), truth, angles, 12.0, 0.01, 1)
pts = synth.scan_points(ranges, angles)
LF = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
n = 200_000
ref = None
for split, margin, curve in itertools.product((0, 3), (1,), (1,)):
    ws = []
    for patch in (2, 0):
        f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), LF, AmclParams(min_particles=n, max_particles=n), seed=11)
        f.set_option("lf_patch", patch)
        f.set_option("lf_split", split)
        f.set_option("lf_margin", margin)
        f.set_option("key_curve", curve)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        f.reweight(pts)
        ws.append(f.particles()[1].copy())
        if patch == 2:
            planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
        f.close()
    bad = ws[0] != ws[1]
    print(f"split {split} margin {margin} curve {curve}: mismatches {int(bad.sum())} of {n}, max rel {np.max(np.abs(ws[0]-ws[1])/ws[1]):.3e}, through {through}/{planned}", flush=True)

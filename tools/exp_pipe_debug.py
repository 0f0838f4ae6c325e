"""Which blocks of k_reweight_lf_pipe differ from the gather kernel (debugging aid): block index = position in the spatial order // 448."""
import sys
import numpy as np
sys.path.insert(0, ".")
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
beams = int(sys.argv[2]) if len(sys.argv) > 2 else 8
grid_wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 7
sigma = (0.1, 0.1, 0.03)
cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
angles = synth.lidar_angles(beams, 270.0)
pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-10.0, -10.0), truth, angles, 12.0, 0.01, 1), angles)
LF = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
out = []
for patch in (2, 0):
    f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), LF, AmclParams(min_particles=n, max_particles=n), seed=11)
    f.set_option("lf_small_particles", 16384)
    f.set_option("lf_patch", patch)
    f.set_option("lf_pipe_grid", grid_wgs)
    f.initialize(truth, np.diag([s * s for s in sigma]))
    f.reweight(pts)
    w = f.particles()[1].copy()
    perm = f.debug_order()[0] if patch == 2 else None
    out.append((w, perm))
    print("patch", patch, "pipe launches", f.counter("lf_pipe_launches"), "planned/through", f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through"))
    f.close()
bad = np.nonzero(out[0][0] != out[1][0])[0]
print("mismatches", len(bad))
if len(bad):
    perm = np.asarray(out[0][1])
    pos = np.empty(n, dtype=np.int64)
    pos[perm] = np.arange(n)
    blocks = np.unique(pos[bad] // 448, return_counts=True)
    nblocks = (n + 447) // 448
    print("blocks (of %d), count:" % nblocks, list(zip(blocks[0].tolist(), blocks[1].tolist()))[:40])
    print("block mod grid:", sorted(set((blocks[0] % max(grid_wgs, 1)).tolist())), "k index:", sorted(set((blocks[0] // max(grid_wgs, 1)).tolist()))[:20])
    r = out[0][0][bad] / out[1][0][bad]
    print("ratio pipe/gather: min %.6g max %.6g" % (r.min(), r.max()))

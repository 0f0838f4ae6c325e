import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "turtlebot3_world_grid.npz"))
ox, oy, ot = z["origin_xytheta"]
cells, res = z["cells"], float(z["resolution"])
grid = OccupancyGrid(cells=cells, resolution=res, origin=se2_from_xytheta(ox, oy, ot))
print("origin", ox, oy, ot, "free", int((cells == 0).sum()))
for seed in (4, 5, 6):
    truth = synth.find_free_pose(cells, res, (ox, oy), seed=seed, clearance_cells=8)
    f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True), AmclParams(min_particles=2000, max_particles=2000), seed=1)
    f.initialize(truth, np.diag([0.04, 0.04, 0.01]))
    angles = synth.lidar_angles(180, 360.0)
    pose, odom = truth, (0.0, 0.0, 0.0)
    print("truth0", truth)
    for c in range(8):
        pose = synth.odometry_step(pose, 0.3, 0.05)
        odom = synth.odometry_step(odom, 0.3, 0.05)
        r = synth.cast_scan(cells, res, (ox, oy), pose, angles, 3.5, 0.01, seed=200 + c)
        e = f.update(se2_from_xytheta(*odom), synth.scan_points(r, angles))
        cx, cy = int((pose[0]-ox)/res), int((pose[1]-oy)/res)
        print(c, "truth", np.round(pose, 3), "cell", cells[cy, cx] if 0<=cy<384 and 0<=cx<384 else None, "est", round(e[0][2],3), round(e[0][3],3), round(math.atan2(e[0][1], e[0][0]),3), "ranges min/max", r.min().round(2), r.max().round(2))
    f.close()

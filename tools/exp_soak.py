"""Soak: contexts of changing sizes and models created, run and closed in a loop; device memory before and after (a leak shows as a trend),
then one filter over thousands of cycles (KLD-adaptive, so the set's size keeps changing).  Usage: python tools/exp_soak.py [contexts] [cycles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beluga_amd import synth
from beluga_amd.amcl import (Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid,
                             se2_from_xytheta)

contexts = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(5)
cells = synth.make_rooms_map(600, 600, seed=3, n_rooms=14)
grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-15.0, -15.0, 0.0))
truth = synth.find_free_pose(cells, 0.05, (-15.0, -15.0), seed=4, clearance_cells=8)
angles = synth.lidar_angles(360, 360.0)
pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-15.0, -15.0), truth, angles, 12.0, 0.01, 1), angles)
free0 = torch.cuda.mem_get_info()[0]
marks = []
for k in range(contexts):
    n = int(rng.choice([500, 2000, 20_000, 70_000, 300_000, 1_000_000]))
    sensor = BeamModelParam(beam_max_range=12.0) if k % 3 == 2 else LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, k % 2 == 0)
    kld = k % 4 == 1
    p = AmclParams(min_particles=max(100, n // 10) if kld else n, max_particles=n)
    f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), sensor, p, seed=k)
    if k % 5 == 0:
        f.initialize_from_map()
    else:
        f.initialize(truth, np.diag([0.05, 0.05, 0.01]))
    pose = np.array(truth, dtype=np.float64)
    for c in range(4):
        pose = pose + np.array([0.3, 0.0, 0.05])
        est = f.update(se2_from_xytheta(*pose), pts)
        assert est is None or np.all(np.isfinite(est[0]))
    if k % 7 == 0:
        f.update_map(grid)
    f.close()
    if k % 20 == 19:
        torch.cuda.synchronize()
        marks.append(free0 - torch.cuda.mem_get_info()[0])
print("device memory in use beyond the start, every 20 contexts (MB):", [round(m / 2**20, 1) for m in marks], flush=True)
f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2),
         AmclParams(min_particles=500, max_particles=200_000, update_min_d=0.0, update_min_a=0.0), seed=9)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
t0 = time.perf_counter()
sizes = []
pose = np.array(truth, dtype=np.float64)
for c in range(cycles):
    pose = pose + np.array([0.02 * np.cos(c / 50), 0.02 * np.sin(c / 50), 0.01])
    est = f.update(se2_from_xytheta(*pose), pts)
    assert est is not None and np.all(np.isfinite(est[0])) and np.all(np.isfinite(est[1]))
    if c % (cycles // 10) == 0:
        sizes.append(f.num_particles())
f.sync()
print(f"{cycles} KLD cycles in {time.perf_counter() - t0:.1f} s; particles at tenths of the run: {sizes}; memory now {round((free0 - torch.cuda.mem_get_info()[0]) / 2**20, 1)} MB")
f.close()
torch.cuda.synchronize()
print("after the last close (MB):", round((free0 - torch.cuda.mem_get_info()[0]) / 2**20, 1))

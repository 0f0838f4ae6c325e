"""Experiment: how much does spatial ordering of the particles buy the likelihood-field kernels?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

def morton3(bx, by, bt, bits=10):
    key = np.zeros(len(bx), dtype=np.uint64)
    for b in range(bits):
        key |= ((bx >> b) & 1).astype(np.uint64) << np.uint64(3 * b)
        key |= ((by >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 1)
        key |= ((bt >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 2)
    return key

cells = synth.make_rooms_map(4000, 4000, seed=42)
grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
angles = synth.lidar_angles(1080, 270.0)
pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-100.0, -100.0), truth, angles, 30.0, 0.01, 1), angles)
n = 1_000_000
states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=9)
theta = np.arctan2(states[:, 1], states[:, 0])
orders = {"random": np.arange(n)}
for name, (rx, rt) in {"morton_xy0.1_t0.005": (0.1, 0.005), "morton_xy0.25_t0.004": (0.25, 0.004), "morton_xy0.05_t0.02": (0.05, 0.02)}.items():
    bx = np.floor((states[:, 2] - states[:, 2].min()) / rx).astype(np.int64)
    by = np.floor((states[:, 3] - states[:, 3].min()) / rx).astype(np.int64)
    bt = np.floor((theta - theta.min()) / rt).astype(np.int64)
    orders[name] = np.argsort(morton3(bx, by, bt), kind="stable")
orders["theta_only"] = np.argsort(theta, kind="stable")
lf = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
for variant in ("0", "1"):
    os.environ["BELUGA_MCL_LF_VARIANT"] = variant
    f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), lf, AmclParams(min_particles=n, max_particles=n), seed=3)
    f.profile_enable(True)
    for name, order in orders.items():
        f.set_particles(states[order], np.ones(n))
        for _ in range(2):
            f.reweight(pts)
        f.sync(); f.profile_read(reset=True)
        for _ in range(5):
            f.reweight(pts)
        f.sync()
        ms, cnt = f.profile_read(reset=True)["reweight"]
        print(f"variant {variant} order {name:24s} reweight {ms / cnt:.3f} ms", flush=True)
    f.close()

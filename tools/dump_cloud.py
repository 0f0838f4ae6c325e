"""Dumps the particle cloud of the default bench workload as the LF kernel of cycle W + 1 sees it - float32 (x, y, theta) -, the
device's ordering of it (perm) and that cycle's scan to gpurun_out/cloud*.npy, for offline studies of the patch planner.
Usage: python tools/dump_cloud.py [--particles N] [--cycles W]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--cycles", type=int, default=8)
ap.add_argument("--out", default="gpurun_out/cloud.npy")
args = ap.parse_args()
cells, truth, odoms, scans, _poses = bench.make_workload(args.cycles + 2)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
p = AmclParams(min_particles=args.particles, max_particles=args.particles)
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), p, seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
controls = [se2_from_xytheta(*o) for o in odoms]
for c in range(args.cycles):
    f.update(controls[c], scans[c])
# one more propagation: the cloud as the reweight kernel of the next cycle sees it, and the order the library gives it
f.propagate(controls[args.cycles], controls[args.cycles - 1], args.cycles + 1)
states, w = f.particles()
perm, keys = f.debug_order()
out = np.stack([states[:, 2], states[:, 3], np.arctan2(states[:, 1], states[:, 0])], axis=1).astype(np.float32)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
np.save(args.out, out)
np.save(args.out.replace(".npy", "_perm.npy"), np.asarray(perm, dtype=np.uint32))
np.save(args.out.replace(".npy", "_scan.npy"), np.asarray(scans[args.cycles], dtype=np.float32))
print("saved", out.shape, "std", out.std(axis=0), "truth", truth)

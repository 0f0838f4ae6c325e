"""Timing of the other BASELINE.json configs on one GPU (not the headline bench):
  config 3: <=10M particles, KLD (eps .05, z 3) + selective resampling (ESS < N/2)
  config 5: 1M particles x 1080 beams, BeamSensorModel (Bresenham on the int8 grid), beam_max_range 30
Usage: python tools/bench_configs.py [3] [5] [--particles N]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["3", "5"])
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--particles5", type=int, default=1_000_000)
ap.add_argument("--max3", type=int, default=10_000_000)
args = ap.parse_args()
cells, truth, odoms, scans, _poses = bench.make_workload(args.steps + 2)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
motion = DifferentialDriveModelParam(*bench.ALPHAS)

def run(name, filt, steps):
    filt.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    filt.profile_enable(True)
    rows = []
    for c in range(steps):
        filt.sync()
        t0 = time.perf_counter()
        est = filt.update(controls[c], scans[c])
        filt.sync()
        dt = time.perf_counter() - t0
        prof = filt.profile_read(reset=True)
        extra = {"cells_visited": filt.beam_cells_visited()} if name == "config5" else {}
        rows.append({"cycle": c, "ms": dt * 1e3, **extra, "n_after": filt.last_info["num_particles"], "resampled": filt.last_info["resampled"],
                     "ess": filt.last_info["ess"], **{k: round(v[0], 3) for k, v in prof.items()}})
        print(name, json.dumps(rows[-1]), flush=True)
    return rows

if "3" in args.configs:
    p = AmclParams(min_particles=100_000, max_particles=args.max3, selective_resampling=True)
    f = Amcl(grid, motion, LikelihoodFieldModelParam(**bench.LF), p, seed=42)
    run("config3", f, args.steps)
    f.close()
if "5" in args.configs:
    n = args.particles5
    p = AmclParams(min_particles=n, max_particles=n)
    f = Amcl(grid, motion, BeamModelParam(beam_max_range=30.0), p, seed=42)
    run("config5", f, args.steps if os.environ.get("CONFIG5_ALL_STEPS") else min(args.steps, 4))
    f.close()

"""A/B of option sets on the bench workload (BASELINE configs[1]; fresh filter per run, same trajectory, the variants alternating):
LF kernel time (HIP events around the sensor kernel of every cycle) and the share of the beam groups through an LDS patch, cycle by
cycle; means over the cycles the driver's bench times (5 .. 24) and over the settled cloud (the last 10); and the whole cycle's rate
over windows of 20 cycles WITHOUT the events (a second pass).

    python tools/exp_r5_ab.py [--cycles 45] [--particles 1000000] [--reps 2] "name:opt=val,opt=val" "name2:..." ...
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=45)
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

variants = []
for v in args.variants:
    name, _, rest = v.partition(":")
    opts = {}
    for kv in filter(None, rest.split(",")):
        k, _, val = kv.partition("=")
        opts[k] = int(val)
    variants.append((name, opts))

steps, n = args.cycles, args.particles
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]


def make(opts):
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in opts.items():
        f.set_option(k, v)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    return f


for rep in range(args.reps):
    for name, opts in variants:
        f = make(opts)
        f.profile_enable(2)
        p0 = t0 = 0
        lf, frac = [], []
        for c in range(steps):
            f.profile_read(reset=True)
            f.update(controls[c], scans[c])
            f.sync()
            p = f.profile_read(reset=True)
            planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
            frac.append((through - t0) / max(planned - p0, 1))
            p0, t0 = planned, through
            lf.append(p["sensor_kernel"][0])
        f.close()
        f = make(opts)  # the whole cycle, no events
        rates = []
        for w0 in range(5, steps - 19, 20):
            if w0 == 5:
                for c in range(5):
                    f.update(controls[c], scans[c])
            f.sync()
            t = time.perf_counter()
            for c in range(w0, w0 + 20):
                f.update(controls[c], scans[c])
            f.sync()
            rates.append(20.0 / (time.perf_counter() - t))
        f.close()
        sel = slice(5, min(25, steps))
        print(f"rep {rep} {name:28s} n {n}: LF ms mean[5:25] {np.mean(lf[sel]):.4f} last10 {np.mean(lf[-10:]):.4f} | patch share [5:25] {np.mean(frac[sel]):.4f} "
              f"last10 {np.mean(frac[-10:]):.4f} | cycles/s per window of 20 from cycle 5: {' '.join(f'{r:.1f}' for r in rates)}", flush=True)
        print("   lf:", " ".join(f"{v:.3f}" for v in lf), flush=True)

"""A/B of the LF kernel's planner / ordering options on the bench workload (fresh filter per variant, same trajectory):
LF kernel time (HIP events) and share of the beam groups through an LDS patch, cycle by cycle, and their means over the cycles
the driver's bench times (5 .. 24).   python tools/exp_lf_ab.py [cycles]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
n = 1_000_000
R2 = dict(key_curve=0, key_bits_xy=6, lf_margin=0, lf_split=0)
VARIANTS = [
    ("round 2 (morton 6/6/8, isotropic margin, no split)", R2),
    ("+ hilbert", dict(R2, key_curve=1)),
    ("+ hilbert + auto bits", dict(R2, key_curve=1, key_bits_xy=0)),
    ("+ hilbert + auto bits + per-axis margin", dict(R2, key_curve=1, key_bits_xy=0, lf_margin=1)),
    ("all (defaults): + split patches", {}),
    ("all, bits 5", dict(key_bits_xy=5)),
    ("all, bits 6", dict(key_bits_xy=6)),
    ("all, bits 4", dict(key_bits_xy=4)),
    ("all, loose_below 128", dict(lf_loose_below=128)),
    ("all, loose_below 224", dict(lf_loose_below=224)),
    ("all but hilbert", dict(key_curve=0)),
    ("all but split", dict(lf_split=0)),
]
only = os.environ.get("ONLY")
for name, opts in VARIANTS:
    if only and only not in name:
        continue
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in opts.items():
        f.set_option(k, v)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
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
    sel = slice(5, min(25, steps))
    print(f"{name:52s} LF ms mean[5:25] {np.mean(lf[sel]):.4f}  patch share {np.mean(frac[sel]):.4f}  | last5 {np.mean(lf[-5:]):.4f} {np.mean(frac[-5:]):.4f}", flush=True)
    print("   lf:", " ".join(f"{v:.3f}" for v in lf), flush=True)
    print("   fr:", " ".join(f"{v:.3f}" for v in frac), flush=True)

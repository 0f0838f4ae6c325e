"""A/B of k_reweight_lf_pipe (option lf_pipe = 1, the default) against k_reweight_lf_patch (lf_pipe = 0) on the bench workload: LF kernel
time by HIP events and cycles per second over the cycles the driver's bench times (5 .. 24), then 20 more (the settled cloud);
the same at 10M particles with PARTICLES=10000000.   python tools/exp_pipe_ab.py [cycles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
n = int(os.environ.get("PARTICLES", "1000000"))
cells, truth, odoms, scans, _poses = bench.make_workload(steps)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
variants = [("pipe", dict(lf_pipe=1)), ("block per workgroup", dict(lf_pipe=0)), ("pipe", dict(lf_pipe=1)), ("block per workgroup", dict(lf_pipe=0))]
extra = os.environ.get("EXTRA")
if extra:
    variants = [(extra, dict(eval(extra)))] + variants[:2]
for name, opts in variants:
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in opts.items():
        f.set_option(k, v)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    # (1) wall clock, no events: warm-up 5, then windows of 20
    for c in range(5):
        f.update(controls[c], scans[c])
    f.sync()
    rates = []
    c = 5
    while c + 20 <= steps:
        t0 = time.perf_counter()
        for k in range(c, c + 20):
            f.update(controls[k], scans[k])
        f.sync()
        rates.append(20 / (time.perf_counter() - t0))
        c += 20
    # (2) the LF kernel by events on a second pass over the same trajectory, on a fresh filter (the first one's odometry is at the
    # trajectory's end: its next update would see the jump back as motion)
    pipe_launches = f.counter('lf_pipe_launches')
    f.close()
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in opts.items():
        f.set_option(k, v)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    f.profile_enable(2)
    lf = []
    for c in range(steps):
        f.profile_read(reset=True)
        f.update(controls[c], scans[c])
        f.sync()
        lf.append(f.profile_read(reset=True)["sensor_kernel"][0])
    print(f"{name:22s} n {n}: cycles/s per window of 20: {' '.join(f'{r:.1f}' for r in rates)} | LF ms mean[5:25] {np.mean(lf[5:25]):.4f} last10 {np.mean(lf[-10:]):.4f} | pipe launches {pipe_launches + f.counter('lf_pipe_launches')}", flush=True)
    f.close()

"""One line per library (BELUGA_MCL_LIB): the LF kernel's time on the bench's set (HIP events: cycles 5 .. 24, the driver's window, and
35 .. 44, the settled cloud), the cycle rate over cycles 25 .. 44, and a checksum of the last estimate (the same for every build that
computes the same thing)."""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = 45
n = int(os.environ.get("N", 1_000_000))
cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
def run(events):
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in [kv.split("=") for kv in os.environ.get("OPTIONS", "").split(",") if kv]:
        f.set_option(k, int(v))
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    lf = []
    if events:
        f.profile_enable(2)
    t25 = None
    for c in range(cycles):
        if c == 25 and not events:
            f.sync(); t25 = time.perf_counter()
        if events:
            f.profile_read(reset=True)
        est = f.update(se2_from_xytheta(*odoms[c]), scans[c])
        if events:
            f.sync()
            p = f.profile_read(reset=True)
            lf.append(1e3 * p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
    f.sync()
    rate = None if events else 20 / (time.perf_counter() - t25)
    digest = hashlib.sha256(np.asarray(est[0], dtype=np.float64).tobytes() + np.asarray(est[1], dtype=np.float64).tobytes()).hexdigest()[:12]
    f.close()
    return lf, rate, digest
lf, _, d1 = run(True)
_, rate, d2 = run(False)
print(f"{os.environ.get('BELUGA_MCL_LIB', 'product'):48s} LF us cycles 5-24 {np.mean(lf[5:25]):6.1f}  35-44 {np.mean(lf[35:45]):6.1f} | {rate:7.1f} cycles/s over 25-44 | estimate {d1} {d2}", flush=True)

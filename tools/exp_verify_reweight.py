import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells, truth, odoms, scans, _ = bench.make_workload(30)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
for rep in range(12):
    f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    for k, v in [kv.split("=") for kv in sys.argv[1:]]:
        f.set_option(k, int(v))
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    f.profile_enable(1)
    for c in range(25):
        f.update(se2_from_xytheta(*odoms[c]), scans[c])
    f.sync()
    f.profile_enable(2)
    s, w = f.particles()
    t0 = time.perf_counter()
    f.reweight(scans[25])
    f.sync()
    dt = (time.perf_counter() - t0) * 1e3
    perm, keys = f.debug_order() if hasattr(f, "debug_order") else (None, None)
    big = int(np.bincount(keys >> 10, minlength=1024).max()) if keys is not None else -1
    print(f"rep {rep}: reweight {dt:.2f} ms; largest bucket {big}; far {f.counter('lf_far_launches')} patch {f.counter('lf_patch_launches')} beams {f.counter('lf_beams_launches')}", flush=True)
    f.close()

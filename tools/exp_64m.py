"""64M particles x 1080 beams on one GPU (the whole 8-GPU job of config 4 on a single device): a few full update cycles
timed, the reweight sampled against the oracle, the resampled set checked to be drawn from the weighted one."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from beluga_amd.amcl import (Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid,  # noqa: E402
                             se2_from_xytheta)
from oracle import binding as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
cells, truth, odoms, scans, _poses = bench.make_workload(6)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
t0 = time.perf_counter()
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
         AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
print(f"create + initialize {n}: {time.perf_counter() - t0:.2f} s", flush=True)
controls = [se2_from_xytheta(*o) for o in odoms]
for c in range(4):
    t = time.perf_counter()
    est = f.update(controls[c], scans[c])
    dt = time.perf_counter() - t
    info = f.last_info
    print(f"cycle {c}: {dt * 1e3:.2f} ms  resampled {info['resampled']}  n {info['num_particles']}  pose {est[0][2]:.3f} {est[0][3]:.3f}",
          flush=True)
# stage level: reweight against the oracle on a sample
states, w0 = f.particles()
assert len(w0) == n and np.all(w0 == 1.0)
f.reweight(scans[4])
w = f.particles()[1]
sample = np.random.Generator(np.random.MT19937(1)).choice(n, 2048, replace=False)
sample = np.concatenate([sample, [0, n - 1]])
want = orc.lf_weights(f.likelihood_field(), bench.RESOLUTION, grid.origin, bench.LF["max_laser_distance"], states[sample], scans[4],
                      threads=orc.max_threads())
err = np.max(np.abs(w[sample] / want - 1.0))
print(f"reweight: max relative difference to the oracle over {len(sample)} sampled particles {err:.3e}", flush=True)
assert err < 1e-12
total = f.weight_sum()
ref = float(np.sum(w, dtype=np.longdouble))
print(f"weight sum {total!r} vs long-double host sum {ref!r}: rel {abs(total - ref) / ref:.2e}", flush=True)
assert abs(total - ref) / ref < 1e-12
print({k: f.counter(k) for k in ("lf_patch_launches", "lf_fast_launches")})
f.close()
print("OK")

"""Where the sharded driver's cycle goes at world size 1 (nccl): rocprof-free breakdown with host timers around phases."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from beluga_amd.amcl import AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
from beluga_amd import sharded as sh

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cells, truth, odoms, scans, _poses = bench.make_workload(40)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = 1_000_000
f = sh.ShardedAmcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42, device=0)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
ctrl = [se2_from_xytheta(*o) for o in odoms]
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
for c in range(5):
    f.update(ctrl[c], scans[c])
torch.cuda.synchronize(); t0 = time.perf_counter()
for c in range(5, 35):
    f.update(ctrl[c], scans[c])
torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 30
e = f.engine
for name in ["propagate", "reweight", "weight_sum_into", "normalize_from", "build_cdf_into", "estimate_sums_into", "resample_targets", "route_targets", "serve_requests", "commit_routed"]:
    setattr(e, name, timed(name, getattr(e, name)))
f._resample = timed("_resample(total)", f._resample)
f._draw = timed("_draw(total)", f._draw)
f.net.all_reduce_sum = timed("all_reduce", f.net.all_reduce_sum)
f.net.all_gather = timed("all_gather", f.net.all_gather)
f.net.all_to_all = timed("all_to_all", f.net.all_to_all)
t0 = time.perf_counter()
for c in range(5, 35):
    f.update(ctrl[c], scans[c])
torch.cuda.synchronize(); inst = (time.perf_counter() - t0) / 30
print("cycle ms: plain", round(base * 1e3, 3), "instrumented (serialised)", round(inst * 1e3, 3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:22s} {v / 30 * 1e3:.3f} ms")
dist.destroy_process_group()

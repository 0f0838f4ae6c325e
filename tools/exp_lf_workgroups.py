"""Measurement build (-DMCL_LF_TIMING -DMCL_LF_TIMING_COARSE): every workgroup of k_reweight_lf_patch, in chosen cycles of the bench's set -
when it started and ended (s_memrealtime; printed in units of 100 ticks: about 0.36 us each on MI355X, a 1M launch of ~460 us spans ~1270),
how many of its groups went through a patch, where it ran.  Prints the launch's schedule:
the spread of the durations by kind of workgroup, the rounds, and what the last workgroups to end were."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd import capi
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = int(os.environ.get("CYCLES", 45))
at = [int(x) for x in os.environ.get("AT", "6,12,18,40").split(",")]
n = int(os.environ.get("N", 1_000_000))
cells, truth, odoms, scans, _poses = bench.make_workload(cycles + 1)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
for k, v in [kv.split('=') for kv in os.environ.get('OPTIONS', '').split(',') if kv]:
    f.set_option(k, int(v))
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
lib = capi.load()
wgs = (n + 447) // 448
out = (C.c_ulonglong * (4 * wgs))()
for c in range(cycles):
    f.update(se2_from_xytheta(*odoms[c]), scans[c])
    if c not in at:
        continue
    f.sync()
    assert lib.mcl_debug_lf_workgroups(out, wgs) == 0
    r = np.frombuffer(out, dtype=np.uint64).reshape(wgs, 4).copy()
    np.save(os.path.join(os.environ.get('OUT', 'gpurun_out'), f'lf_wg_cycle{c}.npy'), r)
    start, end = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
    t0 = start.min()
    start, end = (start - t0) / 100.0, (end - t0) / 100.0  # units of 100 ticks
    groups = (r[:, 2] & 0xFFFF).astype(np.int64)
    fitting = ((r[:, 2] >> 16) & 0x7FFF).astype(np.int64)
    loose = ((r[:, 2] >> 31) & 1).astype(bool)
    dur = end - start
    gathered = np.where(loose, groups, groups - fitting)
    xcc = (r[:, 3] >> 32).astype(np.int64) & 0xF
    print(f"--- cycle {c}: {wgs} workgroups, launch spans {end.max():.1f} units; loose workgroups {loose.sum()}, gathered groups {gathered.sum()} of {groups.sum()} "
          f"({100 * gathered.sum() / groups.sum():.1f} %)")
    for name, m in (("all fitting", (~loose) & (gathered == 0)), ("1-4 gathered", (~loose) & (gathered >= 1) & (gathered <= 4)),
                    ("5-16 gathered", (~loose) & (gathered >= 5) & (gathered <= 16)), (">16 gathered", (~loose) & (gathered > 16)), ("loose", loose)):
        if m.sum():
            d = dur[m]
            print(f"  {name:14s} {m.sum():5d} workgroups: duration min {d.min():6.1f} median {np.median(d):6.1f} p90 {np.percentile(d, 90):6.1f} max {d.max():6.1f}")
    order = np.argsort(start)
    # rounds: the first 768 to start, the next ...
    for k in range(0, wgs, 768):
        sel = order[k:k + 768]
        print(f"  started {k:4d}..{k + len(sel) - 1:4d}: start {start[sel].min():6.1f} .. {start[sel].max():6.1f}, end {end[sel].min():6.1f} .. {end[sel].max():6.1f}, mean duration {dur[sel].mean():6.1f}")
    last = np.argsort(end)[-8:]
    print("  last to end:", ", ".join(f"wg {i} start {start[i]:.0f} dur {dur[i]:.0f} gathered {gathered[i]}{' loose' if loose[i] else ''} xcc {xcc[i]}" for i in last))
    busy = dur.sum() / (end.max() * 768)
    for x in range(8):
        m = xcc == x
        print(f"  xcc {x}: {m.sum()} blocks, last end {end[m].max():.1f}, mean duration {dur[m].mean():.1f}")
    print(f"  slot occupancy (sum of durations / 768 slots x span): {busy:.3f}; per XCC workgroups: {np.bincount(xcc, minlength=8).tolist()}")
    # what a perfectly packed schedule of the same workgroups would take
    print(f"  sum of durations / 768 = {dur.sum() / 768:.1f} units")
f.close()

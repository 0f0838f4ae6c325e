"""Measurement build (-DMCL_LF_TIMING, tools/build_variant.sh lft -DMCL_LF_TIMING): where the waves of k_reweight_lf_patch spend
their cycles - in total, waiting at the workgroup barriers, before the main loop - on the bench's set after `CYCLES` cycles."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd import capi
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cycles = int(os.environ.get("CYCLES", 40))
n = int(os.environ.get("N", 1_000_000))
cells, truth, odoms, scans, _poses = bench.make_workload(cycles + 1)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
lib = capi.load()
out = (C.c_ulonglong * 16)()
f.profile_enable(2)
for c in range(cycles):
    if c in (5, cycles - 1):
        f.sync()
        assert lib.mcl_debug_lf_timing(out, 1) == 0
        f.profile_read(reset=True)
    f.update(se2_from_xytheta(*odoms[c]), scans[c])
    if c in (5, cycles - 1):
        f.sync()
        assert lib.mcl_debug_lf_timing(out, 1) == 0
        v = list(out)
        p = f.profile_read(reset=True)
        ms = p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1)
        cw, cc, cb, pw, pc, pb, pre, plw = v[:8]
        print(f"cycle {c}: kernel {ms * 1e3:.1f} us; consumer waves {cw}: {cc / cw:.0f} cycles each, {100 * cb / cc:.1f} % at barriers, "
              f"{100 * pre / cc:.1f} % before the main loop; producer waves {pw}: {pc / max(pw, 1):.0f} cycles each, {100 * pb / max(pc, 1):.1f} % at barriers, {100 * plw / max(pc, 1):.1f} % waiting for loads"
              + (f"; pipe producer: {v[10]} steps, {v[9] / max(v[10], 1):.0f} cycles per step in slices, {pb / max(v[10], 1):.0f} at barriers, {v[11] / max(pw, 1):.0f} cycles at block starts; consumers {v[6] / max(cw, 1):.0f} cycles at block starts" if v[10] else ""), flush=True)
f.close()

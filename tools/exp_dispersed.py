"""Dispersed 1M-particle set (initialize_from_map on the bench map): one LF reweight per kernel family, timed by the library's
HIP events; run under rocprofv3 --pmc to see what each one asks of the memory system.
Usage: python tools/exp_dispersed.py [variants...]   variants: beams gather far patch lane wave"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

variants = sys.argv[1:] or ["beams", "gather", "lane", "wave"]
cells, truth, odoms, scans, _poses = bench.make_workload(2)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
n = int(os.environ.get("N", 1_000_000))
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize_from_map()
states, w0 = f.particles()
OPTS = {"beams": {"lf_variant": 3}, "gather": {"lf_variant": 2, "lf_patch": 0, "lf_far_tiles": 0, "key_layout": 0},
        "far": {"lf_variant": 2, "lf_patch": 0, "lf_far_tiles": 2, "key_layout": 0},
        "farpos": {"lf_variant": 2, "lf_patch": 0, "lf_far_tiles": 2, "key_layout": 1},
        "gatherpos": {"lf_variant": 2, "lf_patch": 0, "lf_far_tiles": 0, "key_layout": 1}, "patch": {"lf_variant": 2, "lf_patch": 2},
        "lane": {"lf_variant": 1}, "wave": {"lf_variant": 0}}
ref = None
for v in variants:
    for k, val in OPTS[v].items():
        f.set_option(k, val)
    ms = []
    for rep in range(3):
        f.set_particles(states, w0)
        f.profile_enable(2)
        f.profile_read(reset=True)
        f.reweight(scans[0])
        f.sync()
        p = f.profile_read(reset=True)
        ms.append(p["sensor_kernel"][0] / max(p["sensor_kernel"][1], 1))
    w = f.particles()[1]
    if ref is None:
        ref = w
    print(v, "far launches", f.counter("lf_far_launches"), "far tiles", f.counter("lf_far_tiles"), "sensor_kernel_ms", [round(x, 3) for x in ms], "max rel diff vs first", float(np.max(np.abs(w - ref) / ref)), flush=True)
f.close()

"""Randomised check of the LDS-patch kernel against the gather kernel (same weights bit for bit): maps, poses, cloud spreads, scan
lengths and ranges, set sizes and planner options drawn at random.  Usage: python tools/fuzz_patch.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.MT19937(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
LF = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
bad = 0
for case in range(cases):
    size = int(rng.choice([200, 400, 800]))
    res = float(rng.choice([0.05, 0.1, 0.025]))
    cells = synth.make_rooms_map(size, size, seed=int(rng.integers(1, 1000)), n_rooms=int(rng.integers(3, 20)))
    origin = (-size * res / 2 + float(rng.normal(0, 1)), -size * res / 2 + float(rng.normal(0, 1)))
    grid = OccupancyGrid(cells=cells, resolution=res, origin=se2_from_xytheta(origin[0], origin[1], 0.0))
    truth = synth.find_free_pose(cells, res, origin, seed=int(rng.integers(1, 1000)), clearance_cells=4)
    beams = int(rng.choice([57, 180, 360, 720, 1080, 1300, 1537, 1700]))
    max_range = float(rng.choice([3.5, 12.0, 30.0]))
    angles = synth.lidar_angles(beams, float(rng.choice([270.0, 360.0, 180.0])))
    ranges = synth.cast_scan(cells, res, origin, truth, angles, max_range, 0.01, int(rng.integers(1, 100)))
    pts = synth.scan_points(ranges, angles)
    n = int(rng.choice([16_384, 20_000, 66_667, 131_072, 250_000, 300_000, 524_288]))  # (from 262 144 on: one scan segment, the queue applies)
    sig = (float(rng.choice([0.02, 0.1, 0.3, 0.8])), float(rng.choice([0.02, 0.1, 0.3, 0.8])), float(rng.choice([0.01, 0.05, 0.15, 0.5])))
    opts = dict(lf_split=int(rng.integers(0, 4)), lf_margin=int(rng.integers(0, 2)), key_curve=int(rng.integers(0, 2)),
                key_warp=int(rng.integers(0, 2)), key_bits_xy=int(rng.choice([0, 4, 5, 6])), 
                lf_loose_below=int(rng.choice([0, 128, 224, 257])),
                # the queue of blocks, with few resident workgroups so that every one takes many blocks (0 = three per CU: a workgroup per block here)
                lf_queue=int(rng.integers(0, 2)), lf_ends_first=int(rng.integers(0, 2)), lf_queue_grid=int(rng.choice([0, 1, 5, 37, 200])))
    ws = []
    for patch in (2, 0):
        f = Amcl(grid, DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05), LF, AmclParams(min_particles=n, max_particles=n), seed=11)
        f.set_option("lf_small_particles", 16_384)
        f.set_option("lf_patch", patch)
        for k, v in opts.items():
            f.set_option(k, v)
        f.initialize(truth, np.diag([s * s for s in sig]))
        f.reweight(pts)
        ws.append(f.particles()[1].copy())
        if patch == 2:
            share = f.counter("lf_patch_groups_through") / max(f.counter("lf_patch_groups_planned"), 1)
            queued = f.counter("lf_queue_launches")
        f.close()
    ok = np.array_equal(ws[0], ws[1])
    bad += not ok
    print(f"case {case}: map {size} @ {res}, {beams} beams to {max_range} m, n {n}, sigma {sig}, {opts}: through a patch {share:.3f}{' (queue)' if queued else ''} "
          f"{'ok' if ok else 'MISMATCH ' + str(int((ws[0] != ws[1]).sum()))}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)

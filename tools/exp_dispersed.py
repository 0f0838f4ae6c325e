"""1M particles dispersed over the bench map (initialize_from_map): the cycle and the sensor kernel with the ordered-lanes gather kernel
(lf_dispersed = 0) and with the lanes over the beams of the ordered particles (lf_dispersed = 2, particles per wave as given); the
weights of one reweight compared between the two."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
cells, truth, odoms, scans, _ = bench.make_workload(12)
grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
controls = [se2_from_xytheta(*o) for o in odoms]
n = 1_000_000
f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF), AmclParams(min_particles=n, max_particles=n), seed=42)
f.initialize_from_map()
states, w0 = f.particles()
f.reweight(scans[0])
w_ref = f.particles()[1]
variants = [(0, 0)] + [(2, pw) for pw in (int(a) for a in (sys.argv[1:] or ["16", "32", "64"]))]
for mode, per_wave in variants:
    f.set_option("lf_dispersed", mode)
    f.set_option("lf_far_beams_per_wave", per_wave)
    f.initialize_from_map()
    before = f.counter("lf_far_beams_launches")
    f.set_particles(states, w0)
    f.set_option("lf_patch", 0) if False else None
    f.initialize_from_map()
    f.reweight(scans[0])
    w = f.particles()[1]
    rel = float(np.max(np.abs(w - w_ref) / np.abs(w_ref)))
    f.update(controls[0], scans[0])
    f.profile_enable(2)
    f.profile_read(reset=True)
    ms = bench._timed_cycles(f, controls, scans, 1, 8, reinit=f.initialize_from_map)
    prof = f.profile_read(reset=True)
    f.profile_enable(0)
    print(f"lf_dispersed {mode} per_wave {per_wave}: cycle ms median {np.median(ms):.3f} ({1e3 / np.median(ms):.0f}/s)  sensor kernel {prof['sensor_kernel'][0] / max(prof['sensor_kernel'][1], 1):.3f} ms"
          f"  far-beams launches {f.counter('lf_far_beams_launches') - before}  max rel weight diff {rel:.2e}", flush=True)
f.close()

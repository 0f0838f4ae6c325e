"""Generates tests/golden/update_cycles_config1.npz: a frozen end-to-end run of the update cycle (BASELINE config 1 shape:
turtlebot3 grid, 500..2000 particles, KLD + recovery, 180 beams, 20 cycles) as produced by the oracle.

The reference itself pins `Amcl::update` only with smoke tests (test_amcl_core.cpp:73-186), so the oracle is the pin for the
end-to-end level; this fixture freezes the oracle's answer so that neither it nor the HIP path can drift unnoticed.
    python tests/golden/make_cycle_fixture.py
Inputs are regenerated from seeds by the tests (beluga_amd.synth); stored are the per-cycle outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from beluga_amd import synth  # noqa: E402
from beluga_amd.amcl import se2_from_xytheta  # noqa: E402

OUT = os.path.join(HERE, "update_cycles_config1.npz")
SEED, CYCLES, BEAMS, MAX_RANGE = 0xBE1A6A, 20, 180, 3.5
PARAMS = dict(min_particles=500, max_particles=2000, alpha_slow=0.001, alpha_fast=0.1, kld_epsilon=0.05, kld_z=3.0,
              hash_res=(0.5, 0.5, 10.0 * np.pi / 180.0), alphas=(0.1, 0.05, 0.1, 0.05), lf=(2.0, 100.0, 0.5, 0.5, 0.2),
              lf_model_unknown_space=True)


def scenario():
    """-> (cells, resolution, origin se2, truth pose, [(control se2, points)])"""
    z = np.load(os.path.join(HERE, "turtlebot3_world_grid.npz"))
    cells, res = z["cells"], float(z["resolution"])
    ox, oy, ot = z["origin_xytheta"]
    truth = synth.find_free_pose(cells, res, (ox, oy), seed=4, clearance_cells=8)
    angles = synth.lidar_angles(BEAMS, 360.0)
    pose, odom, steps = truth, (0.0, 0.0, 0.0), []
    for c in range(CYCLES):
        fwd, turn = (0.3, 0.05) if c % 5 != 4 else (0.02, 0.01)  # every 5th step is below update_min_d / update_min_a
        pose = synth.odometry_step(pose, fwd, turn)
        odom = synth.odometry_step(odom, fwd, turn)
        ranges = synth.cast_scan(cells, res, (ox, oy), pose, angles, MAX_RANGE, 0.01, seed=100 + c)
        steps.append((se2_from_xytheta(*odom), synth.scan_points(ranges, angles)))
    return cells, res, se2_from_xytheta(ox, oy, ot), truth, steps


def main():
    from oracle import binding as orc
    cells, res, origin, truth, steps = scenario()
    f = orc.Amcl(seed=SEED, **PARAMS)
    f.set_map(cells, res, origin)
    f.initialize(truth, np.diag([0.25, 0.25, 0.0685]))
    updated, poses, covs, counts, sums, probs = [], [], [], [], [], []
    for ctrl, pts in steps:
        out = f.update(ctrl, pts)
        updated.append(out is not None)
        if out is not None:
            poses.append(out[0])
            covs.append(out[1])
            counts.append(len(f.particles()[1]))
            sums.append(f.last_info["weight_sum"])
            probs.append(f.last_info["random_state_probability"])
    states, _ = f.particles()
    np.savez_compressed(OUT, updated=np.array(updated), poses=np.array(poses), covs=np.array(covs), counts=np.array(counts),
                        weight_sums=np.array(sums), random_state_probability=np.array(probs), final_states=states)
    print("wrote", OUT, "cycles with an update:", int(np.sum(updated)), "final particles:", len(states))


if __name__ == "__main__":
    main()

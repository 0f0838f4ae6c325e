"""The sharded driver over the HIP engine with the RCCL ('nccl') backend.  Only one GPU is available to the test
box, so this exercises every collective call (all_reduce / all_gather_into_tensor / all_to_all_single) and the
device-pointer hand-off with world_size 1, and checks the result against the plain single-GPU filter."""
import os

import numpy as np
import pytest

from beluga_amd import synth
from beluga_amd.amcl import Amcl, AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

pytestmark = pytest.mark.gpu


def test_sharded_world1_nccl_matches_single_gpu():
    import torch
    import torch.distributed as dist

    from beluga_amd.sharded import ShardedAmcl

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
        grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
        truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
        n = 50_000
        params = AmclParams(min_particles=n, max_particles=n)
        motion = DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05)
        lf = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
        single = Amcl(grid, motion, lf, params, seed=9)
        sharded = ShardedAmcl(grid, motion, lf, params, seed=9, device=0)
        cov = np.diag([0.25, 0.25, 0.04])
        single.initialize(truth, cov)
        sharded.initialize(truth, cov)
        angles = synth.lidar_angles(360, 270.0)
        pose, odom = truth, (0.0, 0.0, 0.0)
        for c in range(5):
            pose = synth.odometry_step(pose, 0.3, 0.05)
            odom = synth.odometry_step(odom, 0.3, 0.05)
            pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-10.0, -10.0), pose, angles, 12.0, 0.01, seed=c), angles)
            a = single.update(se2_from_xytheta(*odom), pts)
            b = sharded.update(se2_from_xytheta(*odom), pts)
            np.testing.assert_allclose(b[0], a[0], atol=1e-9)
            np.testing.assert_allclose(b[1], a[1], rtol=1e-8, atol=1e-11)
        sa, wa = single.particles()
        sb, wb = sharded.particles()
        assert int(np.any(sa != sb, axis=1).sum()) <= 2 and np.array_equal(wa, wb)
        single.close()
        sharded.close()
    finally:
        dist.destroy_process_group()

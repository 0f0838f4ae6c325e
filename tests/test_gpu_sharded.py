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


# ---- two (three) ranks sharing the one GPU of the test box -------------------------------------------------------
# RCCL refuses two ranks on one device, so these runs use a gloo group: ShardedAmcl stages the device tensors of every
# collective through host memory, everything else (shard offsets, routing, serving, KLD feed, re-balancing) is the HIP
# engine exactly as in a multi-GPU run.  The result must equal the single-context filter.
def _workload(n_cycles=5, beams=360):
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    angles = synth.lidar_angles(beams, 270.0)
    pose, odom, steps = truth, (0.0, 0.0, 0.0), []
    for c in range(n_cycles):
        pose = synth.odometry_step(pose, 0.3, 0.05)
        odom = synth.odometry_step(odom, 0.3, 0.05)
        pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-10.0, -10.0), pose, angles, 12.0, 0.01, seed=c), angles)
        steps.append((se2_from_xytheta(*odom), pts))
    return grid, truth, steps


MOTION = DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05)
LF = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)


def _shared_gpu_worker(rank, world, init_file, params, block, result_file):
    import torch
    import torch.distributed as dist

    from beluga_amd.sharded import ShardedAmcl, shard_bounds
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    grid, truth, steps = _workload()
    f = ShardedAmcl(grid, MOTION, LF, params, seed=9, device=0, kld_block=block)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    outs, counts = [], []
    for c, p in steps:
        outs.append(f.update(c, p))
        counts.append(f.n_total)
        assert (f.first_slot, f.n_local) == shard_bounds(f.n_total, world, rank) and f.engine.num_particles() == f.n_local
    states, w = f.gather_particles()
    if rank == 0:
        np.savez(result_file, states=states, w=w, counts=np.array(counts), poses=np.array([o[0] for o in outs]),
                 covs=np.array([o[1] for o in outs]))
    dist.barrier()
    f.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,min_p,max_p,block", [(2, 50_001, 50_001, None), (2, 300, 200_000, None), (3, 300, 200_000, 4_099)])
def test_sharded_ranks_sharing_one_gpu_match_single_context(world, min_p, max_p, block, tmp_path):
    import torch.multiprocessing as mp
    params = AmclParams(min_particles=min_p, max_particles=max_p)
    init_file, result_file = str(tmp_path / "rendezvous"), str(tmp_path / "result.npz")
    mp.spawn(_shared_gpu_worker, args=(world, init_file, params, block, result_file), nprocs=world, join=True)
    got = np.load(result_file)
    grid, truth, steps = _workload()
    single = Amcl(grid, MOTION, LF, params, seed=9)
    single.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    ref, ref_counts = [], []
    for c, p in steps:
        ref.append(single.update(c, p))
        ref_counts.append(single.num_particles())
    sa, wa = single.particles()
    single.close()
    assert list(got["counts"]) == ref_counts
    if min_p < max_p:
        assert min_p < ref_counts[-1] < max_p  # the KLD cut is a real one
    np.testing.assert_allclose(got["poses"], np.array([o[0] for o in ref]), atol=1e-9)
    np.testing.assert_allclose(got["covs"], np.array([o[1] for o in ref]), rtol=1e-8, atol=1e-11)
    assert got["states"].shape == sa.shape
    assert int(np.any(got["states"] != sa, axis=1).sum()) <= 2  # a CDF-boundary flip (shard-wise summation order) at most
    np.testing.assert_allclose(got["w"], wa, rtol=1e-12)


def test_sharded_kld_world1_nccl_matches_single_gpu():
    import torch
    import torch.distributed as dist

    from beluga_amd.sharded import ShardedAmcl

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29612"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        grid, truth, steps = _workload()
        params = AmclParams(min_particles=300, max_particles=1_000_000)
        single = Amcl(grid, MOTION, LF, params, seed=9)
        sharded = ShardedAmcl(grid, MOTION, LF, params, seed=9, device=0)
        single.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        sharded.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        for c, p in steps:
            a, b = single.update(c, p), sharded.update(c, p)
            assert sharded.n_total == single.num_particles()
            np.testing.assert_allclose(b[0], a[0], atol=1e-9)
            np.testing.assert_allclose(b[1], a[1], rtol=1e-8, atol=1e-11)
        assert 300 < sharded.n_total < 1_000_000
        (sa, wa), (sb, wb) = single.particles(), sharded.particles()
        assert sa.shape == sb.shape and int(np.any(sa != sb, axis=1).sum()) <= 2 and np.array_equal(wa, wb)
        single.close()
        sharded.close()
    finally:
        dist.destroy_process_group()


MOTION = DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05)
LF = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)


def _grid():
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    return OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))


def test_library_rccl_transport_world_one():
    """The built-in RCCL transport of the library (librccl.so loaded at run time; mcl_comm_unique_id / mcl_comm_attach_rccl) on a
    communicator of one rank: the attach succeeds, the collectives' entry points resolve, and update() equals the plain filter
    (a communicator of one needs no exchange)."""
    from beluga_amd.amcl import comm_unique_id
    grid = _grid()
    params = AmclParams(min_particles=30_000, max_particles=30_000)
    plain = Amcl(grid, MOTION, LF, params, seed=5)
    sharded = Amcl(grid, MOTION, LF, params, seed=5, shard_offset=0, shard_capacity=30_000)
    sharded.comm_attach_rccl(comm_unique_id(), 0, 1)
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=4, clearance_cells=8)
    cov = np.diag([0.25, 0.25, 0.04])
    plain.initialize(truth, cov)
    sharded.initialize(truth, cov)
    pose, odom = truth, (0.0, 0.0, 0.0)
    angles = synth.lidar_angles(180, 270.0)
    for c in range(4):
        pose = synth.odometry_step(pose, 0.3, 0.05)
        odom = synth.odometry_step(odom, 0.3, 0.05)
        pts = synth.scan_points(synth.cast_scan(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), pose, angles, 12.0, 0.01, 100 + c), angles)
        a = plain.update(se2_from_xytheta(*odom), pts)
        b = sharded.update(se2_from_xytheta(*odom), pts)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    plain.close()
    sharded.close()


@pytest.mark.parametrize("with_config4", [False, True])
def test_bench_multi_rank_flow_dry_run_on_one_gpu(tmp_path, with_config4):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), here with both ranks on the
    GPU at hand and gloo between them (BELUGA_BENCH_BACKEND=gloo: the timings mean nothing): the multi-rank flow - rendezvous,
    sharded filter, timed region with barriers, max over ranks, ONE JSON line from rank 0 - must not fall over unseen on the
    first 8-GPU node it meets.  The line's contract fields are checked, and the filter still localises."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BELUGA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29617", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--particles", "20000", "--windows", "1",
           "--stage-steps", "2", "--no-cpu-baseline"]
    # with_config4: the branch that several ranks add to the line - BASELINE configs[3], a second sharded filter behind the first - runs
    # as well, at a reduced size (the default 8M per rank is the configuration itself; two of those on one device is the size-realistic
    # test of tests/test_cpp_facade.py)
    cmd += ["--config4-particles", "60000"] if with_config4 else ["--no-other-configs"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["unit"] == "cycles/s" and line["scaling"] == "weak"
    assert line["config"]["particles_total"] == 40000 and line["config"]["particles_per_gpu"] == 20000
    assert line["config"]["collective"]["ranks_seen"] == 2, line["config"]["collective"]  # (the field the driver's "did every rank take part" check reads)
    # value = the whole job in the metric's unit (cycles of particles_per_gpu particles): ranks x the logical filter's own cycle rate
    assert line["config"]["filter_cycles_per_s"] == pytest.approx(4 / line["timed_region_s"], rel=1e-9)
    assert line["value"] == pytest.approx(2 * 4 / line["timed_region_s"], rel=1e-9)
    assert "aggregate over 2 GPUs" in line["metric"] and line["metric"].startswith("MCL update cycles/sec (motion+sensor+resample), N particles x 1080 beams")
    assert line["verified"]["estimate_vs_true_pose"]["ok"], line["verified"]
    if with_config4:
        c4 = line["configs"]["4"]
        assert c4["particles_total"] == 120000 and c4["cycles_per_s"] > 0, c4
    else:
        assert "configs" not in line

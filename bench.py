#!/usr/bin/env python3
"""bench.py — MCL update cycles/sec (motion + sensor + resample), N particles x 1080 beams (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full `Amcl::update` cycle (amcl_core.hpp:165-201): DifferentialDriveModel propagation,
LikelihoodFieldModel reweight over a 1080-beam scan, normalisation + policies, multinomial resample of all
particles, SE2 estimate.  Workload = BASELINE.json configs[1]: 1M particles per GPU, 4000x4000 @ 5 cm grid
(seed 42), resample every cycle.  Map, field and particles are resident in HBM before the timed region; each
step uploads one 1080-point scan (17 KB) and downloads the estimate, as the reference's caller would.
With --gpus N the particle set is sharded N ways (1M per rank, weak scaling) behind one logical filter.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant kernel
(likelihood-field reweight, HIP-event timed on the library's stream) and `cpu_baseline` (the oracle, a CPU
restatement of the reference, timed on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# k_reweight_lf_sorted at 1M x 1080: FETCH_SIZE 109581.8 KB, WRITE_SIZE 34063.3 KB per launch (round-1 PMC run)
LF_KERNEL_HBM_BYTES_PER_LAUNCH = int(2 * 78118.6 * 1024 + 34773.0 * 1024)
HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)

MAP_SIZE, RESOLUTION, ORIGIN = 4000, 0.05, (-100.0, -100.0)
BEAMS, FOV_DEG, MAX_RANGE = 1080, 270.0, 30.0
LF = dict(max_obstacle_distance=2.0, max_laser_distance=100.0, z_hit=0.5, z_random=0.5, sigma_hit=0.2, model_unknown_space=True)
ALPHAS = (0.1, 0.05, 0.1, 0.05)


def lf_algorithmic_bytes(n: int, beams: int) -> int:
    """Algorithmic bytes of ONE launch of the likelihood-field reweight kernel ('K2'):
    one 4-byte field lookup per (particle, beam) + the particle's state read (4 x f64) and weight
    read-modify-write (2 x f64) + the scan itself (beams x 2 x f64)."""
    return n * beams * 4 + n * (32 + 16) + beams * 16


def make_workload(steps_total: int):
    from beluga_amd import synth
    cells = synth.make_rooms_map(MAP_SIZE, MAP_SIZE, seed=42)
    truth = synth.find_free_pose(cells, RESOLUTION, ORIGIN, seed=1)
    angles = synth.lidar_angles(BEAMS, FOV_DEG)
    poses, odoms, scans = [], [], []
    pose, odom = truth, (0.0, 0.0, 0.0)
    for c in range(steps_total):
        pose = synth.odometry_step(pose, 0.3, 0.02 if c % 2 else -0.02)  # 0.3 m > update_min_d: every step updates
        odom = synth.odometry_step(odom, 0.3, 0.02 if c % 2 else -0.02)
        ranges = synth.cast_scan(cells, RESOLUTION, ORIGIN, pose, angles, MAX_RANGE, 0.01, seed=1000 + c)
        poses.append(pose)
        odoms.append(odom)
        scans.append(synth.scan_points(ranges, angles))
    return cells, truth, odoms, scans


def cpu_baseline(cells, truth, odoms, scans, n_full: int, budget_s: float = 12.0):
    """The oracle (CPU restatement of the reference path; kind 'port') on all host cores for the three loops the
    reference parallelises, on a bounded particle sample of the same workload; extrapolated linearly in N."""
    from beluga_amd.amcl import se2_from_xytheta
    from oracle import binding as orc
    threads = orc.max_threads()
    sample_n = 32768 * max(1, threads // 8)
    f = orc.Amcl(min_particles=sample_n, max_particles=sample_n, alphas=ALPHAS, seed=42, threads=threads,
                 lf=(LF["max_obstacle_distance"], LF["max_laser_distance"], LF["z_hit"], LF["z_random"], LF["sigma_hit"]),
                 lf_model_unknown_space=LF["model_unknown_space"])
    f.set_map(cells, RESOLUTION, se2_from_xytheta(ORIGIN[0], ORIGIN[1], 0.0))
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    f.update(se2_from_xytheta(*odoms[0]), scans[0])  # warm-up
    t0 = time.perf_counter()
    done = 0
    for c in range(1, len(odoms)):
        f.update(se2_from_xytheta(*odoms[c]), scans[c])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cycles_per_s_sample = done / dt
    return {
        "value": cycles_per_s_sample * sample_n / n_full,
        "unit": "cycles/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} full update cycles of {sample_n} particles x {BEAMS} beams in {dt:.2f} s "
                  f"({dt / done / sample_n / BEAMS * 1e9:.2f} ns per particle-beam), scaled linearly to {n_full} particles",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windows", type=int, default=5, help="repeated timing windows of --steps cycles behind the timed region")
    ap.add_argument("--stage-steps", type=int, default=6, help="cycles of the per-stage breakdown pass")
    ap.add_argument("--sharded", action="store_true", help="use the sharded driver even with one GPU (measures its overhead)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MCL update has no CPU path")
    # BELUGA_BENCH_BACKEND=gloo lets several ranks share one GPU (a dry run of the multi-rank flow on a 1-GPU box; the
    # collectives are then staged through host memory by beluga_amd/sharded.py and the timings mean nothing)
    backend = os.environ.get("BELUGA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # Native libraries (RCCL's version banner) write to fd 1; the contract is ONE JSON line on stdout, so fd 1 is
    # parked on stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    use_sharded = world > 1 or args.sharded
    if use_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from beluga_amd.amcl import AmclParams, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta

    steps_total = args.warmup + args.steps * (1 + args.windows) + args.stage_steps
    cells, truth, odoms, scans = make_workload(steps_total)
    grid = OccupancyGrid(cells, RESOLUTION, origin=se2_from_xytheta(ORIGIN[0], ORIGIN[1], 0.0))
    n_local = args.particles
    n_total = n_local * world
    params = AmclParams(min_particles=n_total, max_particles=n_total)
    motion = DifferentialDriveModelParam(*ALPHAS)
    sensor = LikelihoodFieldModelParam(**LF)
    if not use_sharded:
        from beluga_amd.amcl import Amcl
        filt = Amcl(grid, motion, sensor, params, seed=42, device=local_rank)
    else:
        from beluga_amd.sharded import ShardedAmcl
        filt = ShardedAmcl(grid, motion, sensor, params, seed=42, device=local_rank)
    filt.initialize(truth, np.diag([0.25, 0.25, 0.04]))

    def sync_all():
        filt.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    controls = [se2_from_xytheta(*o) for o in odoms]
    for c in range(args.warmup):
        assert filt.update(controls[c], scans[c]) is not None
    # Timed region: exactly `steps` update cycles.  Only the dominant kernel carries HIP events here (two records per cycle:
    # an event record costs ~5 us of stream time); the per-stage breakdown comes from a separate pass below.
    filt.profile_enable(1)
    filt.profile_read(reset=True)
    sync_all()
    t0 = time.perf_counter()
    for c in range(args.warmup, args.warmup + args.steps):
        est = filt.update(controls[c], scans[c])
        assert est is not None
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = filt.profile_read(reset=True)
    # Repeated windows of the same length right behind the timed region (same filter, the trajectory continues): the spread
    # says how much a single 20-step sample can be trusted.
    window_rates = []
    c = args.warmup + args.steps
    for _ in range(args.windows):
        sync_all()
        w0 = time.perf_counter()
        for _k in range(args.steps):
            assert filt.update(controls[c], scans[c]) is not None
            c += 1
        sync_all()
        window_rates.append(args.steps / (time.perf_counter() - w0))
    # Per-stage breakdown (events around every stage; these cycles run ~40 us slower and are not part of any rate above).
    filt.profile_enable(2)
    filt.profile_read(reset=True)
    for _k in range(args.stage_steps):
        assert filt.update(controls[c], scans[c]) is not None
        c += 1
    sync_all()
    stage_prof = filt.profile_read(reset=True)
    filt.profile_enable(0)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        lf_ms, lf_count = prof["sensor_kernel"]
        lf_avg_s = (lf_ms / max(lf_count, 1)) * 1e-3
        bytes_lf = lf_algorithmic_bytes(n_local, BEAMS)
        achieved = bytes_lf / lf_avg_s if lf_avg_s > 0 else 0.0
        out = {
            "metric": "MCL update cycles/sec (motion+sensor+resample), N particles x 1080 beams",
            # whole-job aggregate: one unit = one update cycle of 1M particles x 1080 beams (the configuration the metric is
            # quoted on); with N GPUs every global cycle moves N such shards, so it counts N units
            "value": world * args.steps / elapsed,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 1M particles/GPU, 1080-beam LikelihoodFieldModel, 4000x4000@5cm grid (seed 42), "
                            "DifferentialDriveModel, multinomial resample every cycle",
                "particles_per_gpu": n_local,
                "particles_total": n_total,
                "beams": BEAMS,
                "grid": f"{MAP_SIZE}x{MAP_SIZE}@{RESOLUTION}",
                "parallelism": "1 GPU" if not use_sharded else f"particle shards x{world} (RCCL all-reduce of weight sums + all-to-all ancestor exchange)",
                "particle_beam_evals_per_s": n_total * BEAMS * args.steps / elapsed,
                "global_cycles_per_s": args.steps / elapsed,
                "unit_of_work": "one update cycle of 1M particles x 1080 beams; a global cycle over N GPUs = N units",
            },
            "timed_region_s": elapsed,
            "repeat_windows": {"steps_each": args.steps, "cycles_per_s": window_rates,
                               "min": min(window_rates) if window_rates else None,
                               "median": float(np.median(window_rates)) if window_rates else None},
            "stage_ms": {k: (v[0] / max(v[1], 1)) for k, v in stage_prof.items()},
            "roofline": {
                "kernel": "k_reweight_lf_palette (likelihood-field reweight)",
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                # HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/r01_pmc_traffic_calibration.txt):
                # 2 * FETCH_SIZE (gfx950 counts half of every read, calibrated on known streams and gathers) + WRITE_SIZE.
                "traffic": LF_KERNEL_HBM_BYTES_PER_LAUNCH if (n_local == 1_000_000 and not use_sharded) else None,
                "algorithmic_bytes_per_launch": bytes_lf,
                "avg_launch_ms": lf_avg_s * 1e3,
                "launches": int(lf_count),
                # The table is cache resident (see `traffic`): the kernel's real ceiling is VALU issue.  6 v_mul_f64 + 7 v_add_f64
                # + 4 32-bit ops per (particle, beam) at the issue costs measured on this part
                # (profiles/r01_calib_f64_issue_rate.txt: 5.52 / 4.92 / 2.95 nominal cycles per wave64 instruction).
                "valu_issue_floor_ms": n_local * BEAMS / 64 * (6 * 5.52 + 7 * 4.92 + 4 * 2.95) / (1024 * 2.4e9) * 1e3,
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cells, truth, odoms, scans, n_total)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

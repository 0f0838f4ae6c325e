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

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED = 6.29e12  # B/s, the float4-copy rate the same guide measures (79 % of spec)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "lf_kernel_traffic.json")
KERNEL_SOURCE = os.path.join(ROOT, "beluga_amd", "csrc", "kernels.hip")


def lf_kernel_source_sha(path):
    """SHA-256 of the likelihood-field kernels' source: the part of kernels.hip between the [lf-kernels-begin] and
    [lf-kernels-end] markers (the whole file if they are missing)."""
    import hashlib
    with open(path, "rb") as fh:
        text = fh.read()
    a, b = text.find(b"[lf-kernels-begin]"), text.rfind(b"[lf-kernels-end]")  # (the begin marker's own comment names the end marker)
    if 0 <= a < b:
        text = text[a:b]
    return hashlib.sha256(text).hexdigest()


def measured_lf_traffic(n_particles: int):
    """HBM bytes per launch of the LF kernel from the PMC passes of tools/gpu_pmc_traffic.sh (2 x FETCH_SIZE + WRITE_SIZE, see
    profiles/r01_pmc_traffic_calibration.txt), valid only for the kernel source it was collected on: the file records the
    SHA-256 of the LF kernels' source (lf_kernel_source_sha) and the particle count; anything else gives None."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
        sha = lf_kernel_source_sha(KERNEL_SOURCE)
    except (OSError, ValueError):
        return None
    if rec.get("kernels_hip_sha256") != sha or rec.get("particles") != n_particles:
        return None
    return int(2 * rec["fetch_size_kb"] * 1024 + rec["write_size_kb"] * 1024)


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


def _timed_cycles(filt, controls, scans, first, count, reinit=None):
    """Wall time of `count` update cycles, each synchronised; `reinit` (optional) restores the set before every cycle."""
    ms = []
    for k in range(count):
        if reinit is not None:
            reinit()
        filt.sync()
        t0 = time.perf_counter()
        assert filt.update(controls[first + k], scans[first + k]) is not None
        filt.sync()
        ms.append((time.perf_counter() - t0) * 1e3)
    return ms


def other_configs(grid, cells, truth, controls, scans, main_filter, device):
    """The other single-GPU configurations of BASELINE.json, outside the timed headline region (reported, not the metric):
    config 3 (10M particles, KLD + selective resampling), a fixed-size 10M and 8M cycle (the >=10M target and the per-GPU
    share of config 4), config 5 (BeamSensorModel, 1M x 1080) and the worst case for the ordered-lanes kernel: 1M particles
    dispersed over the whole map (initialize_from_map: global localisation)."""
    from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, LikelihoodFieldModelParam
    motion = DifferentialDriveModelParam(*ALPHAS)
    cov = np.diag([0.25, 0.25, 0.04])
    out = {}
    # dispersed cloud on the main filter (same map, same 1M capacity): every cycle starts from a fresh uniform set
    f = main_filter
    f.initialize_from_map()
    assert f.update(controls[0], scans[0]) is not None  # untimed: takes the odometry jump back to the start of the sequence
    f.profile_enable(2)
    f.profile_read(reset=True)
    ms = _timed_cycles(f, controls, scans, 1, 4, reinit=f.initialize_from_map)
    prof = f.profile_read(reset=True)
    out["dispersed_1M"] = {"what": "1M particles from initialize_from_map on the 4000x4000 map, fixed N, one update cycle each",
                           "ms_per_cycle": ms, "cycles_per_s": 1e3 / float(np.median(ms)),
                           "sensor_kernel_ms": prof["sensor_kernel"][0] / max(prof["sensor_kernel"][1], 1)}
    f.profile_enable(0)
    # fixed-size large sets
    for label, n in (("fixed_8M", 8_000_000), ("fixed_10M", 10_000_000)):
        g = Amcl(grid, motion, LikelihoodFieldModelParam(**LF), AmclParams(min_particles=n, max_particles=n), seed=42, device=device)
        g.initialize(truth, cov)
        for c in range(3):
            g.update(controls[c], scans[c])
        g.profile_enable(2)
        g.profile_read(reset=True)
        ms = _timed_cycles(g, controls, scans, 3, 5)
        prof = g.profile_read(reset=True)
        out[label] = {"what": f"{n} particles x {BEAMS} beams, multinomial resample every cycle (same workload as the headline)",
                      "ms_per_cycle": ms, "cycles_per_s": 1e3 / float(np.median(ms)),
                      "sensor_kernel_ms": prof["sensor_kernel"][0] / max(prof["sensor_kernel"][1], 1),
                      "algorithmic_GBps": lf_algorithmic_bytes(n, BEAMS) / (prof["sensor_kernel"][0] / max(prof["sensor_kernel"][1], 1) * 1e-3) / 1e9}
        if n == 10_000_000:
            # config 3 on the same capacity: KLD (eps .05, z 3) + selective resampling from a fresh 10M-particle set
            g.close()
            g = Amcl(grid, motion, LikelihoodFieldModelParam(**LF), AmclParams(min_particles=100_000, max_particles=n, selective_resampling=True),
                     seed=42, device=device)
            first, counts = [], []
            for r in range(3):
                g.initialize(truth, cov)
                g.sync()
                t0 = time.perf_counter()
                assert g.update(controls[r], scans[r]) is not None
                g.sync()
                first.append((time.perf_counter() - t0) * 1e3)
                counts.append(g.last_info["num_particles"])
            steady = _timed_cycles(g, controls, scans, 3, 8)
            out["3"] = {"what": "BASELINE configs[2]: max 10M / min 100k particles, KLD (eps .05, z 3) + selective resampling (ESS < N/2)",
                        "first_cycle_ms_at_10M": first, "cycles_per_s_at_10M": 1e3 / float(np.median(first)),
                        "particles_after_first_cycle": counts, "steady_state_ms_per_cycle": steady,
                        "steady_state_particles": g.last_info["num_particles"]}
        g.close()
    # config 5
    n = 1_000_000
    b = Amcl(grid, motion, BeamModelParam(beam_max_range=MAX_RANGE), AmclParams(min_particles=n, max_particles=n), seed=42, device=device)
    b.initialize(truth, cov)
    b.update(controls[0], scans[0])
    b.beam_cells_visited(reset=True)
    ms = _timed_cycles(b, controls, scans, 1, 3)
    visited = b.beam_cells_visited(reset=True)
    out["5"] = {"what": "BASELINE configs[4]: BeamSensorModel (Bresenham ray casts on the int8 grid), 1M particles x 1080 beams, beam_max_range 30",
                "ms_per_cycle": ms, "cycles_per_s": 1e3 / float(np.median(ms)), "cells_visited_per_cycle": visited / 3,
                "cells_per_s": visited / (sum(ms) * 1e-3)}
    b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 10M / beam / dispersed configurations (reported beside the metric)")
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
    driver_used = [None]

    def make_filter(per_gpu):
        """One logical filter of per_gpu * world particles; with several ranks each holds a contiguous shard and the library runs
        the cycle over RCCL (include/beluga_mcl.h, "Particle shards").  BELUGA_BENCH_DRIVER=python (or a gloo dry run) uses the
        torch.distributed driver of beluga_amd/sharded.py instead."""
        from beluga_amd.amcl import Amcl, comm_unique_id
        total = per_gpu * world
        p = AmclParams(min_particles=total, max_particles=total)
        if not use_sharded:
            return Amcl(grid, motion, sensor, p, seed=42, device=local_rank)
        if backend != "nccl" or os.environ.get("BELUGA_BENCH_DRIVER", "library") == "python":
            from beluga_amd.sharded import ShardedAmcl
            return ShardedAmcl(grid, motion, sensor, p, seed=42, device=local_rank)
        f = Amcl(grid, motion, sensor, p, seed=42, device=local_rank, shard_offset=rank * per_gpu, shard_capacity=per_gpu)
        ok = 1
        try:
            box = [comm_unique_id() if rank == 0 else None]
        except Exception as exc:  # no usable librccl for the library on this box
            print(f"[bench] rank {rank}: {exc}", file=sys.stderr)
            box, ok = [None], 0
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok = 0
        if ok:
            try:
                f.comm_attach_rccl(box[0], rank, world)
            except Exception as exc:
                print(f"[bench] rank {rank}: {exc}", file=sys.stderr)
                ok = 0
        if world > 1:  # every rank takes the same path: the library's communicator everywhere, or the torch.distributed driver everywhere
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            return f
        f.close()
        from beluga_amd.sharded import ShardedAmcl
        driver_used[0] = "torch.distributed driver (beluga_amd/sharded.py): the library's RCCL communicator could not be set up"
        return ShardedAmcl(grid, motion, sensor, p, seed=42, device=local_rank)

    filt = make_filter(n_local)
    filt.initialize(truth, np.diag([0.25, 0.25, 0.04]))

    def sync_all_of(f):
        f.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def sync_all():
        sync_all_of(filt)

    controls = [se2_from_xytheta(*o) for o in odoms]
    for c in range(args.warmup):
        assert filt.update(controls[c], scans[c]) is not None
    # Timed region: exactly `steps` update cycles.  Only the dominant kernel carries HIP events here (two records per cycle:
    # an event record costs ~5 us of stream time); the per-stage breakdown comes from a separate pass below.
    filt.profile_enable(1)
    filt.profile_read(reset=True)
    sync_all()
    patch_before = (filt.counter("lf_patch_groups_planned"), filt.counter("lf_patch_groups_through")) if hasattr(filt, "counter") else None
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
    patch_after = (filt.counter("lf_patch_groups_planned"), filt.counter("lf_patch_groups_through")) if hasattr(filt, "counter") else None
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

    # BASELINE configs[3] with several ranks: 8M particles per GPU behind one logical filter (64M at 8 GPUs); cycles/s of
    # that filter, beside the metric.
    config4 = None
    if world > 1 and not args.no_other_configs:
        filt.close()
        big = make_filter(8_000_000)
        big.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        for c in range(2):
            assert big.update(controls[c], scans[c]) is not None
        sync_all_of(big)
        t0 = time.perf_counter()
        for c in range(2, 8):
            assert big.update(controls[c], scans[c]) is not None
        sync_all_of(big)
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        config4 = {"what": f"BASELINE configs[3]: {8 * world}M particles sharded over {world} GPUs (8M each), 1080 beams, multinomial resample every cycle",
                   "cycles_per_s": 6 / float(t.item()), "ms_per_cycle": float(t.item()) / 6 * 1e3}
        big.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        lf_ms, lf_count = prof["sensor_kernel"]
        lf_avg_s = (lf_ms / max(lf_count, 1)) * 1e-3
        bytes_lf = lf_algorithmic_bytes(n_local, BEAMS)
        achieved = bytes_lf / lf_avg_s if lf_avg_s > 0 else 0.0
        traffic = measured_lf_traffic(n_local) if not use_sharded else None
        patch_fraction = patch_fraction_timed = None
        if hasattr(filt, "counter"):
            planned, through = filt.counter("lf_patch_groups_planned"), filt.counter("lf_patch_groups_through")
            patch_fraction = through / planned if planned else None  # over the whole run (the repeat windows included)
            if patch_before and patch_after and patch_after[0] > patch_before[0]:
                patch_fraction_timed = (patch_after[1] - patch_before[1]) / (patch_after[0] - patch_before[0])
        out = {
            "metric": "MCL update cycles/sec (motion+sensor+resample), N particles x 1080 beams",
            # whole-job aggregate: one unit = one update cycle of 1M particles x 1080 beams (the configuration the metric is
            # quoted on); with N GPUs (weak scaling, 1M particles per GPU) every cycle of the logical N x 1M filter moves N such
            # units.  config.global_cycles_per_s is the rate of the logical filter itself.
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
                "parallelism": "1 GPU" if not use_sharded else f"particle shards x{world}: shard sums / CDF intervals / estimate sums all-gathered, "
                                                                    f"ancestors exchanged all-to-all (RCCL over xGMI, "
                                                                    f"{driver_used[0] or 'inside libbeluga_mcl.so'})",
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
                "kernel": "k_reweight_lf_patch (likelihood-field reweight; look-ups through per-workgroup LDS patches)",
                # The contract's roofline: algorithmic bytes (4 B per particle-beam look-up + the particle's state and
                # weight + the scan) over the kernel's HIP-event time, against the HBM peak.  It is a figure of merit, not HBM
                # utilisation: the table is L2 / LDS resident (see `traffic`, `hbm_utilisation`); the kernel is bound by vector
                # instruction issue (`limiter`).
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "frac_of_measured_copy_rate": achieved / HBM_MEASURED,
                "traffic": traffic,
                "hbm_utilisation": (traffic / lf_avg_s / HBM_PEAK) if (traffic and lf_avg_s > 0) else None,
                "algorithmic_bytes_per_launch": bytes_lf,
                "avg_launch_ms": lf_avg_s * 1e3,
                "launches": int(lf_count),
                "launches_sampled_every": 4,
                # What actually bounds it (profiles/r02_pmc_bench_1M.txt, r02_lf_series.txt): 10.2 vector instructions per wave and
                # beam - 4 v_fma_f64 for the end-point, 1 for the exactness check, 2 for the LDS address, 1 v_add_f64, the rest
                # prologue and gathered groups - at 4 cycles each on 7 of the 8 waves of a workgroup (the eighth fetches the
                # patches).  Groups of beams that do not fit a patch are gathered from global memory instead: they pay the
                # texture-address pipe (16 CU cycles per scattered 64-lane gather, profiles/r02_calib_gather_cost.txt) and their
                # workgroup waits for them at its next barrier - 2.6x the cost of a patched group; the kernel's time follows
                # their share (12-19 % in the first 20 cycles of a run, 2-4 % once the cloud has settled).
                "limiter": "vector instruction issue (VALU): ~10 instructions per wave-beam on 7/8 of the waves; gathered groups (2.6x a patched one): texture-address pipe + barrier waits",
                "groups_through_lds_patch": patch_fraction,
                "groups_through_lds_patch_in_timed_region": patch_fraction_timed,
                "valu_issue_floor_ms": n_local * BEAMS / 64 * 10.2 * 4 * (8 / 7) / (1024 * 2.4e9) * 1e3,
                "ta_floor_ms_if_all_gathered": n_local * BEAMS / 64 * 16 / (256 * 2.4e9) * 1e3,
            },
        }
        if not args.no_other_configs and world == 1 and n_local == 1_000_000:
            out["configs"] = other_configs(grid, cells, truth, controls, scans, filt, local_rank)
        if config4 is not None:
            out["configs"] = {"4": config4}
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

#!/usr/bin/env python3
"""bench.py — MCL update cycles/sec (motion + sensor + resample), N particles x 1080 beams (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full `Amcl::update` cycle (amcl_core.hpp:165-201): DifferentialDriveModel propagation,
LikelihoodFieldModel reweight over a 1080-beam scan, normalisation + policies, multinomial resample of all
particles, SE2 estimate.  Workload = BASELINE.json configs[1]: 1M particles per GPU, 4000x4000 @ 5 cm grid
(seed 42), resample every cycle.  Map, field and particles are resident in HBM before the timed region; each
step uploads one 1080-point scan (17 KB) and downloads the estimate, as the reference's caller would.
With --gpus N the particle set is sharded N ways (1M per rank, weak scaling) behind one logical filter; `value` is the whole
job's rate in units of a 1M-particle cycle: N x that filter's cycle rate (`config.filter_cycles_per_s`).

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
KERNEL_SOURCE = os.path.join(ROOT, "beluga_amd", "csrc", "kernels.hip")
PROFILES = os.path.join(ROOT, "profiles")
# per-launch counters of the LF kernel (rocprofv3 --pmc passes of the default bench, tools/gpu_r3_profiles.sh) and the measured
# issue cost of the instruction classes on this part (tools/calib_f64_rate.hip): what the VALU-issue roofline is computed from
PMC_FILES = [os.path.join(PROFILES, "r06_pmc_bench_1M.txt"), os.path.join(PROFILES, "r05_pmc_bench_1M.txt"), os.path.join(PROFILES, "r04_pmc_bench_1M.txt"), os.path.join(PROFILES, "r03_pmc_bench_1M.txt"), os.path.join(PROFILES, "r02_pmc_bench_1M.txt")]
TRAFFIC_FILES = [os.path.join(PROFILES, "r06_lf_kernel_traffic.json"), os.path.join(PROFILES, "r05_lf_kernel_traffic.json"), os.path.join(PROFILES, "r04_lf_kernel_traffic.json"), os.path.join(PROFILES, "r03_lf_kernel_traffic.json")]
CALIB_FILE = os.path.join(PROFILES, "r01_calib_f64_issue_rate.txt")
# the datasheet's issue rates (MI355X_MICROARCH.md, "Wave scheduling"): a wave64 instruction takes 2 passes of a SIMD-32, f64 at half rate 4
SPEC_CYCLES = {"fma_f64": 4.0, "mul_f64": 4.0, "add_f64": 4.0, "other": 2.0}
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9  # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz (the calibration quotes its cycles at this clock)


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


def calibrated_issue_cycles():
    """Cycles per wave64 instruction and SIMD by class, from profiles/r01_calib_f64_issue_rate.txt (measured with 8 waves per
    SIMD and independent chains: the best the vector unit does, quoted at 2.4 GHz)."""
    out = {}
    with open(CALIB_FILE) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) >= 5 and parts[0].startswith("v_") and parts[3] == "->":
                out[parts[0]] = float(parts[4])
    return {"fma_f64": out["v_fma_f64"], "mul_f64": out["v_mul_f64"], "add_f64": out["v_add_f64"], "other": out["v_add_u32"]}


IN_RUN_COUNTERS = {}  # filled by collect_in_run_counters(): {counter: per-launch average of the LF kernel}, measured by this very run
IN_RUN_BY_KERNEL = {}  # {kernel: {counter: per-launch average, "avg_us": launch duration, "calls": launches}}: every kernel of the cycle


def lf_kernel_counters(kernel="k_reweight_lf_patch"):
    """Per-launch PMC averages of the LF kernel: measured by this run if it could (collect_in_run_counters), otherwise from the newest
    tracked profile that has them: {counter: value}, the source, and whether it was collected on the LF kernels' current source (a
    tracked file's header records the SHA-256 of that part of kernels.hip)."""
    if "SQ_INSTS_VALU" in IN_RUN_COUNTERS:
        return dict(IN_RUN_COUNTERS), "in-run rocprofv3 --pmc pass of this bench.py (--pmc-child)", True
    for path in PMC_FILES:
        counters, sha = {}, None
        try:
            with open(path) as fh:
                for line in fh:
                    parts = line.replace(", ", ",").split()  # (a template argument list is part of the kernel's name)
                    if line.startswith("# lf_kernels_sha256") and len(parts) >= 3:
                        sha = parts[2]
                    if len(parts) >= 4 and parts[0] == "PMC" and parts[1].split("<")[0] == kernel:
                        counters[parts[2]] = float(parts[3])
        except OSError:
            continue
        if "SQ_INSTS_VALU" in counters:
            try:
                current = sha == lf_kernel_source_sha(KERNEL_SOURCE)
            except OSError:
                current = False
            return counters, os.path.relpath(path, ROOT), current
    return None, None, False


def valu_issue_floor_ms(n_particles: int):
    """The time the LF kernel's vector instructions take to ISSUE at the measured rate of each class, all 1024 SIMDs busy:
    sum over classes of (instructions per launch x cycles per instruction) / (SIMDs x clock).  Instruction counts by class from
    SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 where the profile has them (the rest - integer, moves, conversions - at the 32-bit rate);
    a profile with the total only prices everything at the nominal 4 cycles of a wave64 instruction."""
    counters, source, current = lf_kernel_counters()
    if counters is None:
        return None
    scale = n_particles / 1_000_000  # the profile is of the 1M-particle bench; the kernel's work is linear in the particles
    total = counters["SQ_INSTS_VALU"] * scale
    detail = {"source": source, "source_is_current_kernel": current, "valu_instructions_per_launch": total,
              "valu_per_wave_beam": total / (n_particles * BEAMS / 64)}
    try:
        cyc = calibrated_issue_cycles()
    except (OSError, KeyError):
        cyc = None
    classes = {k: counters.get(name) for k, name in (("fma_f64", "SQ_INSTS_VALU_FMA_F64"), ("mul_f64", "SQ_INSTS_VALU_MUL_F64"),
                                                     ("add_f64", "SQ_INSTS_VALU_ADD_F64"))}
    if cyc is not None and all(v is not None for v in classes.values()):
        classes = {k: v * scale for k, v in classes.items()}
        classes["other"] = max(total - sum(classes.values()), 0.0)
        cycles = sum(classes[k] * cyc[k] for k in classes)
        detail.update({"instructions_by_class": classes, "cycles_per_instruction": cyc, "pricing": "measured issue cost per class"})
    else:
        cycles = total * 4.0
        detail.update({"pricing": "4 cycles per wave64 instruction (no class counters in the profile)"})
    detail["issue_floor_ms"] = cycles / (SIMDS * CLOCK_HZ) * 1e3
    detail["peak_winstr_per_s"] = total / (cycles / (SIMDS * CLOCK_HZ))  # instructions of this mix the chip can issue per second
    # the same mix at the datasheet's rates (the calibration above absorbs the clock this load sustains, ~1.65 GHz of the nominal 2.4)
    if isinstance(classes, dict) and "other" in classes:
        spec_cycles = sum(classes[k] * SPEC_CYCLES[k] for k in classes)
    else:
        spec_cycles = total * 2.0
    detail["issue_floor_ms_spec_rates"] = spec_cycles / (SIMDS * CLOCK_HZ) * 1e3
    return detail


def collect_in_run_counters(timeout_s=150, child="lf", by_kernel=None, lf_counters=None):
    """Three more short runs of the headline workload (--pmc-child: 5 + 8 cycles) under `rocprofv3 --kernel-trace --pmc ...`, one per
    counter set as the guide prescribes (the SQ instruction counters; FETCH_SIZE; WRITE_SIZE): the LF kernel's per-launch averages
    land in IN_RUN_COUNTERS.  Returns a short status string; on any failure the line falls back to the tracked profiles and says so."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return "rocprofv3 not found"
    passes = ["SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32", "FETCH_SIZE", "WRITE_SIZE"]
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    try:
        for counters in passes:
            with tempfile.TemporaryDirectory(dir="/tmp") as out:
                cmd = [tool, "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", child]
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
                db_path = None
                for root, _dirs, files in os.walk(out):
                    for name in files:
                        if name.endswith("_results.db"):
                            db_path = os.path.join(root, name)
                if r.returncode != 0 or db_path is None:
                    return f"rocprofv3 pass '{counters}' failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
                db = sqlite3.connect(db_path)
                rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
                try:  # launch durations of the same pass (its kernel trace): microseconds
                    durations = db.execute("select name, total_calls, average from top_kernels").fetchall()
                except sqlite3.Error:
                    durations = []
                db.close()

                def short(name):
                    return name.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", ", ",")

                dest = IN_RUN_BY_KERNEL if by_kernel is None else by_kernel
                for kernel, counter, value, launches in rows:
                    dest.setdefault(short(kernel), {})[counter] = float(value)
                    if "k_reweight_lf_patch" in kernel:
                        got[counter] = float(value)
                        got["launches"] = int(launches)
                for name, calls, average in durations:
                    rec = dest.setdefault(short(name), {})
                    rec.setdefault("calls", int(calls))
                    rec["avg_us"] = min(rec.get("avg_us", float("inf")), float(average))  # (the least perturbed of the three passes)
    except Exception as exc:  # a profiler that hangs or a database of another layout: the tracked profiles serve
        return f"in-run counters failed: {exc!r}"
    if "SQ_INSTS_VALU" not in got:
        return "no LF kernel rows in the profiler's database"
    (IN_RUN_COUNTERS if lf_counters is None else lf_counters).update(got)
    return "ok"


def collect_beam_counters(timeout_s=120):
    """One short run of configuration 5 (--pmc-child beam: 3 cycles) under `rocprofv3 --kernel-trace --pmc` with the SQ instruction
    counters: the ordered beam kernel's per-launch averages -> the roofline entry of configs["5"] (vector issue: the walk is integer
    and LDS work, its HBM traffic a rounding error).  Returns the entry, or a string saying why there is none."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return "rocprofv3 not found"
    counters = "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU"
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as out:
            cmd = [tool, "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "beam"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            db_path = None
            for root, _dirs, files in os.walk(out):
                for name in files:
                    if name.endswith("_results.db"):
                        db_path = os.path.join(root, name)
            if r.returncode != 0 or db_path is None:
                return f"rocprofv3 pass failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            db = sqlite3.connect(db_path)
            rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_reweight_beam_sorted%' group by kernel_name, counter_name").fetchall()
            durations = db.execute("select name, total_calls, average from top_kernels where name like '%k_reweight_beam_sorted%'").fetchall()
            db.close()
    except Exception as exc:
        return f"beam counters failed: {exc!r}"
    rec = {counter: float(value) for _k, counter, value, _n in rows}
    if "SQ_INSTS_VALU" not in rec or not durations:
        return "no beam kernel rows in the profiler's database"
    avg_us = float(durations[0][2])
    f64 = sum(rec.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
    valu = rec["SQ_INSTS_VALU"]
    floor_us = (f64 * 4.0 + (valu - f64) * 2.0) / (SIMDS * CLOCK_HZ) * 1e6
    wave_beams = 1_000_000 * BEAMS / 64
    return {"kernel": "k_reweight_beam_sorted (Bresenham walks over an LDS bit window: integer and LDS work)", "bound": "valu",
            "avg_launch_ms": avg_us * 1e-3, "launches": int(durations[0][1]), "valu_instructions_per_launch": valu,
            "valu_per_wave_beam": valu / wave_beams, "salu_per_wave_beam": rec.get("SQ_INSTS_SALU", 0.0) / wave_beams,
            "f64_per_wave_beam": f64 / wave_beams, "achieved": valu / (avg_us * 1e-6) / 1e9, "peak": SIMDS * CLOCK_HZ / 2.0 / 1e9,
            "unit": "G wave64-instr/s", "issue_floor_ms_spec_rates": floor_us * 1e-3, "frac": floor_us / avg_us,
            "pricing": "datasheet rates: 2 cycles per wave64 instruction, 4 for f64; 1024 SIMDs at 2.4 GHz",
            "source": "in-run rocprofv3 --pmc pass of this bench.py (--pmc-child beam), per launch"}


def fixed_10m_roofline(entry, with_counters):
    """`roofline` and `roofline_by_kernel` of the fixed 10M-particle configuration.  The LF kernel against its vector-issue floor exactly as the
    headline's (`frac` at the calibrated issue costs, `frac_spec` at the datasheet's), from counters of a 10M run of its own where
    rocprofv3 is at hand (`--pmc-child 10m`: three passes of 5 + 8 cycles - SQ instruction classes, FETCH_SIZE, WRITE_SIZE), else from the 1M
    counters scaled by the particle count (the kernel's work is linear in it); every kernel of the 10M cycle with its measured HBM bytes -
    the draw kernel's CDF no longer fits the L2s there: FETCH_SIZE against its algorithmic 80 B per output says what that costs."""
    n = 10_000_000
    by_kernel, lf = {}, {}
    status = collect_in_run_counters(timeout_s=240, child="10m", by_kernel=by_kernel, lf_counters=lf) if with_counters else "skipped (--no-pmc)"
    lf_s = entry["sensor_kernel_ms"] * 1e-3
    if "SQ_INSTS_VALU" in lf:
        saved = dict(IN_RUN_COUNTERS)
        IN_RUN_COUNTERS.clear()
        IN_RUN_COUNTERS.update({k: v * (1_000_000 / n) for k, v in lf.items() if k != "launches"})  # (valu_issue_floor_ms scales by n / 1M)
        floor = valu_issue_floor_ms(n)
        IN_RUN_COUNTERS.clear()
        IN_RUN_COUNTERS.update(saved)
        if floor:
            floor["source"] = "in-run rocprofv3 --pmc pass of this bench.py (--pmc-child 10m), per launch"
    else:
        floor = valu_issue_floor_ms(n)
        if floor:
            floor["source"] = str(floor.get("source")) + " (the 1M-particle counters scaled by the particle count)"
    traffic = (2.0 * lf["FETCH_SIZE"] + lf["WRITE_SIZE"]) * 1024.0 if ("FETCH_SIZE" in lf and "WRITE_SIZE" in lf) else None
    roofline = {"kernel": "k_reweight_lf_patch at 10M particles", "bound": "valu",
                "achieved": (floor["valu_instructions_per_launch"] / lf_s / 1e9) if (floor and lf_s > 0) else None,
                "peak": (floor["peak_winstr_per_s"] / 1e9) if floor else None, "unit": "G wave64-instr/s",
                "frac": (floor["issue_floor_ms"] * 1e-3 / lf_s) if (floor and lf_s > 0) else None,
                "frac_spec": (floor["issue_floor_ms_spec_rates"] * 1e-3 / lf_s) if (floor and lf_s > 0) else None,
                "traffic": traffic, "in_run_counters": status, "avg_launch_ms": entry["sensor_kernel_ms"],
                "launches": entry.get("sensor_kernel_launches_timed"), "valu_issue": floor,
                "algorithmic_bytes_per_launch": lf_algorithmic_bytes(n, BEAMS),
                "algorithmic_over_hbm_peak": lf_algorithmic_bytes(n, BEAMS) / lf_s / HBM_PEAK if lf_s > 0 else None}
    cycle_bytes = n * 4 * BEAMS + n * 44 + n * (4 * math.ceil(math.log2(n)) + 44) + 8 * BEAMS
    roofline["whole_cycle"] = {"algorithmic_bytes_per_cycle": cycle_bytes, "GBps": cycle_bytes / (entry["ms_per_cycle"] * 1e-3) / 1e9,
                               "over_hbm_peak_8.0TBps": cycle_bytes / (entry["ms_per_cycle"] * 1e-3) / HBM_PEAK}
    by = roofline_by_kernel(by_kernel, min_calls=8) if by_kernel else None
    if by and "k_resample_draw<true>" in by:
        by["k_resample_draw<true>"]["algorithmic_bytes"] = 80 * n  # 32 read + 40 written + its share of the CDF, per output
    return {"roofline": roofline, "roofline_by_kernel": by or None}


def roofline_by_kernel(by_kernel=None, min_calls=8):
    """Every kernel of the headline cycle against the two rooflines that can bind it, from the in-run counter passes (per launch):
    HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 tallies a read at half its size) against 8.0 TB/s, and the issue time of its
    vector instructions at the datasheet's rates (f64 classes 4 cycles per wave64 instruction, the rest 2; 1024 SIMDs at 2.4 GHz);
    `bound` = the larger of the two floors, `frac` = that floor / the launch's duration (the counter passes' own kernel trace: the
    first 13 cycles of the filter, cloud still wide).  Kernels that ran in the passes but not in a cycle (map set-up) are left out."""
    out = {}
    for name, rec in sorted((IN_RUN_BY_KERNEL if by_kernel is None else by_kernel).items()):
        if "avg_us" not in rec or "SQ_INSTS_VALU" not in rec or rec.get("calls", 0) < min_calls:
            continue
        f64 = sum(rec.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
        valu = rec["SQ_INSTS_VALU"]
        valu_floor_us = (f64 * 4.0 + max(valu - f64, 0.0) * 2.0) / (SIMDS * CLOCK_HZ) * 1e6
        entry = {"avg_launch_us": rec["avg_us"], "launches": rec["calls"], "valu_instructions": valu, "valu_floor_us": valu_floor_us}
        hbm_floor_us = None
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            hbm = (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
            hbm_floor_us = hbm / HBM_PEAK * 1e6
            entry.update({"hbm_bytes": hbm, "hbm_floor_us": hbm_floor_us, "hbm_GBps": hbm / (rec["avg_us"] * 1e-6) / 1e9})
        bound = "hbm" if (hbm_floor_us or 0.0) > valu_floor_us else "valu"
        floor_us = max(hbm_floor_us or 0.0, valu_floor_us)
        entry.update({"bound": bound, "frac": floor_us / rec["avg_us"] if rec["avg_us"] > 0 else None})
        out[name] = entry
    return out


def pmc_child(kind="lf"):
    """What collect_in_run_counters profiles: the headline filter, 5 cycles of warm-up and 8 more (the same kernels as the timed region);
    kind "beam": three cycles of configuration 5 (BeamSensorModel, 1M x 1080) for collect_beam_counters."""
    from beluga_amd.amcl import Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, LikelihoodFieldModelParam, OccupancyGrid, se2_from_xytheta
    cells, truth, odoms, scans, _poses = make_workload(13 if kind in ("lf", "10m") else 3)
    grid = OccupancyGrid(cells, RESOLUTION, origin=se2_from_xytheta(ORIGIN[0], ORIGIN[1], 0.0))
    n = 10_000_000 if kind == "10m" else 1_000_000  # "10m": the same 5 + 8 cycles on the fixed 10M-particle filter
    if kind == "beam":
        b = Amcl(grid, DifferentialDriveModelParam(*ALPHAS), BeamModelParam(beam_max_range=MAX_RANGE), AmclParams(min_particles=n, max_particles=n), seed=42)
        b.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        for c in range(3):
            assert b.update(se2_from_xytheta(*odoms[c]), scans[c]) is not None
        b.sync()
        b.close()
        return
    f = Amcl(grid, DifferentialDriveModelParam(*ALPHAS), LikelihoodFieldModelParam(**LF), AmclParams(min_particles=n, max_particles=n), seed=42)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    for c in range(13):
        assert f.update(se2_from_xytheta(*odoms[c]), scans[c]) is not None
    f.sync()
    f.close()


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
    return cells, truth, odoms, scans, poses


def cpu_baseline(cells, truth, odoms, scans, n_full: int, budget_s: float = 14.0):
    """The oracle (CPU restatement of the reference path; kind 'port') as BASELINE.md section 2 asks: built -O3 -march=native
    (the timing build: oracle/Makefile `native`; the parity tests keep their -ffp-contract=off build), `seq` = one thread
    (std::execution::seq) and `par` = all host threads on the three loops the reference parallelises (propagate, reweight,
    the normalisation divide; the resampling is sequential in the reference whatever the policy: views/sample.hpp:128-136),
    per-stage milliseconds, median over >= 5 full update cycles, on bounded particle samples of the same workload scaled
    linearly in N."""
    from beluga_amd.amcl import se2_from_xytheta
    from oracle import binding as orc
    flags = orc.use_timing_build(True)
    try:
        threads = orc.max_threads()

        def run(n_sample, n_threads, share):
            f = orc.Amcl(min_particles=n_sample, max_particles=n_sample, alphas=ALPHAS, seed=42, threads=n_threads,
                         lf=(LF["max_obstacle_distance"], LF["max_laser_distance"], LF["z_hit"], LF["z_random"], LF["sigma_hit"]),
                         lf_model_unknown_space=LF["model_unknown_space"])
            f.set_map(cells, RESOLUTION, se2_from_xytheta(ORIGIN[0], ORIGIN[1], 0.0))
            f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
            f.update(se2_from_xytheta(*odoms[0]), scans[0])  # warm-up
            cycle_ms, stages = [], []
            t_begin = time.perf_counter()
            for c in range(1, len(odoms)):
                t0 = time.perf_counter()
                f.update(se2_from_xytheta(*odoms[c]), scans[c])
                cycle_ms.append((time.perf_counter() - t0) * 1e3)
                stages.append(f.stage_times())
                if len(cycle_ms) >= 5 and time.perf_counter() - t_begin > budget_s * share:
                    break
            del f
            med = float(np.median(cycle_ms))
            return {
                "threads": n_threads, "sample_particles": n_sample, "cycles_timed": len(cycle_ms), "ms_per_cycle_median": med,
                "stage_ms_median": {k: float(np.median([st[k] for st in stages]) * 1e3) for k in stages[0]},
                "ns_per_particle_beam": med * 1e6 / (n_sample * BEAMS),
                "cycles_per_s_scaled": 1e3 / med * n_sample / n_full,
            }

        par = run(32768 * max(1, threads // 8), threads, 0.6)
        seq = run(4096, 1, 0.4)
    finally:
        orc.use_timing_build(False)
    return {
        "value": par["cycles_per_s_scaled"],
        "unit": "cycles/s",
        "cores": threads,
        "kind": "port",
        "build": flags,
        "sample": f"par: {par['cycles_timed']} full update cycles of {par['sample_particles']} particles x {BEAMS} beams on {threads} threads "
                  f"(median {par['ms_per_cycle_median']:.1f} ms, {par['ns_per_particle_beam']:.2f} ns per particle-beam); seq: "
                  f"{seq['cycles_timed']} cycles of {seq['sample_particles']} particles on 1 thread (median {seq['ms_per_cycle_median']:.1f} ms); "
                  f"both scaled linearly to {n_full} particles",
        "par": par,
        "seq": seq,
        "note": "value = par.  The reference's resampling is sequential under either execution policy (views/sample.hpp:128-136; the "
                "lazy sample view is materialised by actions::assign on one thread), so `par` carries a serial stage that grows "
                "with N: see stage_ms_median.resample.",
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
    # fixed-size large sets: measured as the headline is - warm-up, ONE timed region of consecutive cycles (each returns its estimate:
    # one host synchronisation per cycle), repeat windows behind it, the LF kernel event-timed in every 4th cycle - so that the >=10M
    # target of BASELINE.json's north_star is a line of its own kind, with its own roofline, not a five-cycle side entry
    for label, n, steps, windows in (("fixed_8M", 8_000_000, 10, 1), ("fixed_10M", 10_000_000, 20, 3)):
        g = Amcl(grid, motion, LikelihoodFieldModelParam(**LF), AmclParams(min_particles=n, max_particles=n), seed=42, device=device)
        g.initialize(truth, cov)
        warm = 5
        for c in range(warm):
            assert g.update(controls[c], scans[c]) is not None
        g.profile_enable(1)
        g.profile_read(reset=True)
        g.sync()
        t0 = time.perf_counter()
        for c in range(warm, warm + steps):
            assert g.update(controls[c], scans[c]) is not None
        g.sync()
        elapsed = time.perf_counter() - t0
        prof = g.profile_read(reset=True)
        g.profile_enable(0)
        rates, c = [], warm + steps
        for _ in range(windows):
            g.sync()
            w0 = time.perf_counter()
            for _k in range(steps):
                assert g.update(controls[c], scans[c]) is not None
                c += 1
            g.sync()
            rates.append(steps / (time.perf_counter() - w0))
        lf_ms = prof["sensor_kernel"][0] / max(prof["sensor_kernel"][1], 1)
        out[label] = {"what": f"{n} particles x {BEAMS} beams, multinomial resample every cycle (same workload as the headline)",
                      "steps": steps, "warmup": warm, "ms_per_cycle": elapsed / steps * 1e3, "cycles_per_s": steps / elapsed,
                      "repeat_windows": {"steps_each": steps, "cycles_per_s": rates},
                      "particle_beam_evals_per_s": n * BEAMS * steps / elapsed,
                      "sensor_kernel_ms": lf_ms, "sensor_kernel_launches_timed": int(prof["sensor_kernel"][1]),
                      "algorithmic_GBps": lf_algorithmic_bytes(n, BEAMS) / (lf_ms * 1e-3) / 1e9 if lf_ms > 0 else None}
        if n == 10_000_000:
            # config 3 on the same capacity: KLD (eps .05, z 3) + selective resampling from a fresh 10M-particle set
            g.close()
            g = Amcl(grid, motion, LikelihoodFieldModelParam(**LF), AmclParams(min_particles=100_000, max_particles=n, selective_resampling=True),
                     seed=42, device=device)
            first, counts, fired_first = [], [], []
            for r in range(3):
                g.initialize(truth, cov)
                g.sync()
                t0 = time.perf_counter()
                assert g.update(controls[r], scans[r]) is not None
                g.sync()
                first.append((time.perf_counter() - t0) * 1e3)
                counts.append(g.last_info["num_particles"])
                fired_first.append(bool(g.last_info["resampled"]))
            cycles_after = []
            for k in range(8):  # the trajectory continues from the third fresh set: the KLD cut, then cycles at N_out
                n_in = g.last_info["num_particles"]
                ms = _timed_cycles(g, controls, scans, 3 + k, 1)[0]
                cycles_after.append({"N_in": n_in, "ms": ms, "resample_fires": bool(g.last_info["resampled"]),
                                     "N_out": g.last_info["num_particles"]})
            out["3"] = {"what": "BASELINE configs[2]: max 10M / min 100k particles, KLD (eps .05, z 3) + selective resampling (ESS < N/2)",
                        "cycle_at_10M_resample_does_not_fire": {"ms": first, "cycles_per_s": 1e3 / float(np.median(first)),
                                                                "N_out": counts, "resampled": fired_first},
                        "cycles_after_it": cycles_after,
                        "note": "from a fresh 10M-particle set the first cycle's ESS stays above N/2 (selective resampling: no "
                                "resample, N stays 10M); the next cycle fires: KLD cut 10M -> N_out; then the filter runs at N_out"}
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


def verify_run(filt, grid, estimates, true_poses, scan, rank):
    """What the timed run just did, checked (reported as `verified`, not timed): (1) the estimate of every timed cycle against
    the workload's true pose (the filter has to be localising, not just running); (2) one more reweight on the set as the run
    left it, 1024 sampled particles + the first and the last against the oracle (likelihood_field_model.hpp:68-91) at 1e-12."""
    from oracle import binding as orc
    out = {}
    pos, ang = [], []
    for est, truth in zip(estimates, true_poses):
        pos.append(math.hypot(est[2] - truth[0], est[3] - truth[1]))
        d = math.atan2(est[1], est[0]) - truth[2]
        ang.append(abs(math.atan2(math.sin(d), math.cos(d))))
    out["estimate_vs_true_pose"] = {"cycles": len(pos), "max_position_error_m": max(pos), "max_heading_error_rad": max(ang),
                                    "ok": bool(max(pos) < 0.25 and max(ang) < 0.05)}
    if hasattr(filt, "reweight") and hasattr(filt, "likelihood_field"):
        states, w0 = filt.particles()
        n = len(w0)
        filt.reweight(scan)
        w1 = filt.particles()[1]
        pick = np.unique(np.concatenate([np.random.Generator(np.random.MT19937(7)).choice(n, min(1024, n), replace=False), [0, n - 1]]))
        want = w0[pick] * orc.lf_weights(filt.likelihood_field(), RESOLUTION, grid.origin, LF["max_laser_distance"], states[pick],
                                        scan, threads=orc.max_threads())
        rel = float(np.max(np.abs(w1[pick] - want) / np.abs(want)))
        out["reweight_sample_vs_oracle"] = {"particles": int(len(pick)), "beams": int(len(scan)), "max_relative_error": rel,
                                            "tolerance": 1e-12, "ok": bool(rel <= 1e-12)}
    out["ok"] = all(v["ok"] for v in out.values())
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
    ap.add_argument("--config4-particles", type=int, default=8_000_000,
                    help="particles per GPU of the BASELINE configs[3] entry that several ranks add to the line (8M = the configuration; tests run it smaller)")
    ap.add_argument("--sharded", action="store_true", help="use the sharded driver even with one GPU (measures its overhead)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes behind the timed region (the line then quotes the tracked profiles)")
    ap.add_argument("--pmc-child", nargs="?", const="lf", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.pmc_child)
        return

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
    cells, truth, odoms, scans, true_poses = make_workload(steps_total)
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
    library_comm = use_sharded and type(filt).__name__ == "Amcl"  # (the torch.distributed driver is beluga_amd.sharded.ShardedAmcl)
    comm_before = (filt.counter("comm_bytes_out"), filt.counter("comm_collectives"), filt.counter("comm_host_syncs")) if library_comm else None
    timed_estimates = []
    t0 = time.perf_counter()
    for c in range(args.warmup, args.warmup + args.steps):
        est = filt.update(controls[c], scans[c])
        assert est is not None
        timed_estimates.append(est[0])  # (a list append: checked against the true poses after the timed region)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = filt.profile_read(reset=True)
    collective = None
    if comm_before is not None:
        # what the communicator moved in the timed region, from the library's own counters: the driver's check that RCCL ran over
        # all ranks (ranks_seen = ncclCommCount of the library's communicator) has a field to read
        backend_name = {0: "none (one rank)", 1: "caller's transport", 2: "RCCL inside libbeluga_mcl.so (ncclAllGather + grouped ncclSend / ncclRecv)"}
        collective = {"backend": backend_name.get(filt.counter("comm_backend"), "?"), "ranks_seen": filt.counter("comm_ranks_seen"),
                      "bytes_out_per_rank_per_cycle": (filt.counter("comm_bytes_out") - comm_before[0]) / args.steps,
                      "collectives_per_cycle": (filt.counter("comm_collectives") - comm_before[1]) / args.steps,
                      # counted by the library (comm_host_syncs): 1 with the fixed-capacity ancestor exchange, 2 with exact counts
                      "host_synchronisations_per_cycle": ((filt.counter("comm_host_syncs") - comm_before[2]) / args.steps) if world > 1 else 1,
                      "exchange_overflows": filt.counter("comm_overflows")}
    elif use_sharded:
        collective = {"backend": f"torch.distributed ({backend}) driver, beluga_amd/sharded.py", "ranks_seen": dist.get_world_size(),
                      "bytes_out_per_rank_per_cycle": None}
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
    verified = verify_run(filt, grid, timed_estimates, true_poses[args.warmup:args.warmup + args.steps], scans[min(c, len(scans) - 1)], rank) if rank == 0 else None

    # (read before the filter is closed below: everything rank 0 reports about it)
    patch_whole_run = (filt.counter("lf_patch_groups_planned"), filt.counter("lf_patch_groups_through")) if hasattr(filt, "counter") else None
    # BASELINE configs[3] with several ranks: 8M particles per GPU behind one logical filter (64M at 8 GPUs); cycles/s of
    # that filter, beside the metric.
    config4 = None
    if world > 1 and not args.no_other_configs:
        filt.close()
        big = make_filter(args.config4_particles)
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
        config4 = {"what": f"BASELINE configs[3]: {args.config4_particles * world / 1e6:g}M particles sharded over {world} GPUs "
                           f"({args.config4_particles / 1e6:g}M each), 1080 beams, multinomial resample every cycle",
                   "particles_per_gpu": args.config4_particles, "particles_total": args.config4_particles * world,
                   "cycles_per_s": 6 / float(t.item()), "ms_per_cycle": float(t.item()) / 6 * 1e3}
        big.close()

    if rank == 0:
        pmc_status = "skipped (--no-pmc)" if args.no_pmc else ("skipped (several ranks or another set size)" if (world != 1 or n_local != 1_000_000) else None)
        if pmc_status is None:
            filt.sync()
            pmc_status = collect_in_run_counters()
        ms_per_step = elapsed / args.steps * 1e3
        lf_ms, lf_count = prof["sensor_kernel"]
        lf_avg_s = (lf_ms / max(lf_count, 1)) * 1e-3
        bytes_lf = lf_algorithmic_bytes(n_local, BEAMS)
        achieved_bytes = bytes_lf / lf_avg_s if lf_avg_s > 0 else 0.0
        floor = valu_issue_floor_ms(n_local)
        traffic_bytes, traffic_source = None, None
        if "FETCH_SIZE" in IN_RUN_COUNTERS and "WRITE_SIZE" in IN_RUN_COUNTERS:
            traffic_bytes = (2.0 * IN_RUN_COUNTERS["FETCH_SIZE"] + IN_RUN_COUNTERS["WRITE_SIZE"]) * 1024.0
            traffic_source = "in-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench.py (--pmc-child), per launch of the LF kernel"
        else:
            for path in TRAFFIC_FILES:
                try:
                    with open(path) as fh:
                        rec = json.load(fh)
                    traffic_bytes = (2.0 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024.0 * (n_local / rec.get("particles", 1_000_000))
                    traffic_source = os.path.relpath(path, ROOT) + (" (collected on the LF kernels' current source)" if rec.get("kernels_hip_sha256") == lf_kernel_source_sha(KERNEL_SOURCE)
                                                                     else " (collected on an EARLIER version of the LF kernels' source)")
                    break
                except (OSError, KeyError, ValueError):
                    continue
        patch_fraction = patch_fraction_timed = None
        if patch_whole_run is not None:
            planned, through = patch_whole_run
            patch_fraction = through / planned if planned else None  # over the whole run (the repeat windows included)
            if patch_before and patch_after and patch_after[0] > patch_before[0]:
                patch_fraction_timed = (patch_after[1] - patch_before[1]) / (patch_after[0] - patch_before[0])
        # SURVEY 8(d)'s whole-cycle figure: algorithmic bytes of ONE cycle at the canonical element sizes over the cycle time
        cycle_bytes = n_local * 4 * BEAMS + n_local * 44 + n_local * (4 * math.ceil(math.log2(max(n_local, 2))) + 44) + 8 * BEAMS
        cycle_rate = cycle_bytes / (ms_per_step * 1e-3)
        roofline = {
            "kernel": "k_reweight_lf_patch (likelihood-field reweight; look-ups through per-workgroup LDS patches), 70 % of the cycle",
            # What binds this kernel is vector-instruction issue, not HBM: its table is read through LDS patches and L2 (measured
            # HBM traffic in profiles/: ~7 % of the algorithmic bytes), so the roofline reported is the issue rate of the kernel's own
            # instruction mix.  achieved / peak are in wave64 vector instructions per second; frac = issue floor / launch time.
            "bound": "valu",
            "achieved": (floor["valu_instructions_per_launch"] / lf_avg_s / 1e9) if (floor and lf_avg_s > 0) else None,
            "peak": (floor["peak_winstr_per_s"] / 1e9) if floor else None,
            "unit": "G wave64-instr/s",
            "frac": (floor["issue_floor_ms"] * 1e-3 / lf_avg_s) if (floor and lf_avg_s > 0) else None,
            # the same instructions priced at the datasheet's issue rates (2 cycles per wave64 instruction, 4 for f64) at the nominal 2.4 GHz
            "frac_spec": (floor["issue_floor_ms_spec_rates"] * 1e-3 / lf_avg_s) if (floor and lf_avg_s > 0) else None,
            "traffic": traffic_bytes,  # HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies a read at half its size), see traffic_source
            "traffic_source": traffic_source,
            "in_run_counters": pmc_status,
            "avg_launch_ms": lf_avg_s * 1e3,
            "launches": int(lf_count),
            "launches_sampled_every": 4,
            "valu_issue": floor,
            # The contract's byte figure, kept beside it (NOT the binding roofline: above 1 where the table never leaves LDS / L2)
            "algorithmic_bytes_per_launch": bytes_lf,
            "algorithmic_GBps": achieved_bytes / 1e9,
            "algorithmic_over_hbm_peak": achieved_bytes / HBM_PEAK,
            "algorithmic_over_measured_copy_rate": achieved_bytes / HBM_MEASURED,
            "whole_cycle": {"algorithmic_bytes_per_cycle": cycle_bytes, "GBps": cycle_rate / 1e9,
                            "over_hbm_peak_8.0TBps": cycle_rate / HBM_PEAK, "over_measured_copy_rate_6.29TBps": cycle_rate / HBM_MEASURED},
            "groups_through_lds_patch": patch_fraction,
            "groups_through_lds_patch_in_timed_region": patch_fraction_timed,
        }
        out = {
            # (N = 1: BASELINE.json's metric as it stands.  Several GPUs: the contract wants the whole job's aggregate in `value` - the
            # string then says that it is an aggregate, so that nobody reads N x the rate as the rate at which estimates come out)
            "metric": "MCL update cycles/sec (motion+sensor+resample), N particles x 1080 beams" + (
                "" if world == 1 else f" [aggregate over {world} GPUs: {world} x config.filter_cycles_per_s, in cycles of "
                                      f"config.particles_per_gpu particles - ONE logical filter of config.particles_total particles produces "
                                      f"estimates at config.filter_cycles_per_s]"),
            # the whole job's rate in the metric's unit, a cycle of particles_per_gpu particles x 1080 beams: with N GPUs the ranks run ONE
            # logical filter of N x particles_per_gpu particles together (weak scaling), every cycle of which is N such units -
            # config.filter_cycles_per_s is that filter's own cycle rate (= value at N = 1)
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
                "collective": collective,
                "particle_beam_evals_per_s": n_total * BEAMS * args.steps / elapsed,
                "filter_cycles_per_s": args.steps / elapsed,
                "value_counts": "cycles of particles_per_gpu particles: n_gpus x filter_cycles_per_s (one logical filter of particles_total particles)",
                "units_of_1M_particle_cycles_per_s": world * args.steps / elapsed * (n_local / 1_000_000),
            },
            "timed_region_s": elapsed,
            "repeat_windows": {"steps_each": args.steps, "cycles_per_s": window_rates,
                               "min": min(window_rates) if window_rates else None,
                               "median": float(np.median(window_rates)) if window_rates else None},
            "stage_ms": {k: (v[0] / max(v[1], 1)) for k, v in stage_prof.items()},
            "roofline": roofline,
            "roofline_by_kernel": roofline_by_kernel() or None,
            "verified": verified,
        }
        if not args.no_other_configs and world == 1 and n_local == 1_000_000:
            out["configs"] = other_configs(grid, cells, truth, controls, scans, filt, local_rank)
            if not args.no_pmc and "5" in out["configs"]:
                out["configs"]["5"]["roofline"] = collect_beam_counters()
            if "fixed_10M" in out["configs"]:
                out["configs"]["fixed_10M"].update(fixed_10m_roofline(out["configs"]["fixed_10M"], not args.no_pmc))
        if config4 is not None:
            out["configs"] = {"4": config4}
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: the other ranks would sit in a barrier behind it)
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

"""f1, the deciding experiment: does the reference's distance map depend on the pop ORDER of equal keys?

Runs tools/exp_field_tie_order.cpp (CPU only) on turtlebot3_world, the 4000x4000 bench map and five adversarial maps and counts,
against the reference's own std::priority_queue (variant 0 = what map_build.cpp runs), the cells whose squared distance differs when
equal keys pop FIFO, LIFO, by cell index, or in a random order; beside them the cells where variant 0 differs from the exact
Euclidean transform (what the device build computes).

    python tools/exp_field_tie_order.py [--out profiles/r05_field_tie_order.json]
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beluga_amd import synth  # noqa: E402

SRC = os.path.join(ROOT, "tools", "exp_field_tie_order.cpp")
LIB = os.path.join(ROOT, "build", "libexp_field_tie_order.so")
VARIANTS = {0: "std::priority_queue (reference)", 1: "FIFO buckets", 2: "LIFO buckets", 3: "ties by cell index", 4: "random ties"}


def load():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    lib = ctypes.CDLL(LIB)
    lib.field_tie_variant.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def run(lib, seeds: np.ndarray, res: float, max_d: float, variant: int, seed: int = 0):
    H, W = seeds.shape
    s = np.ascontiguousarray(seeds, dtype=np.uint8)
    out = np.empty((H, W), dtype=np.float32)
    back = ctypes.c_uint64(0)
    t = time.perf_counter()
    rc = lib.field_tie_variant(s.ctypes.data, W, H, res, max_d, variant, seed, out.ctypes.data, ctypes.byref(back))
    assert rc == 0
    return out, back.value, time.perf_counter() - t


def adversarial_maps():
    rng = np.random.Generator(np.random.MT19937(7))
    maps = {}
    g = np.zeros((512, 512), np.uint8)  # a staircase: obstacle cells touching at corners only
    i = np.arange(40, 470)
    g[i, i] = 1
    maps["diagonal_wall_corner_touching"] = g
    g = (rng.random((768, 768)) < 0.01).astype(np.uint8)  # salt: thousands of equal-distance meetings
    maps["salt_1pct"] = g
    g = np.zeros((600, 600), np.uint8)  # a lattice of single cells, 9 apart: every cell between them is a tie
    g[4::9, 4::9] = 1
    maps["lattice_9"] = g
    g = np.zeros((640, 640), np.uint8)  # rings: all of a circle's cells at (nearly) the same distance from the inside
    yy, xx = np.mgrid[:640, :640]
    r = np.hypot(yy - 320.5, xx - 320.5)
    g[(np.abs(r - 60) < 0.7) | (np.abs(r - 170) < 0.7) | (np.abs(r - 290) < 0.7)] = 1
    maps["rings"] = g
    g = np.zeros((512, 512), np.uint8)  # slanted walls of rational slopes (2:1, 3:1, 3:2) and their mirror images
    for k, (a, b) in enumerate(((2, 1), (3, 1), (3, 2), (1, 2), (1, 3), (2, 3))):
        t = np.arange(0, 150)
        x = 30 + (t * a) // max(a, b) + 70 * k
        y = 30 + (t * b) // max(a, b) + 40 * (k % 2)
        ok = (x < 512) & (y < 512)
        g[y[ok], x[ok]] = 1
        g[511 - y[ok], x[ok]] = 1
    maps["slanted_walls"] = g
    return maps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_field_tie_order.json"))
    ap.add_argument("--skip-bench-map", action="store_true")
    args = ap.parse_args()
    lib = load()
    from scipy import ndimage

    cases = []
    z = np.load(os.path.join(ROOT, "tests", "golden", "turtlebot3_world_grid.npz"))
    cells = z["cells"]
    cases.append(("turtlebot3_world_384x384", (cells == synth.OCCUPIED).astype(np.uint8), float(z["resolution"]), 2.0))
    for name, g in adversarial_maps().items():
        cases.append((name, g, 0.05, 2.0))
    if not args.skip_bench_map:
        cells = synth.make_rooms_map(4000, 4000, seed=42)
        cases.append(("bench_rooms_4000x4000", (cells == synth.OCCUPIED).astype(np.uint8), 0.05, 2.0))

    report = {"what": "cells whose squared distance (float) differs from the reference's std::priority_queue wavefront "
                      "(distance_map.hpp:55-98) when equal keys pop in another order; max_obstacle_distance 2.0 m, 5 cm cells",
              "variants": VARIANTS, "maps": {}}
    for name, seeds, res, max_d in cases:
        ref, back0, t0 = run(lib, seeds, res, max_d, 0)
        reached = int(np.count_nonzero(ref < np.float32(max_d * max_d)))
        edt = ndimage.distance_transform_edt(seeds == 0).astype(np.float64)
        # the exact transform's squared distance through the same expression as the wavefront's (cells -> metres, float)
        edt_sq = np.minimum((edt * res) ** 2, max_d * max_d)
        not_edt = int(np.count_nonzero(np.abs(ref.astype(np.float64) - edt_sq) > 1e-5 * np.maximum(edt_sq, 1e-9)))
        row = {"cells": int(seeds.size), "seeds": int(seeds.sum()), "reached_below_cap": reached,
               "non_monotone_pops_reference": back0, "seconds_reference": round(t0, 3),
               "cells_where_reference_is_not_the_exact_transform": not_edt, "differs_from_reference": {}}
        for v in (1, 2, 3, 4):
            d, back, t = run(lib, seeds, res, max_d, v, seed=11)
            diff = d != ref
            n = int(np.count_nonzero(diff))
            row["differs_from_reference"][VARIANTS[v]] = {
                "cells": n, "share_of_reached": (n / reached if reached else 0.0),
                "max_abs_diff_m2": float(np.max(np.abs(d[diff] - ref[diff]))) if n else 0.0,
                "non_monotone_pops": back, "seconds": round(t, 3)}
        report["maps"][name] = row
        print(name, json.dumps(row, indent=None))
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()

"""Builds libbeluga_mcl.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python -m beluga_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbeluga_mcl.so")
SOURCES = ["kernels.hip", "beam_kernels.hip", "context.hip", "map_build.cpp"]
HEADERS = ["kernels.h", "device_common.hpp", "se2.h", "rng.h", "map_build.h", os.path.join(ROOT, "include", "beluga_mcl.h")]
# -ffp-contract=off: see the header comment of kernels.hip (floor() parity with the reference's arithmetic).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("BELUGA_MCL_EXTRA_CXXFLAGS", "").split()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MCL kernels cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cc = hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [cc] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

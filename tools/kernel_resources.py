"""Registers, scratch and occupancy of every kernel of kernels.hip and beam_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage; CPU only).

    python tools/kernel_resources.py [--all] [--out profiles/r05_kernel_resources.txt]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--out")
    args = ap.parse_args()
    t = ""
    for name in ("kernels.hip", "beam_kernels.hip"):
        src = os.path.join(ROOT, "beluga_amd", "csrc", name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.dirname(src), "-x", "hip", "-c", src, "-o", "/tmp/kernel_resources.o", "-Rpass-analysis=kernel-resource-usage"]
        t += subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
    lines = []
    for b in blocks:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            pass
        name = re.sub(r"^void mcl::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)
        row = (name, g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
        if args.all or row[3] > 0 or re.search(r"patch|draw|propagate|sort_|beam_sorted|palette", name):
            lines.append(f"{row[0]:60s} vgpr {row[1]:3d} sgpr {row[2]:3d} scratch {row[3]:4d} B/lane  occupancy {row[4]}  static lds {row[5]}")
    lines.append(f"{len(blocks)} kernels")
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    sys.exit(main())

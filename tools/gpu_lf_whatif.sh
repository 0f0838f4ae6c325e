#!/bin/bash
# Timing-only what-if builds of the LF patch kernel (results are WRONG in these builds) on a fixed cloud: what each dependency costs.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== variant: $flags"
  BELUGA_MCL_EXTRA_CXXFLAGS="$flags" python -m beluga_amd.build --force > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; continue; }
  timeout 300 python tools/exp_lf_fixed.py 2>/dev/null
done
python -m beluga_amd.build --force > /dev/null 2>&1

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1
BELUGA_MCL_EXTRA_CXXFLAGS="-DMCL_BEAM_STATS" python -m beluga_amd.build --force > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
timeout 600 python tools/exp_beam_stats.py 2>/dev/null | tee gpurun_out/beam_stats.txt
python -m beluga_amd.build --force > /dev/null 2>&1

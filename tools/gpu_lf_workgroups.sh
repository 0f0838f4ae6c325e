#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1
bash tools/build_variant.sh lftw "-DMCL_LF_TIMING -DMCL_LF_TIMING_COARSE" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
BELUGA_MCL_LIB=build/variants/lftw/libbeluga_mcl.so timeout 600 python tools/exp_lf_workgroups.py 2>/dev/null | tee gpurun_out/lf_workgroups.txt

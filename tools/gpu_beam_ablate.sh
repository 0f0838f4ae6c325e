#!/bin/bash
# config 5 (beam model, 1M x 1080) with the product build and with timing builds of the beam kernel (tools/build_variant.sh <name> -DMCL_BEAM_ABLATE=<bits>)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 CONFIG5_ALL_STEPS=1
echo "== product"; python tools/bench_configs.py 5 --steps 6 2>/dev/null | grep -o '"ms": [0-9.]*\|all_ms.*' | tr '\n' ' '; echo
for bits in "$@"; do
  bash tools/build_variant.sh beam$bits "-DMCL_BEAM_ABLATE=$bits" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; continue; }
  echo "== MCL_BEAM_ABLATE=$bits"; BELUGA_MCL_LIB=build/variants/beam$bits/libbeluga_mcl.so python tools/bench_configs.py 5 --steps 6 2>/dev/null | grep -o '"ms": [0-9.]*\|all_ms.*' | tr '\n' ' '; echo
done

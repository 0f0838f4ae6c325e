#!/bin/bash
# tools/gpu_lf_variants.sh "<name>:<flags>" ...: builds each variant of the kernels and runs tools/exp_lf_variant.py with it (and with the product build)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1
: > gpurun_out/lf_variants.txt
timeout 300 python tools/exp_lf_variant.py 2>/dev/null | tee -a gpurun_out/lf_variants.txt
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ -f build/variants/$name/libbeluga_mcl.so ] || bash tools/build_variant.sh $name "$flags" > gpurun_out/build_$name.log 2>&1 || { echo BUILD FAILED $name; tail -20 gpurun_out/build_$name.log; continue; }
  BELUGA_MCL_LIB=build/variants/$name/libbeluga_mcl.so timeout 300 python tools/exp_lf_variant.py 2>/dev/null | tee -a gpurun_out/lf_variants.txt
done

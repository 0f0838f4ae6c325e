#!/bin/bash
# timing-only variants of k_reweight_lf_pipe (tools/build_variant.sh pipe<N> "-DMCL_PIPE_ABLATE=<N>")
export BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then unset BELUGA_MCL_LIB; echo "== product"; else export BELUGA_MCL_LIB=$PWD/build/variants/$v/libbeluga_mcl.so; echo "== $v"; fi
  python tools/exp_pipe_ab.py 30 2>&1 | grep -v amdgpu.ids | head -1
done

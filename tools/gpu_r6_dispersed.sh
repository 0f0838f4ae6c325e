#!/bin/bash
# tools/exp_dispersed.py over library builds (LIBS: names under build/variants/, "product" = the package's own) -> gpurun_out/r06/exp_dispersed.txt
set -u
mkdir -p gpurun_out/r06
O=gpurun_out/r06/exp_dispersed.txt
: > $O
for lib in ${LIBS:-product}; do
  echo "== lib $lib" >> $O
  if [ "$lib" = "product" ]; then
    timeout 600 python tools/exp_dispersed.py "$@" 2>&1 | grep -v "^\[beluga_amd\]" >> $O
  else
    BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 BELUGA_MCL_LIB=build/variants/$lib/libbeluga_mcl.so timeout 600 python tools/exp_dispersed.py "$@" 2>&1 | grep -v "^\[beluga_amd\]" >> $O
  fi
done
cat $O

#!/bin/bash
# A/B of library builds on one box: LIBS = space-separated names under build/variants/ ("product" = the package's own library),
# ARGS = arguments of tools/exp_r5_ab.py (option sets); the builds alternate, ROUNDS times.   -> gpurun_out/$OUT
set -u
mkdir -p gpurun_out
O=gpurun_out/${OUT:-r5_ab.log}
: > $O
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in $LIBS; do
    echo "== round $r lib $lib" >> $O
    if [ "$lib" = "product" ]; then
      python tools/exp_r5_ab.py --reps 1 ${CYCLES:+--cycles $CYCLES} ${PARTICLES:+--particles $PARTICLES} "$@" 2>&1 | grep -v "^\[beluga_amd\]" >> $O
    else
      BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 BELUGA_MCL_LIB=build/variants/$lib/libbeluga_mcl.so python tools/exp_r5_ab.py --reps 1 ${CYCLES:+--cycles $CYCLES} ${PARTICLES:+--particles $PARTICLES} "$@" 2>&1 | grep -v "^\[beluga_amd\]" >> $O
    fi
  done
done
grep "^==\|^rep" $O

#!/bin/bash
# Per-kernel launch averages (rocprofv3 --kernel-trace --stats) of N cycles of the bench workload, for each library build in LIBS
# ("product" = the package's own; others under build/variants/).   -> gpurun_out/$OUT
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${OUT:-r5_kernels.txt}
: > $O
for lib in ${LIBS:-product}; do
  echo "== lib $lib  args ${ARGS:-}" >> $O
  rm -rf /tmp/r5k; mkdir -p /tmp/r5k
  if [ "$lib" = "product" ]; then
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r5k -o t -- python $GRAFT_REPO_ROOT/tools/exp_cycles.py ${ARGS:-} > /tmp/r5k/log 2>&1)
  else
    (cd /tmp && BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 BELUGA_MCL_LIB=$GRAFT_REPO_ROOT/build/variants/$lib/libbeluga_mcl.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r5k -o t -- python $GRAFT_REPO_ROOT/tools/exp_cycles.py ${ARGS:-} > /tmp/r5k/log 2>&1)
  fi
  db=$(find /tmp/r5k -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db >> $O; else tail -5 /tmp/r5k/log >> $O; fi
done
cat $O

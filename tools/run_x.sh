#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for v in da1 da2 da4 da7 ""; do
  if [ -z "$v" ]; then unset BELUGA_MCL_LIB; echo "== product"; else export BELUGA_MCL_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libbeluga_mcl.so; echo "== $v"; fi
  cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/tr
  timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/tr -o trace -- python $GRAFT_REPO_ROOT/tools/exp_fixed.py 6 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_summary.py gpurun_out/prof/tr/trace_results.db | grep "resample_draw"
done
rm -rf gpurun_out/prof/tr

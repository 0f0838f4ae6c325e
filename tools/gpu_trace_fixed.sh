#!/bin/bash
# Kernel trace of a fixed-size filter (N from the environment, default 10M) -> gpurun_out/kernel_trace_fixed_$N.txt
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
N=${N:-10000000}
cd /tmp
N=$N timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/tf -o trace -- python $GRAFT_REPO_ROOT/tools/exp_fixed.py ${1:-8} 2>/dev/null | grep "^N "
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/tf/trace_results.db | tee gpurun_out/kernel_trace_fixed_$N.txt
python tools/timeline.py gpurun_out/prof/tf/trace_results.db 2 > gpurun_out/timeline_fixed_$N.txt
rm -rf gpurun_out/prof/tf

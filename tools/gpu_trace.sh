#!/bin/bash
# rocprofv3 kernel trace of the default bench (short), summary dumped as CSV-ish text under gpurun_out/prof/.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db | tee gpurun_out/prof/kernel_stats.txt

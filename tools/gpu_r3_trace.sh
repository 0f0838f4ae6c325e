#!/bin/bash
# kernel trace + one cycle's dispatch timeline of the default bench
set -u
mkdir -p gpurun_out/r03 gpurun_out/prof
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > $O/trace_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db > $O/kernel_trace_bench_1M.txt
python tools/timeline.py gpurun_out/prof/trace/trace_results.db 2 > $O/timeline_bench_1M.txt
rm -rf gpurun_out/prof/trace
cat $O/kernel_trace_bench_1M.txt; cat $O/timeline_bench_1M.txt

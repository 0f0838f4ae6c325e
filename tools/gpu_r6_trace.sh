#!/bin/bash
# Round 6, quick look: kernel trace + timeline of the default bench (20 cycles after 5) -> gpurun_out/r06/<tag>_*.  Arguments: a tag and
# optional environment assignments for the bench (BELUGA_MCL_<OPTION>=value ...).
set -u
TAG=${1:-trace}; shift || true
mkdir -p gpurun_out/prof gpurun_out/r06
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
BENCH="python $GRAFT_REPO_ROOT/bench.py --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs --no-pmc"
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- $BENCH --steps 20 --warmup 5 > $O/${TAG}_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db > $O/${TAG}_kernel_trace_bench_1M.txt
python tools/timeline.py gpurun_out/prof/trace/trace_results.db 2 > $O/${TAG}_timeline_bench_1M.txt
rm -rf gpurun_out/prof/trace
head -12 $O/${TAG}_kernel_trace_bench_1M.txt

#!/bin/bash
# rocprofv3 kernel traces of BASELINE configs 3 (10M particles, KLD + selective) and 5 (beam model, 1M x 1080)
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace_c3 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 3 --steps 8 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace_c3.err
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace_c5 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace_c5.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace_c3/trace_results.db > gpurun_out/prof/kernel_stats_config3.txt
python tools/rocpd_summary.py gpurun_out/prof/trace_c5/trace_results.db > gpurun_out/prof/kernel_stats_config5.txt
head -12 gpurun_out/prof/kernel_stats_config3.txt; head -6 gpurun_out/prof/kernel_stats_config5.txt

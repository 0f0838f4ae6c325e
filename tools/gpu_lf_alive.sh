#!/bin/bash
set -u
mkdir -p gpurun_out/prof gpurun_out/r06
export TMPDIR=/tmp
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/al
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/prof/al -o pmc -- python $GRAFT_REPO_ROOT/bench.py --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs --no-pmc --steps 40 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/al.err || tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/al.err
cd $GRAFT_REPO_ROOT
python tools/lf_alive.py gpurun_out/prof/al/pmc_results.db > gpurun_out/r06/lf_alive.txt 2>&1
rm -rf gpurun_out/prof/al
cat gpurun_out/r06/lf_alive.txt

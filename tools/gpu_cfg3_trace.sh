#!/bin/bash
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests -m gpu -x -q -k "kld or end_to_end" 2>&1 | tail -4
timeout 300 python tools/bench_configs.py 3 --steps 6 --max3 1000000 2>&1 | grep config3
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace3 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 3 --steps 8 --max3 1000000 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace3.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace3/trace_results.db | head -30

#!/bin/bash
# Round-2 loop: selected parity tests (stop at first failure), a short bench, dispatch timeline and kernel stats.
# usage: gpu_r2_loop.sh [pytest -k expression | all]
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
SEL="${1:-all}"
if [ "$SEL" = "all" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
elif [ "$SEL" != "none" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -k "$SEL" 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>gpurun_out/bench.err | tee gpurun_out/bench_short.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'windows', [round(x) for x in d['repeat_windows']['cycles_per_s']], {k: round(v,4) for k,v in d['stage_ms'].items()}, 'lf_ms', round(d['roofline']['avg_launch_ms'],4))"
tail -3 gpurun_out/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db | head -30 | tee gpurun_out/kernel_stats.txt
python tools/timeline.py gpurun_out/prof/trace/trace_results.db 1 | tee gpurun_out/timeline.txt
rm -rf gpurun_out/prof/trace

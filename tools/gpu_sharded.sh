#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/sharded_tests.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/sharded_tests.log | tail -20
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sharded 2>gpurun_out/bench_sharded.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sharded world1: cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})"
tail -3 gpurun_out/bench_sharded.err

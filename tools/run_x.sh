#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -n 5
bash tools/gpu_r3_trace.sh 2>&1 | sed -n 1,6p
timeout 300 python bench.py --steps 20 --warmup 5 --windows 3 --stage-steps 0 --no-cpu-baseline --no-other-configs 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'], d['repeat_windows']['cycles_per_s'])"
N=10000000 python tools/exp_fixed.py 8 | tail -n 1

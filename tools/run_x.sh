#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -n 3
python tools/exp_lf_fixed.py 2>&1 | grep sigma
timeout 300 python bench.py --steps 20 --warmup 5 --windows 3 --stage-steps 0 --no-cpu-baseline --no-other-configs 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified']['ok'], d['repeat_windows']['cycles_per_s'], d['roofline']['avg_launch_ms'])"

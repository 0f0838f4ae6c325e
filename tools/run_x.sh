#!/bin/bash
# scratch: parity check of the default build, trace, then the draw-occupancy variants
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -n 5
bash tools/gpu_r3_trace.sh 2>&1 | head -n 40
for v in dw4 dw6 dw7; do
  echo "== $v"
  BELUGA_MCL_LIB=$PWD/build/variants/$v/libbeluga_mcl.so timeout 300 python bench.py --steps 20 --warmup 5 --windows 3 --stage-steps 0 --no-cpu-baseline --no-other-configs 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('verified'), d.get('repeat_windows', d.get('windows')))"
done
echo "== default"
timeout 300 python bench.py --steps 20 --warmup 5 --windows 3 --stage-steps 0 --no-cpu-baseline --no-other-configs 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('verified'), d.get('repeat_windows', d.get('windows')))"

#!/bin/bash
# LF-kernel iteration loop: LF parity tests + two short bench runs (box-to-box clocks vary; compare within one call)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lf or likelihood or reweight or end_to_end or million" > gpurun_out/lf_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/lf_tests.log | tail -8
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, 'frac', round(d['roofline']['frac'],3))"
done
if [ -n "${EXTRA_FLAGS:-}" ]; then
  BELUGA_MCL_EXTRA_CXXFLAGS="$EXTRA_FLAGS" python -m beluga_amd.build --force > gpurun_out/build2.log 2>&1 || { tail -30 gpurun_out/build2.log; exit 1; }
  echo "--- with $EXTRA_FLAGS"
  for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, 'frac', round(d['roofline']['frac'],3))"
  done
fi

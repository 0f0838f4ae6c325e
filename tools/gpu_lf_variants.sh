#!/bin/bash
# A/B of compile-time variants of a kernel: each argument is a string of extra compiler flags (-D...) for beluga_amd.build.
# Per variant: rebuild, the patch == gather bit-for-bit test, a short bench (headline + windows + LF kernel time).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== variant: $flags"
  BELUGA_MCL_EXTRA_CXXFLAGS="$flags" python -m beluga_amd.build --force > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; continue; }
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "patch_kernel_equals or fma_variant_is_bit" 2>&1 | tail -2
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', round(d['value'],1), 'windows', [round(x) for x in d['repeat_windows']['cycles_per_s']], 'lf_ms(timed)', round(d['roofline']['avg_launch_ms'],4), 'lf_ms(stage pass)', round(d['stage_ms']['sensor_kernel'],4))"
done
# leave the default build behind
python -m beluga_amd.build --force > /dev/null 2>&1

#!/bin/bash
# A/B of build flags on one box: FLAGS_LIST is a ';'-separated list of flag sets
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "${FLAGS_LIST}"
for flags in "${SETS[@]}"; do
  BELUGA_MCL_EXTRA_CXXFLAGS="$flags" python -m beluga_amd.build --force > gpurun_out/build_ab.log 2>&1 || { tail -30 gpurun_out/build_ab.log; continue; }
  echo "--- flags: $flags"
  for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, 'frac', round(d['roofline']['frac'],3))"
  done
done

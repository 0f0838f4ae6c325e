#!/bin/bash
set -u
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for kb in "7,10" "6,10" "8,10" "7,9" "8,8" "8,9" "9,10" "6,9"; do
  echo -n "key bits $kb: "
  BELUGA_MCL_KEY_BITS=$kb timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cycles/s', round(d['value'],1), {k: round(v,3) for k,v in d['stage_ms'].items() if k in ('reweight','sensor_kernel')})"
done

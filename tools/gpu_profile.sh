#!/bin/bash
# A/B of the two likelihood-field kernel variants + rocprofv3 kernel trace and PMC passes.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for v in 0 1; do
  BELUGA_MCL_LF_VARIANT=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['stage_ms'], d['roofline']['frac'])" | tee -a gpurun_out/ab.log
done
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" ; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag.err
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -50

#!/bin/bash
# Round-2 diagnostics of the LF kernel's FMA variant at 1M particles: gather cost vs lane pattern, PMC groups.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_gather tools/calib_gather_rate.hip && /tmp/calib_gather | tee gpurun_out/calib_gather.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|SQC|TA|TCP|TD|TCC|GRBM)_[A-Z0-9_]+" | sort -u > gpurun_out/counters.txt
wc -l gpurun_out/counters.txt
export BELUGA_MCL_LF_FAST=1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FAST=1', d['value'], d['stage_ms'])"
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_INPUT_VALID_READYB" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/d$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/d$i.err || echo "pass $i ($pmc) failed/timeout"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/d$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_lf"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/d$i
done

#!/bin/bash
# Counters of the LF kernels of tools/exp_dispersed.py (one rocprofv3 --pmc pass per group) -> gpurun_out/r06/dispersed_pmc.txt
set -u
mkdir -p gpurun_out/prof gpurun_out/r06
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06/dispersed_pmc.txt
: > $O
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/d$i -o pmc -- python $GRAFT_REPO_ROOT/tools/exp_dispersed.py "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/d$i.err || { echo "pass $i failed" >> $O; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/d$i.err >> $O; }
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/d$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_lf" >> $O
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/d$i
done
cat $O

#!/bin/bash
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/s$i
  CYCLES=48 timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/s$i -o pmc -- python $GRAFT_REPO_ROOT/tools/exp_lf_converge.py > $GRAFT_REPO_ROOT/gpurun_out/prof/s$i.log 2>&1 || echo "pass $i failed"
  python $GRAFT_REPO_ROOT/tools/pmc_per_launch.py $GRAFT_REPO_ROOT/gpurun_out/prof/s$i/pmc_results.db k_reweight_lf_patch > $GRAFT_REPO_ROOT/gpurun_out/lf_pmc_series_$i.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/s$i
done

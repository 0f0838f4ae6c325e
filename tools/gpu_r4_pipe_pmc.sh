#!/bin/bash
# counters of k_reweight_lf_pipe against k_reweight_lf_patch on a fixed cloud (tools/exp_lf_fixed.py), one rocprofv3 pass per counter set
set -u
mkdir -p gpurun_out/r4 gpurun_out/prof
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4/pipe_pmc.txt
rm -f $O
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_WAVE32_INSTS SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  for pipe in 1 0; do
    BELUGA_MCL_LF_PIPE=$pipe timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/q$i$pipe -o pmc -- python $GRAFT_REPO_ROOT/tools/exp_lf_fixed.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/q$i$pipe.err || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/q$i$pipe.err; }
    echo "== lf_pipe $pipe, set $i" >> $O
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/q$i$pipe/pmc_results.db pmc | grep -E "k_reweight_lf_p" >> $O 2>&1
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/q$i$pipe
  done
done
cat $O

#!/bin/bash
# PMC passes of the default bench (each its own run, --kernel-trace only), LF kernel per dispatch.
set -u
mkdir -p gpurun_out/r03 gpurun_out/prof
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
cd /tmp
i=0
rm -f $O/pmc_series.txt
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 0 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/p$i.err || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/p$i.err; }
  python $GRAFT_REPO_ROOT/tools/pmc_series.py $GRAFT_REPO_ROOT/gpurun_out/prof/p$i/pmc_results.db k_reweight_lf_patch $( [ $i = 1 ] && echo --cols ) >> $O/pmc_series.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/p$i
done
cat $O/pmc_series.txt

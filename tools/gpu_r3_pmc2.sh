#!/bin/bash
set -u
mkdir -p gpurun_out/r03 gpurun_out/prof
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
cd /tmp
i=0
rm -f $O/pmc_series2.txt
for pmc in "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_CYCLES" "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/q$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/q$i.err || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/q$i.err; }
  python $GRAFT_REPO_ROOT/tools/pmc_series.py $GRAFT_REPO_ROOT/gpurun_out/prof/q$i/pmc_results.db k_reweight_lf_patch | tail -4 >> $O/pmc_series2.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_series.py $GRAFT_REPO_ROOT/gpurun_out/prof/q$i/pmc_results.db k_reweight_lf_patch | head -1 >> $O/pmc_series2.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/q$i
done
cat $O/pmc_series2.txt

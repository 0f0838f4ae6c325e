#!/bin/bash
# SQ counters of the likelihood-field kernel on the default bench workload (separate passes; --kernel-trace only).
# usage: [KERNEL_RE=regex] gpu_lf_pmc.sh [extra bench args]   (default kernel: the LF reweight kernel)
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/l$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 6 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/l$i.err || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/l$i.err; }
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/l$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "${KERNEL_RE:-reweight_lf}"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/l$i
done

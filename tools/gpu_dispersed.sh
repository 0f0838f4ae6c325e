#!/bin/bash
# Dispersed-set experiment: timings of the LF kernel families + cache / traffic counters per kernel.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
python tools/exp_dispersed.py "$@" 2>/dev/null | tee gpurun_out/dispersed_timing.txt
cd /tmp
i=0
rm -f $GRAFT_REPO_ROOT/gpurun_out/dispersed_pmc.txt
for pmc in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/d$i -o pmc -- python $GRAFT_REPO_ROOT/tools/exp_dispersed.py "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/d$i.err || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/d$i.err; }
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/d$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_lf" | tee -a $GRAFT_REPO_ROOT/gpurun_out/dispersed_pmc.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/d$i
done

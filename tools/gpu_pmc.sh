#!/bin/bash
# PMC passes (each its own run, --kernel-trace only, hard timeouts) for the default bench.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pmc$i.err || echo "pass $i ($pmc) failed/timeout"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/pmc$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight|resample_draw|sort_blocks|propagate"
done

#!/bin/bash
# End-of-round evidence run: tests, smoke, bench (with cpu baseline), kernel trace + PMC passes, other configs.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 600 python bench.py --steps 30 --warmup 3 2>gpurun_out/bench.err | tee gpurun_out/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/final_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/final_trace.err
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/final_pmc$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/final_pmc$i.err || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/final_trace/trace_results.db > gpurun_out/prof/final_kernel_stats.txt
for i in 1 2 3 4 5 6; do python tools/rocpd_summary.py gpurun_out/prof/final_pmc$i/pmc_results.db pmc | grep "^PMC"; done > gpurun_out/prof/final_pmc.txt
timeout 500 python tools/bench_configs.py 3 5 --steps 6 2>&1 | grep config > gpurun_out/configs.log
head -8 gpurun_out/prof/final_kernel_stats.txt; grep reweight gpurun_out/prof/final_pmc.txt; tail -4 gpurun_out/configs.log

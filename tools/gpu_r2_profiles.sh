#!/bin/bash
# HISTORICAL: ran on the round-2 tree (git history); some of the scripts and build flags it names are gone from the current one.
# Round-2 profile artefacts: kernel trace + timeline of the default bench, HBM traffic of the LF kernel, PMC of the LF kernel,
# gather / hand-off calibrations, traces of the other configurations.  Summaries land in gpurun_out/ (copied to profiles/).
set -u
mkdir -p gpurun_out/prof gpurun_out/r02
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
O=$GRAFT_REPO_ROOT/gpurun_out/r02
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_sync tools/calib_sync.hip 2>/dev/null && /tmp/calib_sync > $O/calib_sync.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_gather tools/calib_gather_rate.hip 2>/dev/null && /tmp/calib_gather > $O/calib_gather_cost.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > $O/trace_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db > $O/kernel_trace_bench_1M.txt
python tools/timeline.py gpurun_out/prof/trace/trace_results.db 2 > $O/timeline_bench_1M.txt
rm -rf gpurun_out/prof/trace
bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1
cp gpurun_out/lf_kernel_traffic.json gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt $O/ 2>/dev/null
cd /tmp
i=0
rm -f $O/pmc_bench_1M.txt
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 5 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/p$i.err || echo "pass $i failed"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/p$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_lf|resample_draw|propagate|sort_scatter" >> $O/pmc_bench_1M.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/p$i
done
# other configurations: kernel traces
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/c5 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 3 > $O/config5.log 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/c5/trace_results.db | head -12 > $O/kernel_trace_config5.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/c5
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/c3 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 3 --steps 6 > $O/config3.log 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/c3/trace_results.db | head -30 > $O/kernel_trace_config3.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/c3
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_1M.json
# beam kernel: instruction counters, then the walk statistics of the measurement build (rebuilds the library twice)
cd /tmp
rm -f $O/pmc_config5.txt
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/b$i -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/b$i.err || echo "beam pass $i failed"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/b$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_beam" >> $O/pmc_config5.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/b$i
done
cd $GRAFT_REPO_ROOT
python tools/exp_small.py 2>/dev/null > $O/small_filters.txt
python tools/exp_lf_series.py 2>/dev/null > $O/lf_series.txt
bash tools/gpu_beam_stats.sh > /dev/null 2>&1
cp gpurun_out/beam_stats.txt $O/beam_walk_stats.txt 2>/dev/null
ls -la $O

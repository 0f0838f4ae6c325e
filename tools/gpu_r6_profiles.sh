#!/bin/bash
# Round-6 profile artefacts (one GPU call): kernel trace + timeline of the default bench, PMC passes of the LF kernel
# (instruction classes for the VALU-issue roofline, LDS / wait counters, cache counters, FETCH_SIZE / WRITE_SIZE), the full
# bench line, traces of the other configurations.  Summaries land in gpurun_out/r06 (copied to profiles/r06_*).
set -u
mkdir -p gpurun_out/prof gpurun_out/r06
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
SHA=$(python -c "import bench; print(bench.lf_kernel_source_sha(bench.KERNEL_SOURCE))")
BENCH="python $GRAFT_REPO_ROOT/bench.py --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs --no-pmc"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- $BENCH --steps 20 --warmup 5 > $O/trace_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db > $O/kernel_trace_bench_1M.txt
python tools/timeline.py gpurun_out/prof/trace/trace_results.db 2 > $O/timeline_bench_1M.txt
rm -rf gpurun_out/prof/trace
cd /tmp
echo "# lf_kernels_sha256 $SHA particles 1000000   (rocprofv3 --kernel-trace --pmc <group>, one pass per group, of: bench.py --steps 20 --warmup 5; averages over the launches)" > $O/pmc_bench_1M.txt
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/p$i -o pmc -- $BENCH --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/p$i.err || echo "pass $i failed"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/p$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_lf|resample_draw|propagate|sort_|normalize|k_cdf" >> $O/pmc_bench_1M.txt
  if [ "$pmc" = "FETCH_SIZE" ]; then cp -r $GRAFT_REPO_ROOT/gpurun_out/prof/p$i $GRAFT_REPO_ROOT/gpurun_out/prof/t_fetch; fi
  if [ "$pmc" = "WRITE_SIZE" ]; then cp -r $GRAFT_REPO_ROOT/gpurun_out/prof/p$i $GRAFT_REPO_ROOT/gpurun_out/prof/t_write; fi
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/p$i
done
cd $GRAFT_REPO_ROOT
python tools/make_traffic_record.py gpurun_out/prof/t_fetch/pmc_results.db gpurun_out/prof/t_write/pmc_results.db $O/lf_kernel_traffic.json
rm -rf gpurun_out/prof/t_fetch gpurun_out/prof/t_write
# other configurations: kernel traces
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/c5 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 3 > $O/config5.log 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/c5/trace_results.db | head -12 > $O/kernel_trace_config5.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/c5
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/c3 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 3 --steps 6 > $O/config3.log 2> /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/c3/trace_results.db | head -30 > $O/kernel_trace_config3.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/c3
i=0
echo "# rocprofv3 --kernel-trace --pmc <group> of: tools/bench_configs.py 5 --steps 2 (BeamSensorModel, 1M x 1080)" > $O/pmc_config5.txt
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/b$i -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/b$i.err || echo "beam pass $i failed"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/b$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_beam" >> $O/pmc_config5.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/b$i
done
cd $GRAFT_REPO_ROOT
# fixed 10M: kernel trace + timeline
cd /tmp
N=10000000 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/tf -o trace -- python $GRAFT_REPO_ROOT/tools/exp_fixed.py 8 2>/dev/null | grep "^N " > $O/fixed_10M.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/tf/trace_results.db | head -14 > $O/kernel_trace_fixed_10M.txt
python tools/timeline.py gpurun_out/prof/tf/trace_results.db 1 >> $O/kernel_trace_fixed_10M.txt
rm -rf gpurun_out/prof/tf
python tools/exp_host_time.py 2>/dev/null | tail -n 1 > $O/host_time_per_cycle.txt
python tools/exp_lf_converge.py 2>/dev/null > $O/lf_converge.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err > $O/bench_1M.json
python tools/exp_small.py 2>/dev/null > $O/small_filters.txt
python tools/exp_dispersed.py 32 2>/dev/null > $O/dispersed.txt
python tools/exp_order_ahead.py 2>/dev/null > $O/order_ahead_counters.txt
python tools/exp_cluster.py 2>/dev/null > $O/cluster_estimate.txt
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
ls -la $O

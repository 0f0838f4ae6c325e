#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the LF kernel (separate PMC passes) -> gpurun_out/lf_kernel_traffic.json (copy to profiles/).
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/prof/t_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 5 --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/t_$c.err || echo "pass $c failed"
done
cd $GRAFT_REPO_ROOT
python tools/make_traffic_record.py gpurun_out/prof/t_FETCH_SIZE/pmc_results.db gpurun_out/prof/t_WRITE_SIZE/pmc_results.db gpurun_out/lf_kernel_traffic.json
python tools/rocpd_summary.py gpurun_out/prof/t_FETCH_SIZE/pmc_results.db pmc | grep "^PMC" > gpurun_out/pmc_fetch.txt
python tools/rocpd_summary.py gpurun_out/prof/t_WRITE_SIZE/pmc_results.db pmc | grep "^PMC" > gpurun_out/pmc_write.txt
rm -rf gpurun_out/prof/t_FETCH_SIZE gpurun_out/prof/t_WRITE_SIZE

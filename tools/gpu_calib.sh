#!/bin/bash
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib || exit 1
cd /tmp
/tmp/calib
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/prof/calib_$c -o pmc -- /tmp/calib > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/calib_$c.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/calib_$c/pmc_results.db pmc | grep "^PMC"
done
cd $GRAFT_REPO_ROOT
# PMC traffic of the default bench's dominant kernel (after the cube-table change)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/prof/bench_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/bench_$c.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/bench_$c/pmc_results.db pmc | grep "^PMC" | grep -E "reweight|resample_draw|sort_blocks"
done

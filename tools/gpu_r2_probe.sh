#!/bin/bash
# Round-2 first probe: hand-off costs, the cloud the LF kernel sees, and a dispatch timeline of the bench.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_sync tools/calib_sync.hip && /tmp/calib_sync | tee gpurun_out/calib_sync.txt
python tools/dump_cloud.py 2>&1 | tail -2
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/tl_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/tl.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/prof/tl/tl_results.db 2 | tee gpurun_out/timeline.txt
rm -rf gpurun_out/prof/tl

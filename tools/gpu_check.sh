#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof kernel trace. Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q -k "beam" 2>&1 | tail -8
timeout 600 python tools/bench_configs.py 5 --steps 4 2>&1 | grep config5

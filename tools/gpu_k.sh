#!/bin/bash
# usage: gpu_k.sh "<pytest -k expression>" [extra pytest args]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" ${2:-} 2>&1 | tail -40

#!/bin/bash
# Round 3, first GPU call: parity tests on the new tree, the bench line, the LF planner / ordering A/B.
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
O=gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -30 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs 2>$O/bench.err | tee $O/bench_a.json
tail -3 $O/bench.err
timeout 600 python tools/exp_lf_ab.py 25 2>&1 | tee $O/lf_ab.txt

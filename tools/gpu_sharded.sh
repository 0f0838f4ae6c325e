#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sharded 2>gpurun_out/bench_sharded.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sharded world1: cycles/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})"
tail -3 gpurun_out/bench_sharded.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sharded 2>&1 | tail -2 | cut -c1-300

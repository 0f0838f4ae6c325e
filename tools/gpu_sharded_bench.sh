#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
echo "--- world 1 through the library's RCCL transport"
timeout 300 python bench.py --sharded --steps 10 --warmup 3 --windows 1 --stage-steps 2 --no-cpu-baseline --no-other-configs 2>gpurun_out/bs1.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['parallelism'][:60], d['repeat_windows']['cycles_per_s'])" || tail -5 gpurun_out/bs1.err
echo "--- 2 ranks sharing the GPU, gloo dry run of the torch.distributed driver"
BELUGA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 4 --warmup 2 --windows 0 --stage-steps 0 --particles 200000 --no-cpu-baseline --no-other-configs 2>gpurun_out/bs2.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['n_gpus'], d['config']['particles_total'])" || tail -5 gpurun_out/bs2.err

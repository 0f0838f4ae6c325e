#!/bin/bash
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 300 python tools/bench_configs.py 5 --steps 2 --particles5 262144 2>&1 | grep config5
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/prof/beam1 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 2 --particles5 262144 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/beam1.err
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d $GRAFT_REPO_ROOT/gpurun_out/prof/beam2 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 2 --particles5 262144 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/beam2.err
cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/rocpd_summary.py gpurun_out/prof/beam$i/pmc_results.db pmc | grep "^PMC" | grep beam; done
python tools/rocpd_summary.py gpurun_out/prof/beam1/pmc_results.db | head -4

#!/bin/bash
# VALU instructions and time of the LF patch kernel's block prologue alone (timing build -DMCL_ABLATE=32: every workgroup returns behind its plan),
# beside the whole kernel's: rocprofv3 --kernel-trace --pmc of the bench workload, a workgroup per block (lf_queue = 0).
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 BELUGA_MCL_LF_QUEUE=0
bash tools/build_variant.sh pro32 "-DMCL_ABLATE=32" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
BENCH="python $GRAFT_REPO_ROOT/bench.py --pmc-child"
cd /tmp
for lib in "" build/variants/pro32/libbeluga_mcl.so; do
  [ -n "$lib" ] && export BELUGA_MCL_LIB=$GRAFT_REPO_ROOT/$lib
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/pp
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/prof/pp -o pmc -- $BENCH > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pp.err || echo "pass failed: $(tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof/pp.err)"
  echo "== ${lib:-product}"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/pp/pmc_results.db pmc 2>/dev/null | grep -E "reweight_lf" 
done

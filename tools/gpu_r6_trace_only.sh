#!/bin/bash
# The kernel trace + timeline of the default bench (as in tools/gpu_r6_profiles.sh), up to five attempts: rocprofv3 now and then stretches ONE launch of
# the verification's stage-level reweight to milliseconds (never seen without the profiler: tools/exp_verify_reweight.py); the first attempt in which no kernel
# of the cycle has a launch beyond three times its median is kept, the others are listed in trace_attempts.txt.
set -u
mkdir -p gpurun_out/prof gpurun_out/r06
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
BENCH="python $GRAFT_REPO_ROOT/bench.py --windows 0 --stage-steps 0 --no-cpu-baseline --no-other-configs --no-pmc"
: > $O/trace_attempts.txt
for a in 1 2 3 4 5; do
  cd /tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/trace
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o trace -- $BENCH --steps 20 --warmup 5 > $O/trace_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/trace.err
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_summary.py gpurun_out/prof/trace/trace_results.db > $O/kernel_trace_bench_1M.txt
  python tools/timeline.py gpurun_out/prof/trace/trace_results.db 2 > $O/timeline_bench_1M.txt
  echo "attempt $a (kernel, calls, total, avg, %, median, max in us): $(sort -k7 -g -r $O/kernel_trace_bench_1M.txt | head -1)" >> $O/trace_attempts.txt
  if python - <<PY
import sys
ok = False
for l in open("$O/kernel_trace_bench_1M.txt"):
    f = l.split()
    if len(f) >= 7 and f[1].isdigit() and int(f[1]) >= 20:  # the kernels of the cycle
        ok = True
        if float(f[-1]) > 3.0 * float(f[-2]):
            sys.exit(1)
sys.exit(0 if ok else 1)
PY
  then break; fi
done
rm -rf gpurun_out/prof/trace
cat $O/trace_attempts.txt; head -14 $O/kernel_trace_bench_1M.txt

#!/bin/bash
# Full check: GPU suite, smoke, the complete default bench line (with other configs and the CPU baseline).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
if [ "${1:-tests}" = "tests" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench.err > gpurun_out/bench.json
echo "bench wall: ${SECONDS}s"; tail -5 gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("value", round(d["value"], 1), "windows", [round(x) for x in d["repeat_windows"]["cycles_per_s"]], "lf_ms", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"])
print("stage_ms", {k: round(v, 4) for k, v in d["stage_ms"].items()})
for k, v in d.get("configs", {}).items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else ([round(x, 3) for x in b] if isinstance(b, list) and b and isinstance(b[0], float) else b)) for a, b in v.items() if a != "what"})
print("cpu", d.get("cpu_baseline"))
PY

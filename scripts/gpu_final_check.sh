#!/bin/bash
# bench.py (default line with every secondary record) + the whole -m gpu suite: the round-end check
mkdir -p gpurun_out/final_check
s=$(date +%s)
timeout 1500 python bench.py > gpurun_out/final_check/bench.json 2> gpurun_out/final_check/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final_check/bench.json").read().strip().splitlines()[-1])
print("c3", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["secondary"].items():
    print(k, v.get("error") or (v.get("ms_per_step"), v.get("value"), v.get("unit")))
PY
s=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/final_check/pytest_gpu.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - s )) s"; tail -3 gpurun_out/final_check/pytest_gpu.txt

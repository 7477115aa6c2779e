#!/bin/bash
# bench.py (default line with every secondary record) + the whole -m gpu suite + smoke: the round-end check
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final_check
export TMPDIR=/tmp
s=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/final_check/bench.json 2> gpurun_out/final_check/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final_check/bench.json").read().strip().splitlines()[-1])
print("c3", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["secondary"].items():
    print(k, v.get("error") or (v.get("ms_per_step"), v.get("value"), v.get("unit")))
print(d["summary"])
PY
s=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/final_check/pytest_gpu.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - s )) s"; tail -3 gpurun_out/final_check/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/final_check/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/final_check/smoke.txt

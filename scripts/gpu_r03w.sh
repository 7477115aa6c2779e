#!/bin/bash
mkdir -p gpurun_out/r03w
timeout 300 python scripts/probes/spmm_narrow_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03w/probe.log; cat gpurun_out/r03w/probe.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mofa.py -x -q > gpurun_out/r03w/tests.log 2>&1; tail -3 gpurun_out/r03w/tests.log
timeout 600 python bench.py --workload c4 > gpurun_out/r03w/c4.json 2> gpurun_out/r03w/c4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03w/c4.json").read().strip().splitlines()[-1])
print("c4", d["value"], d["unit"], d.get("parity"))
PY

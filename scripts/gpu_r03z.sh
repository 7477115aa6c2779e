#!/bin/bash
mkdir -p gpurun_out/r03z
timeout 900 python -m pytest tests/test_gpu_wnn.py -x -q > gpurun_out/r03z/tests.log 2>&1; tail -5 gpurun_out/r03z/tests.log
timeout 600 python scripts/wnn_probe.py 100000 2>&1 | grep -v amdgpu.ids > gpurun_out/r03z/wnn.log; cat gpurun_out/r03z/wnn.log

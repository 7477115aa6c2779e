#!/bin/bash
mkdir -p gpurun_out/r03ab
timeout 600 python bench.py --workload wnn > gpurun_out/r03ab/wnn.json 2> gpurun_out/r03ab/wnn.err; tail -c 1500 gpurun_out/r03ab/wnn.json
timeout 900 python -m pytest tests/test_gpu_wnn.py tests/test_gpu_lsi.py -x -q -k "wnn or widened or neighbours or filter" 2>&1 | tail -3
f=$(find gpurun_out/r03y/prof -name "*kernel_trace.csv" 2>/dev/null | head -1)

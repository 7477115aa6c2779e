#!/bin/bash
mkdir -p gpurun_out/r03af
timeout 900 python -m pytest tests/test_gpu_wnn.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload wnn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wnn ms', round(d['ms'],1), d['parity'])"

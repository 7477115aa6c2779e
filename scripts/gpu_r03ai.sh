#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "spmm" 2>&1 | tail -2
timeout 300 python scripts/probes/spmm_narrow_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "NB=1|accounting"
for i in 1 2; do for off in 0 1; do MUON_AMD_BENCH_TUNE="spmm_narrow_off=$off" timeout 300 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 narrow_off=$off', round(d['value'],4))"; done; done

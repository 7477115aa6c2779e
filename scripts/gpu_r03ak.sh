#!/bin/bash
timeout 300 python scripts/probes/spmm_narrow_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "step stream|NB=1"
for i in 1 2; do for t in 1 0; do MUON_AMD_MOFA_TILES=$t timeout 300 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 tiles=$t', round(d['value'],4))"; done; done

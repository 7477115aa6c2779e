#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_mofa.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload mofa_ng 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mofa_ng ms/iter', round(d['ms_per_iteration'],2), d['parity'])"

#!/bin/bash
mkdir -p gpurun_out/r03ae
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lsi.py -x -q -k "chol or lsi or rank_deficient or baseline_shape or fixture" > gpurun_out/r03ae/tests.log 2>&1; tail -3 gpurun_out/r03ae/tests.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03ae/prof -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03ae/c2_prof.json 2>/dev/null)
f=$(find gpurun_out/r03ae/prof -name "*kernel_trace.csv" | head -1); python scripts/kstats.py "$f" c2 | grep -E "chol|total kernel" | cut -c1-150
find gpurun_out/r03ae -name "*.csv" -size +2M -delete
for i in 1 2; do timeout 300 python bench.py --workload c2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 ms/step', round(d['ms_per_step'],3))"; done
timeout 300 python bench.py --workload c3shard --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3shard ms/step', round(d['ms_per_step'],3))"

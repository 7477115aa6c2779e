#!/bin/bash
# r05 first GPU visit: the new transposition + stream emission - correctness first, then timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05a}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tpack4.py -x -q > "$OUT/pytest_tpack4.txt" 2>&1
echo "tpack4 rc=$?"; tail -15 "$OUT/pytest_tpack4.txt"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "transpose or stream" > "$OUT/pytest_kernels.txt" 2>&1
echo "kernels rc=$?"; tail -5 "$OUT/pytest_kernels.txt"
timeout 600 python scripts/probes/tpack4_probe.py 125000 > "$OUT/probe_125k.txt" 2>&1
echo "probe125 rc=$?"; grep -v amdgpu.ids "$OUT/probe_125k.txt"
timeout 900 python scripts/probes/tpack4_probe.py 1000000 --sweep > "$OUT/probe_1m.txt" 2>&1
echo "probe1m rc=$?"; cat "$OUT/probe_1m.txt"
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('c3 ms/step', d['ms_per_step'], 'spmm avg', d['roofline']['avg_launch_ms'], d['config']['lsi'])
"; tail -3 "$OUT/bench.err"

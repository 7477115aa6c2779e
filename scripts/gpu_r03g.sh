#!/bin/bash
# r03g: drain-free revisits in the row-stream SpMM: kernel tests (bit-identity, overflowing rows), probe, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03g}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lsi.py tests/test_gpu_mofa.py -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.txt"
timeout 600 python scripts/spmm_probe.py --modes 64 > "$OUT/spmm_125k.txt" 2>&1; echo "probe rc=$?"; cat "$OUT/spmm_125k.txt"
timeout 900 python bench.py --no-secondary --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['allocator'], d['config']['lsi'])"

#!/bin/bash
# r03c: full gpu suite (new: general MOFA engine, rowstats kernel, self-launching bench), c4 line, B = 16 K sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03c}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_gpu.txt"
timeout 600 python bench.py --workload c4 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "c4 rc=$?"; cat "$OUT/bench_c4.json"; tail -3 "$OUT/bench_c4.err"
timeout 600 python scripts/spmm_probe.py --cells 100000 --peaks 100000 --B 16 --modes "" --ks 2,3,4,5,6,8 > "$OUT/spmm_b16.txt" 2>&1; echo "probe rc=$?"; cat "$OUT/spmm_b16.txt"

#!/bin/bash
# r03a: VALU issue-rate probe, full gpu suite, default bench line (c3 + secondary c2 / c4)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03a}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 120 scripts/probes/valu_rate > "$OUT/valu_rate.txt" 2>&1; echo "valu rc=$?"; cat "$OUT/valu_rate.txt"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_gpu.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"

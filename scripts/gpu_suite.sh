#!/bin/bash
# the whole -m gpu suite + smoke on one box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-suite}
mkdir -p "$OUT"
export TMPDIR=/tmp
s=$(date +%s)
timeout 2700 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - s )) s"; tail -15 "$OUT/pytest_gpu.txt"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?"; tail -4 "$OUT/smoke.txt"

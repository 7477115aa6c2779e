#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06i
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lsi.py -x -q -m gpu -k "two_ranks_on_one_gpu or bench_starts or rccl_branch" -s > "$OUT/dist.txt" 2>&1; echo "dist rc=$?"; grep -a "rsag\|rsqr\|warm start\|ranks 2\|passed\|failed\|Error" "$OUT/dist.txt" | head -20

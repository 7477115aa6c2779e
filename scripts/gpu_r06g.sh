#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06g
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o r8 -- python "$OLDPWD/scripts/bench_rank8.py" c5_rank8 > "$OLDPWD/$OUT/prof.out" 2> "$OLDPWD/$OUT/prof.err")
echo "prof rc=$?"; cat "$OUT/prof.out" | cut -c1-300
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
db=$(find "$OUT/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python scripts/bench_rank8.py c5_rank8 (one rank of eight emulated on one GPU: 12 500 cells, 3 warm-up + 100 timed iterations)" > "$OUT/c5_rank8_kernel_stats.md"
  rm -f "$db"
  head -50 "$OUT/c5_rank8_kernel_stats.md"
fi

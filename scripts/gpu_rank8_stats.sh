#!/bin/bash
# rocprofv3 kernel statistics of the two emulated rank-of-eight records (scripts/bench_rank8.py)
# -> gpurun_out/<tag>/<tag>_c3_rank8_kernel_stats.md, <tag>_c5_rank8_kernel_stats.md
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r06}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in c3_rank8 c5_rank8; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$v" -o r8 -- python "$OLDPWD/scripts/bench_rank8.py" $v > "$OUT/prof_$v.out" 2> "$OUT/prof_$v.err")
  find "$OUT/prof_$v" -name "*kernel_trace.csv" -size +20M -delete
  db=$(find "$OUT/prof_$v" -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python scripts/bench_rank8.py $v (one rank of eight emulated on one GPU; warm-up + timed steps; k_synth = input generation)" > "$OUT/${R}_${v}_kernel_stats.md"
  rm -rf "$OUT/prof_$v"
done
ls "$OUT"

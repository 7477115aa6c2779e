#!/bin/bash
# rocprofv3 kernel statistics of the WNN probe (100 000 cells) -> gpurun_out/<tag>/<tag>_wnn_kernel_stats.md
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r04}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_wnn" -o wnn -- python "$OLDPWD/scripts/probes/wnn_host_profile.py" 100000 > "$OUT/prof_wnn.txt" 2> "$OUT/prof_wnn.err")
db=$(find "$OUT/prof_wnn" -name "*.db" | head -1)
[ -n "$db" ] && python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python scripts/probes/wnn_host_profile.py 100000 (1 GPU; pp.knn x 2 + pp.neighbors, run twice: warm-up + profiled)" > "$OUT/${R}_wnn_kernel_stats.md"
rm -rf "$OUT/prof_wnn"
head -24 "$OUT/${R}_wnn_kernel_stats.md" | cut -c1-170
grep -n "total\|knn\|neighbors" "$OUT/prof_wnn.txt" | head -12

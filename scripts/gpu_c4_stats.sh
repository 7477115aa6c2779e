#!/bin/bash
# rocprofv3 kernel statistics of the MOFA bench (c4, f32 and f64) -> gpurun_out/<tag>/<tag>_c4*_kernel_stats.md
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r04}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in c4 c4_f64; do
  extra=""; [ $v = c4_f64 ] && extra="--f64"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$v" -o bench -- python "$OLDPWD/scripts/bench_mofa.py" --iters 10 --warmup 2 --no-cpu-baseline $extra > "$OUT/prof_bench_$v.json" 2> "$OUT/prof_$v.err")
  db=$(find "$OUT/prof_$v" -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python scripts/bench_mofa.py --iters 10 --warmup 2 --no-cpu-baseline $extra (1 GPU; 12 iterations incl. warm-up + set-up; k_synth = input generation)" > "$OUT/${R}_${v}_kernel_stats.md"
  rm -rf "$OUT/prof_$v"
done
ls "$OUT"

#!/bin/bash
# The evidence behind bench.py's roofline block, one GPU-box visit: kernel stats of the default
# bench (c3) and of c3shard, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, with the 4 GiB
# calibration copy) of the SpMM at 1e6 and 125k rows, SQ counters of the SpMM at 125k rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r02}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for wl in c3 c3shard; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$wl" -o bench -- python "$OLDPWD/bench.py" --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/prof_bench_$wl.json" 2> "$OUT/prof_$wl.err")
  db=$(find "$OUT/prof_$wl" -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (1 GPU; 3 steps incl. warm-up; k_synth = input generation outside the timed region)" > "$OUT/${R}_${wl}_kernel_stats.md"
  rm -rf "$OUT/prof_$wl"
done
[ "${SKIP_PMC:-0}" = "1" ] && { ls "$OUT"; exit 0; }
for cells in 1000000 125000; do
  CELLS=$cells bash scripts/pmc_traffic.sh $R/traffic_$cells > "$OUT/traffic_$cells.log" 2>&1
  python scripts/pmc_traffic_summary.py "$OUT/traffic_$cells" $cells 200000 > "$OUT/traffic_$cells.json" 2> "$OUT/traffic_${cells}_summary.err"
  find "$OUT/traffic_$cells" -name "*.csv" -size +2M -delete
done
PROBE_ARGS="--no-packed" bash scripts/pmc_spmm.sh $R/pmc > "$OUT/pmc_summary.txt" 2>&1
find "$OUT/pmc" -name "*.csv" -size +2M -delete
du -sh "$OUT"; ls "$OUT"

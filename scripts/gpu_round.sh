#!/bin/bash
# One GPU-box visit: full gpu suite, bench (c3 + c3shard + c2), rocprof kernel stats.  Everything is
# wrapped in `timeout` so that a hung kernel cannot hold the box; outputs land in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-run}
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > "$OUT/device.txt" 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== full gpu suite" | tee "$OUT/steps.txt"
timeout 1200 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
echo "rc=$?" | tee -a "$OUT/steps.txt"; tail -5 "$OUT/pytest_gpu.txt"
fi
echo "== bench c3" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
echo "== bench c3shard (weak-scaling shard)" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c3shard.json" 2> "$OUT/bench_c3shard.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench_c3shard.json"
echo "== bench c2" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench_c2.json"
if [ "${SKIP_PROF:-0}" != "1" ]; then
echo "== rocprof kernel stats" | tee -a "$OUT/steps.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
echo "rc=$?" | tee -a "$OUT/steps.txt"
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
db=$(find "$OUT/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (c3, 1 GPU)" > "$OUT/kernel_stats.md"
  rm -f "$db"
  head -30 "$OUT/kernel_stats.md"
fi
fi
du -sh "$OUT"

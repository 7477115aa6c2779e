#!/bin/bash
# One GPU-box visit: kernel tests, full gpu suite, bench, rocprof kernel stats.  Everything is
# wrapped in `timeout` so that a hung kernel cannot hold the box; outputs land in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-run}
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > "$OUT/device.txt" 2>&1
echo "== packed spmm tests" | tee "$OUT/steps.txt"
timeout 420 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pack" > "$OUT/pytest_pack.txt" 2>&1
echo "rc=$?" | tee -a "$OUT/steps.txt"; tail -5 "$OUT/pytest_pack.txt"
echo "== full gpu suite" | tee -a "$OUT/steps.txt"
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
echo "rc=$?" | tee -a "$OUT/steps.txt"; tail -5 "$OUT/pytest_gpu.txt"
echo "== bench (packed)" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
echo "== bench c3shard (weak-scaling shard, packed)" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c3shard.json" 2> "$OUT/bench_c3shard.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench_c3shard.json"
echo "== bench (csr kernel, ablation)" | tee -a "$OUT/steps.txt"
timeout 600 python bench.py --workload c3shard --steps 2 --warmup 1 --no-pack --no-cpu-baseline > "$OUT/bench_nopack.json" 2> "$OUT/bench_nopack.err"
echo "rc=$?" | tee -a "$OUT/steps.txt"; cat "$OUT/bench_nopack.json"
echo "== rocprof kernel stats" | tee -a "$OUT/steps.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
echo "rc=$?" | tee -a "$OUT/steps.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -3
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep the merged output small: traces can be large
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
du -sh "$OUT"
db=$(find "$OUT/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (c3, 1 GPU)" > "$OUT/kernel_stats.md"
  rm -f "$db"
fi
echo "== lsi convergence" | tee -a "$OUT/steps.txt"
{ echo "# 125000 x 200000 (one of eight shards as a matrix of its own)"; timeout 300 python scripts/lsi_convergence.py 2>&1 | grep -v amdgpu.ids
  echo "# 1000000 x 200000 (configs[2])"; timeout 400 python scripts/lsi_convergence.py --cells 1000000 2>&1 | grep -v amdgpu.ids; } > "$OUT/lsi_convergence.txt"
tail -2 "$OUT/lsi_convergence.txt"
echo "== mofa" | tee -a "$OUT/steps.txt"
timeout 300 python scripts/bench_mofa.py --iters 100 --warmup 4 2>/dev/null | tail -1 > "$OUT/mofa_f32.json"
timeout 300 python scripts/bench_mofa.py --f64 --iters 100 --warmup 4 2>/dev/null | tail -1 > "$OUT/mofa_f64.json"
cat "$OUT/mofa_f32.json" "$OUT/mofa_f64.json" | cut -c1-300
du -sh "$OUT"

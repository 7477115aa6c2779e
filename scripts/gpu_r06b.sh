#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lsi.py -x -q -m gpu -k "slice_products or warm_start or two_ranks" > "$OUT/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -15 "$OUT/pytest.txt"
timeout 300 python scripts/probes/lsi_rank_of_8_probe.py > "$OUT/rank8.txt" 2>&1; echo "rank8 rc=$?"; cat "$OUT/rank8.txt"
MUON_AMD_LSI_WARM_SLICE=operands timeout 300 python bench.py --workload c3shard --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/c3shard_old.json" 2> "$OUT/c3shard_old.err"; echo "c3shard(old) rc=$?"
python -c "import json;d=json.load(open('$OUT/c3shard_old.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['lsi'])"
timeout 300 python bench.py --workload c3shard --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/c3shard.json" 2> "$OUT/c3shard.err"; echo "c3shard rc=$?"
python -c "import json;d=json.load(open('$OUT/c3shard.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['lsi'])"
timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$OUT/c3.json" 2> "$OUT/c3.err"; echo "c3 rc=$?"
python -c "import json;d=json.load(open('$OUT/c3.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['lsi'])"

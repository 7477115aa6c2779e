#!/bin/bash
# r06 first visit: the touched tests, the rank-of-8 probe as it stands, c3shard + c3 bench lines (baseline of this round's boxes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lsi.py tests/test_gpu_tpack4.py tests/test_gpu_tfidf.py tests/test_gpu_mofa.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -3 "$OUT/pytest.txt"
timeout 300 python scripts/probes/lsi_rank_of_8_probe.py > "$OUT/rank8.txt" 2>&1; echo "rank8 rc=$?"; cat "$OUT/rank8.txt"
timeout 300 python bench.py --workload c3shard --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/c3shard.json" 2> "$OUT/c3shard.err"; echo "c3shard rc=$?"
python -c "import json;d=json.load(open('$OUT/c3shard.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['lsi'])"
timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary > "$OUT/c3.json" 2> "$OUT/c3.err"; echo "c3 rc=$?"
python -c "import json;d=json.load(open('$OUT/c3.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['lsi'], d.get('summary'))"

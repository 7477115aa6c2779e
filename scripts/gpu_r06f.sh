#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06f
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mofa.py -x -q -m gpu -k "segments or spikeslab or 48_factors or two_ranks or captured" > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.txt"
timeout 300 python scripts/bench_rank8.py c5_rank8 > "$OUT/rank8.jsonl" 2> "$OUT/rank8.err"; echo "rank8 rc=$?"; tail -3 "$OUT/rank8.err"
python - <<'PY'
import json
for l in open("gpurun_out/r06f/rank8.jsonl"):
    d=json.loads(l)
    for k,v in d.items(): print(k, v["value"], {a:b for a,b in v.items() if a in ("graph","segments","elbo_monotone")})
PY
timeout 900 python -m pytest tests/test_gpu_lsi.py -x -q -m gpu -k "two_ranks_on_one_gpu" > "$OUT/dist.txt" 2>&1; echo "dist rc=$?"; tail -4 "$OUT/dist.txt"

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python scripts/bench_rank8.py c3_rank8 > "$OUT/rank8.jsonl" 2> "$OUT/rank8.err"; echo "rank8 rc=$?"; tail -3 "$OUT/rank8.err"
python - <<'PY'
import json
for l in open("gpurun_out/r06d/rank8.jsonl"):
    d=json.loads(l)
    for k,v in d.items(): print(k, v["value"], {a:b for a,b in v.items() if a in ("tfidf_ms","steps_ms","allocator")}, v["config"])
PY

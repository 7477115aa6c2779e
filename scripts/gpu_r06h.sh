#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06h
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ell.py tests/test_gpu_mofa.py tests/test_gpu_tpack4.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.txt"
timeout 300 python scripts/bench_rank8.py c5_rank8 > "$OUT/rank8.jsonl" 2> "$OUT/rank8.err"; echo "rank8 rc=$?"; tail -3 "$OUT/rank8.err"
python - <<'PY'
import json
for l in open("gpurun_out/r06h/rank8.jsonl"):
    d=json.loads(l)
    for k,v in d.items(): print(k, v["value"], {a:b for a,b in v.items() if a in ("graph","segments","elbo_monotone")})
PY
timeout 600 python bench.py --workload c4 --no-cpu-baseline > "$OUT/c4.json" 2> "$OUT/c4.err"; echo "c4 rc=$?"
python -c "import json;d=json.load(open('$OUT/c4.json'));print('c4', d['value'], d['ms_per_step'])"

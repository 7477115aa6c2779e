#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06e
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python scripts/bench_spectra.py > "$OUT/spectra.jsonl" 2> "$OUT/spectra.err"; echo "spectra rc=$?"; tail -5 "$OUT/spectra.err"
python - <<'PY'
import json
for l in open("gpurun_out/r06e/spectra.jsonl"):
    d=json.loads(l)
    for k,v in d.items():
        print(k, {a:b for a,b in v.items() if a not in ("metric","config","parity","roofline")})
        print("   config", v["config"]); print("   parity", v["parity"]); print("   roofline", v.get("roofline"))
PY
timeout 900 python -m pytest tests/test_gpu_lsi.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.txt"

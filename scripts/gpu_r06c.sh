#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python scripts/bench_rank8.py > "$OUT/rank8.jsonl" 2> "$OUT/rank8.err"; echo "rank8 rc=$?"; tail -3 "$OUT/rank8.err"
python - <<'PY'
import json
for l in open("gpurun_out/r06c/rank8.jsonl"):
    d=json.loads(l)
    for k,v in d.items(): print(k, v["value"], {a:b for a,b in v.items() if a in ("tfidf_ms","graph","elbo_monotone")}, v["config"])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o r8 -- python "$OLDPWD/scripts/bench_rank8.py" c3_rank8 > "$OLDPWD/$OUT/prof.out" 2> "$OLDPWD/$OUT/prof.err")
echo "prof rc=$?"
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
db=$(find "$OUT/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python scripts/bench_rank8.py c3_rank8 (one rank of eight emulated on one GPU: 2 warm-up + 5 timed + 1 split step = 8 steps; k_synth = input generation)" > "$OUT/rank8_kernel_stats.md"
  rm -f "$db"
  head -45 "$OUT/rank8_kernel_stats.md"
fi

#!/bin/bash
# the default bench line with every secondary record, timed like the driver runs it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-bench_full}
mkdir -p "$OUT"
export TMPDIR=/tmp
s=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$? wall=$(( $(date +%s) - s )) s"; tail -3 "$OUT/bench.err"
python - "$OUT" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench.json").read().strip().splitlines()[-1])
print("line length", len(json.dumps(d, separators=(",",":"))))
print("c3", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("config scalars", {k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
for k,v in d["secondary"].items():
    print(k, v.get("error") or (v.get("ms_per_step"), v.get("value"), v.get("unit")))
print("summary", d["summary"])
print("tail", json.dumps(d, separators=(",",":"))[-1800:])
PY

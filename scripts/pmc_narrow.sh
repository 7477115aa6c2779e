#!/bin/bash
# PMC passes over the narrow-block SpMM probe (counters only: no tracing domains alongside --pmc).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-pmc_narrow}
mkdir -p "$OUT"
export TMPDIR=/tmp NARROW_REPS=2
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 170 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- python "$OLDPWD/scripts/probes/spmm_narrow_probe.py" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmm" not in k:
            continue
        agg[(k[:70], r["Counter_Name"], r.get("Dispatch_Id"))] += float(r["Counter_Value"])
per = collections.defaultdict(list)
for (k, c, d), v in agg.items():
    per[(k, c)].append(v)
for (k, c), vs in sorted(per.items()):
    print(f"{k:72s} {c:24s} n={len(vs):3d} mean={sum(vs)/len(vs):.4e}")
PY
find "$OUT" -name "*.csv" -size +2M -delete

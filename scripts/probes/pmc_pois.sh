#!/bin/bash
# PMC passes over scripts/probes/pois_mfma_probe.py (the dense sweep of a poisson view): per kernel averages
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/pmc_pois; mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d "$OUT/p$i" -o pmc --output-format csv -- python "$ROOT/${PROBE:-scripts/probes/pois_mfma_probe.py}" "$@" > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$ROOT"
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc_pois/p*/**/*counter_collection.csv', recursive=True)):
    per = defaultdict(lambda: defaultdict(float)); names = {}
    for r in csv.DictReader(open(f)):
        if 'pois' not in r['Kernel_Name']: continue
        m = re.search(r'k_pois_\w+<[^>]*>', r['Kernel_Name'])
        names[r['Dispatch_Id']] = m.group(0) if m else r['Kernel_Name'][:40]
        per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
    for d, cs in per.items():
        for c, v in cs.items(): agg[names[d]][c].append(v)
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf $OUT/p*/

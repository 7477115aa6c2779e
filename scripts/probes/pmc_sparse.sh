#!/bin/bash
# memory-side PMC passes over the poisson kernels of scripts/mofa_general_probe.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/pmc_sparse; mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA|TCP|TCC|TD)_[A-Z0-9_a-z]+" | sort -u > "$OUT/counters.txt"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d "$OUT/p$i" -o pmc --output-format csv -- python "$ROOT/scripts/mofa_general_probe.py" 20000 f32 iters=3 > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$? ($grp)"
done
cd "$ROOT"
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc_sparse/p*/**/*counter_collection.csv', recursive=True)):
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
        print(f"   {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf $OUT/p*/

#!/bin/bash
# r03f: SpMM fabric traffic with the K = 6 layouts (PMC, separate passes) + where the excess comes from
# (FETCH_SIZE of the ablations: 1 = no gathers, 9 = no gathers and no window requests: the Q slabs alone)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-r03f}
mkdir -p "$OUT"
export TMPDIR=/tmp
python muon_amd/csrc/build.py > /dev/null 2>&1
CELLS=1000000 bash scripts/pmc_traffic.sh r03f/traffic_1000000 > "$OUT/t1m.log" 2>&1
python scripts/pmc_traffic_summary.py "$OUT/traffic_1000000" 1000000 200000 > "$OUT/traffic_1000000.json" 2> "$OUT/t1m.err"; tail -25 "$OUT/traffic_1000000.json"; tail -3 "$OUT/t1m.err"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/abl" -o pmc --output-format csv -- python "$ROOT/scripts/spmm_probe.py" --reps 1 --modes "1,9" --no-packed --calibrate --cells 122880 > "$OUT/abl.log" 2>&1
echo "abl rc=$?"; tail -12 "$OUT/abl.log"
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/abl/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_spmm_win" in r["Kernel_Name"]:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:70], r["Grid_Size"])
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
for k, v in per.items():
    print(k, f"{2 * v * 1024 / 1e9:.2f} GB (FETCH_SIZE x2)")
PY
find "$OUT" -name "*.csv" -size +5M -delete; du -sh "$OUT"

#!/bin/bash
# HBM traffic of the sliced-ELL SpMM (csrc/spmm_ell.hip) at MOFA c4's sparse view: FETCH_SIZE in its own rocprofv3 --pmc
# pass (counters only), summed per kernel name; x2 on gfx950 (128-byte requests tallied as 64 B, MI355X_MICROARCH.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-r04}/pmc_ell
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o pmc --output-format csv -- python "$OLDPWD/scripts/probes/ell_probe.py" > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmm" in k:
            agg[k[:70]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v.sort()
    med = v[len(v) // 2]
    print(f"{k}: {len(v)} dispatches, FETCH_SIZE median {med:.0f} KiB -> x2 = {2 * med * 1024 / 1e9:.3f} GB per launch")
PY
find "$OUT" -name "*.csv" -size +2M -delete

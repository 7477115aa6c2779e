#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fourth-generation fill with its real stores, without stores, with every run stored
# into one 8 KiB region (separate counter passes; FETCH_SIZE x2 on gfx950, KiB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-pmc_t4}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d "$OUT/$c" -o pmc --output-format csv -- python "$OLDPWD/scripts/probes/tpack4_abl_pmc.py" > "$OUT/$c.log" 2>&1
  echo "$c rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.OrderedDict()
    for f in glob.glob(out + f"/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_t4_fill" in r["Kernel_Name"] and r["Counter_Name"] == c:
                agg[int(r["Dispatch_Id"])] = agg.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for (d, v), name in zip(sorted(agg.items()), ("real stores", "no stores", "stores into one 8 KiB region")):
        gb = v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
        print(f"k_t4_fill {name:30s} {c:11s} raw {v:.4e} KiB -> {gb:7.1f} GB")
PY
find "$OUT" -name "*.csv" -size +2M -delete

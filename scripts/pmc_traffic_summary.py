#!/usr/bin/env python
"""Turn the two PMC passes of scripts/pmc_traffic.sh into the per-shape entry of
profiles/r03_spmm_traffic.json (--out NAME): HBM bytes per SpMM launch = FETCH_SIZE x2 (gfx950 tallies 128-byte
requests as 64 B; checked against the 4 GiB calibration copy of the same pass) + WRITE_SIZE, KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, cells, peaks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 200000


def load(counter):
    f = glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    per = defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        i = int(r["Dispatch_Id"])
        per[i] += float(r["Counter_Value"])
        meta[i] = (r["Kernel_Name"], int(r["Grid_Size"]))
    return per, meta


res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per, meta = load(counter)
    ids = sorted(per)
    spmm = [i for i in ids if "k_spmm_win" in meta[i][0]]
    xq_key = meta[spmm[0]]  # (kernel instance, grid): the first product of scripts/spmm_probe.py is X * Q
    xq = [per[i] for i in spmm if meta[i] == xq_key]
    xt = [per[i] for i in spmm if meta[i] != xq_key]
    copies = [per[i] for i in ids if "copyBuffer" in meta[i][0] or "elementwise" in meta[i][0].lower() and per[i] > 1.5e6]
    res[counter] = {"xq": sum(xq) / len(xq), "xt": sum(xt) / max(len(xt), 1), "copy_max": max(copies) if copies else None,
                    "n": (len(xq), len(xt))}
KiB = 1024.0
xq_b = (2 * res["FETCH_SIZE"]["xq"] + res["WRITE_SIZE"]["xq"]) * KiB
xt_b = (2 * res["FETCH_SIZE"]["xt"] + res["WRITE_SIZE"]["xt"]) * KiB
entry = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/pmc_traffic.sh + pmc_traffic_summary.py, {os.path.basename(out_dir.rstrip('/'))}); sorted + dealt layout",
    "units": f"KiB; FETCH_SIZE x2 (gfx950: 128-byte requests tallied as 64 B; the 4 GiB calibration copy of the same pass reports {res['FETCH_SIZE']['copy_max']} KiB fetched, {res['WRITE_SIZE']['copy_max']} KiB written), WRITE_SIZE x1",
    "workload": f"{cells} x {peaks}, B=64",
    "spmm_xq_bytes_per_launch": xq_b,
    "spmm_xty_bytes_per_launch": xt_b,
    "launches_measured": {"xq": res["FETCH_SIZE"]["n"][0], "xty": res["FETCH_SIZE"]["n"][1]},
    "raw_KiB": {"spmm_xq.FETCH_SIZE": res["FETCH_SIZE"]["xq"], "spmm_xty.FETCH_SIZE": res["FETCH_SIZE"]["xt"],
                "spmm_xq.WRITE_SIZE": res["WRITE_SIZE"]["xq"], "spmm_xty.WRITE_SIZE": res["WRITE_SIZE"]["xt"]},
}
print(json.dumps(entry, indent=1))
if "--update" in sys.argv:
    name = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "r03_spmm_traffic.json"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
    d = json.load(open(path)) if os.path.exists(path) else {}
    old = d.get(f"{cells}x{peaks}", {})
    if "algorithmic_bytes_per_launch" in old:
        entry["algorithmic_bytes_per_launch"] = old["algorithmic_bytes_per_launch"]
    if "--alg" in sys.argv:
        entry["algorithmic_bytes_per_launch"] = int(sys.argv[sys.argv.index("--alg") + 1])
    entry["spmm_mean_bytes_per_launch"] = (4 * xq_b + 3 * xt_b) / 7  # 4 X*Q + 3 X^T*Y per lsi call
    d[f"{cells}x{peaks}"] = entry
    json.dump(d, open(path, "w"), indent=1)
    print("updated", path)

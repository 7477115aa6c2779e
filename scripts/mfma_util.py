#!/usr/bin/env python
"""Summarise the MFMA PMC passes of scripts/pmc_mfma.sh: per kernel, MFMA busy fraction
(rocprofv3's MfmaUtil expression: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs))
and the MFMA flop rate (SQ_INSTS_VALU_MFMA_MOPS_* x 512 / kernel duration)."""
import csv
import re
import sys
from collections import defaultdict

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def load(path):
    rows = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            d = rows[int(r["Dispatch_Id"])]
            d["name"] = short(r["Kernel_Name"])
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows


def main(pmc_dir, grbm_dir):
    a, g = load(f"{pmc_dir}/pmc_counter_collection.csv"), load(f"{grbm_dir}/pmc_counter_collection.csv")
    occ = defaultdict(list)
    for i in sorted(g):
        occ[g[i]["name"]].append(g[i].get("GRBM_GUI_ACTIVE", 0.0))
    seen = defaultdict(int)
    agg = defaultdict(lambda: defaultdict(float))
    for i in sorted(a):
        d = a[i]
        n = d["name"]
        k = seen[n]
        seen[n] += 1
        mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) + d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
        if mops == 0:
            continue
        s = agg[n]
        s["calls"] += 1
        s["ns"] += d["ns"]
        s["busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        s["flop"] += 512.0 * mops
        s["f64"] += d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
        s["gui"] += occ[n][k] if k < len(occ[n]) else float("nan")
    print(f"{'kernel':70s} {'calls':>5s} {'avg us':>9s} {'MFMA busy %':>11s} {'TFLOP/s':>8s} {'dtype':>5s}")
    for n, s in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        util = 100.0 * s["busy"] / (s["gui"] * SIMDS) if s["gui"] else float("nan")
        print(f"{n:70s} {int(s['calls']):5d} {s['ns'] / s['calls'] / 1e3:9.1f} {util:11.1f} "
              f"{s['flop'] / s['ns'] / 1e3:8.2f} {'f64' if s['f64'] else 'f32':>5s}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

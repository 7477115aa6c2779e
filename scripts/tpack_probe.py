#!/usr/bin/env python
"""Timing ablations of the transpose-pack fill on the bench matrix (tune knobs tpack_abl / tpack_c)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend
be = HipBackend(0)
X = be.synth_counts(0, 125000, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, 125000, 3, 1e4)
import sys
SORT = "--natural" not in sys.argv
def t(label):
    be.transpose_pack(T, sort_rows=SORT); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): be.transpose_pack(T, sort_rows=SORT)
    e.record(); torch.cuda.synchronize()
    print(f"{label}: {s.elapsed_time(e)/3:.2f} ms (count + layout + scan + fill + pads; sorted layout = {SORT})", flush=True)
t("full")
be.tune("tpack_abl", 8); t("row loads from the cursor (not line aligned)"); be.tune("tpack_abl", 0)
for abl, name in ((1, "no count walk"), (2, "no place walk"), (4, "no write-out"), (3, "no walks"), (7, "setup/scans/barriers only")):
    be.tune("tpack_abl", abl); t(name)
be.tune("tpack_abl", 0)
for c in (256, 384, 640, 768):
    be.tune("tpack_c", c); t(f"C={c}")
be.tune("tpack_c", 0)

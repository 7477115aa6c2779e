#!/usr/bin/env python
"""Timing of the transpose-pack on the bench matrix: third-generation fill (count rides on the
previous tile's place walk) against the two-walk kernel, with its ablations and tile widths."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 125000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)
SORT = "--natural" not in sys.argv


def t(label):
    be.transpose_stream(T, sort_rows=SORT)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        be.transpose_stream(T, sort_rows=SORT)
    e.record()
    torch.cuda.synchronize()
    print(f"{label}: {s.elapsed_time(e) / 3:.2f} ms (count + layout + scan + fill + pads; sorted layout = {SORT})", flush=True)


t("v3 full")
for c in (384, 512, 640):
    be.tune("tpack_c", c)
    t(f"v3 C={c}")
be.tune("tpack_c", 0)
be.tune("tpack_v2", 1)
t("v2 full")
for abl, name in ((1, "v2 no count walk"), (2, "v2 no place walk"), (4, "v2 no write-out"), (3, "v2 no walks"),
                  (7, "v2 setup/scans/barriers only")):
    be.tune("tpack_abl", abl)
    t(name)
be.tune("tpack_abl", 0)
for c in (640, 768):
    be.tune("tpack_c", c)
    t(f"v2 C={c}")
be.tune("tpack_c", 0)
be.tune("tpack_v2", 0)

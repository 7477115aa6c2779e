#!/usr/bin/env python
"""Timing of the transposition (csrc/tpack.hip) on the bench matrix: the third-generation fill at
several tile widths and its per-phase cycle accounting (tune tpack_dbg)."""
import ctypes
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
NAMES = ["header+scan", "count walk", "prefix", "place walk", "wait others", "write-out"]


def t(label, dbg=False):
    be.transpose_stream(T, sort_rows=SORT)
    torch.cuda.synchronize()
    if dbg:
        be.tune("tpack_dbg", 1)
        be.lib.mu_csr_tpack_phase_cycles(None, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        be.transpose_stream(T, sort_rows=SORT)
    e.record()
    torch.cuda.synchronize()
    print(f"{label}: {s.elapsed_time(e) / 3:.2f} ms (count + layout + scan + fill; sorted layout = {SORT})", flush=True)
    if dbg:
        out = (ctypes.c_ulonglong * 6)()
        be.lib.mu_csr_tpack_phase_cycles(ctypes.cast(out, ctypes.c_void_p), 0)
        be.tune("tpack_dbg", 0)
        tot = sum(out)
        print("   phase share of the fill (wave 0 of every workgroup): " +
              ", ".join(f"{n} {100.0 * v / tot:.1f} %" for n, v in zip(NAMES, out)), flush=True)


t("v3 full")
t("v3 full, phases", dbg=True)
for c in (256, 384, 512, 640):
    be.tune("tpack_c", c)
    t(f"v3 C={c}", dbg=(c == 256))
be.tune("tpack_c", 0)
for rows in (8, 16, 24, 48):
    be.tune("tpack_rows", rows)
    t(f"v3 {rows}e5 entries per row block", dbg=(rows == 16))
be.tune("tpack_rows", 0)

#!/usr/bin/env python
"""What bounds the fourth-generation fill: tile-width sweep (visits per row scale with 1 / width) and the run without
its global stores.  Usage: tpack4_abl.py [cells]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
peaks = 200000
X = be.synth_counts(0, cells, peaks, 50, 0.03, 0)
out = torch.empty_like(X.values)
T = tfidf_device(be, X, cells, 3, 1e4, out=out, emit_stream=True)
xs, row_dst = be._xstream_of(T)


def fill_ms(reps=2):
    r = be.transpose_stream(T, src=(xs, row_dst))
    del r
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        r = be.transpose_stream(T, src=(xs, row_dst))
        del r
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


circ = int(os.environ.get("TPACK4_CIRC", "0"))  # (0: default = circular windows, 2: plain)
be.tune("tpack4_circ", circ)
print(f"{cells} x {peaks}: count + layout + fill, stream source" + (", plain windows" if circ == 2 else ", circular windows"), flush=True)
for c in (448, 480, 512):
    be.tune("tpack4_c", c)
    a = fill_ms()
    be.tune("tpack4_abl", 2)
    b = fill_ms()
    be.tune("tpack4_abl", 4)
    c4 = fill_ms()
    be.tune("tpack4_abl", 0)
    print(f"tile {c:4d} columns: {a:7.2f} ms, without the write-out's stores {b:7.2f} ms, all runs stored to one 8 KiB region {c4:7.2f} ms", flush=True)
be.tune("tpack4_c", 0)

#!/usr/bin/env python
"""The fill once per store variant (tune tpack4_abl 0 / 2 / 4) for a counter pass: run under
rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE) and read the k_t4_fill dispatches in order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
out = torch.empty_like(X.values)
T = tfidf_device(be, X, cells, 3, 1e4, out=out, emit_stream=True)
xs, row_dst = be._xstream_of(T)
be.tune("tpack4_circ", int(os.environ.get("TPACK4_CIRC", "0"))  # (0: default = circular windows, 2: plain))
for abl in (0, 2, 4):
    be.tune("tpack4_abl", abl)
    r = be.transpose_stream(T, src=(xs, row_dst))
    del r
    torch.cuda.synchronize()
be.tune("tpack4_abl", 0)
be.tune("tpack4_circ", 0)

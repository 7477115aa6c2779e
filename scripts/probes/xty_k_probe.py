#!/usr/bin/env python
"""X^T Y at 1e6 x 200k with the layout of X^T dealt for K = 5 .. 8 row-sets per wave (K = 6 as for
X Q?).  One operand at a time (the probe of r03 held them all and does not fit any more)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)
del X
Y = torch.randn((cells, 64), device="cuda")
for K in (7, 6, 5, 8, 7, 6):
    Xt = be.transpose_stream(T, K=K)
    for _ in range(2):
        be.spmm(Xt, Y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        Z = be.spmm(Xt, Y)
    e.record()
    torch.cuda.synchronize()
    print(f"K = {K}: {s.elapsed_time(e) / 5:.2f} ms per X^T Y ({Xt.n_pos // (64 * K)} workgroups)", flush=True)
    del Xt, Z
    torch.cuda.empty_cache()

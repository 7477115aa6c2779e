#!/usr/bin/env python
"""Where a small LSI call (10k x 30k) spends its time: host Ritz steps vs waits vs the rest."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
n, d = 10000, 30000
X = be.synth_counts(0, n, d, 50, 0.03, 0)
T = tfidf_device(be, X, n, 3, 1e4)
for mb in (3, 2):
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        U, s, V, info = lsi_device(be, T, n_comps=50, return_info=True, max_blocks=mb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    print(f"max_blocks={mb}: {dt:.2f} ms, iterations {info['iterations']}, spmm {info['spmm']}, host {info['host']}")
torch.cuda.synchronize()
t0 = time.perf_counter()
Tp = be.stream(T); Tt = be.transpose_stream(T)
torch.cuda.synchronize()
print(f"pack + transpose_stream: {(time.perf_counter() - t0) * 1e3:.2f} ms")
t0 = time.perf_counter()
T2 = tfidf_device(be, X, n, 3, 1e4)
torch.cuda.synchronize()
print(f"tfidf: {(time.perf_counter() - t0) * 1e3:.2f} ms")

#!/usr/bin/env python
"""Host-side profile of a small LSI call (10k x 30k): cProfile of lsi_device, top cumulative entries."""
import cProfile
import os
import pstats
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
for _ in range(3):
    lsi_device(be, T, n_comps=50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    lsi_device(be, T, n_comps=50)
torch.cuda.synchronize()
print(f"lsi_device: {(time.perf_counter() - t0) * 100:.2f} ms per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    lsi_device(be, T, n_comps=50)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(38)

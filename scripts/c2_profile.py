#!/usr/bin/env python
"""Where a 10k x 30k tfidf + lsi call spends its time on the host (BASELINE configs[1]; other shapes: argv cells
peaks): ms per step, the host's wait / Ritz shares, cProfile of 20 calls."""
import cProfile
import io
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
CELLS = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
PEAKS = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
X = be.synth_counts(0, CELLS, PEAKS, 50, 0.03, 0)
out = torch.empty_like(X.values)


def step():
    T = tfidf_device(be, X, CELLS, 3, 1e4, out=out)
    return lsi_device(be, T, n_comps=50, n_obs=CELLS, return_info=True)


for _ in range(5):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        r = step()
    torch.cuda.synchronize()
    print("ms/step", round((time.perf_counter() - t0) / 20 * 1e3, 2), {k: round(v, 2) for k, v in r[3]["host"].items()},
          r[3]["iterations"], r[3]["restarts"], flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])

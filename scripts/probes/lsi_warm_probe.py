#!/usr/bin/env python
"""The subsampled power warm start of the block Lanczos iteration (muon_amd/_atac/tools.py, r05) on the bench matrix:
products, bounds, time per lsi call and the angle between the warm and the cold run's top-k subspaces, per (cell
fraction, power steps).  Usage: lsi_warm_probe.py [cells] [spec ...]   (spec = "frac:q"; default a small sweep)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
specs = [a for a in sys.argv[2:]] or ["0", "16:1", "16:2", "8:1", "32:1", "32:2", "64:2"]
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
out = torch.empty_like(X.values)
T = tfidf_device(be, X, cells, 3, 1e4, out=out)
ref = None
for spec in specs:
    os.environ["MUON_AMD_LSI_WARM"] = spec
    lsi_device(be, T, n_comps=50, n_obs=cells)  # warm-up (allocator)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        U, sd, V, info = lsi_device(be, T, n_comps=50, n_obs=cells, return_info=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 2 * 1e3
    q, _ = torch.linalg.qr(V.double())
    if ref is None:
        ref, ref_sd = q, sd
    ang = float(torch.linalg.matrix_norm(ref - q @ (q.T @ ref), ord=2))
    print(f"warm {spec:5s}: {ms:7.1f} ms per lsi call, expansions {info['iterations']}, products {info['spmm']} (+ warm start {info['warm_start']}), "
          f"bounds {[float(f'{b:.2g}') for b in info['bounds']]}, converged {info['converged']}, angle to the first run {ang:.2e}, "
          f"stdev max rel diff {float(np.max(np.abs(sd - ref_sd) / ref_sd)):.1e}", flush=True)

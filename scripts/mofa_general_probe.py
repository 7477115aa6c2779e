#!/usr/bin/env python
"""GeneralMofaEngine (poisson / bernoulli pseudo-data, chunk passes as device tensor operations) at a
moderate size on one MI355X: seconds per iteration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from muon_amd._backend import HipBackend
from muon_amd._core.mofa_general import GeneralMofaEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
be = HipBackend(0)
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, 5)).astype(np.float32)
y1 = (Z @ rng.standard_normal((2000, 5)).T.astype(np.float32) + rng.standard_normal((N, 2000))).astype(np.float32)
rate = np.logaddexp(0, Z @ (0.3 * rng.standard_normal((20000, 5))).T.astype(np.float32) - 3.0)
y2 = sp.csr_matrix(rng.poisson(rate).astype(np.float32))
print(f"N={N}: gaussian {y1.shape}, poisson {y2.shape} ({y2.nnz} nnz, {y2.nnz / N / 20000:.3f} dense)", flush=True)
dts = (torch.float32,) if 'f32' in sys.argv else (torch.float32, torch.float64)
ITERS = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('iters=')), 5)
for dt in dts:
    eng = GeneralMofaEngine(be, [y1, y2], ["gaussian", "poisson"], np.zeros(N, dtype=int), 10, dtype=dt, seed=1)
    eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        eng.step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / ITERS
    e = np.asarray(eng.elbo)
    print(f"{dt}: {per * 1e3:.1f} ms per iteration; ELBO monotone {bool(np.all(np.diff(e) > -1e-5 * abs(e[0])))}", flush=True)
    del eng

#!/usr/bin/env python
"""GeneralMofaEngine on a binarised sparse view (the likelihood mofapy2 guesses for `ac.pp.binarize`d ATAC data):
ms per iteration, kernel-level cost by torch profiler-free timing of the three passes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import torch

from muon_amd._backend import HipBackend
from muon_amd._core.mofa_general import GeneralMofaEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
be = HipBackend(0)
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, 5)).astype(np.float32)
logit = Z @ (0.5 * rng.standard_normal((D, 5))).T.astype(np.float32) - 3.0
y = sp.csr_matrix((rng.random((N, D)) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32))
print(f"N={N}, D={D}: bernoulli view with {y.nnz} ones ({y.nnz / N / D:.3f} dense)", flush=True)
for dt in (torch.float32, torch.float64):
    for fused in (False, True):
        eng = GeneralMofaEngine(be, [y], ["bernoulli"], np.zeros(N, dtype=int), 10, dtype=dt, seed=1)
        eng.views[0].fusedb = bool(fused and eng.views[0].fusedb)  # (False: the chunk passes of r03-r05)
        eng.step()
        torch.cuda.synchronize()
        for name, fn in (("W update", lambda: eng._update_w(0)), ("Z update", eng._update_z), ("tau / ELBO", eng._update_rest_and_elbo)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            print(f"  {name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step()
        torch.cuda.synchronize()
        e = np.asarray(eng.elbo)
        print(f"{dt}, {'one sweep per update, sparse products' if eng.views[0].fusedb else 'dense chunk passes'}: "
              f"{(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per iteration; ELBO monotone "
              f"{bool(np.all(np.diff(e) > -1e-5 * abs(e[0])))}; last ELBO {e[-1]:.10e}", flush=True)
        del eng
        torch.cuda.empty_cache()

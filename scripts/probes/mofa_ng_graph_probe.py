#!/usr/bin/env python
"""bench.py --workload mofa_ng's model: the iteration replayed as a HIP graph against eager launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from muon_amd._backend import get_backend
from muon_amd._core.mofa_general import GeneralMofaEngine
from scripts.bench_widened import _ng_views

be = get_backend()
y1, y2 = _ng_views(20000, 2000, 20000, 0)
y1[::97, ::13] = np.nan
for graph in ("0", "1"):
    os.environ["MUON_AMD_MOFA_NG_GRAPH"] = graph
    eng = GeneralMofaEngine(be, [y1, y2], ["gaussian", "poisson"], np.zeros(20000, dtype=int), 10, dtype=torch.float32, seed=1)
    for _ in range(2):
        eng.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.step()
    torch.cuda.synchronize()
    first = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(50):
        eng.step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t) / 50
    print(f"graph {graph}: captured {eng._graph is not None}, third step {1e3 * first:.1f} ms, then {1e3 * per:.2f} ms per iteration, ELBO {eng.elbo[-1]:.6e}", flush=True)

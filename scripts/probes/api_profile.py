#!/usr/bin/env python
"""cProfile of the API path ac.pp.tfidf(adata); ac.tl.lsi(adata) from a host scipy CSR (bench.py --workload c3_api):
where the host spends the time that is neither PCIe nor kernels.  Usage: api_profile.py [cells]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import torch

from muon_amd import AnnData
from muon_amd import atac as ac
from muon_amd._backend import get_backend

be = get_backend()
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
m = sp.csr_matrix((be.to_host(X.values).astype(np.float32), be.to_host(X.indices), be.to_host(X.indptr)), shape=X.shape)
m.has_sorted_indices = True
m.has_canonical_format = True
del X
torch.cuda.empty_cache()
for it in range(3):
    ad = AnnData(m.copy())
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if it == 1:
        pr.enable()
    ac.pp.tfidf(ad, backend=be)
    ta = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ac.tl.lsi(ad, backend=be)
    torch.cuda.synchronize()
    if it == 1:
        pr.disable()
    t2 = time.perf_counter()
    print(f"run {it}: tfidf {1e3 * (t1 - t0):.0f} ms (of which the synchronize after it {1e3 * (t1 - ta):.0f}), lsi {1e3 * (t2 - t1):.0f} ms"
          + ("  [profiled]" if it == 1 else ""), flush=True)
    if it == 1:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue())
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
        print(s.getvalue())
    del ad

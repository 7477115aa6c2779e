#!/usr/bin/env python
"""How many cells share a neighbour with a cell (the candidates of its kernel bandwidth, csrc/wnn.hip): with
repetitions (sum over its neighbours of their reverse degrees) and distinct (row lengths of A A^T)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp

from muon_amd import AnnData
from muon_amd._backend import HipBackend
from muon_amd._core import preproc as pp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
be = HipBackend(0)
rng = np.random.default_rng(0)
lab = rng.integers(0, 30, n)
for p in (50, 30):
    x = rng.standard_normal((30, p))[lab] * 2 + rng.standard_normal((n, p))
    a = AnnData(x)
    pp.knn(a, n_neighbors=20, use_rep="X", backend=be)
    G = a.obsp["distances"].tocsr()
    rdeg = np.bincount(G.indices, minlength=n)
    cand = np.add.reduceat(rdeg[G.indices], G.indptr[:-1])
    A = sp.csr_matrix((np.ones(G.nnz, dtype=np.float32), G.indices, G.indptr), shape=G.shape)
    uniq = np.diff((A @ A.T).tocsr().indptr)
    q = lambda v: (v.mean(), np.percentile(v, 99), np.percentile(v, 99.9), v.max())  # noqa: E731
    print(f"p={p}: reverse degree max {rdeg.max()}; with repetitions mean/p99/p99.9/max {q(cand)}; distinct {q(uniq)}", flush=True)

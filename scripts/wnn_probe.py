#!/usr/bin/env python
"""Weighted nearest neighbours at scale on one MI355X: two modalities (50- and 30-dimensional embeddings
of planted clusters), pp.knn per modality + pp.neighbors, wall time per stage."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from muon_amd import AnnData, MuData
from muon_amd._backend import HipBackend
from muon_amd._core import preproc as pp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
be = HipBackend(0)
rng = np.random.default_rng(0)
lab = rng.integers(0, 30, n)
x1 = rng.standard_normal((30, 50))[lab] * 2 + rng.standard_normal((n, 50))
x2 = rng.standard_normal((30, 30))[lab] * 2 + rng.standard_normal((n, 30))
md = MuData({"rna": AnnData(x1), "atac": AnnData(x2)})


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"{name}: {time.perf_counter() - t0:.2f} s", flush=True)


for m in md.mod:
    timed(f"knn {m} (n={n}, k=20)", lambda m=m: pp.knn(md.mod[m], n_neighbors=20, use_rep="X", backend=be))
timed("l2norm", lambda: pp.l2norm(md, rep="X"))
timed(f"neighbors (n_multineighbors=200)", lambda: pp.neighbors(md, backend=be))
g = md.obsp["distances"]
print("graph", g.shape, "nnz/row", g.nnz // n, "same-cluster fraction", float(np.mean(lab[g.indices] == np.repeat(lab, g.nnz // n))))
print("mean weight rna", float(md.obs["rna:mod_weight"].mean()))

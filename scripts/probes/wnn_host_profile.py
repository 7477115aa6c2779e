"""cProfile of pp.knn x 2 + pp.neighbors at 100 000 cells (what bench.py --workload wnn times), host side."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from muon_amd import AnnData, MuData
from muon_amd._backend import HipBackend
from muon_amd._core import preproc as pp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
be = HipBackend(0)
rng = np.random.default_rng(0)
lab = rng.integers(0, 30, n)
x1 = rng.standard_normal((30, 50))[lab] * 2 + rng.standard_normal((n, 50))
x2 = rng.standard_normal((30, 30))[lab] * 2 + rng.standard_normal((n, 30))


def run():
    md = MuData({"rna": AnnData(x1), "atac": AnnData(x2)})
    t = {}
    for m in md.mod:
        t0 = time.perf_counter()
        pp.knn(md.mod[m], n_neighbors=20, use_rep="X", backend=be)
        torch.cuda.synchronize()
        t["knn " + m] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pp.neighbors(md, backend=be)
    torch.cuda.synchronize()
    t["neighbors"] = time.perf_counter() - t0
    return t


run()
print({k: round(v, 3) for k, v in run().items()}, flush=True)
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

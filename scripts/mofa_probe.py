#!/usr/bin/env python
"""MOFA timing on a device-generated two-view MuData-like input (BASELINE.json configs[3]:
rna N x 20k dense + atac N x 100k sparse (TF-IDF'd planted counts), K = 10, fixed iterations)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend
from muon_amd._core.mofa_engine import MofaEngine

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=100000)
ap.add_argument("--rna", type=int, default=20000)
ap.add_argument("--atac", type=int, default=100000)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--f64", action="store_true")
args = ap.parse_args()
be = HipBackend(0)
T = torch.float64 if args.f64 else torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
N, K0 = args.cells, 10
Z = torch.randn((N, K0), generator=g, device="cuda", dtype=torch.float32)
W = torch.randn((args.rna, K0), generator=g, device="cuda") * (torch.rand((args.rna, K0), generator=g, device="cuda") < 0.3)
rna = Z @ W.T
rna += torch.randn(rna.shape, generator=g, device="cuda")
X = be.synth_counts(0, N, args.atac, 50, 0.03, 0)
atac = tfidf_device(be, X, N, 3, 1e4)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng = MofaEngine(be, [rna, atac], np.zeros(N, dtype=np.int64), 10, dtype=T, seed=1)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(args.iters):
    eng.step()
torch.cuda.synchronize()
t3 = time.perf_counter()
dense_b = 4 * N * args.rna * (2 if args.f64 else 1)
sparse_b = atac.nnz * (12 if args.f64 else 8)
alg = 2 * (dense_b + sparse_b)
per = (t3 - t2) / args.iters
print(f"MOFA {N} x ({args.rna} dense + {args.atac} sparse, nnz={atac.nnz}) K=10 dtype={T}: setup {t1 - t0:.2f}s, "
      f"{per * 1e3:.2f} ms/iter, {args.iters} iters {t3 - t2:.2f}s, algorithmic {alg / 1e9:.1f} GB/iter -> {alg / per / 1e9:.0f} GB/s; "
      f"ELBO {eng.elbo[0]:.6e} -> {eng.elbo[-1]:.6e} monotone={bool(np.all(np.diff(eng.elbo) > -1e-5 * abs(eng.elbo[0])))}")

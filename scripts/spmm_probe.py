#!/usr/bin/env python
"""Micro-benchmark of the CSR SpMM in both directions (X*Q and X^T*Y) on the bench matrix.
Used with rocprofv3 (--kernel-trace --stats, or --pmc in a separate pass)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=125000)
ap.add_argument("--peaks", type=int, default=200000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--B", type=int, default=64)
args = ap.parse_args()

be = HipBackend(0)
X = be.synth_counts(0, args.cells, args.peaks, 50, 0.03, 0)
T = tfidf_device(be, X, args.cells, 3, 1e4)
Tt = be.transpose(T)
Q = be.randn(args.peaks, args.B, 1)
Y = be.spmm(T, Q)
Z = be.spmm(Tt, Y)
torch.cuda.synchronize()
for name, M, D in (("X*Q", T, Q), ("Xt*Y", Tt, Y)):
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.reps):
        be.spmm(M, D)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / args.reps
    n, d = M.shape
    byt = 8 * M.nnz + 8 * (n + 1) + 4 * args.B * (n + d)
    print(f"{name}: {n}x{d} nnz={M.nnz} {ms:.3f} ms  {byt / ms / 1e6:.0f} GB/s algorithmic  "
          f"{M.nnz / ms / 1e6:.1f} Gnnz/s")

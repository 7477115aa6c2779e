#!/usr/bin/env python
"""Subspace angle of the top-k right singular subspace after q expansions of the block Lanczos
iteration, against a long run (q = 14), on the bench matrix.  Shows how conservative the stopping
rule of lsi_device is."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=125000)
ap.add_argument("--peaks", type=int, default=200000)
ap.add_argument("--k", type=int, default=50)
args = ap.parse_args()
be = HipBackend(0)
X = be.synth_counts(0, args.cells, args.peaks, 50, 0.03, 0)
T = tfidf_device(be, X, args.cells, 3, 1e4)


def angle(Va, Vb):
    qa, _ = torch.linalg.qr(Va.double())
    qb, _ = torch.linalg.qr(Vb.double())
    s = torch.linalg.svdvals(qa.T @ qb).clamp(max=1.0)
    # sin of the largest principal angle, computed stably from the residual
    r = qb - qa @ (qa.T @ qb)
    return float(torch.linalg.matrix_norm(r, ord=2))


_, sref, Vref, _ = lsi_device(be, T, n_comps=args.k, n_iter=14, return_info=True)
for q in range(1, 8):
    _, s, V, info = lsi_device(be, T, n_comps=args.k, n_iter=q, return_info=True)
    print(f"n_iter={q}: sin(max angle) vs n_iter=14: {angle(Vref, V):.3e}   max rel stdev err {np.max(np.abs(s - sref) / sref):.2e}", flush=True)
_, s, V, info = lsi_device(be, T, n_comps=args.k, return_info=True)
print("default rule: iterations", info["iterations"], "angle", f"{angle(Vref, V):.3e}",
      "measured s_j", ["%.2e" % a for a in info["angles"]], "predicted err", info["predicted_angle"],
      "spmm", info["spmm"], "unused", info["spmm_unused"], "restarts", info["restarts"], "host", info["host"])

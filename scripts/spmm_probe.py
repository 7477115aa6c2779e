#!/usr/bin/env python
"""Micro-benchmark of the packed SpMM in both directions (X*Q and X^T*Y) on the bench matrix,
with the kernel's timing ablations.  Used bare or under rocprofv3 (--kernel-trace --stats, or
--pmc in a separate pass)."""
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
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--modes", default="0")
ap.add_argument("--fma", default="0")
ap.add_argument("--ks", default="0")
ap.add_argument("--csr", action="store_true", help="also time the CSR kernel")
args = ap.parse_args()

be = HipBackend(0)
X = be.synth_counts(0, args.cells, args.peaks, 50, 0.03, 0)
T = tfidf_device(be, X, args.cells, 3, 1e4)
Tt = be.transpose(T)
Tp, Ttp = be.pack(T), be.pack(Tt)
Q = be.randn(args.peaks, 64, 1)
Y = be.spmm(T, Q)
Yp = be.spmm(Tp, Q)
print("packed vs csr kernel max abs diff:", float((Y - Yp).abs().max()), "scale", float(Y.abs().max()))
torch.cuda.synchronize()


def timeit(M, D):
    be.spmm(M, D)
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.reps):
        be.spmm(M, D)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / args.reps


ops = [("X*Q ", Tp, Q), ("Xt*Y", Ttp, Y)]
if args.csr:
    ops += [("X*Q  csr", T, Q), ("Xt*Y csr", Tt, Y)]
for fma in [int(x) for x in args.fma.split(",")]:
    for K in [int(x) for x in args.ks.split(",")]:
        for mode in [int(x) for x in args.modes.split(",")]:
            be.tune("spmm_fma", fma)
            be.tune("spmm_k", K)
            be.tune("spmm_mode", mode)
            for name, M, D in ops:
                ms = timeit(M, D)
                n, d = M.shape
                byt = 8 * M.nnz + 8 * (n + 1) + 4 * 64 * (n + d)
                print(f"fma={fma} K={K} mode={mode:2d} {name}: {ms:7.3f} ms  {byt / ms / 1e6:6.0f} GB/s alg  "
                      f"{M.nnz / ms / 1e6:6.1f} Gnnz/s", flush=True)
be.tune("spmm_mode", 0)
be.tune("spmm_k", 0)
be.tune("spmm_fma", 0)

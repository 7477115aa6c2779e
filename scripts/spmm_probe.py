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
ap.add_argument("--waves", default="0")
ap.add_argument("--pipe", default="0")
ap.add_argument("--ks", default="0")
ap.add_argument("--csr", action="store_true", help="also time the CSR kernel")
ap.add_argument("--no-round-fill", action="store_true", help="layout without the whole-rounds rule")
ap.add_argument("--unstructured", action="store_true", help="uniform random pattern, values 1 + Poisson(0.5) (SURVEY 8d)")
ap.add_argument("--calibrate", action="store_true",
                help="run a 4 GiB device copy first (known HBM byte count for PMC calibration)")
args = ap.parse_args()

be = HipBackend(0)
be._no_round_fill = args.no_round_fill
if args.calibrate:
    src = torch.empty(1 << 30, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)  # reads 4 GiB, writes 4 GiB (far beyond the 256 MiB Infinity Cache)
    torch.cuda.synchronize()
    del src, dst
X = be.synth_counts(0, args.cells, args.peaks, 0 if args.unstructured else 50, 0.03, 0)
print(f"{'unstructured' if args.unstructured else 'planted-topic'} {args.cells} x {args.peaks}, nnz {X.nnz}")
T = tfidf_device(be, X, args.cells, 3, 1e4)
Tp, Ttp = be.pack(T), be.transpose_pack(T)
Q = be.randn(args.peaks, 64, 1)
Yp = be.spmm(Tp, Q)
Y = Yp
if args.csr:
    Tt = be.transpose(T)
    Y = be.spmm(T, Q)
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
Zp = be.spmm(Ttp, Y)
ref = {"X*Q ": Yp.clone(), "Xt*Y": Zp.clone()}
from muon_amd._ffi import MuonAmdError
for waves in [int(x) for x in args.waves.split(",")]:
  for pipe in [int(x) for x in args.pipe.split(",")]:
    for K in [int(x) for x in args.ks.split(",")]:
        for mode in [int(x) for x in args.modes.split(",")]:
            be.tune("spmm_waves", waves)
            be.tune("spmm_pipe", pipe)
            be.tune("spmm_k", K)
            be.tune("spmm_mode", mode)
            for name, M, D in ops:
                try:
                    ms = timeit(M, D)
                except MuonAmdError as e:
                    print(f"waves={waves} pipe={pipe} K={K} mode={mode} {name}: {e}")
                    continue
                same = ""
                if mode in (0, 4) and name in ref:
                    same = " bit-identical" if torch.equal(be.spmm(M, D), ref[name]) else " DIFFERS"
                n, d = M.shape
                byt = 8 * M.nnz + 8 * (n + 1) + 4 * 64 * (n + d)
                print(f"waves={waves:2d} pipe={pipe} K={K:2d} mode={mode:2d} {name}: {ms:7.3f} ms  {byt / ms / 1e6:6.0f} GB/s alg  "
                      f"{M.nnz / ms / 1e6:6.1f} Gnnz/s{same}", flush=True)
for k in ("spmm_mode", "spmm_k", "spmm_waves", "spmm_pipe"):
    be.tune(k, 0)

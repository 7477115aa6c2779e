#!/usr/bin/env python
"""r02: the CSR-direct windowed SpMM (mu_spmm_csr_f32) against the r01 packed kernel
(mu_spmm_packed_f32) in both directions on the bench matrix: bit-identity, launch times, and the cost
of building the operands (transpose_csr vs transpose_pack + pack)."""
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
ap.add_argument("--no-packed", action="store_true")
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--ks", default="", help="also time the pair-stream kernel with layouts dealt for these K")
args = ap.parse_args()

be = HipBackend(0)
X = be.synth_counts(0, args.cells, args.peaks, 50, 0.03, 0)
print(f"planted-topic {args.cells} x {args.peaks}, nnz {X.nnz}", flush=True)
T = tfidf_device(be, X, args.cells, 3, 1e4)


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize()
    s, e = ev(), ev()
    s.record()
    for _ in range(reps):
        r = fn()
    e.record()
    torch.cuda.synchronize()
    return r, s.elapsed_time(e) / reps


Tl, ms = timed(lambda: be.spmm_layout(T))
print(f"spmm_layout(X):      {ms:8.3f} ms", flush=True)
Tt, ms = timed(lambda: be.transpose_csr(T))
print(f"transpose_csr(X):    {ms:8.3f} ms", flush=True)
Tpr, ms = timed(lambda: be.pairs(Tl))
print(f"pairs(X):            {ms:8.3f} ms", flush=True)
Ttpr, ms = timed(lambda: be.transpose_pairs(T))
print(f"transpose_pairs(X):  {ms:8.3f} ms", flush=True)
B = args.B
Q = be.randn(args.peaks, B, 1)
Yn = be.spmm(Tl, Q)
Zn = be.spmm(Tt, Yn)
torch.cuda.synchronize()
print("X*Q  pairs vs csr-win:", "bit-identical" if torch.equal(be.spmm(Tpr, Q), Yn) else "DIFFERS")
print("Xt*Y pairs vs csr-win:", "bit-identical" if torch.equal(be.spmm(Ttpr, Yn), Zn) else "DIFFERS", flush=True)
if not args.no_packed:
    Tp, ms = timed(lambda: be.pack(T))
    print(f"pack(X):             {ms:8.3f} ms", flush=True)
    Ttp, ms = timed(lambda: be.transpose_pack(T))
    print(f"transpose_pack(X):   {ms:8.3f} ms", flush=True)
    Yp = be.spmm(Tp, Q)
    Zp = be.spmm(Ttp, Yp)
    print("X*Q  csr-win vs packed:", "bit-identical" if torch.equal(Yn, Yp) else
          f"DIFFERS max abs {float((Yn - Yp).abs().max()):.3e} (scale {float(Yp.abs().max()):.3e})")
    print("Xt*Y csr-win vs packed:", "bit-identical" if torch.equal(Zn, Zp) else
          f"DIFFERS max abs {float((Zn - Zp).abs().max()):.3e} (scale {float(Zp.abs().max()):.3e})", flush=True)
# natural row order gives the same sums
Y0 = be.spmm(T, Q)
print("layout vs natural order:", "bit-identical" if torch.equal(Y0, Yn) else "DIFFERS", flush=True)


def bench(name, M, D):
    _, ms = timed(lambda: be.spmm(M, D), args.reps)
    n, d = M.shape
    byt = 8 * M.nnz + 8 * (n + 1) + 4 * B * (n + d)
    print(f"{name}: {ms:8.3f} ms  {byt / ms / 1e6:6.0f} GB/s alg ({byt / ms / 8e7:5.1f} % of 8 TB/s)  {M.nnz / ms / 1e6:6.1f} Gnnz/s", flush=True)


for mode in [int(m) for m in args.modes.split(",")]:
    be.tune("spmm_mode", mode)
    print(f"-- spmm_mode {mode}")
    bench("X*Q  csr-win (layout) ", Tl, Q)
    bench("X*Q  pairs   (layout) ", Tpr, Q)
    if mode == 0:
        bench("Xt*Y pairs   (layout) ", Ttpr, Yn)
        bench("Xt*Y csr-win (layout) ", Tt, Yn)
        bench("X*Q  csr-win (natural)", T, Q)
        if not args.no_packed:
            bench("X*Q  packed r01       ", Tp, Q)
            bench("Xt*Y packed r01       ", Ttp, Yn)
be.tune("spmm_mode", 0)
from muon_amd._backend import DevicePairs
for K in [int(k) for k in args.ks.split(",") if k]:
    perm, _inv, _ = be.packed_layout(T.indptr[1:] - T.indptr[:-1], k_fn=lambda n: K)
    A = DevicePairs(Tpr.indptr, Tpr.ent, Tpr.shape, perm, K)
    lens_t = Ttpr.indptr[1:] - Ttpr.indptr[:-1]
    perm_t, _inv, _ = be.packed_layout(lens_t, k_fn=lambda n: K)
    At = DevicePairs(Ttpr.indptr, Ttpr.ent, Ttpr.shape, perm_t, K)
    assert torch.equal(be.spmm(A, Q), Yn) and torch.equal(be.spmm(At, Yn), Zn)
    print(f"-- K = {K}: {perm.numel() // (64 * K)} / {perm_t.numel() // (64 * K)} workgroups")
    bench(f"X*Q  pairs K={K}         ", A, Q)
    bench(f"Xt*Y pairs K={K}         ", At, Yn)
    if K in (4, 6, 8):
        for mode in ((1, 64, 32, 96) if K == 6 else (1, 9, 64)):
            be.tune("spmm_mode", mode)
            bench(f"X*Q  pairs K={K} mode {mode:2d} ", A, Q)
            if mode & 64:
                t = be.spmm(A, Q)
                nw = perm.numel() // (64 * K) * 16
                tt = t.reshape(-1)[: nw * 64].reshape(nw, 64)[:, :5].double()
                passes = (args.peaks + 255) // 256 * K
                names = ["wait window", "stage A", "stage B", "barrier+dma wait", "dma issue"]
                print("   cycles per pass and wave: " + ", ".join(f"{n} {float(tt[:, i].mean()) / passes:7.1f}" for i, n in enumerate(names))
                      + f"; sum {float(tt.sum(dim=1).mean()) / passes:7.1f}")
        be.tune("spmm_mode", 0)

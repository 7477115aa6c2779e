#!/usr/bin/env python
"""The row-stream SpMM (mu_spmm_stream_f32) in both directions on the bench matrix: launch times,
the cost of building the operands, timing ablations and the per-wave cycle accounting of the kernel
(spmm_mode 64).  Used bare or under rocprofv3 (--kernel-trace --stats, or --pmc in a separate pass).
(r02a-j compared it with the r01 packed kernel, bit for bit: gpurun_out/r02h.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import DeviceStream, HipBackend

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=125000)
ap.add_argument("--peaks", type=int, default=200000)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--modes", default="1,9,32,64,96")
ap.add_argument("--no-packed", action="store_true")
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--ks", default="", help="also time layouts dealt for these K (row-sets per wave)")
ap.add_argument("--k-modes", default="", help="ablation modes to time on the --ks layouts as well")
ap.add_argument("--calibrate", action="store_true",
                help="run a 4 GiB device copy first (known HBM byte count for PMC calibration)")
args = ap.parse_args()

be = HipBackend(0)
if args.calibrate:
    src = torch.empty(1 << 30, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)  # reads 4 GiB, writes 4 GiB (far beyond the 256 MiB Infinity Cache)
    torch.cuda.synchronize()
    del src, dst
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


Ts, ms = timed(lambda: be.stream(T))
print(f"stream(X):           {ms:8.3f} ms", flush=True)
Tts, ms = timed(lambda: be.transpose_stream(T))
print(f"transpose_stream(X): {ms:8.3f} ms", flush=True)
B = args.B
Q = be.randn(args.peaks, B, 1)
Yn = be.spmm(Ts, Q)
Zn = be.spmm(Tts, Yn)
torch.cuda.synchronize()
if args.cells <= 500000:  # (a third 8 B/nnz copy: not next to the 1e6-row operands)
    Y0 = be.spmm(be.stream(T, sort_rows=False), Q)
    print("layout vs natural order:", "bit-identical" if torch.equal(Y0, Yn) else "DIFFERS", flush=True)
    del Y0

def bench(name, M, D):
    _, ms = timed(lambda: be.spmm(M, D), args.reps)
    n, d = M.shape
    byt = 8 * M.nnz + 8 * (n + 1) + 4 * B * (n + d)
    print(f"{name}: {ms:8.3f} ms  {byt / ms / 1e6:6.0f} GB/s alg ({byt / ms / 8e7:5.1f} % of 8 TB/s)  {M.nnz / ms / 1e6:6.1f} Gnnz/s", flush=True)


bench("X*Q  stream      ", Ts, Q)
bench("Xt*Y stream      ", Tts, Yn)
names = ["wait window", "stage A", "stage B", "barrier+dma wait", "dma issue"]
for M, D, tag in ((Ts, Q, "X*Q "), (Tts, Yn, "Xt*Y")):
    if M.k != 8 and not all(int(m) in (4096, 16384, 262144) for m in args.modes.split(",") if m):
        continue
    for mode in [int(m) for m in args.modes.split(",") if m]:
        be.tune("spmm_mode", mode)
        bench(f"{tag} stream mode {mode:2d}", M, D)
        if mode & 64:
            t = be.spmm(M, D)
            nw = M.n_pos // (64 * M.k) * 16
            tt = t.reshape(-1)[: nw * B].reshape(nw, B)[:, :5].double()
            passes = (M.shape[1] + 255) // 256 * M.k
            print("   cycles per pass and wave: " + ", ".join(f"{n} {float(tt[:, i].mean()) / passes:7.1f}" for i, n in enumerate(names))
                  + f"; sum {float(tt.sum(dim=1).mean()) / passes:7.1f}", flush=True)
    be.tune("spmm_mode", 0)

for K in [int(k) for k in args.ks.split(",") if k]:
    for mode in [0] + [int(m) for m in args.k_modes.split(",") if m]:
        be.tune("spmm_mode", mode)
        A = be.stream(T, K=K)
        bench(f"X*Q  stream K={K} mode {mode} ({A.n_pos // (64 * K)} workgroups)", A, Q)
        if mode:
            print("   vs mode 0:", "bit-identical" if torch.equal(be.spmm(A, Q), Yn) else "DIFFERS", flush=True)
        del A
        At = be.transpose_stream(T, K=K)
        bench(f"Xt*Y stream K={K} mode {mode} ({At.n_pos // (64 * K)} workgroups)", At, Yn)
        if mode:
            print("   vs mode 0:", "bit-identical" if torch.equal(be.spmm(At, Yn), Zn) else "DIFFERS", flush=True)
        del At
    be.tune("spmm_mode", 0)

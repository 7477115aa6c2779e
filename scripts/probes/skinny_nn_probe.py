#!/usr/bin/env python
"""MOFA Z-update product A = Y (tau o W) on the dense view (100k x 20k by 20k x K, f32): hipBLASLt
through torch.matmul for K = 10 / 16 / 32 against mu_skinny_nn."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)
N, D = 100000, 20000
Y = torch.randn(N, D, device="cuda", dtype=torch.float32)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for K in (10, 16, 32):
    W = torch.randn(D, K, device="cuda", dtype=torch.float32)
    ms = timed(lambda: Y @ W)
    print(f"torch.matmul K={K}: {ms:.3f} ms  {4 * N * D / ms / 1e6:.0f} GB/s", flush=True)
    Wt = torch.randn(K, D, device="cuda", dtype=torch.float32)
    ms = timed(lambda: Y @ Wt.T)
    print(f"torch.matmul K={K} (W stored K x D): {ms:.3f} ms  {4 * N * D / ms / 1e6:.0f} GB/s", flush=True)
W16 = torch.randn(D, 16, device="cuda", dtype=torch.float32)
ms = timed(lambda: be.skinny_nn(Y, W16))
print(f"mu_skinny_nn K=16: {ms:.3f} ms  {4 * N * D / ms / 1e6:.0f} GB/s", flush=True)
Z16 = torch.randn(N, 16, device="cuda", dtype=torch.float32)
ms = timed(lambda: be.skinny_tn(Y, Z16))
print(f"mu_skinny_tn K=16: {ms:.3f} ms  {4 * N * D / ms / 1e6:.0f} GB/s", flush=True)

# f64 (mofapy2's default precision): rocBLAS with either operand layout against mu_skinny_nn
del Y
Yd = torch.randn(N, D, device="cuda", dtype=torch.float64)
for K in (10, 16):
    W = torch.randn(D, K, device="cuda", dtype=torch.float64)
    Wt = torch.randn(K, D, device="cuda", dtype=torch.float64)
    ms = timed(lambda: Yd @ W, reps=3)
    print(f"f64 torch.matmul K={K}: {ms:.3f} ms  {8 * N * D / ms / 1e6:.0f} GB/s", flush=True)
    ms = timed(lambda: Yd @ Wt.T, reps=3)
    print(f"f64 torch.matmul K={K} (W stored K x D): {ms:.3f} ms  {8 * N * D / ms / 1e6:.0f} GB/s", flush=True)
W16 = torch.randn(D, 16, device="cuda", dtype=torch.float64)
ms = timed(lambda: be.skinny_nn(Yd, W16))
print(f"f64 mu_skinny_nn K=16: {ms:.3f} ms  {8 * N * D / ms / 1e6:.0f} GB/s", flush=True)
# (r02: 2 / 4 tiles of 16 rows per wave, so that an element of the D x 16 operand feeds 2 / 4 matrix
#  instructions: 5.8 / 8.1 ms against 4.4 - the tiles' LDS halves / quarters the waves per CU)
Z16 = torch.randn(N, 16, device="cuda", dtype=torch.float64)
ms = timed(lambda: be.skinny_tn(Yd, Z16))
print(f"f64 mu_skinny_tn K=16: {ms:.3f} ms  {8 * N * D / ms / 1e6:.0f} GB/s", flush=True)

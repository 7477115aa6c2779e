#!/usr/bin/env python
"""Times mu_gram_f32 / mu_gram_cross_f32 / mu_dense_apply_f32 on tall 64-column blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)


def timeit(f, reps=20):
    f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for wg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]:
    be.tune("gram_wg", wg)
    for n in (125000, 200000, 1000000):
        A, Bm = be.randn(n, 64, 1), be.randn(n, 64, 2)
        M = torch.eye(64, device="cuda")
        ref = (A.double().T @ Bm.double())
        C = be.gram_cross(A, Bm)
        err = float((C - ref).abs().max() / ref.abs().max())
        print(f"gram_wg={wg} n={n}: gram {timeit(lambda: be.gram(A)):7.1f} us  cross {timeit(lambda: be.gram_cross(A, Bm)):7.1f} us  "
              f"apply {timeit(lambda: be.apply(A, M)):7.1f} us   (cross rel err {err:.1e}; ideal stream: gram {n * 256 / 5e6:.0f} us, cross {n * 512 / 5e6:.0f} us)")
be.tune("gram_wg", 0)

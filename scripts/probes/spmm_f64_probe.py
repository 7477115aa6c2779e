"""Row-stream SpMM with f64 blocks (mu_spmm_stream_f64, B = 16 / 32) against the f32 B = 64 product at 1e6 x 200 000 / 125k shard.
usage: python scripts/probes/spmm_f64_probe.py [cells]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muon_amd._backend import HipBackend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
d = 200_000
be = HipBackend(0)
X = be.synth_counts(0, n, d, 50, 0.03, 0)
X = X.with_values(torch.log1p(X.values.to(torch.float32)))
Xs, Xts = be.stream_both(X)
print(f"{n} x {d}, {X.nnz} entries; stream K {Xs.k} / {Xts.k}", flush=True)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for name, S, rows in (("X Q", Xs, d), ("X^T Y", Xts, n)):
    for dt, B in ((torch.float32, 64), (torch.float32, 32), (torch.float64, 32), (torch.float64, 16)):
        Q = torch.randn((rows, B), device="cuda", dtype=dt)
        try:
            ms = timed(lambda: be.spmm(S, Q))
            print(f"{name}: {dt} B={B}: {ms:.1f} ms ({ms / B:.2f} ms per column)", flush=True)
        except Exception as e:
            print(f"{name}: {dt} B={B}: {e}")
        del Q

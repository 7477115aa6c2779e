"""Dense sweep of a poisson view: matrix-core kernel (k_pois_mfma) against the vector kernel (k_pois_dense), f32.
usage: python scripts/probes/pois_mfma_probe.py [N] [D] [K]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from muon_amd._backend import get_backend, check, _p, _dt  # noqa: E402

be = get_backend()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
KP = next(k for k in (4, 8, 12, 16, 32) if k >= K)
g = torch.Generator(device="cuda").manual_seed(1)
Z = torch.zeros((N, KP), device="cuda"); Z[:, :K] = torch.randn((N, K), generator=g, device="cuda") * 0.7
W = torch.zeros((D, KP), device="cuda"); W[:, :K] = torch.randn((D, K), generator=g, device="cuda") * 0.5
kap_d = torch.rand((D,), generator=g, device="cuda") + 0.25


def dense(mode, Eo, Et, kap):
    n_own, n_other = Eo.shape[0], Et.shape[0]
    blk = int(be.lib.mu_mofa_poisson_blocks_for(_dt(Eo), mode, K, n_own, n_other))
    nb = -(-n_other // blk)
    part = torch.empty((nb, n_own) if mode == 2 else (nb, n_own, K + 1 if mode == 3 else K), device="cuda")
    check(be.lib.mu_mofa_poisson_dense(_dt(Eo), mode, n_own, n_other, K, blk, _p(Eo), _p(Et), _p(kap), _p(part), be._stream()))
    return part.sum(dim=0)


def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


cases = [(0, Z, W, kap_d), (1, W, Z, kap_d), (2, Z, W, None), (3, W, Z, kap_d)]
for mode, Eo, Et, kap in cases:
    out = {}
    for valu in (1, 0):
        be.lib.mu_tune_set(b"pois_valu", valu)
        out[valu] = dense(mode, Eo, Et, kap)
        ms = timed(lambda: dense(mode, Eo, Et, kap))
        print(f"mode {mode} {'valu' if valu else 'mfma'}: {ms:.3f} ms (with the fold of the partials)", flush=True)
    a, b = out[1].double(), out[0].double()
    print(f"   max |mfma - valu| / max |valu| = {float((a - b).abs().max() / a.abs().max()):.2e}")
be.lib.mu_tune_set(b"pois_valu", 0)

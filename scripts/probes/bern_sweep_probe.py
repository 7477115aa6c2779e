"""The Jaakkola sweep of a bernoulli view (csrc/mofa_bernoulli.hip) against numpy f64, and its time at 20 000 x 20 000.
usage: python scripts/probes/bern_sweep_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from muon_amd._backend import get_backend  # noqa: E402

be = get_backend()


def ref(Eo, Eo2, Et, Et2):
    zeta = Eo @ Et.T
    xi2 = np.maximum(zeta ** 2 + Eo2 @ Et2.T - (Eo ** 2) @ (Et ** 2).T, 0)
    xi = np.maximum(np.sqrt(xi2), 1e-8)
    Om = np.tanh(0.5 * xi) / (2 * xi)
    P = Et[:, :, None] * Et[:, None, :]
    i = np.arange(Et.shape[1])
    P[:, i, i] = Et2
    return np.einsum("ot,tkl->okl", Om, P)


for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
    for K in (1, 3, 5, 8, 10, 12, 13, 16):
        for n_own, n_other in ((1, 1), (37, 150), (300, 517)):
            rng = np.random.default_rng(K * 7 + n_own)
            Eo = rng.standard_normal((n_own, K)) * 0.8
            Et = rng.standard_normal((n_other, K)) * 0.6
            Eo2 = Eo ** 2 + rng.random((n_own, K)) * 0.3
            Et2 = Et ** 2 + rng.random((n_other, K)) * 0.2
            dev = lambda a: torch.from_numpy(a).to(be.device).to(dt)
            got = be.to_host(be.mofa_jaakkola_sweep(dev(Eo), dev(Eo2), dev(Et), dev(Et2))).astype(np.float64)
            npdt = np.float32 if dt == torch.float32 else np.float64
            c = lambda a: a.astype(npdt).astype(np.float64)
            want = ref(c(Eo), c(Eo2), c(Et), c(Et2))
            err = np.max(np.abs(got - want)) / np.max(np.abs(want))
            flag = "" if err <= tol else "   <-- FAIL"
            if flag or (n_own, n_other) == (300, 517):
                print(f"{dt} K={K} {n_own}x{n_other}: max rel err {err:.2e}{flag}", flush=True)
N = D = 20000
for dt in (torch.float32, torch.float64):
    for K in (10, 16):
        g = torch.Generator(device="cuda").manual_seed(1)
        Z = (torch.randn((N, K), generator=g, device="cuda") * 0.7).to(dt)
        W = (torch.randn((D, K), generator=g, device="cuda") * 0.5).to(dt)
        Z2, W2 = Z * Z + 0.1, W * W + 0.05
        f = lambda: be.mofa_jaakkola_sweep(W, W2, Z, Z2)
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        print(f"{dt} K={K}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per sweep at {N} x {D} (pack + sweep + fold + expand)")

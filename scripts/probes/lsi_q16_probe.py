#!/usr/bin/env python
"""16-bit dense operands in the LATE products of the block Lanczos iteration, measured on the GPU
(VERDICT r02 item 2 iv).  The arithmetic stays f32: the dense operand of every product from product number
p0 on is rounded to f16 / bf16 (MUON_AMD_LSI_Q16, muon_amd/_atac/tools.py) - exactly what a 16-bit Q slab
in LDS would feed the FMAs.  Reported: the largest principal angle between the top-50 right singular
subspace of that run and of the plain f32 run (itself within ~1e-6 rad of f64 ARPACK on the shapes the
tests cover), the number of products the stopping rule took, and the bound it reported.

    python scripts/probes/lsi_q16_probe.py [cells] [peaks]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import get_backend


def angle(V0, V1):
    q0, _ = torch.linalg.qr(V0.double())
    q1, _ = torch.linalg.qr(V1.double())
    s = torch.linalg.svdvals(q0.T @ q1).clamp(max=1.0)
    # sin of the largest angle from the residual (acos is blind below 1e-8)
    r = q1 - q0 @ (q0.T @ q1)
    return float(torch.linalg.matrix_norm(r, ord=2)), float(s.min())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    be = get_backend()
    X = tfidf_device(be, be.synth_counts(0, n, d, 50, 0.03, 0), n, 3, 1e4)
    os.environ.pop("MUON_AMD_LSI_Q16", None)
    _, s0, V0, info0 = lsi_device(be, X, 50, return_info=True)
    print(f"{n} x {d}: f32 run: {info0['spmm']} products, bound {info0['angle_bound']:.1e}", flush=True)
    last = info0["spmm"]
    for name in ("f16", "bf16"):
        for p0 in (last - 2, last - 4, 0):  # the last expansion (X^T Y and X Q), the last two, every product
            os.environ["MUON_AMD_LSI_Q16"] = f"{name}:{p0}"
            _, s1, V1, info = lsi_device(be, X, 50, return_info=True)
            a, _ = angle(V0, V1)
            ds = float(abs(s1 - s0).max() / s0.max())
            print(f"{name} from product {p0:2d}: angle to the f32 subspace {a:.2e} rad, stdev max rel diff {ds:.1e}, "
                  f"{info['spmm']} products, converged {info['converged']}, bound {info['angle_bound']:.1e}", flush=True)
    os.environ.pop("MUON_AMD_LSI_Q16", None)


if __name__ == "__main__":
    main()

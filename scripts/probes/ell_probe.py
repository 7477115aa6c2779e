"""Sliced-ELL narrow SpMM (csrc/spmm_ell.hip) against the row-stream kernel (csrc/spmm_narrow.hip) at MOFA c4's
sparse view: atac 100k x 100k, ~3e8 stored entries, block of 16 f32 columns; both directions.
usage: python scripts/probes/ell_probe.py [n_cells] [n_features]"""
import sys
import time

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muon_amd._backend import get_backend, ell16_layout  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    be = get_backend()
    X = be.synth_counts(0, n, d, 50, 0.03, 0)
    X = type(X)(X.indptr, X.indices, torch.log1p(X.values.to(torch.float32)) + 0.25, X.shape)
    print(f"X {X.shape} nnz {X.nnz}", flush=True)
    for name, M in (("X", X), ("Xt", be.transpose(X))):
        rows, cols = M.shape
        Q = torch.randn(cols, 16, device=be.device, dtype=torch.float32)
        S = be.stream(M)
        ref = be.spmm(S, Q)
        t_stream = timed(lambda: be.spmm(S, Q))
        print(f"[{name}] row stream: {t_stream:.3f} ms   ({M.nnz * 8 / t_stream / 1e6:.0f} GB/s of 8-byte entries)", flush=True)
        del S
        t0 = time.perf_counter()
        E = be.ell16(M)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        out = be.spmm(E, Q)
        err = float((out - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(out, be.spmm(E, Q)))
        line = (f"[{name}] plan waves {E.waves}; slots/nnz {E.slots / M.nnz:.3f} build {t_build:.2f}s err {err:.2e} "
                f"reproducible {same}")
        n_waves = -(-rows // 16)
        for w, ragged in ((15, 0), (14, 0), (13, 0), (12, 0)):
            E.waves = w
            ts = []
            for mode in (0, 1, 2, 3):
                be.tune("ell_mode", mode | ragged)
                ts.append(timed(lambda: be.spmm(E, Q)))
            be.tune("ell_mode", 0)
            line += (f"\n      waves {w:2d} : {ts[0]:.3f} ms ({E.slots * 6 / ts[0] / 1e6:.0f} GB/s)"
                     f" | no gathers {ts[1]:.3f} | no slab copies {ts[2]:.3f} | neither {ts[3]:.3f}")
        print(line, flush=True)
        del E, out
        # f64 blocks (512-column slabs)
        Q64 = Q.to(torch.float64)
        E = be.ell16(M, wide=True)
        ref64 = be.spmm(E, Q64)
        err = float((ref64 - ref.to(torch.float64)).abs().max() / ref.abs().max())
        line = f"[{name}] f64 blocks: slots/nnz {E.slots / M.nnz:.3f} waves {E.waves} err vs f32 {err:.2e}"
        for w in (15, 14, 12):
            E.waves = w
            ts = []
            for mode in (0, 1, 2, 3):
                be.tune("ell_mode", mode)
                ts.append(timed(lambda: be.spmm(E, Q64)))
            be.tune("ell_mode", 0)
            line += (f"\n      waves {w:2d}: {ts[0]:.3f} ms ({E.slots * 6 / ts[0] / 1e6:.0f} GB/s)"
                     f" | no gathers {ts[1]:.3f} | no slab copies {ts[2]:.3f} | neither {ts[3]:.3f}")
        print(line, flush=True)
        del Q, ref, E, Q64, ref64


if __name__ == "__main__":
    main()

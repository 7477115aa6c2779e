#!/usr/bin/env python
"""B = 16 products of MOFA's sparse view (BASELINE configs[3]: 100 000 x 100 000 at 3 %): the narrow-block
kernel (csrc/spmm_narrow.hip) against the NB = 1 instance of the B = 64 kernel (tune spmm_narrow_off),
both directions, same operands; results compared, launches timed with HIP events.

    python scripts/probes/spmm_narrow_probe.py [n_rows] [n_cols]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import get_backend


REPS = int(os.environ.get("NARROW_REPS", "20"))


def timed(fn, reps=None):
    reps = reps or REPS
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    be = get_backend()
    X = tfidf_device(be, be.synth_counts(0, n, d, 50, 0.03, 0), n, 3, 1e4)
    P, Pt = be.stream(X), be.transpose_stream(X)
    print(f"{n} x {d}, nnz {X.nnz}, layout K = {P.k} / {Pt.k}", flush=True)
    for name, S, rows in (("X Q", P, d), ("X^T Y", Pt, n)):
        Q = be.randn(rows, 16, 3)
        be.tune("spmm_narrow_off", 1)
        ref = be.spmm(S, Q).clone()
        t_old = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_narrow_off", 0)
        new = be.spmm(S, Q).clone()
        again = be.spmm(S, Q).clone()
        t_new = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_mode", 2)  # timing ablation: half windows (results wrong on purpose)
        t_half = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_mode", 0)
        be.tune("spmm_mode", 3)  # A/B: the LDS staging row instead of lane swaps (same results up to summation order)
        t_swap = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_mode", 4)  # ablation: no gathers / FMAs
        t_nog = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_mode", 6)  # cycle accounting: per wave sums of s_memtime differences replace the product
        acct = be.spmm(S, Q)
        be.tune("spmm_mode", 0)
        nw = 16 * ((S.n_pos + 64 * S.k - 1) // (64 * S.k))
        a = acct[:nw, :5].double()
        tot = a[:, 4].mean().item()
        print(f"{name:6s}: accounting (share of a wave's time): window wait + count {a[:, 0].mean().item() / tot:.2f}, "
              f"mask + spread + request {a[:, 1].mean().item() / tot:.2f}, gathers + FMAs {a[:, 2].mean().item() / tot:.2f}, "
              f"slab barrier {a[:, 3].mean().item() / tot:.2f}; ticks per wave {tot:.0f}", flush=True)
        be.tune("spmm_mode", 5)  # ablation: every request = the next 32 pairs of one sequential stream per wave
        t_seq = timed(lambda: be.spmm(S, Q))
        be.tune("spmm_mode", 0)
        print(f"{name:6s}: sequential 32-pair requests per wave (ablation) {t_seq:.3f} ms", flush=True)
        print(f"{name:6s}: narrow kernel with 32-entry requests (ablation) {t_half:.3f} ms, through an LDS staging row {t_swap:.3f} ms, "
              f"without gathers / FMAs (ablation) {t_nog:.3f} ms", flush=True)
        scale = ref.abs().max().item()
        err = (new - ref).abs().max().item() / scale
        gb = (8.0 * X.nnz + 4 * 16 * (n + d)) / 1e9
        print(f"{name:6s}: NB=1 instance {t_old:.3f} ms ({gb / t_old * 1e3:.0f} GB/s)  narrow {t_new:.3f} ms "
              f"({gb / t_new * 1e3:.0f} GB/s)  max |diff| / max |ref| = {err:.2e}  reproducible = {bool(torch.equal(new, again))}",
              flush=True)


if __name__ == "__main__":
    main()

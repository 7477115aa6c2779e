#!/usr/bin/env python
"""Where the set-up of a MOFA sparse view goes at the c4 shape (100 000 x 100 000, 3.1e8 entries): the pieces of
`transpose_csr` + 2 x `ell16` timed with a synchronisation after each (warm: second round)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._backend import get_backend
from muon_amd._atac.preproc import tfidf_device

be = get_backend()
X = be.synth_counts(0, 100000, 100000, 50, 0.03, 0)
X = tfidf_device(be, X, 100000, 3, 1e4)
X = type(X)(X.indptr, X.indices, X.values.to(torch.float32), X.shape)


def timed(label, fn, out):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    out.append((label, (time.perf_counter() - t) * 1e3))
    return r


for rnd in range(2):
    t = []
    Xt = timed("transpose_csr", lambda: be.transpose_csr(X), t)
    for name, M in (("X", X), ("Xt", Xt)):
        for cols in (1024, 512):
            sp = timed(f"{name} slab_ptr_width({cols})", lambda: be.slab_ptr_width(M, cols), t)
            del sp
        lens = M.indptr[1:] - M.indptr[:-1]
        timed(f"{name} argsort", lambda: torch.argsort(lens, descending=True, stable=True), t)
        timed(f"{name} ell16 f32", lambda: be.ell16(M), t)
        timed(f"{name} ell16 wide", lambda: be.ell16(M, wide=True), t)
    if rnd:
        for k, v in t:
            print(f"{k:32s} {v:8.2f} ms")

#!/usr/bin/env python
"""The sliced-ELL product of a MOFA shard (12 500 x 100 000 and its transpose, B = 16) with and without column parts (r06)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend, _p, check

be = HipBackend(0)
for n in (12500, 25000, 50000):
    X = tfidf_device(be, be.synth_counts(0, n, 100000, 50, 0.03, 0), n, 3, 1e4)
    E, Et = be.ell16_pair(X)
    for name, op, rows, cols in (("X W", E, n, 100000), ("X^T Z", Et, 100000, n)):
        Q = torch.randn((cols, 16), device="cuda")
        wv, parts = C.c_int(0), C.c_int(0)
        be.lib.mu_spmm_ell16_parts(rows, cols, 0, C.byref(wv), C.byref(parts))
        res = {}
        for label, (w, p) in (("plain", (op.waves, 1)), ("auto", (wv.value, parts.value)), ("15 x 4", (15, 4)), ("15 x 24", (15, min(24, -(-cols // 1024) // 2 or 1)))):
            slabs = -(-cols // 1024)
            p = max(1, min(p, slabs))
            ny = -(-slabs // -(-slabs // p))
            part = torch.empty((ny, rows, 16), device="cuda")
            def run():
                check(be.lib.mu_spmm_ell16_parts_f32(w, p, int(op.perm.numel()), cols, _p(op.hdr), _p(op.wave_base), _p(op.ent),
                                                     _p(op.perm), _p(Q), _p(part), rows * 16, None))
                return part.sum(dim=0) if ny > 1 else part[0]
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                y = run()
            torch.cuda.synchronize()
            res[label] = (1e3 * (time.perf_counter() - t) / 20, w, p, y)
        ref = res["plain"][3]
        print(f"n={n:6d} {name:6s} " + "  ".join(f"{k}: {v[0]:.3f} ms (waves {v[1]}, parts {v[2]}, maxdiff {float((v[3] - ref).abs().max()):.1e})" for k, v in res.items()), flush=True)

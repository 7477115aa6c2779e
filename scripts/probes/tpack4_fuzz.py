#!/usr/bin/env python
"""Randomised comparison of the fourth-generation fill's two window schemes (and, for small shapes, scipy): random shapes,
densities, bursts of consecutive columns, empty rows / column ranges, forced tile widths.  Bit-identical streams or it
stops.  Usage: tpack4_fuzz.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import torch

from muon_amd._backend import get_backend

be = get_backend()
be.keep_tpack4_work = True
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def make(n, d, dens):
    m = sp.random(n, d, density=dens, format="lil", random_state=rng, dtype=np.float32)
    for _ in range(int(rng.integers(0, 12))):
        r = int(rng.integers(0, n))
        L = int(rng.integers(20, min(d, 700) + 1)) if d > 20 else d
        c0 = int(rng.integers(0, d - L + 1))
        m[r, c0:c0 + L] = rng.random(L).astype(np.float32) + 0.25
    if n > 40 and rng.random() < 0.5:  # a run of neighbouring rows with the same burst (one wave)
        r0 = int(rng.integers(0, n - 34))
        L = int(rng.integers(33, min(d, 200) + 1)) if d > 33 else d
        c0 = int(rng.integers(0, d - L + 1))
        for r in range(r0, r0 + int(rng.integers(2, 34))):
            m[r, c0:c0 + L] = 1.5
    m = m.tocsr()
    if d > 50 and rng.random() < 0.5:  # an empty column range
        a = int(rng.integers(0, d - 10))
        keep = np.ones(d, dtype=np.float32)
        keep[a:a + int(rng.integers(1, d - a))] = 0
        m = sp.csr_matrix(m @ sp.diags(keep))
    if n > 10 and rng.random() < 0.5:  # empty rows
        keep = np.ones(n, dtype=np.float32)
        keep[::int(rng.integers(2, 9))] = 0
        m = sp.csr_matrix(sp.diags(keep) @ m)
    m.eliminate_zeros()
    m.sort_indices()
    return m.astype(np.float32)


def stream_src(X):
    Xs = be.stream(X)
    inv = torch.empty(X.shape[0], dtype=torch.int64, device=Xs.sptr.device)
    perm = Xs.perm.long()
    ok = perm >= 0
    inv[perm[ok]] = torch.nonzero(ok).reshape(-1)
    return Xs, Xs.sptr[inv].contiguous()


done = 0
for it in range(cases):
    n = int(rng.choice([1, 3, 17, 100, 513, 2000, 9000, 40000, 120000]))
    d = int(rng.choice([1, 5, 33, 200, 1025, 5000, 30000, 200000]))
    dens = float(rng.choice([0.3, 0.05, 0.01, 0.002]))
    if n * d * dens > 3e7:
        dens = 3e7 / (n * d)
    m = make(n, d, dens)
    if m.nnz == 0:
        continue
    X = be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float32)
    if not be._use_tpack4(X):
        continue
    src = stream_src(X)
    C = int(rng.choice([0, 0, 16, 48, 160, 512]))
    outs = []
    try:
        be.tune("tpack4_c", C)
        for mode in (2, 0):
            be.tune("tpack4_circ", mode)
            R = be.transpose_stream(X, sort_rows=False, src=src)
            assert be.tpack4_status() == 0, (it, n, d, dens, C, mode, be.tpack4_status())
            outs.append((R.sptr.clone(), R.ent[: m.nnz].clone()))
    finally:
        be.tune("tpack4_c", 0)
        be.tune("tpack4_circ", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (it, n, d, dens, C)
    if m.nnz < 3e6:  # ... and the transpose itself
        mt = m.T.tocsr()
        mt.sort_indices()
        ent = be.to_host(outs[1][1]).view(np.uint64)
        cols = (ent & np.uint64(0xffffffff)).astype(np.int64)
        vals = (ent >> np.uint64(32)).astype(np.uint32).view(np.float32)
        assert np.array_equal(cols, mt.indices) and np.array_equal(vals.view(np.uint32), mt.data.view(np.uint32)), (it, n, d)
        assert np.array_equal(be.to_host(outs[1][0]), mt.indptr), (it, n, d)
    done += 1
print(f"{done} random cases: circular and plain windows write the same streams (and scipy's transpose where it was computed)")

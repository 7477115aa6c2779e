#!/usr/bin/env python
"""Timing ablations of the TF-IDF sweeps (tune key "tfidf_abl"; wrong results on purpose):
1 sum sweep without the LDS atomics, 2 with f32 atomics, 3 scale sweep without arithmetic, 4 without stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
nnz = X.nnz
print(f"{cells} x 200000, nnz {nnz}")


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


rs, cs = be.row_col_sums(X)
idf = be.idf(cs, cells, 3, torch.float32)
out = torch.empty_like(X.values)
for abl in (0, 1, 2, 0):
    be.tune("tfidf_abl", abl)
    t = timeit(lambda: be.row_col_sums(X))
    print(f"abl {abl}: sum sweep (+ pointers) {t:.2f} ms")
for abl in (0, 3, 4, 0):
    be.tune("tfidf_abl", abl)
    be.row_col_sums(X)
    kept = be.__dict__.get("_sweep_work")

    def scale_only():
        be._sweep_work = kept
        be.tfidf_scale(X, rs, idf, 1e4, 3, out=out)

    t = timeit(scale_only)
    print(f"abl {abl}: scale sweep alone {t:.2f} ms")
be.tune("tfidf_abl", 0)
rs0, cs0 = be.row_col_sums(X)
for m in (0, 2, 0, 2):
    be.tune("tfidf_sum_m", m)
    rs1, cs1 = be.row_col_sums(X)
    t = timeit(lambda: be.row_col_sums(X))
    print(f"tfidf_sum_m {m}: sum sweep (+ pointers) {t:.2f} ms, identical {bool(torch.equal(rs1, rs0))} {bool(torch.equal(cs1, cs0))}")
be.tune("tfidf_sum_m", 0)
for m in (1, 0, 1, 0):
    be.tune("tfidf_sum_m", m)
    rs1, cs1 = be.row_col_sums(X)
    t = timeit(lambda: be.row_col_sums(X))
    print(f"tfidf_sum_m {m} (1: 8192-column bins + k_slab_ptr, 0: 16384-column bins, pointers made in the sweep): {t:.2f} ms, identical {bool(torch.equal(rs1, rs0))} {bool(torch.equal(cs1, cs0))}")
be.tune("tfidf_sum_m", 0)

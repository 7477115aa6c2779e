#!/usr/bin/env python
"""Times the TF-IDF passes on the bench matrix: sum sweep, scale (slab sweep vs per-lane gather)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 125000
uns = "--unstructured" in sys.argv  # uniform random pattern, values 1 + Poisson(0.5)
X = be.synth_counts(0, cells, 200000, 0 if uns else 50, 0.03, 0)
print(f"{'unstructured' if uns else 'planted-topic'} {cells} x 200000, nnz {X.nnz}")
nnz = X.nnz


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
t = timeit(lambda: be.row_col_sums(X))
print(f"sum sweep (with slab pointers): {t:.2f} ms  {8 * nnz / t / 1e9:.2f} TB/s")
be._scale_gather = True
t = timeit(lambda: be.tfidf_scale(X, rs, idf, 1e4, 3, out=out))
print(f"scale, per-lane idf gather:     {t:.2f} ms  {12 * nnz / t / 1e9:.2f} TB/s")
be._scale_gather = False
t = timeit(lambda: be.tfidf_scale(X, rs, idf, 1e4, 3, out=out))
print(f"scale, slab sweep (+ pointers): {t:.2f} ms  {12 * nnz / t / 1e9:.2f} TB/s")


def both():
    be.row_col_sums(X)
    be.tfidf_scale(X, rs, idf, 1e4, 3, out=out)


t = timeit(both)
print(f"sum sweep + scale sweep (pointers shared): {t:.2f} ms  {20 * nnz / t / 1e9:.2f} TB/s")

# r04: the scale sweep's idf slabs (tune key "tfidf_wide": 1 = the 8192-column kernel, else 4 x 8192 columns)
be.tune("tfidf_wide", 1)
be.row_col_sums(X)
ref, _ = be.tfidf_scale(X, rs, idf, 1e4, 3)
for wide in (1, 0):
    be.tune("tfidf_wide", wide)
    be.row_col_sums(X)
    got, _ = be.tfidf_scale(X, rs, idf, 1e4, 3)
    same = bool(torch.equal(got, ref))
    be.row_col_sums(X)
    kept = be.__dict__.get("_sweep_work")

    def scale_only():
        be._sweep_work = kept
        be.tfidf_scale(X, rs, idf, 1e4, 3, out=out)

    t = timeit(scale_only) if kept is not None else float("nan")
    t2 = timeit(both)
    print(f"tfidf_wide {wide}: scale sweep alone {t:.2f} ms ({12 * nnz / t / 1e9:.2f} TB/s), sum + scale {t2:.2f} ms, bit-identical {same}")
be.tune("tfidf_wide", 0)

# r04: software-pipelined walks (tune key "tfidf_pipe": 1 = the kernels of before)
be.tune("tfidf_pipe", 1)
rs_ref, cs_ref = be.row_col_sums(X)
val_ref, _ = be.tfidf_scale(X, rs_ref, idf, 1e4, 3)
for pipe in (1, 0, 1, 0):
    be.tune("tfidf_pipe", pipe)
    rs2, cs2 = be.row_col_sums(X)
    got, _ = be.tfidf_scale(X, rs2, idf, 1e4, 3)
    same = (bool(torch.equal(rs2, rs_ref)), bool(torch.equal(cs2, cs_ref)), bool(torch.equal(got, val_ref)))
    t1 = timeit(lambda: be.row_col_sums(X))
    t2 = timeit(both)
    print(f"tfidf_pipe {pipe}: sum sweep (+ pointers) {t1:.2f} ms ({8 * nnz / t1 / 1e9:.2f} TB/s), sum + scale {t2:.2f} ms "
          f"({20 * nnz / t2 / 1e9:.2f} TB/s), row sums / column sums / values bit-identical {same}")
be.tune("tfidf_pipe", 0)

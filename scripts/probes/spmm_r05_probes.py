#!/usr/bin/env python
"""r05, the two gated SpMM probes of VERDICT r04 item 2, measured as far as a measurement can decide them:
 (b) B = 32 x 2 passes: the B = 32 instance of k_spmm_win (256-column slabs) against the B = 64 one on the same operand -
     two passes have to come in under 0.9 of one B = 64 pass;
 (a) padded e-steps: the slot statistics of the launch as it is (four rows in lock step through a slab) against what
     pairing row-sets (a group moves on to its next row inside one pass) and what no padding at all would need.
Usage: spmm_r05_probes.py [cells]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
peaks = 200000
X = be.synth_counts(0, cells, peaks, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)
Xs, Xt = be.stream_both(T)
print(f"{cells} x {peaks}, {T.nnz} stored entries; K = {Xs.k} (X), {Xt.k} (X^T)", flush=True)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, S, rows in (("X Q  ", Xs, peaks), ("X^T Y", Xt, cells)):
    t = {}
    for B in (64, 32):
        Q = torch.randn((rows, B), dtype=torch.float32, device="cuda")
        out = torch.empty((S.shape[0], B), dtype=torch.float32, device="cuda")
        t[B] = timed(lambda: be.spmm(S, Q, out=out))
    print(f"(b) {name}: B = 64 {t[64]:.3f} ms, B = 32 {t[32]:.3f} ms -> two B = 32 passes = {2 * t[32] / t[64]:.2f} x one B = 64 pass "
          f"(gate: <= 0.90 with 512-column slabs, which halve the {S.shape[1] // 256} slab visits of a pass)", flush=True)

# (a) e-steps of the launch as it is: per (row-set of 4 positions, 256-column slab) the window costs
# ceil-to-batch(max over its 4 rows of the entries in the slab); pairing row-sets k, k + 1 of a wave: max over the 4
# groups of the SUM of the two rows' entries
lens = Xs.sptr[1:] - Xs.sptr[:-1]
perm = Xs.perm.long()
n_pos = perm.numel()
S = -(-peaks // 256)
rows_of_pos = torch.where(perm >= 0, perm, torch.zeros_like(perm))
rid = torch.repeat_interleave(torch.arange(T.shape[0], device="cuda"), T.indptr[1:] - T.indptr[:-1])
key = rid * S + (T.indices.long() >> 8)
cnt = torch.bincount(key, minlength=T.shape[0] * S).view(T.shape[0], S)  # entries per (row, slab)
cp = cnt[rows_of_pos] * (perm >= 0)[:, None]                             # per position
del key, rid
sets = cp.view(n_pos // 4, 4, S)
mx = sets.amax(dim=1)                                                     # e-steps a set needs in a slab (unrounded)
used = float(cp.sum())
lock = float(mx.sum()) * 4


def batched(m):  # the kernel gates its gathers in batches: 4 / 8 slots, then pairs up to 12, then 16
    r = torch.where(m <= 4, torch.full_like(m, 4), torch.where(m <= 8, torch.full_like(m, 8), ((m + 1) // 2) * 2))
    r = torch.where(m == 0, torch.zeros_like(m), r)
    return torch.where(m > 12, ((m + 15) // 16) * 16, r)


gated = float(batched(mx).sum()) * 4
K = Xs.k
w = sets.view(-1, K, 4, S)  # (wave, row-set, group, slab)
if K % 2 == 0:
    pair = (w[:, 0::2] + w[:, 1::2]).amax(dim=2)  # max over the groups of the sum of the two rows
    paired = float(pair.sum()) * 4
else:
    paired = float("nan")
allk = w.sum(dim=1).amax(dim=1)  # a group walks ALL its K rows of the slab back to back
print(f"(a) X Q: stored entries {used:.4g}; gather slots with 4 rows in lock step {lock:.4g} (slot use {used / lock:.3f}), "
      f"as gated by the kernel {gated:.4g} ({used / gated:.3f}); row-sets paired {paired:.4g} ({used / paired:.3f}); "
      f"a group through all its {K} rows per slab {float(allk.sum()) * 4:.4g} ({used / (float(allk.sum()) * 4):.3f})", flush=True)

#!/usr/bin/env python
"""Is the row-stream SpMM limited by the chip's power budget (MI355X_MICROARCH.md "DVFS give-back")?
Same launch, same instruction stream, three operand fills: random Q (production), Q = 0 (the gathered
bytes and the FMA operands do not toggle), Q = 0 and all stored values 0.  A power-limited kernel runs
faster on the quiet operands; a latency- or pipe-limited one does not care."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells, peaks = int(os.environ.get("CELLS", 125000)), 200000
X = be.synth_counts(0, cells, peaks, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)
S = be.stream(T)
Q = be.randn(peaks, 64, 1)


def t(M, D, reps=6):
    be.spmm(M, D)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        be.spmm(M, D)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print(f"X*Q {cells} x {peaks}: random Q {t(S, Q):.3f} ms", flush=True)
print(f"                      Q = 0    {t(S, torch.zeros_like(Q)):.3f} ms", flush=True)
ent = S.ent.view(torch.int32).view(-1, 2)
ent[:, 1] = 0  # value bits of every pair
print(f"            Q = 0, values = 0  {t(S, torch.zeros_like(Q)):.3f} ms", flush=True)
print(f"       random Q, values = 0    {t(S, Q):.3f} ms", flush=True)

#!/usr/bin/env python
"""The streaming copy of X (mu_csr_stream_fill) with the pipelined asm loop against the loop of before (tune
stream_pipe = 1): alone, and next to the transposition's fill (backend.stream_both, what lsi runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)


def timeit(f, reps=3):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


del X
be.tune("stream_pipe", 1)
ref = be.stream(T)
for mode in (1, 0, 1, 0):
    be.tune("stream_pipe", mode)
    P = be.stream(T)
    torch.cuda.synchronize()
    same = bool(torch.equal(P.ent, ref.ent))
    del P
    torch.cuda.empty_cache()
    t1 = timeit(lambda: be.stream(T))
    torch.cuda.empty_cache()
    t2 = timeit(lambda: be.stream_both(T))
    torch.cuda.empty_cache()
    print(f"stream_pipe {mode}: copy alone {t1:.2f} ms, copy + transposition (stream_both) {t2:.2f} ms, same bytes {same}", flush=True)
be.tune("stream_pipe", 0)

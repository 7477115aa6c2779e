#!/usr/bin/env python
"""Why one step in five of the emulated rank-of-8 loop took 121 instead of 44 ms (r06): per-step wall time with the
cyclic garbage collector's runs logged (gc.callbacks) and, second pass, with the collector frozen."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend
from scripts.bench_rank8 import EightAlike

be = HipBackend(0)
X = be.synth_counts(0, 125000, 200000, 50, 0.03, 0)
comm = EightAlike()
out = torch.empty_like(X.values)
events = []


def cb(phase, info):
    if phase == "start":
        events.append([info["generation"], time.perf_counter(), None])
    else:
        events[-1][2] = time.perf_counter()


gc.callbacks.append(cb)


def step():
    T = tfidf_device(be, X, 1000000, 3, 1e4, comm=comm, out=out)
    lsi_device(be, T, n_comps=50, n_obs=1000000, comm=comm, return_info=True)


for mode in ("gc on", "gc frozen"):
    if mode == "gc frozen":
        gc.collect()
        gc.freeze()
        gc.disable()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    per = []
    for i in range(14):
        del events[:]
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t)
        per.append(round(ms, 1))
        if ms > 60:
            print(f"  step {i}: {ms:.1f} ms, gc runs: {[(g, round(1e3 * (b - a), 1)) for g, a, b in events]}")
    print(mode, per, flush=True)

#!/usr/bin/env python
"""What ONE rank of an 8-GPU c3 run does per lsi call, on one GPU: the 125 000-cell shard with n_obs = 1e6 and a
stand-in communicator that behaves like eight identical ranks (sums x 8).  The subspace it converges to is the shard's
own (the timing of the phases is the point, not the answer): warm start with the global slice floor against the cold
start."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend
from muon_amd._comm import LocalComm


class EightAlike(LocalComm):
    world_size = 1  # (no collectives are issued; the sums are scaled as eight identical ranks' would be)

    def all_reduce_sum(self, *tensors):
        for t in tensors:
            t *= 8
        return tensors[0] if len(tensors) == 1 else tensors

    def all_reduce_sum_big(self, t):
        t *= 8
        return t

    def sum_scalar(self, x):
        return 8 * x


be = HipBackend(0)
X = be.synth_counts(0, 125000, 200000, 50, 0.03, 0)
comm = EightAlike()
T = tfidf_device(be, X, 1000000, 3, 1e4, comm=comm)
for spec in ("0", "32:2", "0", "32:2", "8:2", "16:2", "8:2", "16:2"):  # (8:2 = the 16 384-row slice of the per-rank floor)
    os.environ["MUON_AMD_LSI_WARM"] = spec
    torch.cuda.synchronize()
    t = time.perf_counter()
    U, sd, V, info = lsi_device(be, T, n_comps=50, n_obs=1000000, comm=comm, return_info=True)
    torch.cuda.synchronize()
    print(f"warm {spec:5s}: {1e3 * (time.perf_counter() - t):7.2f} ms, products {info['spmm']}, warm_start {info['warm_start']}, "
          f"bounds {info.get('lanczos_bounds')}", flush=True)

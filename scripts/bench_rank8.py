#!/usr/bin/env python
"""What ONE rank of an 8-GPU run does per step, emulated on one MI355X (VERDICT r05 item 1 c): the shard of
BASELINE.json configs[2] (125 000 of 1e6 cells x 200 000 peaks: tfidf + lsi) and of configs[4] (12 500 of 100 000 cells:
one MOFA iteration), with a stand-in communicator that behaves like eight IDENTICAL ranks - every sum over the ranks is
the local value x 8, nothing crosses a link.  What the figures say: the per-rank compute (kernels, launches, host steps)
that bounds the 8-GPU speed-up BEFORE communication; what they do not: the collectives' time (four 51 MB all-reduces of a
d x 64 block and a dozen small ones per tfidf + lsi step; one 5 MB all-reduce per MOFA iteration).  Conservative in two
ways: the warm start's slice holds one rank's distinct cells (eight real ranks' slices together hold eight times as many),
and the replicated d x 64 work (orthonormalisation, projection) is NOT divided by eight here.

`bench.py` carries both as `secondary.c3_rank8` / `secondary.c5_rank8`; no scaling curve is claimed from them.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from muon_amd._comm import LocalComm


class EightAlike(LocalComm):
    """Eight identical ranks: sums are the local value x 8 (one small kernel per tensor where a collective would be)."""

    world_size = 8
    rank = 0

    def all_reduce_sum(self, *tensors):
        for t in tensors:
            t *= 8
        return tensors[0] if len(tensors) == 1 else tensors

    def all_reduce_sum_big(self, t):
        t *= 8
        return t

    def sum_scalar(self, x):
        return 8 * x


def run_c3_rank8(be, steps=10, warmup=3):
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device

    n_local, n_global, d = 125_000, 1_000_000, 200_000
    comm = EightAlike()
    X = be.synth_counts(0, n_local, d, 50, 0.03, 0)
    out_vals = torch.empty_like(X.values)
    info = {}
    t_tfidf = [0.0]

    def step():
        t0 = time.perf_counter()
        T = tfidf_device(be, X, n_global, 3, 1e4, comm=comm, out=out_vals)
        if info.get("_split"):
            torch.cuda.synchronize()
            t_tfidf[0] += time.perf_counter() - t0
        U, sd, V, inf = lsi_device(be, T, n_comps=50, n_obs=n_global, comm=comm, return_info=True)
        info.update(inf)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    ms0 = torch.cuda.memory_stats(dev)
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per.append(1e3 * (time.perf_counter() - ts))
    ms = 1e3 * (time.perf_counter() - t0) / steps
    ms1 = torch.cuda.memory_stats(dev)
    # (a hipMalloc / hipFree inside a step synchronises the device and costs tens of ms: the counters say whether one did)
    alloc = {"device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
             "device_frees": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0))}
    info["_split"] = True  # (one more step with a synchronisation between the two calls: the split, not the figure)
    step()
    torch.cuda.synchronize()
    return {"metric": "ms per tfidf + lsi(k=50) step of ONE rank of eight (emulated on one GPU, no communication)",
            "value": ms, "unit": "ms", "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
            "ms_per_step": ms, "median_ms": float(np.median(per)), "tfidf_ms": 1e3 * t_tfidf[0],
            "steps_ms": [round(v, 2) for v in per], "allocator": alloc,
            "config": {"workload": f"c3_rank8: the {n_local}-cell shard of configs[2] ({n_global} x {d}, {X.nnz} stored entries on "
                                   f"this rank), n_obs = {n_global}, a stand-in communicator of eight identical ranks",
                       "lsi_spmm_per_step": int(info.get("spmm", 0)), "lsi_converged": bool(info.get("converged")),
                       "warm_start": info.get("warm_start")},
            "note": "per-rank compute only: 8-GPU speed-up before communication = (1-GPU c3 ms per step) / this; the "
                    "warm start's slice holds one rank's distinct cells and the replicated d x 64 work is not divided by 8"}


def run_c5_rank8(be, iters=100, warmup=3, f64=False):
    import importlib.util

    from muon_amd._core.mofa_engine import MofaEngine

    spec = importlib.util.spec_from_file_location("bench_mofa", os.path.join(ROOT, "scripts", "bench_mofa.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    comm = EightAlike()
    n_local, n_total = 12_500, 100_000
    rna, atac = bm.make_views(be, 0, n_local, n_total, 20_000, 100_000, 0, comm)
    T = torch.float64 if f64 else torch.float32
    eng = MofaEngine(be, [rna, atac], np.zeros(n_local, dtype=np.int64), 10, dtype=T, seed=1, comm=comm, row_offset=0,
                     n_total=n_total)
    for _ in range(warmup):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    e = np.asarray(eng.elbo)
    return {"metric": "ms per MOFA ELBO iteration of ONE rank of eight (emulated on one GPU, no communication)",
            "value": ms, "unit": "ms", "higher_is_better": False, "n_gpus": 1, "dtype": "f64" if f64 else "f32",
            "data": "synthetic", "ms_per_iteration": ms, "graph": bool(eng._graph is not None),
            "segments": bool(getattr(eng, "_seg_graphs", None) is not None),
            "elbo_monotone": bool(np.all(np.diff(e) > -1e-6 * abs(e[0]))),
            "config": {"workload": f"c5_rank8: the {n_local}-cell shard of configs[4] (rna {n_local} x 20000 dense + atac "
                                   f"{n_local} x 100000 sparse, {atac.nnz} stored entries), K = 10, a stand-in communicator of "
                                   "eight identical ranks (statistics x 8 where the all-reduce would be)"},
            "note": "per-rank compute only; configs[4]'s one packed all-reduce per iteration (~5 MB) is not in it"}


RUNNERS = {"c3_rank8": run_c3_rank8, "c5_rank8": run_c5_rank8}

if __name__ == "__main__":
    import json

    from muon_amd._backend import HipBackend

    be = HipBackend(0)
    for name in (sys.argv[1:] or list(RUNNERS)):
        print(json.dumps({name: RUNNERS[name](be)}))

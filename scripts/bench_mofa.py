#!/usr/bin/env python
"""MOFA+ timing (BASELINE.json configs[3]/[4]): two-view synthetic MuData-like input - rna N x 20k
dense + atac N x 100k sparse (TF-IDF'd planted counts) - K = 10 factors, a fixed number of ELBO
iterations, cells sharded over the ranks (strong scaling) with RCCL all-reduce of the expectation
sufficient statistics.  Prints ONE JSON line on rank 0 in the schema of bench.py.

    python scripts/bench_mofa.py [--iters 100] [--f64]
    python -m torch.distributed.run --nproc-per-node N ... scripts/bench_mofa.py --gpus N
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--cells", type=int, default=100000)
ap.add_argument("--rna", type=int, default=20000)
ap.add_argument("--atac", type=int, default=100000)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--f64", action="store_true", help="float64 like the reference default (use_float32=False)")
args = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
shared_gpu = os.environ.get("MUON_AMD_BENCH_SHARED_GPU") == "1"  # test hook: all ranks on GPU 0 over gloo
if shared_gpu:
    local_rank = 0
torch.cuda.set_device(local_rank)
comm = None
if world > 1:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if shared_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from muon_amd._comm import TorchDistComm

    comm = TorchDistComm()

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend
from muon_amd._core.mofa_engine import MofaEngine

be = HipBackend(local_rank)
T = torch.float64 if args.f64 else torch.float32
row0 = rank * args.cells // world
N = (rank + 1) * args.cells // world - row0
K0 = 10
g = torch.Generator(device="cuda").manual_seed(0)
W = torch.randn((args.rna, K0), generator=g, device="cuda") * (torch.rand((args.rna, K0), generator=g, device="cuda") < 0.3)
gz = torch.Generator(device="cuda").manual_seed(1 + rank)
Z = torch.randn((N, K0), generator=gz, device="cuda", dtype=torch.float32)
rna = Z @ W.T
rna += torch.randn(rna.shape, generator=gz, device="cuda")
X = be.synth_counts(row0, N, args.atac, 50, 0.03, 0)
atac = tfidf_device(be, X, args.cells, 3, 1e4, comm=comm)
eng = MofaEngine(be, [rna, atac], np.zeros(N, dtype=np.int64), 10, dtype=T, seed=1, comm=comm,
                 row_offset=row0, n_total=args.cells)
for _ in range(args.warmup):
    eng.step()


def sync():
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


sync()
t0 = time.perf_counter()
for _ in range(args.iters):
    eng.step()
sync()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
if rank == 0:
    vb = 8 if args.f64 else 4
    dense_b = vb * args.cells * args.rna
    sparse_b = int(atac.nnz * world) * (4 + vb)  # approx.: rank 0's nnz x world
    alg = 2 * (dense_b + sparse_b)  # two passes over every view per iteration
    per = dt / args.iters
    mono = bool(np.all(np.diff(eng.elbo) > -1e-5 * abs(eng.elbo[0])))
    print(json.dumps({
        "metric": "seconds per 100 ELBO iterations of mu.tl.mofa (10 factors)",
        "value": 100 * per, "unit": "s", "n_gpus": world, "steps": args.iters, "warmup": args.warmup,
        "ms_per_step": per * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if args.f64 else "f32", "data": "synthetic",
        "config": {"workload": f"c4: rna {args.cells} x {args.rna} dense + atac {args.cells} x {args.atac} sparse "
                               f"({atac.nnz} nnz on rank 0), K=10, gaussian likelihoods, {args.iters} iterations",
                   "parallelism": f"cells row-sharded x{world}" if world > 1 else "1 GPU"},
        "roofline": {"kernel": "whole iteration (two passes over every view: A = Y (tau o W), B = Y^T Z)",
                     "bound": "hbm", "achieved": alg / per / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / per / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_iteration": alg},
        "elbo": {"first": eng.elbo[0], "last": eng.elbo[-1], "monotone": mono},
    }))
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()

#!/usr/bin/env python
"""MOFA+ timing (BASELINE.json configs[3]/[4]): two-view synthetic MuData-like input - rna N x 20k
dense + atac N x 100k sparse (TF-IDF'd planted counts) - K = 10 factors, a fixed number of ELBO
iterations, cells sharded over the ranks (strong scaling) with RCCL all-reduce of the expectation
sufficient statistics.  Prints ONE JSON line on rank 0 in the schema of bench.py.

    python scripts/bench_mofa.py [--iters 100] [--f64]
    python -m torch.distributed.run --nproc-per-node N ... scripts/bench_mofa.py --gpus N

cpu_baseline + parity (rank 0, one GPU): the numpy f64 restatement (oracle/mofa_oracle.py) runs a few
iterations on the first --cpu-sample-cells cells with the REAL feature dimensions; the GPU engine runs
the same iterations on the same sample from the same initialisation (f64 like the oracle, and in the
timed precision), and the line carries `parity`: ELBO trace / <Z> / <W> differences.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_views(be, row0, n_rows, cells_total, n_rna, n_atac, rank=0, comm=None):
    """The c4 generator: rna = Z W^T + N(0, 1) dense f32 (W 30 % dense), atac = TF-IDF of planted-topic
    counts (device CSR).  Deterministic per (row0, n_rows, rank)."""
    from muon_amd._atac.preproc import tfidf_device

    K0 = 10
    g = torch.Generator(device="cuda").manual_seed(0)
    W = torch.randn((n_rna, K0), generator=g, device="cuda") * (torch.rand((n_rna, K0), generator=g, device="cuda") < 0.3)
    gz = torch.Generator(device="cuda").manual_seed(1 + rank)
    Z = torch.randn((n_rows, K0), generator=gz, device="cuda", dtype=torch.float32)
    rna = Z @ W.T
    rna += torch.randn(rna.shape, generator=gz, device="cuda")
    X = be.synth_counts(row0, n_rows, n_atac, 50, 0.03, 0)
    atac = tfidf_device(be, X, cells_total, 3, 1e4, comm=comm)
    return rna, atac


def _fold(stages):
    out = {}
    n = {}
    for k, v in stages:
        n[k] = n.get(k, 0) + 1
        out[k if n[k] == 1 else f"{k} #{n[k]}"] = v
    return out


def sample_views(be, rna, atac, n):
    """First n cells of both views: (dense host f64, densified host f64) for the oracle and
    (device dense, device CSR) for the engine."""
    import scipy.sparse as sp
    from muon_amd._backend import DeviceCSR

    n = min(n, rna.shape[0])
    hi = int(atac.indptr[n].item())
    y1 = be.to_host(rna[:n]).astype(np.float64)
    y2 = sp.csr_matrix((be.to_host(atac.values[:hi]).astype(np.float64), be.to_host(atac.indices[:hi]),
                        be.to_host(atac.indptr[: n + 1])), shape=(n, atac.shape[1])).toarray()
    dev = [rna[:n].contiguous(), DeviceCSR(atac.indptr[: n + 1].contiguous(), atac.indices[:hi].contiguous(),
                                           atac.values[:hi].contiguous(), (n, atac.shape[1]))]
    return n, [y1, y2], dev


def oracle_parity(be, host_views, dev_views, n, iters, dtypes=(torch.float64, torch.float32)):
    """Oracle (numpy f64) and GPU engine on the same sample, same seed, same number of iterations."""
    from muon_amd._core.mofa_engine import MofaEngine
    from oracle import mofa_oracle

    t0, c0 = time.perf_counter(), time.process_time()
    ref = mofa_oracle.run(host_views, groups=np.zeros(n, dtype=np.int64), n_factors=10, n_iterations=iters,
                          convergence_mode="slow", min_iterations=iters + 1)
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    par = {"sample": f"first {n} cells x ({host_views[0].shape[1]} dense + {host_views[1].shape[1]} sparse) features, "
                     f"K=10, {len(ref['elbo'])} iterations, same seed and initialisation",
           "oracle": "oracle/mofa_oracle.py (numpy f64 restatement of the published MOFA+ updates; mofapy2 itself "
                     "is not installable here: parity unpinned)"}
    for dt in dtypes:
        eng = MofaEngine(be, dev_views, np.zeros(n, dtype=np.int64), 10, dtype=dt, seed=1)
        for _ in range(len(ref["elbo"])):
            eng.step()
        res = eng.results(sort_factors=False)
        e, r = np.asarray(res["elbo"]), np.asarray(ref["elbo"])
        key = "f64" if dt == torch.float64 else "f32"
        par[key] = {"elbo_max_rel": float(np.max(np.abs(e - r) / np.abs(r))),
                    "Z_max_abs": float(np.max(np.abs(res["Z"] - ref["Z"]))),
                    "W_max_abs": float(max(np.max(np.abs(a - b)) for a, b in zip(res["W"], ref["W"])))}
        del eng
    return ref, wall, cpu, par


def cpu_baseline(be, rna, atac, n_cells_total, sample_cells, iters):
    """mofapy2 is not installable here, so the CPU baseline is the numpy f64
    restatement of the MOFA+ updates (oracle/mofa_oracle.py: checker-side code) on the first
    `sample_cells` cells with the SAME feature dimensions and K = 10 - the reference's data path
    (tools.py:117-141 densifies the sparse modality) -, timed per iteration and extrapolated linearly
    in the number of cells to the full workload; labelled as such.  The same run is the parity
    reference of the GPU engine on that sample."""
    n, host, dev = sample_views(be, rna, atac, sample_cells)
    ref, wall, cpu, par = oracle_parity(be, host, dev, n, iters)
    done = max(len(ref["elbo"]), 1)
    per_iter_full = wall / done * (n_cells_total / n)
    return {
        "value": 100 * per_iter_full, "unit": "s", "cores": max(1, int(round(cpu / max(wall, 1e-9)))),
        "kind": "port",
        "sample": f"numpy f64 MOFA+ updates (oracle/mofa_oracle.py) on the first {n} cells x ({rna.shape[1]} dense + "
                  f"{atac.shape[1]} densified) features, K=10, {done} iterations in {wall:.1f} s ({wall / done:.2f} s per "
                  f"iteration incl. set-up); extrapolated linearly in cells to {n_cells_total} and to 100 iterations; "
                  f"measured CPU/wall {cpu / max(wall, 1e-9):.1f} (host has {os.cpu_count()} cores)",
    }, par


def run(argv=None, init_dist=True):
    """Runs the workload; returns the JSON object on rank 0 (None elsewhere)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cells", type=int, default=100000)
    ap.add_argument("--rna", type=int, default=20000)
    ap.add_argument("--atac", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--f64", action="store_true", help="float64 like the reference default (use_float32=False)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=1500)
    ap.add_argument("--cpu-sample-iters", type=int, default=4)
    args = ap.parse_args(argv)

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
        if init_dist and not dist.is_initialized():
            if shared_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from muon_amd._comm import TorchDistComm

        comm = TorchDistComm()

    from muon_amd._backend import HipBackend
    from muon_amd._core.mofa_engine import MofaEngine

    be = HipBackend(local_rank)
    for kv in filter(None, os.environ.get("MUON_AMD_BENCH_TUNE", "").split(",")):  # A/B runs: key=value of mu_tune_set
        be.tune(kv.split("=")[0], int(kv.split("=")[1]))
    T = torch.float64 if args.f64 else torch.float32
    row0 = rank * args.cells // world
    N = (rank + 1) * args.cells // world - row0
    rna, atac = make_views(be, row0, N, args.cells, args.rna, args.atac, rank, comm)
    # set-up (moments, centring, transposition, operand layouts: once per fit) timed twice: the first construction in a
    # process also pays for loading every kernel it uses for the first time and for the allocator's first hipMallocs
    t_setups = []
    for _ in range(2):
        eng = None
        torch.cuda.synchronize()
        t_setup = time.perf_counter()
        eng = MofaEngine(be, [rna, atac], np.zeros(N, dtype=np.int64), 10, dtype=T, seed=1, comm=comm,
                         row_offset=row0, n_total=args.cells)
        torch.cuda.synchronize()
        t_setups.append(time.perf_counter() - t_setup)
    t_setup = t_setups[1]
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
    out = None
    if rank == 0:
        vb = 8 if args.f64 else 4
        # bytes as STORED: the f64 fit keeps a dense view whose values are exact in f32 (this generator's, like
        # AnnData's default dtype) in f32 (r04), and its sparse view as f32 values + 16-bit offsets
        dense_vb = eng.views[0].Y.element_size()
        dense_b = dense_vb * args.cells * args.rna
        sparse_b = int(atac.nnz * world) * (4 + vb)  # approx.: rank 0's nnz x world
        alg = 2 * (dense_b + sparse_b)  # two passes over every view per iteration
        per = dt / args.iters
        mono = bool(np.all(np.diff(eng.elbo) > -1e-5 * abs(eng.elbo[0])))
        out = {
            "metric": "seconds per 100 ELBO iterations of mu.tl.mofa (10 factors)",
            "value": 100 * per, "unit": "s", "n_gpus": world, "steps": args.iters, "warmup": args.warmup,
            "ms_per_step": per * 1e3, "setup_ms": t_setup * 1e3, "setup_first_ms": t_setups[0] * 1e3,
            **({"setup_profile_ms": {k: round(v, 2) for k, v in _fold(eng.setup_profile).items()}} if eng.setup_profile else {}),
            "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.f64 else "f32", "data": "synthetic",
            "config": {"workload": f"c4: rna {args.cells} x {args.rna} dense + atac {args.cells} x {args.atac} sparse "
                                   f"({atac.nnz} nnz on rank 0), K=10, gaussian likelihoods, {args.iters} iterations"
                                   + (", dense view stored in f32 (exact), f64 arithmetic" if args.f64 and dense_vb == 4 else ""),
                       "parallelism": f"cells row-sharded x{world}" if world > 1 else "1 GPU"},
            "roofline": {"kernel": "whole iteration (two passes over every view: A = Y (tau o W), B = Y^T Z)",
                         "bound": "hbm", "achieved": alg / per / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg / per / 1e9 / 8000.0, "traffic": None,
                         "algorithmic_bytes_per_iteration": alg},
            "elbo": {"first": eng.elbo[0], "last": eng.elbo[-1], "monotone": mono},
        }
        del eng
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity"] = cpu_baseline(be, rna, atac, args.cells, args.cpu_sample_cells,
                                                              args.cpu_sample_iters)
    return out


def main(argv=None):
    out = run(argv)
    if out is not None:
        print(json.dumps(out))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py - TF-IDF + LSI(k=50) throughput on synthetic planted-topic CSR (BASELINE.json).

A "step" is one pass of the hot path over one batch: tfidf (counts -> TF-IDF values) followed
by lsi (row streams of X and X^T, block Lanczos to convergence, Rayleigh-Ritz over the Krylov space),
with the count matrix already resident in HBM when the timed region starts and U / stdev / V
left in HBM at the end.

roofline: the dominant kernel is the row-stream SpMM (both X*Q and X^T*Y).  `achieved` =
algorithmic bytes per launch (8 B per stored entry + row pointers + the two dense blocks,
SURVEY.md 8d / DESIGN.md 4) / mean launch time from HIP events recorded on the launch stream
inside the timed region.  `traffic` = HBM-side bytes per launch from rocprofv3 PMC counters
(FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction) for the default workload, read
from the newest profiles/r*_spmm_traffic.json - a builder-run measurement, NOT a counter of this
run: the top-level `traffic_source` says so; null for any other shape.  `lds_frac` = the kernel's
own bound: every stored entry gathers a 256-byte Q row from LDS (256 B/clk/CU), so a launch
cannot take less than nnz / (CUs x clock) whatever the HBM does.

parity: the CPU leg's f32 ARPACK result on its sample is kept, the GPU path runs tfidf + lsi on the
same sample, and the largest principal angle between the two top-k right singular subspaces and
the largest relative difference of the singular values go into the JSON line (`parity`).

secondary: the default run (c3, one GPU) also carries the two other single-GPU configurations of
BASELINE.json as complete sub-records (ms, roofline, cpu_baseline, parity): `c2` (configs[0]/[1]) and
`c4` (configs[3], mu.tl.mofa, f32) + `c4_f64` (the same in the reference's default precision), and
`ingest` / `mofa_ng` / `mofa_bern` / `wnn`: one measured record with parity per widened row of SURVEY 8f, `c3_api`: the
API path from a host matrix.
--no-secondary skips them.

Workloads (--workload):
  c3 (default)       BASELINE.json configs[2], the shape the metric is quoted on: 1 000 000 cells x
                     200 000 peaks, 3 % nnz (6.3e9 stored entries), row-sharded over the N GPUs
                     (rank r holds cells [r*1M/N, (r+1)*1M/N)) with RCCL all-reduce of the per-peak
                     sums and of Z = X^T Y.  STRONG scaling: the whole matrix sits on one GPU at
                     N = 1 (~170 GB of the 288 GB).
  c3shard            125 000 cells x 200 000 peaks PER GPU (weak scaling; N = 8 is the same 1M x
                     200k matrix)
  c2                 10 000 x 30 000, 3 % nnz (configs[0]/[1]; launch/latency bound at this size);
                     cpu_baseline = the scipy path on the SAME 10 000 cells
  c4                 configs[3]/[4]: mu.tl.mofa on rna 100k x 20k dense + atac 100k x 100k sparse, K = 10;
                     metric = seconds per 100 ELBO iterations (lower is better), --steps = iterations
                     (default 100), cells sharded over the N GPUs; cpu_baseline = the numpy f64
                     restatement on a cell sample, extrapolated (scripts/bench_mofa.py)

  ingest, mofa_ng,   the widened rows of SURVEY 8f on one GPU (scripts/bench_widened.py): 10x arrays -> device CSR
  wnn                (PCIe included), MOFA+ with a poisson view and missing values, weighted nearest neighbours
  c3_api             ac.pp.tfidf(adata); ac.tl.lsi(adata) through the public API from a HOST scipy CSR (250 000 x
                     200 000, a quarter of configs[2]): upload, fingerprints and write-back inside the clock, split
                     into upload / fingerprint / download / kernels

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either run it under
torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env) or
just call `python bench.py --gpus N`: it then starts its own N ranks (torch.distributed.run on
127.0.0.1 with a free port).  Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
LDS_BYTES_PER_CLK_CU = 256.0  # ds_read_b128, conflict free (same guide, LDS table)
SHADER_CLOCK_HZ = 2.4e9
WORKLOADS = {  # cells = TOTAL cells for strong scaling, cells PER GPU for weak scaling
    "c3": dict(cells=1_000_000, peaks=200_000, scaling="strong"),
    "c3shard": dict(cells=125_000, peaks=200_000, scaling="weak"),
    "c2": dict(cells=10_000, peaks=30_000, scaling="weak"),
    "c4": None,  # BASELINE.json configs[3]/[4]: mu.tl.mofa, 100 ELBO iterations (scripts/bench_mofa.py)
    "ingest": None, "mofa_ng": None, "mofa_bern": None, "wnn": None,  # SURVEY 8f.2 - 8f.4 (scripts/bench_widened.py)
    "c3_api": None,  # tfidf + lsi through the public API from a host scipy CSR (upload, fingerprints, write-back)
    "c3_rank8": None, "c5_rank8": None,  # one rank of eight, emulated on one GPU (scripts/bench_rank8.py)
    "unstructured": None, "hard": None,  # configs[2]'s shape on other spectra (scripts/bench_spectra.py)
}


class TimedBackend:
    """Wraps the backend to time every SpMM launch with HIP events on the launch stream."""

    def __init__(self, be):
        self._be = be
        self.events = []
        self.enabled = False
        self.min_nnz = 0

    def __getattr__(self, name):
        return getattr(self._be, name)

    def spmm(self, X, Q, out=None):
        if not self.enabled:
            return self._be.spmm(X, Q, out=out)
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        r = self._be.spmm(X, Q, out=out)
        e.record()
        B = Q.shape[1]
        n, d = X.shape
        if X.nnz >= self.min_nnz:  # (the products on the WHOLE operands: the warm start's two on a cell slice are not the dominant kernel)
            self.events.append((s, e, 8 * X.nnz + 8 * (n + 1) + 4 * B * (n + d), n, 4.0 * B * X.nnz))
        return r


def cpu_baseline(be, peaks, density, seed, sample_cells, n_comps):
    """The reference's CPU path (scipy tfidf + ARPACK svds, f32 like sc.read_10x_h5 data) on a
    bounded sample of the same generator, and the GPU path's parity with it on that sample.
    Checker-side code: uses oracle/."""
    import scipy.sparse as sp
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device
    from oracle import lsi_oracle, tfidf_oracle

    Xs = be.synth_counts(0, sample_cells, peaks, 50, density, seed)
    m = sp.csr_matrix((be.to_host(Xs.values), be.to_host(Xs.indices), be.to_host(Xs.indptr)), shape=Xs.shape)
    t0 = time.perf_counter()
    tf = tfidf_oracle.tfidf(m)
    t1 = time.perf_counter()
    ref = lsi_oracle.lsi(tf, n_comps=n_comps, dtype=np.float32)
    t2 = time.perf_counter()
    # threads the timed legs could use: scipy's sparse kernels and ARPACK's matvec loop are serial
    # (measured: user+sys CPU time / wall time of the two legs)
    tfidf_oracle.tfidf(m[:2000])
    w0 = time.perf_counter()
    c0 = time.process_time()
    tfidf_oracle.tfidf(m[:2000])
    busy = (time.process_time() - c0) / max(time.perf_counter() - w0, 1e-9)
    # parity of the product path on the same sample (tolerances of BASELINE.json north_star)
    T = tfidf_device(be, Xs, sample_cells, 3, 1e4)
    _U, stdev, V, info = lsi_device(be, T, n_comps=n_comps, n_obs=sample_cells, return_info=True)
    tfc = tfidf_oracle.canonical(tf)
    Th = be.to_host(T.values)
    parity = {
        "sample": f"first {sample_cells} cells of the bench matrix (the cpu_baseline sample)",
        "tfidf_pattern_identical": bool(np.array_equal(tfc.indices, be.to_host(T.indices))
                                        and np.array_equal(tfc.indptr, be.to_host(T.indptr))),
        "tfidf_values_max_rel": float(np.max(np.abs(Th - tfc.data) / np.abs(tfc.data))),
        "lsi_angle_rad": lsi_oracle.max_subspace_angle(be.to_host(V), ref["LSI"]),
        "lsi_stdev_max_rel": float(np.max(np.abs(stdev - ref["stdev"]) / ref["stdev"])),
        "lsi_converged": bool(info["converged"]),
        "lsi_angle_bound": float(info["angle_bound"]),
        "oracle": "scipy svds(k, f32) - the reference's own precision on f32 data; its repeatability is ~1e-5 rad on this spectrum",
    }
    return {
        "value": sample_cells / (t2 - t0),
        "unit": "cells/s",
        "cores": max(1, int(round(busy))),
        "kind": "port",
        "sample": f"{sample_cells} cells x {peaks} peaks ({m.nnz} nnz, same generator, rows 0..{sample_cells - 1}); "
                  f"scipy tfidf {t1 - t0:.2f}s + svds(k={n_comps},f32) {t2 - t1:.2f}s; measured CPU/wall of the scipy "
                  f"leg {busy:.2f} (host has {os.cpu_count()} cores)",
    }, parity


def newest_traffic_profile():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_spmm_traffic.json")))
    return files[-1] if files else None


def run_lsi(args, workload, rank, world, local_rank, comm, steps, warmup, cpu_sample_cells):
    """tfidf + lsi on one workload; returns the JSON object on rank 0 (None elsewhere)."""
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device
    from muon_amd._backend import HipBackend

    wl = dict(WORKLOADS[workload])
    if args.cells and workload == args.workload:
        wl["cells"] = args.cells
    if args.peaks and workload == args.workload:
        wl["peaks"] = args.peaks
    d = wl["peaks"]
    if wl["scaling"] == "strong":
        n_global = wl["cells"]
        row0 = rank * n_global // world
        n_local = (rank + 1) * n_global // world - row0
    else:
        n_local = wl["cells"]
        n_global = n_local * world
        row0 = rank * n_local

    be = TimedBackend(HipBackend(local_rank))
    X = be.synth_counts(row0, n_local, d, 50, args.density, args.seed)
    nnz_local = X.nnz
    be.min_nnz = nnz_local // 2
    tf_vals = torch.empty_like(X.values)
    flags = 3  # log_tf | log_idf (reference defaults)
    info = {}

    def step():
        T = tfidf_device(be, X, n_global, flags, 1e4, comm=comm, out=tf_vals)
        U, stdev, V, inf = lsi_device(be, T, n_comps=args.n_comps, n_obs=n_global, comm=comm,
                                      return_info=True, pack=False if args.no_pack else None)
        info.update(inf)
        return U, stdev, V

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()

    def sync():
        if dist_on:  # (also at world size 1 under MUON_AMD_BENCH_FORCE_DIST: the RCCL dry run)
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    torch.cuda.reset_peak_memory_stats(local_rank)
    be.events.clear()
    be.enabled = True
    sync()
    ms0 = torch.cuda.memory_stats(local_rank)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    be.enabled = False
    ms1 = torch.cuda.memory_stats(local_rank)
    # device allocations inside the timed region: 0 segments / 0 retries = every buffer of a step came
    # from the caching allocator (no hipMalloc / hipFree, which synchronise the device)
    alloc = {"device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
             "device_frees": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
             "alloc_retries": int(ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0)),
             "reserved_gb": round(ms1.get("reserved_bytes.all.peak", 0) / 1e9, 1),
             "allocated_peak_gb": round(ms1.get("allocated_bytes.all.peak", 0) / 1e9, 1)}
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: the row-stream SpMM (both X*Q and X^T*Y run through it)
    ms = [s.elapsed_time(e) for s, e, _, _, _ in be.events]
    byt = [b for _, _, b, _, _ in be.events]
    avg_ms = float(np.mean(ms))
    achieved = float(np.mean(byt)) / (avg_ms * 1e-3) / 1e9
    spmm_total_ms = float(np.sum(ms))
    n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    lds_floor_ms = float(np.mean([g for *_, g in be.events])) / (LDS_BYTES_PER_CLK_CU * n_cus * SHADER_CLOCK_HZ) * 1e3

    traffic = traffic_detail = traffic_source = None
    tpath = newest_traffic_profile()
    tj = None
    if not args.cells and not args.peaks and not args.no_pack and tpath:
        with open(tpath) as f:
            tj = json.load(f).get(f"{n_local}x{d}")  # measured per shape of the rank-0 shard
    if tj:
        # bytes per launch, like algorithmic_bytes_per_launch: the two directions weighted as launched
        n_xq = sum(1 for _, _, _, rows, _ in be.events if rows == n_local)
        n_xt = len(be.events) - n_xq
        traffic = (tj["spmm_xq_bytes_per_launch"] * n_xq + tj["spmm_xty_bytes_per_launch"] * n_xt) / max(len(be.events), 1)
        traffic_source = (f"profiles/{os.path.basename(tpath)}: builder-run rocprofv3 --pmc FETCH_SIZE (x2 gfx950 "
                          "correction, calibrated on a copy) + WRITE_SIZE, separate passes - not a counter of this run")
        traffic_detail = {"x_q": tj["spmm_xq_bytes_per_launch"], "xt_y": tj["spmm_xty_bytes_per_launch"],
                          "vs_algorithmic": traffic / tj["algorithmic_bytes_per_launch"]}

    out = None
    if rank == 0:
        out = {
            "metric": "cells/sec for TF-IDF+LSI(k=50)",
            "value": n_global * steps / dt,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": wl["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{workload}: planted-topic CSR, {n_global} cells x {d} peaks in total, "
                            f"{n_local} cells on rank 0 ({wl['scaling']} scaling), {nnz_local} nnz on rank 0 "
                            f"({nnz_local / n_local / d:.4f} dense), tfidf + lsi(n_comps={args.n_comps})",
                "parallelism": (f"cells row-sharded x{world}; Z collective = "
                                f"{os.environ.get('MUON_AMD_Z_COLLECTIVE', 'allreduce')} (variants: MUON_AMD_Z_COLLECTIVE=rsag "
                                "= reduce-scatter + all-gather, rsqr = SURVEY 8e's form with the CholeskyQR on row slices in "
                                "between); NOTE no 8-GPU node was available to the builder in rounds 1-6: the multi-GPU path "
                                "is tested with gloo and with 2 ranks on one GPU, its scaling is unmeasured") if world > 1 else "1 GPU",
                "allocator": alloc,
                "lsi": {"block": info.get("block"), "iterations": info.get("iterations"),
                        "converged": info.get("converged"), "spmm_per_step": len(ms) // max(steps, 1),
                        "spmm_unused": info.get("spmm_unused"), "warm_start": info.get("warm_start"),
                        "angle_bound": info.get("angle_bound"),
                        "lanczos_bounds": [float(f"{b:.3g}") for b in info.get("bounds", [])]},
            },
            "roofline": {
                "kernel": ("k_spmm_rowwave (CSR SpMM, f32, B=64: one wave per row, Q rows gathered through L2)"
                           if args.no_pack else
                           "k_spmm_win (row-stream SpMM, f32, B=64: Q column slabs in LDS via LDS-DMA, one "
                           "counted-vmcnt window request per row-set and slab, DPP broadcast)"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_peak": achieved / 6300.0,  # the guide's achievable HBM rate
                "lds_floor_ms": lds_floor_ms,  # B x 4 bytes gathered from LDS per stored entry at 256 B/clk/CU
                "lds_frac": lds_floor_ms / avg_ms,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "avg_launch_ms": avg_ms,
                "launches": len(ms),
                "share_of_step": spmm_total_ms / (dt * 1e3),
                "algorithmic_bytes_per_launch": float(np.mean(byt)),
            },
            "traffic_source": traffic_source,
        }
        if not args.no_cpu_baseline and world == 1:  # (the CPU leg and the sample parity ride on the N = 1 line only)
            out["cpu_baseline"], out["parity"] = cpu_baseline(be._be, d, args.density, args.seed,
                                                              min(cpu_sample_cells, n_global), args.n_comps)
    del X, tf_vals, be
    torch.cuda.empty_cache()
    return out


def flatten_for_driver(out, sec):
    """The driver's record of the line keeps SCALAR keys of `config` / `roofline` / `cpu_baseline` and a 2 000-character
    tail of stdout: parity and the figures of the LSI iteration go into `config` as flat scalars, and
    the LAST key of the line is a compact `summary` (headline, parity, one number per secondary record) so that the tail
    shows them whatever the length of the sub-records in front."""
    cfg = out["config"]
    lsi = cfg.get("lsi") or {}
    for k in ("spmm_per_step", "converged", "angle_bound", "iterations", "spmm_unused"):
        if k in lsi:
            cfg["lsi_" + k] = lsi[k]
    par = out.get("parity") or {}
    for k in ("tfidf_pattern_identical", "tfidf_values_max_rel", "lsi_angle_rad", "lsi_stdev_max_rel", "lsi_converged",
              "lsi_angle_bound"):
        if k in par:
            cfg["parity_" + k] = par[k]
    summ = {"ms_per_step": round(out["ms_per_step"], 2), "cells_per_s": round(out["value"]),
            "roofline_frac": round(out["roofline"]["frac"], 4), "spmm_ms": round(out["roofline"]["avg_launch_ms"], 3)}
    for k in ("tfidf_pattern_identical", "tfidf_values_max_rel", "lsi_angle_rad", "lsi_stdev_max_rel"):
        if k in par:
            summ["parity_" + k] = par[k]
    for name, rec in (sec or {}).items():
        rec = rec or {}
        if "error" in rec:
            cfg[name + "_error"] = summ[name + "_error"] = str(rec["error"])[:120]
            continue
        # one scalar per sub-record under `config` (the driver drops nested objects), the same in the summary
        key = name + ("_ms_per_step" if rec.get("unit") == "cells/s" and "ms_per_step" in rec else "_value")
        val = rec.get("ms_per_step") if key.endswith("_ms_per_step") else rec.get("value")
        cfg[key] = val
        summ[key] = val if not isinstance(val, float) else float(f"{val:.5g}")
        rf = (rec.get("roofline") or {}).get("frac")
        if rf is not None:
            cfg[name + "_roofline_frac"] = rf
            summ[name + "_roofline_frac"] = round(rf, 4)
        for pk, pv in (rec.get("parity") or {}).items():
            if isinstance(pv, (bool, int, float)):
                cfg[f"{name}_parity_{pk}"] = pv
        for ck, cv in (rec.get("config") or {}).items():  # (the hard spectrum continued in f64: time, verdict, bound)
            if ck.startswith("f64_continuation_") and isinstance(cv, (bool, int, float)):
                cfg[f"{name}_{ck}"] = cv
                if ck in ("f64_continuation_ms_per_step", "f64_continuation_converged"):
                    summ[f"{name}_{ck}"] = cv if not isinstance(cv, float) else float(f"{cv:.5g}")
    r8 = ((sec or {}).get("c3_rank8") or {}).get("ms_per_step")
    if r8:
        # one rank of eight against the whole matrix on one GPU: the 8-GPU speed-up BEFORE communication (an emulation
        # on one GPU - scripts/bench_rank8.py - not a measured curve)
        cfg["speedup_8gpu_before_comm_emulated"] = summ["speedup_8gpu_before_comm_emulated"] = round(out["ms_per_step"] / r8, 3)
    out.pop("summary", None)
    out["summary"] = summ  # (last key: survives the tail)


def run_c4(args, steps, warmup, f64=False):
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mofa", os.path.join(ROOT, "scripts", "bench_mofa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ["--gpus", str(args.gpus), "--iters", str(steps), "--warmup", str(max(warmup, 3))]
    if args.cells and args.workload == "c4":
        argv += ["--cells", str(args.cells)]
    if args.no_cpu_baseline or f64:
        argv += ["--no-cpu-baseline"]
    if f64:
        argv += ["--f64"]
    return mod.run(argv, init_dist=False)


def run_widened(name):
    """Sub-records of the widened rows: scripts/bench_widened.py; of one rank of eight: scripts/bench_rank8.py."""
    import importlib.util

    from muon_amd._backend import get_backend

    script = "bench_rank8" if name.endswith("_rank8") else ("bench_spectra" if name in ("unstructured", "hard") else "bench_widened")
    spec = importlib.util.spec_from_file_location(script, os.path.join(ROOT, "scripts", script + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.RUNNERS[name](get_backend())


def self_launch(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; c4: 100 iterations)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cells", type=int, default=None, help="override cells (total if strong, per GPU if weak)")
    ap.add_argument("--peaks", type=int, default=None)
    ap.add_argument("--density", type=float, default=0.03)
    ap.add_argument("--n-comps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample-cells", type=int, default=12000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default workload on one GPU: do not append the c2 / c4 sub-records")
    ap.add_argument("--no-pack", action="store_true",
                    help="ablation: run the SpMM on plain CSR (k_spmm_rowwave, one wave per row) instead of the row streams")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test hook (not a measurement mode): all ranks on GPU 0 over gloo, to exercise the multi-rank
    # GPU path on a one-GPU box
    shared_gpu = os.environ.get("MUON_AMD_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = None
    # MUON_AMD_BENCH_FORCE_DIST=1: build the RCCL process group and route every exchange step through it also
    # at world size 1 - the dry run of the multi-GPU branch on a one-GPU box (tests/test_gpu_lsi.py), so that the
    # first real 8-GPU lease does not find a typo here
    force_dist = os.environ.get("MUON_AMD_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1 and "MASTER_PORT" not in os.environ:
            import socket

            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from muon_amd._comm import TorchDistComm

        comm = TorchDistComm()

    if args.workload == "c4":
        out = run_c4(args, args.steps or 100, args.warmup)
    elif args.workload in ("ingest", "mofa_ng", "mofa_bern", "wnn", "c3_api", "c3_rank8", "c5_rank8", "unstructured", "hard"):
        out = run_widened(args.workload) if rank == 0 else None
    else:
        default_line = (args.workload == "c3" and world == 1 and not (args.cells or args.peaks or args.no_pack
                                                                       or args.no_secondary))
        sec = {}
        if default_line:
            # The 10k x 30k record first: that call is host-bound (a host eigensolve and ~330 launches per call) and
            # measures 0.9 ms slower per step after the 1M-cell workload has been through the process (10.6 against 9.7
            # ms on one box, r04 - same code, same steps; the 1M-cell figure does not care about the order).
            try:
                sec["c2"] = run_lsi(args, "c2", rank, world, local_rank, comm, 20, 3, 10_000)
            except Exception as e:  # noqa: BLE001  (the headline line must survive a failing extra)
                sec["c2"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        out = run_lsi(args, args.workload, rank, world, local_rank, comm, args.steps or 3, args.warmup,
                      args.cpu_sample_cells)
        if default_line and out is not None:
            # the other single-GPU configurations of BASELINE.json, as complete sub-records
            try:
                sec["c4"] = run_c4(args, 100, 3)
            except Exception as e:  # noqa: BLE001
                sec["c4"] = {"error": repr(e)}
            try:  # the reference's default precision (use_float32=False, tools.py:308)
                sec["c4_f64"] = run_c4(args, 100, 3, f64=True)
            except Exception as e:  # noqa: BLE001
                sec["c4_f64"] = {"error": repr(e)}
            # the widened rows, the API path and one rank of eight (configs[2] / [4] shards), seconds each
            for name in ("ingest", "mofa_ng", "mofa_bern", "wnn", "c3_api", "c3_rank8", "c5_rank8", "unstructured", "hard"):
                try:
                    torch.cuda.empty_cache()
                    sec[name] = run_widened(name)
                except Exception as e:  # noqa: BLE001
                    sec[name] = {"error": repr(e)}
            out["secondary"] = sec
        if out is not None:
            flatten_for_driver(out, sec if default_line else None)
    if rank == 0 and out is not None:
        print(json.dumps(out, separators=(",", ":")))
    if world > 1 or force_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

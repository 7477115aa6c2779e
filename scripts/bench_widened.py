#!/usr/bin/env python
"""Measured sub-records for the widened rows of SURVEY.md 8f, in the schema of bench.py (one dict each:
metric / value / unit / parity / cpu_baseline), for `bench.py --workload ingest | mofa_ng | wnn` and the
`secondary` block of the default line.  One GPU, synthetic data, bounded to seconds.

  ingest   8f.2  host arrays of a 10x `matrix` group -> device CSR of the cells (peak columns selected on the
                 device) -> tfidf: entries/s including the PCIe upload; parity = the scipy route
                 (tocsr, column slice) entry by entry.
  mofa_ng  8f.3  GeneralMofaEngine: a gaussian dense view + a poisson sparse view (+ element-wise NaN handled by
                 the same engine), seconds per 100 iterations; parity = oracle/mofa_oracle.run_general on a
                 cell sample with the real feature dimensions.
  wnn      8f.4  pp.knn per modality + pp.neighbors (weighted nearest neighbours), cells/s; parity =
                 oracle/wnn_oracle.py (exhaustive numpy loops) on a cell sample.

The oracles are test infrastructure: they run after the timed region, as the checker and the CPU leg.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch


def _sync():
    torch.cuda.synchronize()


def run_ingest(be, n_cells=60000, n_feat=200000, density=0.03, seed=0):
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._core import io as mio

    # a 10x matrix group as the file holds it: CSC over (features x barcodes) = CSR of cells x features,
    # int32 counts, int64 indices / indptr; every fifth feature is not a peak
    X = be.synth_counts(0, n_cells, n_feat, 50, density, seed)
    host = {"data": be.to_host(X.values).astype(np.int32), "indices": be.to_host(X.indices).astype(np.int64),
            "indptr": be.to_host(X.indptr).astype(np.int64), "shape": np.array([n_feat, n_cells])}
    ft = np.array(["Peaks" if j % 5 else "Gene Expression" for j in range(n_feat)])
    del X
    nnz = int(host["indptr"][-1])
    best = None
    for it in range(4):  # the first call pins the staging buffers and starts the copy threads; then the best of three
        D = None
        _sync()
        t0 = time.perf_counter()
        D, keep, _ = mio.device_csr_from_10x(host, None, be, True, ft)
        _sync()
        t1 = time.perf_counter()
        if it > 0 and (best is None or t1 - t0 < best[1] - best[0]):
            best = (t0, t1)
    t0, t1 = best
    ta = time.perf_counter()
    T = tfidf_device(be, D, n_cells, 3, 1e4)
    _sync()
    tfidf_ms = 1e3 * (time.perf_counter() - ta)
    # the reference's route on the host (scanpy's reader ends in the same two scipy calls), timed on a sample
    ns = min(n_cells, 6000)
    hi = int(host["indptr"][ns])
    c0 = time.perf_counter()
    ref = sp.csr_matrix((host["data"][:hi].astype(np.float32), host["indices"][:hi], host["indptr"][: ns + 1]),
                        shape=(ns, n_feat))[:, np.nonzero(ft == "Peaks")[0]]
    c1 = time.perf_counter()
    got = sp.csr_matrix((be.to_host(D.values), be.to_host(D.indices), be.to_host(D.indptr)), shape=D.shape)[:ns]
    same = bool((got != ref).nnz == 0 and np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices))
    in_bytes = host["data"].nbytes + host["indices"].nbytes + host["indptr"].nbytes
    return {"metric": "stored entries/sec, host 10x arrays -> device CSR of the peak columns (PCIe upload included)",
            "value": nnz / (t1 - t0), "unit": "entries/s", "higher_is_better": True, "n_gpus": 1, "data": "synthetic",
            "dtype": "int32 counts -> f32, int64 -> int32 indices", "ms": 1e3 * (t1 - t0), "tfidf_after_ms": tfidf_ms,
            "host_bytes_per_s": in_bytes / (t1 - t0),
            "config": {"workload": f"ingest: {n_cells} cells x {n_feat} features, {nnz} stored entries, 4/5 of the columns are peaks",
                       "kept_columns": int(len(keep)), "device_entries": int(D.nnz)},
            "parity": {"sample": f"first {ns} cells against scipy csr_matrix(...)[:, peaks]", "identical": same},
            "cpu_baseline": {"value": hi / (c1 - c0), "unit": "entries/s", "cores": 1, "kind": "port",
                             "sample": f"scipy csr_matrix + column slice of the first {ns} cells (what scanpy's reader + "
                                       f"`atac_only` do on the host)"},
            # what crosses PCIe: int32 indices + f32 values (the staging threads narrow the host's int64 / int32 arrays
            # on the way) + the row pointers - NOT the host bytes, which r04 put in the numerator (frac 1.02)
            "roofline": {"bound": "pcie", "achieved": (8.0 * nnz + host["indptr"].nbytes) / (t1 - t0) / 1e9, "peak": 64.0,
                         "unit": "GB/s", "frac": (8.0 * nnz + host["indptr"].nbytes) / (t1 - t0) / 64e9,
                         "note": "bytes on the bus (8 B per stored entry + row pointers) over PCIe 5 x16 (64 GB/s per "
                                 "direction nominal, 56 measured); `host_bytes_per_s` is the rate in host bytes"}}


def _ng_views(n, d_dense, d_sparse, seed):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, 5)).astype(np.float32)
    y1 = (Z @ rng.standard_normal((d_dense, 5)).T.astype(np.float32) + rng.standard_normal((n, d_dense))).astype(np.float32)
    rate = np.logaddexp(0, Z @ (0.3 * rng.standard_normal((d_sparse, 5))).T.astype(np.float32) - 3.0)
    y2 = sp.csr_matrix(rng.poisson(rate).astype(np.float32))
    return y1, y2


def run_mofa_ng(be, n=20000, d_dense=2000, d_sparse=20000, iters=20, sample=400, seed=0):
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from oracle import mofa_oracle

    y1, y2 = _ng_views(n, d_dense, d_sparse, seed)
    y1[::97, ::13] = np.nan  # element-wise missing values in the dense view
    lik = ["gaussian", "poisson"]
    eng = GeneralMofaEngine(be, [y1, y2], lik, np.zeros(n, dtype=int), 10, dtype=torch.float32, seed=1)
    for _ in range(2):
        eng.step()
    _sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.step()
    _sync()
    per = (time.perf_counter() - t0) / iters
    e = np.asarray(eng.elbo)
    mono = bool(np.all(np.diff(e) > -1e-5 * abs(e[0])))
    del eng
    # parity + CPU leg: the same model on the first `sample` cells, f64, against the oracle
    s1, s2 = y1[:sample].astype(np.float64), y2[:sample].astype(np.float64)
    k = 4
    c0 = time.perf_counter()
    ref = mofa_oracle.run_general([s1, s2.toarray()], lik, groups=np.zeros(sample, dtype=np.int64), n_factors=10,
                                  n_iterations=k, convergence_mode="slow", min_iterations=k + 1)
    c1 = time.perf_counter()
    g = GeneralMofaEngine(be, [s1, sp.csr_matrix(s2)], lik, np.zeros(sample, dtype=int), 10, dtype=torch.float64, seed=1)
    for _ in range(len(ref["elbo"])):
        g.step()
    res = g.results(sort_factors=False)
    ee, rr = np.asarray(res["elbo"]), np.asarray(ref["elbo"])
    cpu_per = (c1 - c0) / len(ref["elbo"]) * (n / sample)
    return {"metric": "seconds per 100 ELBO iterations, MOFA+ with a poisson view and element-wise missing values (K=10)",
            "value": per * 100, "unit": "s", "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
            "ms_per_iteration": per * 1e3, "elbo_monotone": mono,
            "config": {"workload": f"mofa_ng: {n} cells x ({d_dense} gaussian dense with NaN entries + {d_sparse} poisson sparse, "
                                   f"{y2.nnz} stored counts), K = 10"},
            "parity": {"sample": f"first {sample} cells, real feature dimensions, f64, {len(rr)} iterations, same seed",
                       "oracle": "oracle/mofa_oracle.run_general (numpy restatement of mofapy2's pseudo-data nodes; parity unpinned)",
                       "elbo_max_rel": float(np.max(np.abs(ee - rr) / np.abs(rr))),
                       "Z_max_abs": float(np.max(np.abs(res["Z"] - ref["Z"]))),
                       "W_max_abs": float(max(np.max(np.abs(a - b)) for a, b in zip(res["W"], ref["W"])))},
            "cpu_baseline": {"value": cpu_per * 100, "unit": "s", "cores": 1, "kind": "port",
                             "sample": f"oracle on {sample} cells (densified), {len(rr)} iterations, scaled by cells"},
            # the poisson view's passes as tile work on the matrix cores (prediction zeta = Z W^T, the
            # pseudo-data transform in registers, one reduction against a K-column block - since r06 that IS the kernel,
            # k_pois_mfma): three passes over the N x D predictions of both views' size, 4 N D K flops each, against
            # the f32 matrix-core peak.  An iteration runs two such sweeps (the likelihood shares the W update's), the
            # stored-entry corrections, the gaussian view's statistics and ~50 small kernels.
            "roofline": (lambda fl: {"bound": "mfma", "achieved": fl / per / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                     "frac": fl / per / 157.3e12, "traffic": None,
                                     "algorithmic_flops_per_iteration": fl,
                                     "note": "12 N D K flops per iteration (3 passes x (prediction + one reduction)) against the "
                                             "f32-input MFMA peak of MI355X_MICROARCH.md; the sweeps themselves: 0.29 / 0.32 ms "
                                             "(profiles/r06_mofa_ng_kernel_stats.md)"})(
                12.0 * n * (d_dense + d_sparse) * 10)}


def run_mofa_bern(be, n=20000, d=20000, iters=20, sample=400, seed=0):
    """8f.3, the binary case (what mofapy2 guesses for `ac.pp.binarize`d ATAC data, /root/reference/muon/_core/tools.py:
    272-280): one bernoulli view stored sparse, K = 10 - fitted without anything N x D (csrc/mofa_bernoulli.hip)."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from oracle import mofa_oracle

    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, 5)).astype(np.float32)
    logit = Z @ (0.5 * rng.standard_normal((d, 5))).T.astype(np.float32) - 3.0
    y = sp.csr_matrix((rng.random((n, d)) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32))
    del logit
    lik = ["bernoulli"]
    eng = GeneralMofaEngine(be, [y], lik, np.zeros(n, dtype=int), 10, dtype=torch.float32, seed=1)
    fused = bool(eng.views[0].fusedb)
    for _ in range(2):
        eng.step()
    _sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.step()
    _sync()
    per = (time.perf_counter() - t0) / iters
    e = np.asarray(eng.elbo)
    mono = bool(np.all(np.diff(e) > -1e-5 * abs(e[0])))
    del eng
    s = y[:sample].astype(np.float64)
    k = 4
    c0 = time.perf_counter()
    ref = mofa_oracle.run_general([s.toarray()], lik, groups=np.zeros(sample, dtype=np.int64), n_factors=10,
                                  n_iterations=k, convergence_mode="slow", min_iterations=k + 1)
    c1 = time.perf_counter()
    g = GeneralMofaEngine(be, [sp.csr_matrix(s)], lik, np.zeros(sample, dtype=int), 10, dtype=torch.float64, seed=1)
    for _ in range(len(ref["elbo"])):
        g.step()
    res = g.results(sort_factors=False)
    ee, rr = np.asarray(res["elbo"]), np.asarray(ref["elbo"])
    cpu_per = (c1 - c0) / len(ref["elbo"]) * (n / sample)
    pc = 55  # distinct entries of a 10 x 10 moment block
    fl = 2.0 * (2.0 * n * d * (3 * 10 + pc)) + 2.0 * n * d * 10  # two Jaakkola sweeps + the likelihood sweep
    return {"metric": "seconds per 100 ELBO iterations, MOFA+ with one sparse bernoulli view (K=10)",
            "value": per * 100, "unit": "s", "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
            "ms_per_iteration": per * 1e3, "elbo_monotone": mono, "without_dense_chunks": fused,
            "config": {"workload": f"mofa_bern: {n} cells x {d} binary features, {y.nnz} ones, K = 10"},
            "parity": {"sample": f"first {sample} cells, real feature dimension, f64, {len(rr)} iterations, same seed",
                       "oracle": "oracle/mofa_oracle.run_general (numpy restatement of mofapy2's Jaakkola node; parity unpinned)",
                       "elbo_max_rel": float(np.max(np.abs(ee - rr) / np.abs(rr))),
                       "Z_max_abs": float(np.max(np.abs(res["Z"] - ref["Z"]))),
                       "W_max_abs": float(max(np.max(np.abs(a - b)) for a, b in zip(res["W"], ref["W"])))},
            "cpu_baseline": {"value": cpu_per * 100, "unit": "s", "cores": 1, "kind": "port",
                             "sample": f"oracle on {sample} cells (dense), {len(rr)} iterations, scaled by cells"},
            "roofline": {"bound": "mfma", "achieved": fl / per / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": fl / per / 157.3e12, "traffic": None, "algorithmic_flops_per_iteration": fl,
                         "note": "two precision sweeps (prediction + two variance products + the 55 distinct moment columns) "
                                 "and the likelihood sweep, 2 flops per multiply-add, against the f32-input MFMA peak; r05's "
                                 "dense chunk passes took 16.4 ms per iteration on this model"}}


def run_wnn(be, n=100000, sample=1500, seed=0):
    from muon_amd import AnnData, MuData
    from muon_amd._core import preproc as pp
    from oracle import wnn_oracle

    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 30, n)
    c1, c2 = rng.standard_normal((30, 50)) * 2, rng.standard_normal((30, 30)) * 2
    x1 = (c1[lab] + rng.standard_normal((n, 50))).astype(np.float32)
    x2 = (c2[lab] + rng.standard_normal((n, 30))).astype(np.float32)

    def pipeline(a, b, k_multi):
        md = MuData({"rna": AnnData(a.copy()), "atac": AnnData(b.copy())})
        for m in md.mod.values():
            pp.knn(m, n_neighbors=20, use_rep="X", backend=be)
        pp.neighbors(md, n_multineighbors=k_multi, backend=be)
        return md

    pipeline(x1[:4000], x2[:4000], 200)  # warm-up
    _sync()
    t0 = time.perf_counter()
    md = pipeline(x1, x2, 200)
    _sync()
    dt = time.perf_counter() - t0
    g = md.obsp["distances"]
    per_row = g.nnz // n
    same_cluster = float(np.mean(lab[g.indices] == np.repeat(lab, per_row)))
    # parity + CPU leg on a sample (the oracle is O(n^2) numpy loops)
    s1, s2 = x1[:sample].astype(np.float64), x2[:sample].astype(np.float64)
    ms = pipeline(s1, s2, 100)
    c0 = time.perf_counter()
    graphs = {}
    for name, x in (("rna", s1), ("atac", s2)):
        graphs[name] = wnn_oracle.knn_graph(x, 20)[0]
    D, C, W, _sig, _k = wnn_oracle.neighbors({"rna": s1, "atac": s2}, graphs, n_multineighbors=100)
    c1_ = time.perf_counter()
    got = ms.obsp["distances"]
    eq = got.indices == D.indices
    return {"metric": "cells/sec for knn per modality + weighted nearest neighbours (mu.pp.neighbors)",
            "value": n / dt, "unit": "cells/s", "higher_is_better": True, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
            "ms": dt * 1e3,
            "config": {"workload": f"wnn: {n} cells, two modalities (50 and 30 dimensions, 30 planted clusters), 20 neighbours per "
                                   f"modality, n_multineighbors = 200", "neighbours_per_row": int(per_row),
                       "same_cluster_fraction": same_cluster},
            "parity": {"sample": f"first {sample} cells, n_multineighbors = 100",
                       "oracle": "oracle/wnn_oracle.py (exhaustive search; the reference's NN-descent is approximate: parity unpinned)",
                       "modality_weight_max_abs": float(np.max(np.abs(ms.obs["rna:mod_weight"].values - W[:, 0]))),
                       "graph_identical_fraction": float(np.mean(eq)),
                       "distance_max_rel_where_identical": float(np.max(np.abs(got.data[eq] - D.data[eq]) / np.maximum(D.data[eq], 1e-12))),
                       "connectivities_max_abs": float(abs(ms.obsp["connectivities"] - C).max())},
            "cpu_baseline": {"value": sample / (c1_ - c0), "unit": "cells/s", "cores": 1, "kind": "port",
                             "sample": f"oracle on {sample} cells (quadratic in the cell count: not an extrapolation)"},
            # the exhaustive searches are what the call computes: two per modality (its own kNN graph, the
            # n_multineighbors candidates), 2 n^2 p flops each in GEMM form, f64 - against the f64 matrix-core peak.
            # Top-k merges, the kernel bandwidths and the fuzzy simplicial set ride on top (profiles/r03_wnn_kernel_stats.md).
            "roofline": (lambda fl: {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                     "frac": fl / dt / 78.6e12, "traffic": None, "algorithmic_flops": fl,
                                     "note": "sum over 4 searches of 2 n^2 p flops; MI355X f64 matrix peak 78.6 TFLOP/s"})(
                2.0 * 2.0 * n * n * (50 + 30))}


def run_c3_api(be, n_cells=250_000, n_feat=200_000, density=0.03, seed=0):
    """End to end through the PUBLIC API from a host scipy CSR (VERDICT r03 missing #6; SURVEY 7 hard part 7):
    ``ac.pp.tfidf(adata); ac.tl.lsi(adata)`` (/root/reference/muon/_atac/preproc.py:16-129, tools.py:29-71) with the
    upload, the fingerprints of the resident copies and the write-back inside the clock.  The matrix is a quarter
    of configs[2] (what a host with 64 GB holds next to its copies), same generator, same density."""
    import muon_amd
    from muon_amd import AnnData
    from muon_amd import atac as ac
    from muon_amd._atac import preproc as P

    X = be.synth_counts(0, n_cells, n_feat, 50, density, seed)
    m = sp.csr_matrix((be.to_host(X.values).astype(np.float32), be.to_host(X.indices), be.to_host(X.indptr)),
                      shape=X.shape)
    m.has_sorted_indices = True
    m.has_canonical_format = True
    nnz = m.nnz
    del X
    torch.cuda.empty_cache()
    spans = {"upload": 0.0, "download": 0.0, "fingerprint": 0.0}

    def timed(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                _sync()
                spans[key] += time.perf_counter() - t
        setattr(obj, name, g)
        return f

    best = None
    kept = None
    for it in range(3):  # the first call pins staging buffers and warms the allocator
        for k in spans:
            spans[k] = 0.0
        ad = AnnData(m.copy())
        where = ad.X.data.ctypes.data
        # it == 2: the opt-in takeover of the replaced matrix's host arrays (r06: off by default - the reference never
        # writes into the matrix it replaces and a raw-pointer holder is invisible to reference counts)
        saved = [(be, "upload_csr", timed(be, "upload_csr", "upload")), (be, "to_host", timed(be, "to_host", "download")),
                 (P, "_fingerprint", timed(P, "_fingerprint", "fingerprint"))]
        _sync()
        t0 = time.perf_counter()
        ac.pp.tfidf(ad, backend=be, reuse_host=(it == 2))
        _sync()
        t1 = time.perf_counter()
        ac.tl.lsi(ad, backend=be)
        _sync()
        t2 = time.perf_counter()
        for obj, name, f in saved:
            setattr(obj, name, f)
        if it == 2:
            taken = ad.X.data.ctypes.data == where
            kept = (t1 - t0, t2 - t1)
        else:
            assert ad.X.data.ctypes.data != where  # the default never writes into the replaced matrix
            best = (t1 - t0, t2 - t1, dict(spans))
        del ad
    t_tfidf, t_lsi, sp_ = best
    total = t_tfidf + t_lsi
    return {"metric": "cells/sec through the public API from a host scipy CSR: ac.pp.tfidf(adata); ac.tl.lsi(adata)",
            "value": n_cells / total, "unit": "cells/s", "higher_is_better": True, "n_gpus": 1, "dtype": "f32",
            "data": "synthetic", "ms": total * 1e3,
            "split_ms": {"tfidf_call": t_tfidf * 1e3, "lsi_call": t_lsi * 1e3, "upload_pcie": sp_["upload"] * 1e3,
                         "download_pcie": sp_["download"] * 1e3, "fingerprints_xxh3": sp_["fingerprint"] * 1e3,
                         "kernels_and_host_logic": (total - sum(sp_.values())) * 1e3},
            "host_arrays": "default: fresh arrays for the result (a 6 GB index copy, 12 GB of first-touch page faults), the "
                           "replaced matrix released inside the call - the reference's behaviour (preproc.py:121-127)",
            "reuse_host_opt_in_ms": {"tfidf_call": kept[0] * 1e3, "lsi_call": kept[1] * 1e3, "total": (kept[0] + kept[1]) * 1e3,
                                     "taken_over": bool(taken),
                                     "note": "tfidf(..., reuse_host=True): nothing else referenced the replaced matrix, the result "
                                             "took its index arrays over and was downloaded into its value array "
                                             "(preproc._dies_with_rebinding; opt-in since r06)"},
            "config": {"workload": f"c3_api: {n_cells} cells x {n_feat} peaks ({nnz} stored entries, a quarter of configs[2]) as a host "
                                   f"scipy CSR in an AnnData; tfidf writes adata.X back, lsi finds the device copy "
                                   f"(fingerprint check) and writes obsm / varm / uns"},
            "roofline": {"bound": "pcie", "achieved": (12.0 * nnz + 4.0 * nnz) / total / 1e9, "peak": 64.0, "unit": "GB/s",
                         "frac": (16.0 * nnz) / total / 64e9, "traffic": None,
                         "note": "12 B per entry up (indices, values; the row pointers are noise) + 4 B per entry down (TF-IDF values) over "
                                 "PCIe 5 x16 against the WHOLE call sequence: what the API costs when nothing is resident"}}


RUNNERS = {"ingest": run_ingest, "mofa_ng": run_mofa_ng, "mofa_bern": run_mofa_bern, "wnn": run_wnn, "c3_api": run_c3_api}

if __name__ == "__main__":
    import json

    from muon_amd._backend import get_backend

    be = get_backend()
    for name in (sys.argv[1:] or list(RUNNERS)):
        print(json.dumps({name: RUNNERS[name](be)}), flush=True)

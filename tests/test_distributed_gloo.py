"""The row(cell)-sharded path with world_size 2 on CPU (gloo): two processes each hold half of
the cells, exchange the per-peak sums (tfidf), Z = X^T Y and the Gram (lsi) and the MOFA
statistics through TorchDistComm, and must reproduce the single-process result.  Kernels are
replaced by the CPU test operator set (tests/cpu_backend.py); the collectives, sharding and
host algebra are the product code."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from muon_amd._atac.preproc import canonical_csr, tfidf_device
        from muon_amd._atac.tools import lsi_device
        from muon_amd._comm import TorchDistComm
        from muon_amd._core.mofa_engine import MofaEngine
        from tests.cpu_backend import CpuTestBackend
        from tests.synth import planted_topics_csr

        be = CpuTestBackend()
        comm = TorchDistComm()
        X = planted_topics_csr(600, 400, n_topics=8, density=0.08, seed=5, dtype=np.float32)
        n = X.shape[0]
        lo, hi = (0, 290) if rank == 0 else (290, n)  # uneven shards
        Xs = canonical_csr(X[lo:hi])
        Xd = be.upload_csr(Xs.indptr, Xs.indices, Xs.data, Xs.shape)
        T = tfidf_device(be, Xd, n, 3, 1e4, comm=comm)
        U, stdev, V, info = lsi_device(be, T, n_comps=8, n_obs=n, comm=comm, return_info=True)
        Uall = comm.all_gather_rows(U.contiguous())
        # the Z all-reduce as reduce-scatter + all-gather over row chunks: same sums
        os.environ["MUON_AMD_Z_COLLECTIVE"] = "rsag"
        U2, stdev2, V2, info2 = lsi_device(be, T, n_comps=8, n_obs=n, comm=comm, return_info=True)
        os.environ.pop("MUON_AMD_Z_COLLECTIVE")
        rsag_same = bool(torch.equal(V2, V) and torch.equal(U2, U) and info2["iterations"] == info["iterations"])
        # SURVEY 8e's form (r06): reduce-scatter -> projection + CholeskyQR on row slices -> all-gather.  The sums over the
        # ranks are the same numbers in another grouping: the subspace to 1e-6 rad, the same expansions.
        os.environ["MUON_AMD_Z_COLLECTIVE"] = "rsqr"
        U3, stdev3, V3, info3 = lsi_device(be, T, n_comps=8, n_obs=n, comm=comm, return_info=True)
        os.environ.pop("MUON_AMD_Z_COLLECTIVE")
        from oracle import lsi_oracle as _lo

        rsqr = {"angle": _lo.max_subspace_angle(V3.numpy(), V.numpy()), "iters": info3["iterations"],
                "stdev": float(np.max(np.abs(stdev3 - stdev) / stdev)), "converged": bool(info3["converged"]),
                "replicated": bool(torch.equal(comm.all_gather_rows(V3[:3].contiguous())[:3],
                                               comm.all_gather_rows(V3[:3].contiguous())[3:6]))}
        # f64 arithmetic for f64 input (r06, tools._refine_f64): Y = X Q local, Z = X^T Y summed over the ranks, the d x w
        # blocks replicated; the stop decisions are rank 0's
        U4, stdev4, V4, info4 = lsi_device(be, T, n_comps=8, n_obs=n, comm=comm, return_info=True, refine_f64=True)
        refine = {"dtype": str(V4.dtype), "bound": info4["refine_f64"]["angle_bound"], "blocks": info4["refine_f64"]["blocks"],
                  "V": V4.numpy(), "stdev": stdev4, "U": comm.all_gather_rows(U4.contiguous()).numpy(),
                  "replicated": bool(torch.equal(comm.all_gather_rows(V4[:3].contiguous())[:3],
                                                 comm.all_gather_rows(V4[:3].contiguous())[3:6]))}
        rows_probe = torch.arange(14, dtype=torch.float64).reshape(7, 2) * (rank + 1)  # (7 rows, 2 ranks: chunks of 4 and 3)
        r0, r1 = comm.reduce_scatter_rows(rows_probe)
        own_ok = bool(torch.equal(rows_probe[r0:r1], 3 * torch.arange(14, dtype=torch.float64).reshape(7, 2)[r0:r1]))
        comm.all_gather_rows_into(rows_probe, (r0, r1))
        rsqr["halves"] = own_ok and bool(torch.equal(rows_probe, 3 * torch.arange(14, dtype=torch.float64).reshape(7, 2)))
        probe = torch.arange(10, dtype=torch.float64) + rank  # (a length the world size does not divide)
        comm.all_reduce_sum_big(probe, mode="rsag")
        rsag_same = rsag_same and bool(torch.equal(probe, 2 * torch.arange(10, dtype=torch.float64) + 1))

        # MOFA: samples sharded, two views (one sparse), two groups
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((120, 4))
        y1 = Z @ rng.standard_normal((30, 4)).T + rng.standard_normal((120, 30))
        y2 = Z @ rng.standard_normal((50, 4)).T + rng.standard_normal((120, 50))
        y2[np.abs(y2) < 0.8] = 0
        groups = rng.integers(0, 2, 120)
        a, b = (0, 55) if rank == 0 else (55, 120)
        eng = MofaEngine(be, [y1[a:b], sp.csr_matrix(y2[a:b])], groups[a:b], 6, seed=1, comm=comm,
                         row_offset=a, n_total=120)
        eng.run(12, "slow")
        res = eng.results(sort_factors=False)
        Zall = comm.all_gather_rows(torch.from_numpy(res["Z"]))
        # rank 0's shard of the dense view exact in f32, rank 1's not: the f32-storage / implicit-centring mode must be
        # ONE decision of all ranks (per-rank modes all-reduce centred with uncentred statistics)
        y1x = y1.copy()
        y1x[:55] = y1x[:55].astype(np.float32)
        engx = MofaEngine(be, [y1x[a:b], sp.csr_matrix(y2[a:b])], groups[a:b], 6, seed=1, comm=comm,
                          row_offset=a, n_total=120)
        modes = [bool(getattr(v, "implicit", False)) for v in engx.views]
        engx.run(6, "slow")
        elbo_x = engx.results(sort_factors=False)["elbo"]
        # the element-wise-precision engine (poisson counts, sparse; gaussian view with NaN entries)
        from muon_amd._core.mofa_general import GeneralMofaEngine

        yc = rng.poisson(np.logaddexp(0, Z @ rng.standard_normal((40, 4)).T)).astype(float)
        yn = y1.copy()
        yn[rng.random(yn.shape) < 0.1] = np.nan
        ge = GeneralMofaEngine(be, [yn[a:b], sp.csr_matrix(yc[a:b])], ["gaussian", "poisson"], groups[a:b], 5,
                               seed=1, comm=comm, row_offset=a, n_total=120, chunk_elems=900)
        ge.run(6, "slow", min_iterations=100)
        gres = ge.results(sort_factors=False)
        gZ = comm.all_gather_rows(torch.from_numpy(gres["Z"]))
        if rank == 0:
            q.put({"tfidf": T.values.numpy(), "U": Uall.numpy(), "stdev": stdev, "V": V.numpy(),
                   "elbo": res["elbo"], "Z": Zall.numpy(), "W": res["W"], "iters": info["iterations"],
                   "elbo_x": elbo_x, "modes_x": modes, "rsag_same": rsag_same, "rsqr": rsqr, "refine": refine,
                   "g_elbo": gres["elbo"], "g_Z": gZ.numpy(), "g_W": gres["W"], "g_r2": gres["r2"]})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_matches_single_process():
    sys.path.insert(0, ROOT)
    from muon_amd._atac.preproc import canonical_csr, tfidf_device
    from muon_amd._atac.tools import lsi_device
    from muon_amd._core.mofa_engine import MofaEngine
    from oracle import lsi_oracle
    from tests.cpu_backend import CpuTestBackend
    from tests.synth import planted_topics_csr

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    be = CpuTestBackend()
    X = planted_topics_csr(600, 400, n_topics=8, density=0.08, seed=5, dtype=np.float32)
    Xs = canonical_csr(X)
    Xd = be.upload_csr(Xs.indptr, Xs.indices, Xs.data, Xs.shape)
    T = tfidf_device(be, Xd, 600, 3, 1e4)
    # rank 0 held rows 0..289: its TF-IDF values equal the first rows of the global result
    n0 = int(Xs.indptr[290])
    np.testing.assert_allclose(got["tfidf"], T.values.numpy()[:n0], rtol=1e-6)
    U, stdev, V = lsi_device(be, T, n_comps=8, n_obs=600)
    np.testing.assert_allclose(got["stdev"], stdev, rtol=1e-5)
    assert lsi_oracle.max_subspace_angle(got["V"], V.numpy()) < 1e-4
    assert got["U"].shape == (600, 8)
    assert got["rsag_same"]
    r = got["rsqr"]
    assert r["halves"] and r["replicated"] and r["converged"] and r["iters"] == got["iters"], r
    assert r["angle"] < 1e-6 and r["stdev"] < 1e-6, r
    assert lsi_oracle.max_subspace_angle(got["U"], U.numpy()) < 5e-4
    # the f64 continuation on two ranks against the same on one, and against f64 ARPACK of the same operand
    f = got["refine"]
    U64, stdev64, V64, i64 = lsi_device(be, T, n_comps=8, n_obs=600, return_info=True, refine_f64=True)
    assert f["dtype"] == "torch.float64" and f["replicated"] and f["bound"] <= 1e-6 and f["blocks"] == i64["refine_f64"]["blocks"]
    assert lsi_oracle.max_subspace_angle(f["V"], V64.numpy()) < 1e-7  # (two starts from two f32 runs, each inside its bound)
    np.testing.assert_allclose(f["stdev"], stdev64, rtol=1e-10)
    np.testing.assert_allclose(np.abs(f["U"]), np.abs(U64.numpy()), atol=1e-7)
    Tsp = sp.csr_matrix((T.values.numpy().astype(np.float64), T.indices.numpy(), T.indptr.numpy()), shape=T.shape)
    assert lsi_oracle.max_subspace_angle(f["V"], lsi_oracle.lsi(Tsp, n_comps=8)["LSI"]) < 1e-7

    rng = np.random.default_rng(0)
    Z = rng.standard_normal((120, 4))
    y1 = Z @ rng.standard_normal((30, 4)).T + rng.standard_normal((120, 30))
    y2 = Z @ rng.standard_normal((50, 4)).T + rng.standard_normal((120, 50))
    y2[np.abs(y2) < 0.8] = 0
    groups = rng.integers(0, 2, 120)
    eng = MofaEngine(be, [y1, sp.csr_matrix(y2)], groups, 6, seed=1)
    eng.run(12, "slow")
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(got["elbo"], res["elbo"], rtol=1e-9)
    np.testing.assert_allclose(got["Z"], res["Z"], atol=1e-8)
    for a, b in zip(got["W"], res["W"]):
        np.testing.assert_allclose(a, b, atol=1e-8)

    y1x = y1.copy()
    y1x[:55] = y1x[:55].astype(np.float32)
    engx = MofaEngine(be, [y1x, sp.csr_matrix(y2)], groups, 6, seed=1)
    engx.run(6, "slow")
    assert got["modes_x"] == [False, False]
    np.testing.assert_allclose(got["elbo_x"], engx.results(sort_factors=False)["elbo"], rtol=1e-9)

    # general engine: the same model on the whole data in one process, and the oracle
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from oracle import mofa_oracle

    yc = rng.poisson(np.logaddexp(0, Z @ rng.standard_normal((40, 4)).T)).astype(float)
    yn = y1.copy()
    yn[rng.random(yn.shape) < 0.1] = np.nan
    ref = mofa_oracle.run_general([yn, yc], ["gaussian", "poisson"], groups=groups, n_factors=5, n_iterations=6,
                                  convergence_mode="slow", min_iterations=100)
    np.testing.assert_allclose(got["g_elbo"], ref["elbo"], rtol=1e-9)
    np.testing.assert_allclose(got["g_Z"], ref["Z"], atol=1e-8)
    for a, b in zip(got["g_W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-8)
    np.testing.assert_allclose(got["g_r2"], ref["r2"], atol=1e-6)

#!/usr/bin/env python
"""Multi-rank GPU path against the single-process result, on ONE GPU (all ranks on cuda:0, gloo):
tfidf + lsi of a row-sharded matrix must give the single-process values / subspace, mofa on
row-sharded samples the single-process ELBO trace, factors and weights.

  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dist_gpu_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend
from muon_amd._comm import TorchDistComm

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = TorchDistComm()
be = HipBackend(0)
n, d, k = 60000, 40000, 50
cuts = [0] + [int(n * (r + 1) / world * (0.9 if r + 1 < world else 1.0)) for r in range(world)]  # uneven shards
r0, r1 = cuts[rank], cuts[rank + 1]
X = be.synth_counts(r0, r1 - r0, d, 50, 0.03, 0)
T = tfidf_device(be, X, n, 3, 1e4, comm=comm)
U, sd, V, info = lsi_device(be, T, n_comps=k, n_obs=n, comm=comm, return_info=True)
# the Z all-reduce as reduce-scatter + all-gather (MUON_AMD_Z_COLLECTIVE=rsag): the same sums, element by element
os.environ["MUON_AMD_Z_COLLECTIVE"] = "rsag"
U2, sd2, V2, info2 = lsi_device(be, T, n_comps=k, n_obs=n, comm=comm, return_info=True)
os.environ.pop("MUON_AMD_Z_COLLECTIVE")
assert info2["iterations"] == info["iterations"] and torch.equal(V2, V) and torch.equal(U2, U) and np.array_equal(sd2, sd), \
    "reduce-scatter + all-gather changed the result"
if rank == 0:
    print("rsag == allreduce: bit-identical U, V, stdev")
# SURVEY 8e's form (r06): reduce-scatter -> projection + CholeskyQR2 on row slices (device Cholesky of the all-reduced Gram)
# -> all-gather.  The same expansions, the subspace to 1e-6 rad (sums over the ranks grouped differently)
os.environ["MUON_AMD_Z_COLLECTIVE"] = "rsqr"
U3, sd3, V3, info3 = lsi_device(be, T, n_comps=k, n_obs=n, comm=comm, return_info=True)
os.environ.pop("MUON_AMD_Z_COLLECTIVE")
qa, _ = torch.linalg.qr(V.double())
qb, _ = torch.linalg.qr(V3.double())
ang3 = float(torch.linalg.matrix_norm(qb - qa @ (qa.T @ qb), ord=2))
assert info3["iterations"] == info["iterations"] and info3["converged"] and ang3 < 1e-6 and float(np.max(np.abs(sd3 - sd) / sd)) < 1e-6, \
    (ang3, info3["bounds"])
if rank == 0:
    print(f"rsqr (reduce-scatter, sliced CholeskyQR2, all-gather): angle to the all-reduce run {ang3:.2e}, same expansions")
# the subsampled power warm start across ranks (each rank's first cells; one all-reduce per power step)
os.environ["MUON_AMD_LSI_WARM"] = "8:2"
Uw, sdw, Vw, infow = lsi_device(be, T, n_comps=k, n_obs=n, comm=comm, return_info=True)
os.environ.pop("MUON_AMD_LSI_WARM")
assert infow["warm_start"] is not None and infow["converged"] and infow["iterations"] <= info["iterations"], infow
qa, _ = torch.linalg.qr(V.double())
qb, _ = torch.linalg.qr(Vw.double())
angw = float(torch.linalg.matrix_norm(qb - qa @ (qa.T @ qb), ord=2))
assert angw < 5e-5 and float(np.max(np.abs(sdw - sd) / sd)) < 1e-6, (angw, infow["bounds"])
if rank == 0:
    print(f"warm start ({infow['warm_start']}): {infow['iterations']} expansions against {info['iterations']}, angle to the cold run {angw:.2e}")
# f64 arithmetic for f64 input (r06, tools._refine_f64) on the row streams of the shards: the f32 process continued in f64
# blocks, Z = X^T Y summed over the ranks in f64, the stop decisions rank 0's; against the same on one rank below
U6, sd6, V6, info6 = lsi_device(be, T, n_comps=k, n_obs=n, comm=comm, return_info=True, refine_f64=True)
assert V6.dtype == torch.float64 and info6["refine_f64"]["angle_bound"] <= 1e-6, info6["refine_f64"]
if rank == 0:
    Xf = be.synth_counts(0, n, d, 50, 0.03, 0)
    Tf = tfidf_device(be, Xf, n, 3, 1e4)
    Uf, sdf, Vf, inff = lsi_device(be, Tf, n_comps=k, return_info=True)
    lo, hi = int(Xf.indptr[r0].item()), int(Xf.indptr[r1].item())
    dv = float((Tf.values[lo:hi] - T.values).abs().max() / Tf.values.abs().max())
    qa, _ = torch.linalg.qr(V.double())
    qb, _ = torch.linalg.qr(Vf.double())
    ang = float(torch.linalg.matrix_norm(qb - qa @ (qa.T @ qb), ord=2))
    ds = float(np.max(np.abs(sd - sdf) / sdf))
    du = float((U.abs() - Uf[r0:r1].abs()).abs().max())
    print(f"ranks {world}: tfidf max rel diff {dv:.2e}, subspace angle vs single process {ang:.2e}, stdev rel diff {ds:.2e}, "
          f"|U| max diff {du:.2e}, iterations {info['iterations']} / {inff['iterations']}")
    assert dv < 1e-6 and ang < 1e-4 and ds < 1e-5
    U6f, sd6f, V6f, inf6 = lsi_device(be, Tf, n_comps=k, return_info=True, refine_f64=True)
    qa, _ = torch.linalg.qr(V6)
    qb, _ = torch.linalg.qr(V6f)
    ang6 = float(torch.linalg.matrix_norm(qb - qa @ (qa.T @ qb), ord=2))
    print(f"f64 continuation, ranks {world} against 1: angle {ang6:.2e} (bounds {info6['refine_f64']['angle_bound']:.1e} / "
          f"{inf6['refine_f64']['angle_bound']:.1e}, {info6['refine_f64']['blocks']} / {inf6['refine_f64']['blocks']} blocks), "
          f"stdev rel diff {float(np.max(np.abs(sd6 - sd6f) / sd6f)):.2e}")
    assert ang6 < 2e-6 and float(np.max(np.abs(sd6 - sd6f) / sd6f)) < 1e-9

# MOFA: two views (dense + sparse), two groups, samples sharded by rows; the sufficient statistics,
# the factor column sums and the ELBO part of the samples are the collectives (float64 engine)
import scipy.sparse as sp

from muon_amd._core.mofa_engine import MofaEngine

rng = np.random.default_rng(0)
N = 3000
Z0 = rng.standard_normal((N, 4))
y1 = Z0 @ rng.standard_normal((300, 4)).T + rng.standard_normal((N, 300))
y2 = Z0 @ rng.standard_normal((500, 4)).T + rng.standard_normal((N, 500))
y2[np.abs(y2) < 0.8] = 0
groups = np.sort(rng.integers(0, 2, N))
a, b = (0, 1300) if rank == 0 else (1300, N)
if world == 1:
    a, b = 0, N
eng = MofaEngine(be, [y1[a:b], sp.csr_matrix(y2[a:b])], groups[a:b], 6, seed=1, comm=comm, row_offset=a, n_total=N)
eng.run(12, "slow")
res = eng.results(sort_factors=False)
Zall = comm.all_gather_rows(torch.from_numpy(res["Z"]))
if rank == 0:
    one = MofaEngine(be, [y1, sp.csr_matrix(y2)], groups, 6, seed=1)
    one.run(12, "slow")
    ref = one.results(sort_factors=False)
    de = float(np.max(np.abs((np.asarray(res["elbo"]) - np.asarray(ref["elbo"])) / np.asarray(ref["elbo"]))))
    dz = float(np.max(np.abs(Zall.numpy() - ref["Z"])))
    dw = max(float(np.max(np.abs(x - y))) for x, y in zip(res["W"], ref["W"]))
    print(f"mofa ranks {world}: ELBO trace max rel diff {de:.2e}, |Z| diff {dz:.2e}, |W| diff {dw:.2e}")
    assert de < 1e-9 and dz < 1e-7 and dw < 1e-7
# the element-wise-precision engine with the views that need nothing N x D (r06): masked gaussian statistics, a fused
# poisson view (matrix-core sweeps + stored entries), a fused bernoulli view (Jaakkola sweeps + sparse products) - T and b
# summed over the ranks, S and a local - against the same fit in one process, f64
from muon_amd._core.mofa_general import GeneralMofaEngine

yg = y1.copy()
yg[rng.random(yg.shape) < 0.1] = np.nan
yp = sp.csr_matrix(rng.poisson(np.logaddexp(0, Z0 @ rng.standard_normal((400, 4)).T - 1.0)).astype(float))
yb = sp.csr_matrix((rng.random((N, 350)) < 1 / (1 + np.exp(-(Z0 @ rng.standard_normal((350, 4)).T)))).astype(float))
liks = ["gaussian", "poisson", "bernoulli"]
ge = GeneralMofaEngine(be, [yg[a:b], yp[a:b], yb[a:b]], liks, groups[a:b], 6, seed=1, comm=comm, row_offset=a, n_total=N,
                       dtype=torch.float64)
assert ge.views[1].fused and ge.views[2].fusedb
ge.run(8, "slow", min_iterations=100)
gres = ge.results(sort_factors=False)
gZ = comm.all_gather_rows(torch.from_numpy(gres["Z"]))
if rank == 0:
    g1 = GeneralMofaEngine(be, [yg, yp, yb], liks, groups, 6, seed=1, dtype=torch.float64)
    g1.run(8, "slow", min_iterations=100)
    gref = g1.results(sort_factors=False)
    de = float(np.max(np.abs((np.asarray(gres["elbo"]) - np.asarray(gref["elbo"])) / np.asarray(gref["elbo"]))))
    dz = float(np.max(np.abs(gZ.numpy() - gref["Z"])))
    dw = max(float(np.max(np.abs(x - y))) for x, y in zip(gres["W"], gref["W"]))
    print(f"general engine (masked gaussian + fused poisson + fused bernoulli) ranks {world}: ELBO trace max rel diff {de:.2e}, "
          f"|Z| diff {dz:.2e}, |W| diff {dw:.2e}")
    assert de < 1e-9 and dz < 1e-7 and dw < 1e-7
    print("dist gpu check ok")
dist.barrier()

"""mu.tl.mofa on the GPU: HIP sweeps + SpMM statistics against the numpy oracle under identical
initialisation (iteration-level ELBO / <Z> / <W>), the reference's structural test, and the
reference's union / groups smoke tests (tests/test_muon_tools.py:12-87).
Tolerances: float64 path 1e-8 relative on the ELBO trace, 1e-7 absolute on expectations;
float32 path (use_float32=True) 1e-3."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import muon_amd as mu
from muon_amd import AnnData, MuData
from muon_amd._core.mofa_engine import MofaEngine
from oracle import mofa_oracle
from tests.test_mofa_host import r2_per_factor, simple_views

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["dense", "groups", "sparse_missing", "noard_nospike", "k17"])
def test_engine_matches_oracle_f64(hip, case):
    y1, y2 = simple_views()
    rng = np.random.default_rng(0)
    groups, kw, K = None, {}, 8
    views = [y1, y2]
    if case == "groups":
        groups = rng.integers(0, 3, 100)
    if case == "sparse_missing":
        groups = rng.integers(0, 2, 100)
        y1 = y1.copy(); y1[85:] = np.nan
        y2 = y2.copy(); y2[np.abs(y2) < 1.0] = 0
        views = [y1, sp.csr_matrix(y2)]
    if case == "noard_nospike":
        kw = dict(ard_weights=False, ard_factors=False, spikeslab_weights=False)
    if case == "k17":
        K = 17
    dense = [v.toarray() if sp.issparse(v) else v for v in views]
    ref = mofa_oracle.run(dense, groups=groups, n_factors=K, n_iterations=25, convergence_mode="slow", **kw)
    eng = MofaEngine(hip, views, np.zeros(100, dtype=int) if groups is None else groups, K, seed=1, **kw)
    eng.run(25, "slow")
    res = eng.results(sort_factors=False)
    n = min(len(ref["elbo"]), len(res["elbo"]))
    np.testing.assert_allclose(res["elbo"][:n], ref["elbo"][:n], rtol=1e-8)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-7)
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-7)
    np.testing.assert_allclose(res["r2"], ref["r2"], atol=1e-6)


def test_engine_f32_and_larger_sparse_view(hip):
    rng = np.random.default_rng(5)
    N, K0 = 3000, 6
    Z = rng.standard_normal((N, K0))
    W1 = rng.standard_normal((400, K0)) * (rng.random((400, K0)) < 0.3)
    W2 = rng.standard_normal((2500, K0)) * (rng.random((2500, K0)) < 0.3)
    y1 = Z @ W1.T + rng.standard_normal((N, 400))
    y2 = Z @ W2.T + rng.standard_normal((N, 2500))
    y2[rng.random(y2.shape) < 0.9] = 0  # sparse "atac-like" view, stays CSR on the device
    views = [y1, sp.csr_matrix(y2)]
    ref = mofa_oracle.run([y1, y2], n_factors=10, n_iterations=15, convergence_mode="slow")
    for dt, tol in ((torch.float64, 1e-8), (torch.float32, 2e-3)):
        eng = MofaEngine(hip, views, np.zeros(N, dtype=int), 10, seed=1, dtype=dt)
        eng.run(15, "slow")
        np.testing.assert_allclose(eng.elbo, ref["elbo"], rtol=tol)
    e = np.array(eng.elbo)
    assert np.all(np.diff(e) > -1e-4 * abs(e[0]))


def test_captured_iteration_equals_eager(hip):
    # from the third iteration on the sweep is replayed as a HIP graph; the ELBO trajectory and the
    # final state must be the ones of the eager sweeps
    y1, y2 = simple_views()
    y2 = y2.copy(); y2[np.abs(y2) < 1.0] = 0
    views = [y1, sp.csr_matrix(y2)]
    groups = np.random.default_rng(2).integers(0, 2, 100)
    out = []
    for graph in (True, False):
        eng = MofaEngine(hip, views, groups, 6, seed=1)
        eng._graph_ok = graph
        eng.run(12, "slow")
        assert (eng._graph is not None) == graph
        out.append((np.array(eng.elbo), eng.results(sort_factors=False)))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-13)
    np.testing.assert_allclose(out[0][1]["Z"], out[1][1]["Z"], atol=1e-12)
    np.testing.assert_allclose(out[0][1]["r2"], out[1][1]["r2"], atol=1e-9)


class TestWrapperOnGpu:
    def setup_method(self):
        y1, y2 = simple_views()
        self.mdata = MuData({"y1": AnnData(y1), "y2": AnnData(y2)})

    def test_mofa_nfactors(self, tmp_path):
        mu.tl.mofa(self.mdata, n_factors=10, quiet=True, verbose=False, outfile=str(tmp_path / "m.hdf5"))
        y = np.concatenate([self.mdata.mod["y1"].X, self.mdata.mod["y2"].X], axis=1)
        r2 = r2_per_factor(y, self.mdata.obsm["X_mofa"], self.mdata.varm["LFs"])
        assert all(i > 0.1 for i in r2[:5]) and not any(i > 0.1 for i in r2[5:])

    def test_mofa_anndata_groups_union(self):
        a = self.mdata["y1"].copy()
        np.random.seed(3)
        a.obs["ab"] = np.random.choice(["a", "b"], a.n_obs)
        mu.tl.mofa(a, groups_label="ab", n_factors=10, use_float32=True)
        assert a.obsm["X_mofa"].shape == (100, 10) and a.varm["LFs"].shape == (90, 10)
        y1, y2 = self.mdata["y1"], self.mdata["y2"]
        y2.X = sp.csr_matrix(y2.X)
        p, q = y1[:-10], y2[10:]
        p._init_as_actual(); q._init_as_actual()
        m = MuData({"y1": p, "y2": q})
        mu.tl.mofa(m, n_factors=10, use_obs="union", likelihoods="gaussian")
        assert m.obsm["X_mofa"].shape == (100, 10) and np.all(np.isfinite(m.obsm["X_mofa"]))

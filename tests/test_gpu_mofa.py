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


@pytest.mark.parametrize("case", ["plain", "groups_missing", "offset"])
def test_f64_fit_of_an_f32_valued_dense_view_streams_f32(hip, case, monkeypatch):
    """r04: a dense view whose values are exact in f32 (AnnData's default dtype) stays in f32 on the device under the
    f64 fit and is centred inside the products (like the sparse views).  Against the oracle (centres first, all f64)
    and against the same engine with the f64 copy (MUON_AMD_MOFA_F32_STORAGE=0): rounding differences only."""
    rng = np.random.default_rng(12)
    N, K0 = 600, 5
    Z = rng.standard_normal((N, K0))
    W1 = rng.standard_normal((300, K0)) * (rng.random((300, K0)) < 0.4)
    W2 = rng.standard_normal((170, K0)) * (rng.random((170, K0)) < 0.4)
    y1 = (Z @ W1.T + rng.standard_normal((N, 300))).astype(np.float32).astype(np.float64)
    y2 = (Z @ W2.T + rng.standard_normal((N, 170))).astype(np.float32).astype(np.float64)
    groups = np.zeros(N, dtype=int)
    if case == "groups_missing":
        groups = rng.integers(0, 3, N)
        y1[500:] = np.nan  # samples missing from view 1
    if case == "offset":
        y1 = (y1 + 40.0).astype(np.float32).astype(np.float64)  # |mean| = 40 std: the cancellation case
    ref = mofa_oracle.run([y1, y2], groups=groups if case == "groups_missing" else None, n_factors=6, n_iterations=20,
                          convergence_mode="slow")
    eng = MofaEngine(hip, [y1, y2], groups, 6, seed=1)
    assert all(v.Y.dtype == torch.float32 and v.implicit for v in eng.views)
    eng.run(20, "slow")
    res = eng.results(sort_factors=False)
    monkeypatch.setenv("MUON_AMD_MOFA_F32_STORAGE", "0")
    eng64 = MofaEngine(hip, [y1, y2], groups, 6, seed=1)
    assert all(v.Y.dtype == torch.float64 and not getattr(v, "implicit", False) for v in eng64.views)
    eng64.run(20, "slow")
    tol = 1e-8 if case != "offset" else 1e-7
    np.testing.assert_allclose(res["elbo"], ref["elbo"][:len(res["elbo"])], rtol=tol)
    np.testing.assert_allclose(eng.elbo, eng64.elbo, rtol=tol)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-6)
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-6)
    for a, b in zip(res["intercepts"], eng64.results(sort_factors=False)["intercepts"]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,G,ard,ss", [(10, 1, True, True), (10, 2, True, False), (20, 2, False, True), (3, 3, False, False)])
def test_fused_small_nodes_match_tensor_formulas(hip, dtype, K, G, ard, ss):
    """csrc/mofa_elbo.hip against the same equations as tensor operations (tests/cpu_backend.py):
    tau / <ln tau>, alpha_w, theta, alpha_z and every ELBO term."""
    from tests.cpu_backend import CpuTestBackend

    cpu = CpuTestBackend()
    g = torch.Generator().manual_seed(5)
    D, N = 1234, 777
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    EW = (r(D, K) - 0.5).to(dtype)
    EW2 = (EW.double() ** 2 + 0.1 * r(D, K)).to(dtype)
    Zm = r(G, 50, K) - 0.5
    Gz = torch.einsum("gnk,gnl->gkl", Zm, Zm).to(dtype)
    Z2 = (Zm ** 2).sum(dim=1).add(0.3).to(dtype)
    B = (4 * (r(G, D, K) - 0.5)).to(dtype)
    yy = (60 + 30 * r(G, D)).to(dtype)
    Ngm = torch.randint(40, 51, (G,), generator=g).to(dtype)
    gamma = r(D, K)
    gamma[:5] = 1.0
    gamma[5:9] = 0.0
    gamma = gamma.to(dtype)
    EWh2 = (0.2 + r(D, K)).to(dtype)
    sig2 = (0.01 + r(D, K)).to(dtype)
    EZ2 = (0.1 + r(N, K)).to(dtype)
    sig2z = (0.01 + r(N, K)).to(dtype)
    cuts = [0] + sorted(torch.randint(1, N, (G - 1,), generator=g).tolist()) + [N]
    Ng = torch.tensor([cuts[i + 1] - cuts[i] for i in range(G)], dtype=torch.float64)

    def run(be, dev):
        to = lambda t: t.to(dev).contiguous()
        out = {"tau": torch.zeros((G, D), dtype=dtype, device=dev), "ltau": torch.zeros((G, D), dtype=dtype, device=dev)}
        for n in ("alpha", "lalpha", "lth", "l1mth"):
            out[n] = torch.full((K,), 0.25, dtype=dtype, device=dev)
        out["alpha_z"] = torch.ones((G, K), dtype=dtype, device=dev)
        out["lalpha_z"] = torch.zeros((G, K), dtype=dtype, device=dev)
        parts = []
        work = be.mofa_elbo_work(K)
        e = torch.zeros((), dtype=torch.float64, device=dev)
        be.mofa_tau_elbo(to(yy), to(Ngm), to(EW), to(EW2), to(B), to(Gz), to(Z2), 1e-14, 1e-14, out["tau"], out["ltau"], e, work)
        parts.append(float(e))
        e = torch.zeros((), dtype=torch.float64, device=dev)
        be.mofa_w_elbo(to(EWh2), to(gamma), to(sig2), ard, ss, 1e-14 + 0.5 * D, 1e-14, 1e-14, 1.0, 1.0,
                       out["alpha"], out["lalpha"], out["lth"], out["l1mth"], e, work)
        parts.append(float(e))
        zs = torch.zeros((G, 2, K), dtype=torch.float64, device=dev)
        for i in range(G):
            be.mofa_z_sums(to(EZ2), to(sig2z), cuts[i], cuts[i + 1], zs[i], work)
        e = torch.zeros((), dtype=torch.float64, device=dev)
        be.mofa_z_elbo(zs, to(Ng), ard, 1e-14, 1e-14, out["alpha_z"], out["lalpha_z"], e)
        parts.append(float(e))
        return {k: v.cpu() for k, v in out.items()}, parts, zs.cpu()

    ref, pref, zref = run(cpu, torch.device("cpu"))
    got, pgot, zgot = run(hip, hip.device)
    tol = 1e-10 if dtype == torch.float64 else 2e-6
    assert torch.allclose(zgot, zref, rtol=1e-12, atol=0)
    for n in ref:
        assert torch.allclose(got[n].double(), ref[n].double(), rtol=tol, atol=tol), n
    for a, b in zip(pgot, pref):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (pgot, pref)
    # the partial sums are folded in a fixed order
    _again, p2, _z = run(hip, hip.device)
    assert p2 == pgot


@pytest.mark.timeout(900)
def test_engine_matches_oracle_at_the_baseline_feature_dimensions(hip):
    """BASELINE.json configs[3] at its REAL feature dimensions (20 000 dense + 100 000 sparse features,
    K = 10) on a 1500-cell sample - what the oracle finishes in seconds: the engine must follow the
    oracle iteration by iteration from the same initialisation (f64: ELBO 1e-8 relative, <Z>/<W> 1e-6;
    f32, the timed precision: ELBO 2e-3).  Same code as bench.py's `secondary.c4.parity`."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mofa", os.path.join(root, "scripts", "bench_mofa.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    n = 1500
    rna, atac = bm.make_views(hip, 0, n, n, 20_000, 100_000)
    n, host, dev = bm.sample_views(hip, rna, atac, n)
    ref, _wall, _cpu, par = bm.oracle_parity(hip, host, dev, n, 3)
    assert len(ref["elbo"]) == 3
    assert par["f64"]["elbo_max_rel"] < 1e-8 and par["f64"]["Z_max_abs"] < 1e-6 and par["f64"]["W_max_abs"] < 1e-6
    assert par["f32"]["elbo_max_rel"] < 2e-3


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-9), (torch.float32, 5e-3)])
def test_general_engine_on_the_gpu_matches_oracle(hip, dt, tol):
    """SURVEY 8f.3 on the device: gaussian view with element-wise NaN + sparse poisson view (stays CSR
    in HBM, chunk-wise densified) + bernoulli view, two groups, against oracle.run_general."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from tests.test_mofa_host import _mixed_views

    _, y1, y2, y3 = _mixed_views(n=400, seed=2)
    groups = np.random.default_rng(1).integers(0, 2, 400)
    liks = ["gaussian", "poisson", "bernoulli"]
    ref = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=5, n_iterations=6,
                                  convergence_mode="slow", min_iterations=100)
    eng = GeneralMofaEngine(hip, [y1, sp.csr_matrix(y2), y3], liks, groups, 5, seed=1, dtype=dt, chunk_elems=6000)
    eng.run(6, "slow", min_iterations=100)
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=tol)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=max(tol * 10, 1e-8))


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-9), (torch.float32, 5e-3)])
@pytest.mark.parametrize("K", [5, 10, 16])
def test_sparse_bernoulli_view_without_dense_chunks_matches_oracle(hip, dt, tol, K):
    """r06: a bernoulli view stored sparse is fitted without anything N x D - its
    Jaakkola precision in one dense sweep over the factor blocks per update (csrc/mofa_bernoulli.hip, f32 and f64 on the
    matrix cores), the data through sparse products, the likelihood through the poisson view's softplus sweep - and gives
    what the dense restatement gives (oracle.run_general), iteration by iteration; next to a masked gaussian view and a
    fused poisson view, two groups."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from tests.test_mofa_host import _mixed_views

    _, y1, y2, y3 = _mixed_views(n=400, seed=2)
    groups = np.random.default_rng(1).integers(0, 2, 400)
    liks = ["gaussian", "poisson", "bernoulli"]
    ref = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=K, n_iterations=6,
                                  convergence_mode="slow", min_iterations=100)
    eng = GeneralMofaEngine(hip, [y1, sp.csr_matrix(y2), sp.csr_matrix(y3)], liks, groups, K, seed=1, dtype=dt,
                            chunk_elems=6000)
    assert eng.views[2].fusedb and eng.views[1].fused
    eng.run(6, "slow", min_iterations=100)
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=tol)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=max(tol * 10, 1e-8))
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=max(tol * 10, 1e-8))
    # the same view given dense takes the chunk passes: the same fit
    eng2 = GeneralMofaEngine(hip, [y1, sp.csr_matrix(y2), y3], liks, groups, K, seed=1, dtype=dt, chunk_elems=6000)
    assert not eng2.views[2].fusedb
    eng2.run(6, "slow", min_iterations=100)
    np.testing.assert_allclose(res["elbo"], eng2.results(sort_factors=False)["elbo"], rtol=tol)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("K", [1, 3, 5, 8, 10, 12, 13, 16])
def test_jaakkola_sweep_kernel(hip, dt, tol, K):
    """mu_mofa_pack_moments + mu_mofa_jaakkola_sweep against the element-wise definition in numpy f64: every instantiated
    (padded width, column tiles) pair, sizes off the 16-row tiles / the LDS stages / the workgroups, moments with zero
    variance (xi = |zeta|) and tiny predictions (the series of tanh(xi / 2) / (2 xi))."""
    for n_own, n_other in ((1, 1), (37, 150), (300, 517), (130, 64)):
        rng = np.random.default_rng(K * 7 + n_own)
        Eo = rng.standard_normal((n_own, K)) * 0.8
        Et = rng.standard_normal((n_other, K)) * 0.6
        Eo2 = Eo ** 2 + rng.random((n_own, K)) * 0.3
        Et2 = Et ** 2 + rng.random((n_other, K)) * 0.2
        Eo2[::3] = Eo[::3] ** 2  # (no variance)
        Et[::5] *= 1e-3          # (xi ~ 1e-3 .. 1e-2)
        Et2[::5] = Et[::5] ** 2 + 1e-6
        npdt = np.float32 if dt == torch.float32 else np.float64
        c = lambda a: a.astype(npdt).astype(np.float64)
        Eo, Eo2, Et, Et2 = c(Eo), c(Eo2), c(Et), c(Et2)
        dev = lambda a: torch.from_numpy(a).to(hip.device).to(dt)
        got = hip.to_host(hip.mofa_jaakkola_sweep(dev(Eo), dev(Eo2), dev(Et), dev(Et2))).astype(np.float64)
        zeta = Eo @ Et.T
        xi = np.maximum(np.sqrt(np.maximum(zeta ** 2 + Eo2 @ Et2.T - (Eo ** 2) @ (Et ** 2).T, 0)), 1e-8)
        Om = np.tanh(0.5 * xi) / (2 * xi)
        P = Et[:, :, None] * Et[:, None, :]
        i = np.arange(K)
        P[:, i, i] = Et2
        want = np.einsum("ot,tkl->okl", Om, P)
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= tol * max(np.max(np.abs(want)), 1e-300), (n_own, n_other)
        assert np.array_equal(got, got.transpose(0, 2, 1))  # symmetric by construction


@pytest.mark.parametrize("shape", [(5, 3, 2, 1), (17, 33, 1, 1), (40, 16, 16, 2), (130, 257, 7, 3)])
def test_sparse_bernoulli_view_edge_shapes(hip, shape):
    """the fused bernoulli path at shapes below every tile of its kernels (a handful of samples / features, one factor,
    several groups, a feature without a single one): f64 against the dense restatement to rounding, f32 inside its bar"""
    from muon_amd._core.mofa_general import GeneralMofaEngine

    N, D, K, G = shape
    rng = np.random.default_rng(N + D)
    Z = rng.standard_normal((N, 3))
    y = (rng.random((N, D)) < 1 / (1 + np.exp(-(Z @ rng.standard_normal((D, 3)).T)))).astype(float)
    y[:, 0] = 0
    groups = np.sort(np.concatenate([np.arange(G), rng.integers(0, G, N - G)]))
    ref = mofa_oracle.run_general([y], ["bernoulli"], groups=groups, n_factors=K, n_iterations=5, convergence_mode="slow",
                                  min_iterations=100)
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 5e-3)):
        eng = GeneralMofaEngine(hip, [sp.csr_matrix(y)], ["bernoulli"], groups, K, seed=1, dtype=dt)
        assert eng.views[0].fusedb
        eng.run(5, "slow", min_iterations=100)
        res = eng.results(sort_factors=False)
        np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=tol)
        np.testing.assert_allclose(res["Z"], ref["Z"], atol=max(10 * tol, 1e-9))


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
def test_poisson_passes_edge_shapes(dt, tol):
    """mofa_poisson_pass (matrix-core sweep + stored entries, every padded width) at shapes of one or a few rows on either
    side, against the element-wise definition: nothing reads past a block, padded lanes add nothing"""
    from muon_amd._backend import get_backend

    be = get_backend()
    rng = np.random.default_rng(0)
    for K in (1, 2, 9, 12, 16, 17, 32):
        for N, D in ((1, 1), (2, 5), (3, 1), (1, 40), (65, 2), (16, 16)):
            Z = rng.standard_normal((N, K)) * 0.7
            W = rng.standard_normal((D, K)) * 0.5
            Y = sp.csr_matrix(rng.poisson(np.logaddexp(0, Z @ W.T)).astype(np.float64))
            Y.sort_indices()
            kappa = 0.25 + 0.17 * np.asarray(Y.max(axis=0).todense()).ravel()
            zeta = Z @ W.T
            r = np.maximum(np.logaddexp(0, zeta), 1e-300)
            Yd = Y.toarray()
            R = kappa[None, :] * zeta - 1.0 / (1.0 + np.exp(-zeta)) * (1.0 - Yd / r)
            want = (R @ W, R.T @ Z, (Yd * np.log(r) - r).sum(axis=1))
            X = be.upload_csr(Y.indptr, Y.indices, Y.data, Y.shape, values_dtype=np.float64)
            X = X.with_values(X.values.to(dt))
            Xt = be.transpose(X)
            Zd, Wd, kd = (torch.from_numpy(a).to(be.device).to(dt).contiguous() for a in (Z, W, kappa))
            got = (be.mofa_poisson_pass(0, Zd, Wd, kd, X), be.mofa_poisson_pass(1, Wd, Zd, kd, Xt),
                   be.mofa_poisson_pass(2, Zd, Wd, None, X))
            for mode, (g, w) in enumerate(zip(got, want)):
                g = be.to_host(g).astype(np.float64)
                assert g.shape == w.shape and np.all(np.isfinite(g)), (K, N, D, mode)
                assert np.max(np.abs(g - w)) <= tol * max(np.max(np.abs(w)), 1e-30), (K, N, D, mode)


def test_wrapper_fits_count_likelihoods_on_the_gpu():
    from tests.test_mofa_host import _mixed_views

    _, y1, y2, y3 = _mixed_views(n=300, seed=3)
    md = MuData({"rna": AnnData(np.nan_to_num(y1)), "counts": AnnData(sp.csr_matrix(y2)), "acc": AnnData(y3)})
    mu.tl.mofa(md, n_factors=6, n_iterations=25, quiet=True)
    assert list(md.uns["mofa"]["params"]["data"]["likelihoods"]) == ["gaussian", "poisson", "bernoulli"]
    assert np.all(np.isfinite(md.obsm["X_mofa"])) and md.obsm["X_mofa"].shape == (300, 6)
    e = md.uns["mofa"]["elbo"]
    assert np.all(np.diff(e) > -1e-6 * abs(e[0]))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,R", [(10, 5000), (16, 777), (3, 70000), (20, 1234)])
def test_rowstats_kernel_matches_tensor_formulas(hip, dtype, K, R):
    """csrc/mofa_stats.hip: padded operand, transposed operand, weighted Gram, second- and first-moment
    sums of a factor / weight block in one pass, against the same formulas as tensor operations."""
    g = torch.Generator(device="cuda").manual_seed(K * 1000 + R)
    E = torch.randn((R, K), generator=g, device="cuda", dtype=dtype)
    E2 = E ** 2 + torch.rand((R, K), generator=g, device="cuda", dtype=dtype)
    w = torch.rand((R,), generator=g, device="cuda", dtype=dtype) + 0.5
    a = torch.randn((R,), generator=g, device="cuda", dtype=dtype)
    work = hip.mofa_rowstats_work(K)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for r0, r1, wgt, aux, scale in ((0, R, w, a, True), (R // 3, R - 5, None, None, False), (7, 7, w, None, True)):
        ld, col0 = 32, 3
        pad = torch.full((R, ld), 7.0, device="cuda", dtype=dtype)
        out_t = torch.full((K, R), 7.0, device="cuda", dtype=dtype)
        gram = torch.empty((K, K), device="cuda", dtype=dtype)
        s2 = torch.empty((K,), device="cuda", dtype=dtype)
        s1 = torch.empty((K,), device="cuda", dtype=dtype)
        hip.mofa_rowstats(E, E2, r0, r1, work, wgt=wgt, aux=aux, scale_out=scale, out_pad=pad, col0=col0,
                          out_t=out_t, gram=gram, s2=s2, s1=s1)
        e, e2 = E[r0:r1].double(), E2[r0:r1].double()
        ww = wgt[r0:r1].double() if wgt is not None else torch.ones(r1 - r0, device="cuda", dtype=torch.float64)
        aa = aux[r0:r1].double() if aux is not None else torch.ones(r1 - r0, device="cuda", dtype=torch.float64)
        o = (ww[:, None] * e) if scale else e
        want_pad = torch.full((R, ld), 7.0, device="cuda", dtype=torch.float64)
        want_pad[r0:r1, col0:col0 + K] = o
        want_t = torch.full((K, R), 7.0, device="cuda", dtype=torch.float64)
        want_t[:, r0:r1] = o.T
        scale_g = max(float(((ww[:, None] * e).T @ e).abs().max()), 1.0)
        assert torch.allclose(pad.double(), want_pad, rtol=tol, atol=tol)
        assert torch.allclose(out_t.double(), want_t, rtol=tol, atol=tol)
        assert float((gram.double() - (ww[:, None] * e).T @ e).abs().max()) <= tol * scale_g
        assert float((s2.double() - (ww[:, None] * e2).sum(0)).abs().max()) <= tol * scale_g
        assert float((s1.double() - ((aa * ww)[:, None] * e).sum(0)).abs().max()) <= tol * scale_g


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("K,spike", [(5, True), (10, False), (20, True)])
def test_gauss_seidel_row_kernel_matches_the_tensor_loop(hip, dt, K, spike):
    """mu_mofa_gs_update (a thread per row, row-wise K x K statistics) against the per-factor tensor loop it replaced."""
    g = torch.Generator(device="cpu").manual_seed(K)
    n = 3001
    A = torch.randn((n, K, K), generator=g, dtype=torch.float64)
    Tm = (A @ A.transpose(1, 2) + 0.5 * torch.eye(K, dtype=torch.float64)).to(dt).to(hip.device).contiguous()
    b = torch.randn((n, K), generator=g, dtype=torch.float64).to(dt).to(hip.device).contiguous()
    E0 = torch.randn((n, K), generator=g, dtype=torch.float64).to(dt).to(hip.device)
    prior = (torch.rand((K,), generator=g, dtype=torch.float64) + 0.5).to(hip.device)
    lth = -torch.rand((K,), generator=g, dtype=torch.float64).to(hip.device)
    l1m = -torch.rand((K,), generator=g, dtype=torch.float64).to(hip.device)
    E, E2, gam, Eh2, s2 = (E0.clone(), torch.zeros_like(E0), torch.zeros_like(E0), torch.zeros_like(E0), torch.zeros_like(E0))
    hip.mofa_gs_update(Tm, b, prior, lth, l1m, spike, E, E2, gam, Eh2, s2)
    R, T64, b64 = E0.double().clone(), Tm.double(), b.double()
    for k in range(K):
        t = b64[:, k] - (R * T64[:, k, :]).sum(dim=1) + R[:, k] * T64[:, k, k]
        prec = T64[:, k, k] + prior[k]
        mu = t / prec
        gm = torch.ones_like(mu)
        if spike:
            gm = torch.sigmoid(lth[k] - l1m[k] + 0.5 * torch.log(prior[k]) - 0.5 * torch.log(prec) + 0.5 * t * t / prec)
        R[:, k] = gm * mu
        tol = 1e-12 if dt == torch.float64 else 2e-5
        assert torch.allclose(E[:, k].double(), R[:, k], rtol=tol, atol=tol)
        assert torch.allclose(E2[:, k].double(), gm * (mu * mu + 1 / prec), rtol=tol, atol=tol)
        assert torch.allclose(gam[:, k].double(), gm, rtol=tol, atol=tol)
        assert torch.allclose(Eh2[:, k].double(), gm * (mu * mu + 1 / prec) + (1 - gm) / prior[k], rtol=tol, atol=tol)
        assert torch.allclose(s2[:, k].double(), 1 / prec, rtol=tol, atol=tol)
        R[:, k] = E[:, k].double()  # (continue from the kernel's rounded value, as the kernel does not)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("K", [3, 10, 16, 20])
def test_poisson_passes_without_the_dense_chunk(dt, tol, K):
    """mu_mofa_poisson_dense + _sparse (a dense sweep over the factor blocks + the stored entries) against the
    element-wise definition on the densified view: a = R <W>, b = R^T <Z>, the likelihood terms"""
    import scipy.sparse as sp

    from muon_amd._backend import get_backend

    be = get_backend()
    rng = np.random.default_rng(K)
    N, D = 700, 1300
    Z = rng.standard_normal((N, K)) * 0.7
    W = rng.standard_normal((D, K)) * 0.5
    W[::7] = 0
    rate = np.logaddexp(0, Z @ W.T - 1.5)
    Y = sp.csr_matrix(rng.poisson(rate).astype(np.float64))
    Y.sort_indices()
    kappa = 0.25 + 0.17 * np.asarray(Y.max(axis=0).todense()).ravel()
    zeta = Z @ W.T
    r = np.maximum(np.where(zeta > 20, zeta, np.log1p(np.exp(zeta))), 1e-300)
    Yd = Y.toarray()
    R = kappa[None, :] * zeta - 1.0 / (1.0 + np.exp(-zeta)) * (1.0 - Yd / r)
    want = (R @ W, R.T @ Z, (Yd * np.log(r) - r).sum(axis=1))
    X = be.upload_csr(Y.indptr, Y.indices, Y.data, Y.shape, values_dtype=np.float64)
    X = X.with_values(X.values.to(dt))
    Xt = be.transpose(X)
    Zd, Wd, kd = (torch.from_numpy(a).to(be.device).to(dt).contiguous() for a in (Z, W, kappa))
    got = (be.mofa_poisson_pass(0, Zd, Wd, kd, X), be.mofa_poisson_pass(1, Wd, Zd, kd, Xt),
           be.mofa_poisson_pass(2, Zd, Wd, None, X))
    for g, w in zip(got, want):
        g = be.to_host(g).astype(np.float64)
        assert g.shape == w.shape
        assert np.max(np.abs(g - w)) <= tol * np.max(np.abs(w))
    # deterministic
    again = be.mofa_poisson_pass(0, Zd, Wd, kd, X)
    assert torch.equal(again, got[0])
    # mode 3: b and the likelihood term in one sweep over the same predictions
    both = be.mofa_poisson_pass(3, Wd, Zd, kd, Xt)
    assert both.shape == (D, K + 1) and torch.equal(both[:, :K].contiguous(), got[1])
    lik = float(both[:, K].sum(dtype=torch.float64))
    assert abs(lik - want[2].sum()) <= tol * abs(want[2].sum())


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
@pytest.mark.parametrize("K,ld", [(3, 4), (7, 8), (10, 12), (10, 16), (13, 16), (16, 16), (16, 20), (20, 32)])
def test_poisson_stored_entries_ragged_rows(dt, tol, K, ld):
    """mu_mofa_poisson_sparse_ld alone (the correction over the stored entries, added to a given array) against numpy f64:
    rows of 0, 1, .. entries around every batch size of the two kernels (16 / 64 entries per step, 128 / 256 per batch),
    both kernels where both exist (four lanes per entry at 9 <= K <= 16 - tune key pois_lane selects the lane-per-entry
    one), row strides at and above the padded width, predictions far below zero (the rate needs its RELATIVE accuracy
    there: ln(1 + e^z) by Kahan's quotient), every mode."""
    from muon_amd._backend import _dt, _p, check, get_backend

    be = get_backend()
    rng = np.random.default_rng(K * 100 + ld)
    lens = [0, 1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 0, 513, 700]
    n_own, n_other = len(lens), 900
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_other, size=n, replace=False)) for n in lens] + [[]]).astype(np.int32)
    vals = rng.integers(1, 9, size=indices.size).astype(np.float64)
    Eo = np.zeros((n_own, ld)); Eo[:, :K] = rng.standard_normal((n_own, K)) * 1.5
    Et = np.zeros((n_other, ld)); Et[:, :K] = rng.standard_normal((n_other, K)) * 1.5
    for j in indices[indptr[5]:indptr[5] + 6]:  # a row with predictions of exactly -40 (and whatever the rest gives)
        Et[j, :K] = -40.0 * Eo[5, :K] / np.sum(Eo[5, :K] ** 2)
    npdt = np.float32 if dt == torch.float32 else np.float64
    Eo, Et = Eo.astype(npdt).astype(np.float64), Et.astype(npdt).astype(np.float64)
    dev = lambda a, t=dt: torch.from_numpy(np.ascontiguousarray(a)).to(be.device).to(t).contiguous()
    ip, ix, vd, Eod, Etd = dev(indptr, torch.int64), dev(indices, torch.int32), dev(vals), dev(Eo), dev(Et)
    want = {0: np.zeros((n_own, K)), 2: np.zeros(n_own)}
    for i in range(n_own):
        j = indices[indptr[i]:indptr[i + 1]]
        y = vals[indptr[i]:indptr[i + 1]]
        z = Et[j, :K] @ Eo[i, :K]
        rate = np.maximum(np.logaddexp(0.0, z), 1e-300 if dt == torch.float64 else 1e-30)
        want[0][i] = ((1.0 / (1.0 + np.exp(-z))) * y / rate) @ Et[j, :K]
        want[2][i] = np.sum(y * np.log(rate))
    assert np.min(Et[indices[indptr[5]:indptr[6]], :K] @ Eo[5, :K]) < -39
    kernels = (0, 1) if 8 < K <= 16 else (0,)
    try:
        for lane_kernel in kernels:
            be.lib.mu_tune_set(b"pois_lane", lane_kernel)
            for mode in (0, 1, 2, 3):
                shape = (n_own,) if mode == 2 else (n_own, K + 1 if mode == 3 else K)
                out = torch.full(shape, 0.5, dtype=dt, device=be.device)
                check(be.lib.mu_mofa_poisson_sparse_ld(_dt(Eod), mode, n_own, K, ld, _p(ip), _p(ix), _p(vd), _p(Eod),
                                                       _p(Etd), _p(out), be._stream()))
                got = be.to_host(out).astype(np.float64) - 0.5
                if mode == 2:
                    w = want[2]
                elif mode == 3:
                    w = np.concatenate([want[0], want[2][:, None]], axis=1)
                else:
                    w = want[0]
                assert np.all(np.isfinite(got))
                # per row: a row's terms against the row's own scale (the 0.5 the result is added to costs f32 its
                # last bits: an absolute 1e-7 on top)
                scale = np.maximum(np.max(np.abs(w.reshape(n_own, -1)), axis=1), 1.0)
                err = np.max(np.abs(got - w).reshape(n_own, -1), axis=1)
                bad = np.nonzero(err > tol * scale + (0 if dt == torch.float64 else 2e-7))[0]
                assert bad.size == 0, (lane_kernel, mode, bad, (err / scale)[bad], [lens[i] for i in bad])
    finally:
        be.lib.mu_tune_set(b"pois_lane", 0)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-13), (torch.float32, 2e-6)])
@pytest.mark.parametrize("K,D,G,masked", [(1, 1, 1, True), (5, 700, 1, True), (10, 2000, 2, True), (10, 513, 3, False),
                                          (20, 300, 2, True)])
def test_stats_view_residual_and_tau_finish_kernels(dt, tol, K, D, G, masked):
    """mu_mofa_stats_resid + mu_mofa_tau_finish (r06: the tau node of a dense gaussian view with missing entries in the
    general engine, two kernels for ~35 tensor launches) against the tensor expressions they replace
    (muon_amd/_core/mofa_general.py, the `elif V.stats` / `elif V.lik == "gaussian"` branches), in f64."""
    import math

    from muon_amd._backend import get_backend
    from muon_amd._core.mofa_general import A0, B0, _gamma_kl

    be = get_backend()
    g = torch.Generator(device="cuda").manual_seed(K * 1000 + D)
    r = lambda *shape: torch.randn(shape, generator=g, device="cuda", dtype=torch.float64)
    EW = (r(D, K) * 0.7).to(dt)
    EW2 = (EW.double() ** 2 + torch.rand((D, K), generator=g, device="cuda", dtype=torch.float64) * 0.1).to(dt)
    S = torch.empty((G, D), dtype=torch.float64, device="cuda")
    Ngd = torch.randint(50, 400, (G, D), generator=g, device="cuda").double()
    yy = (r(G, D) ** 2 * 300 + 200).contiguous()
    want_S = torch.empty_like(S)
    idx = torch.arange(K, device="cuda")
    WW = EW.double()[:, :, None] * EW.double()[:, None, :]
    WW[:, idx, idx] = EW2.double()
    WW = WW.reshape(D, K * K)
    for gi in range(G):
        B = (r(D, K) * 3).to(dt).contiguous()
        A = r(D if masked else 1, K, K)
        Q = (A @ A.transpose(1, 2) * 10).reshape(-1, K * K).to(dt).contiguous()  # (sums of outer products: symmetric, psd)
        be.mofa_stats_resid(yy[gi], EW, EW2, B, Q, S[gi])
        want_S[gi] = yy[gi] - 2.0 * (EW.double() * B.double()).sum(dim=1) + (Q.double() * WW).sum(dim=1)
    assert float((S - want_S).abs().max()) <= tol * float(want_S.abs().max())
    # the node's finish on the kernel's own S
    tau = torch.empty((G, D), dtype=dt, device="cuda")
    ltau = torch.empty_like(tau)
    elbo = torch.full((), 3.25, dtype=torch.float64, device="cuda")
    S.clamp_(min=1.0)
    be.mofa_tau_finish(S, Ngd, A0, B0, tau, ltau, elbo, be.mofa_elbo_work(K))
    a, b = A0 + 0.5 * Ngd, B0 + 0.5 * S
    wt, wl = a / b, torch.digamma(a) - torch.log(b)
    want = 3.25 + float((0.5 * Ngd * (wl - math.log(2 * math.pi)) - 0.5 * wt * S).sum()
                        + _gamma_kl(A0, B0, a, b, wt, wl).sum())
    assert float((tau.double() - wt).abs().max()) <= max(tol, 1e-7 if dt == torch.float32 else 0) * float(wt.abs().max())
    assert float((ltau.double() - wl).abs().max()) <= max(tol, 1e-7 if dt == torch.float32 else 1e-12) * float(wl.abs().max())
    assert abs(float(elbo) - want) <= 1e-11 * abs(want)
    again = torch.full((), 3.25, dtype=torch.float64, device="cuda")
    be.mofa_tau_finish(S, Ngd, A0, B0, tau, ltau, again, be.mofa_elbo_work(K))
    assert float(again) == float(elbo)  # fixed-order fold


def _pois_dense_sweep(be, mode, Eo, Et, kap, K, blk=None):
    """mu_mofa_poisson_dense alone (no stored entries): the partial results folded in block order"""
    from muon_amd._backend import _dt, _p, check

    n_own, n_other = Eo.shape[0], Et.shape[0]
    if blk is None:
        blk = int(be.lib.mu_mofa_poisson_blocks_for(_dt(Eo), mode, K, n_own, n_other))
    assert blk >= 128 and blk % 128 == 0
    nb = -(-n_other // blk)
    part = torch.full((nb, n_own) if mode == 2 else (nb, n_own, K + 1 if mode == 3 else K), float("nan"),
                      dtype=Eo.dtype, device=Eo.device)
    check(be.lib.mu_mofa_poisson_dense(_dt(Eo), mode, n_own, n_other, K, blk, _p(Eo), _p(Et),
                                       _p(kap) if kap is not None else None, _p(part), be._stream()))
    return part.sum(dim=0)


def _pois_dense_reference(mode, Eo, Et, kap, K):
    """the element-wise definition in numpy f64 (include/muon_amd.h): own x other predictions, y = 0 everywhere"""
    zeta = Eo[:, :K] @ Et[:, :K].T
    sig = 1.0 / (1.0 + np.exp(-zeta))
    sp_ = np.logaddexp(0.0, zeta)
    if mode == 0:
        return (kap[None, :] * zeta - sig) @ Et[:, :K]
    if mode == 2:
        return -sp_.sum(axis=1)
    b = (kap[:, None] * zeta - sig) @ Et[:, :K]
    return b if mode == 1 else np.concatenate([b, -sp_.sum(axis=1, keepdims=True)], axis=1)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("K", [1, 4, 5, 8, 10, 13, 16])
@pytest.mark.parametrize("shape", [(1, 1), (17, 15), (255, 16), (257, 17), (700, 127), (64, 129), (300, 1300)])
def test_poisson_dense_sweep_on_the_matrix_cores(K, shape, dt, tol):
    """k_pois_mfma (r06: the dense sweep of an f32 model as two small matrix products around the transform, csrc/
    mofa_poisson.hip) against the element-wise definition in f64 AND against the vector kernel it replaces (tune key
    pois_valu): every mode, own / other sizes off the 16-row tiles, the 128-row stages and the 256-row workgroups, every
    padded width (K = 1 .. 16 -> KP = 4, 8, 12, 16), f32 and f64 models (the f64 matrix cores, later in r06), several column blocks (their last one partial: the padding rows'
    share of the likelihood sum is taken off after the loop)."""
    from muon_amd._backend import get_backend

    be = get_backend()
    n_own, n_other = shape
    rng = np.random.default_rng(1000 * K + n_own + n_other)
    KP = next(k for k in (4, 8, 12, 16) if k >= K)
    Eo = np.zeros((n_own, KP)); Eo[:, :K] = rng.standard_normal((n_own, K)) * 0.8
    Et = np.zeros((n_other, KP)); Et[:, :K] = rng.standard_normal((n_other, K)) * 0.6
    dev = lambda a: torch.from_numpy(a).to(be.device).to(dt).contiguous()
    Eod, Etd = dev(Eo), dev(Et)
    try:
        for mode in (0, 1, 2, 3):
            kap = 0.25 + rng.random(n_other if mode == 0 else n_own)
            kd = None if mode == 2 else dev(kap)
            want = _pois_dense_reference(mode, Eo, Et, kap, K)
            scale = max(1.0, float(np.max(np.abs(want))))
            got = {}
            for valu in (0, 1):
                be.lib.mu_tune_set(b"pois_valu", valu)
                for blk in (None, 128):
                    g = be.to_host(_pois_dense_sweep(be, mode, Eod, Etd, kd, K, blk)).astype(np.float64)
                    assert g.shape == want.shape and np.all(np.isfinite(g))
                    assert np.max(np.abs(g - want)) <= tol * scale, (mode, valu, blk)
                    got[valu, blk] = g
            assert np.max(np.abs(got[0, None] - got[1, None])) <= 0.5 * tol * scale
            if mode == 3:  # mode 1 and mode 3 give the same b, bit for bit
                be.lib.mu_tune_set(b"pois_valu", 0)
                b1 = _pois_dense_sweep(be, 1, Eod, Etd, kd, K)
                b3 = _pois_dense_sweep(be, 3, Eod, Etd, kd, K)
                assert torch.equal(b1, b3[:, :K].contiguous())
    finally:
        be.lib.mu_tune_set(b"pois_valu", 0)


def test_poisson_dense_sweep_with_extreme_predictions():
    """predictions far outside anything a fit produces (|zeta| up to ~300): the hardware exp2 / log2 / rcp of the
    matrix-core sweep must stay finite and right - sigmoid 0 / 1, softplus 0 / zeta"""
    from muon_amd._backend import get_backend

    be = get_backend()
    K, KP, n_own, n_other = 3, 4, 40, 50
    rng = np.random.default_rng(5)
    Eo = np.zeros((n_own, KP)); Eo[:, :K] = rng.standard_normal((n_own, K)) * 12
    Et = np.zeros((n_other, KP)); Et[:, :K] = rng.standard_normal((n_other, K)) * 12
    assert np.max(np.abs(Eo[:, :K] @ Et[:, :K].T)) > 200
    dev = lambda a: torch.from_numpy(a).to(be.device).to(torch.float32).contiguous()
    Eo32, Et32 = Eo.astype(np.float32).astype(np.float64), Et.astype(np.float32).astype(np.float64)
    for mode in (0, 1, 2, 3):
        kap = 0.25 + rng.random(n_other if mode == 0 else n_own)
        want = _pois_dense_reference(mode, Eo32, Et32, kap.astype(np.float32).astype(np.float64), K)
        g = be.to_host(_pois_dense_sweep(be, mode, dev(Eo), dev(Et), None if mode == 2 else dev(kap), K)).astype(np.float64)
        assert np.all(np.isfinite(g))
        assert np.max(np.abs(g - want)) <= 1e-5 * np.max(np.abs(want)), mode


def test_poisson_block_rule_fills_whole_rounds():
    """mu_mofa_poisson_blocks_for: a multiple of the 128-row stage, never more column blocks than stages, and the
    workgroups of the sweep fit the places the kernel's occupancy gives (or come in whole rounds of them)"""
    from muon_amd._backend import get_backend

    be = get_backend()
    for dt in (0, 1):
        for mode in (0, 1, 2, 3):
            for K in (3, 10, 16, 24):
                for n_own, n_other in ((20000, 20000), (1, 1), (100, 100000), (100000, 100), (513, 129)):
                    blk = int(be.lib.mu_mofa_poisson_blocks_for(dt, mode, K, n_own, n_other))
                    assert blk >= 128 and blk % 128 == 0
                    assert blk <= 128 * -(-n_other // 128)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(1, 1), (37, 5), (1000, 1024), (4099, 2052), (513, 1027)])
def test_dense_column_moments_kernel(hip, dtype, shape):
    """mu_dense_col_moments (the group means / sums of squares of a fit's set-up) against numpy f64 on the same values:
    row ranges off the chunking, column counts off the vector width, empty ranges."""
    n, D = shape
    g = torch.Generator(device="cuda").manual_seed(n + D)
    Y = (torch.randn((n, D), generator=g, device="cuda", dtype=torch.float64) * 3 + 1).to(dtype)
    host = hip.to_host(Y).astype(np.float64)
    for a, b in ((0, n), (n // 3, n - n // 5), (n // 2, n // 2)):
        s1, s2 = hip.col_moments(Y, a, b)
        assert s1.dtype == torch.float64 and s1.shape == (D,)
        w1, w2 = host[a:b].sum(axis=0), (host[a:b] ** 2).sum(axis=0)
        assert np.max(np.abs(hip.to_host(s1) - w1)) <= 1e-13 * max(1.0, np.max(np.abs(host)) * (b - a))
        assert np.max(np.abs(hip.to_host(s2) - w2)) <= 1e-13 * max(1.0, np.max(host ** 2) * (b - a))
    again = hip.col_moments(Y, 0, n)
    assert torch.equal(again[0], hip.col_moments(Y, 0, n)[0])  # fixed-order fold: bit-reproducible


def test_device_resident_dense_view_is_not_modified_and_centred_once(hip):
    """A device-resident dense view of the fit's own type is read in place (no copy, one centring pass): the caller's
    tensor keeps its values and the fit equals the fit of a host copy of it."""
    from muon_amd._core.mofa_engine import MofaEngine

    n, D = 600, 96
    g = torch.Generator(device="cuda").manual_seed(3)
    Y = torch.randn((n, D), generator=g, device="cuda", dtype=torch.float64) + 2.0
    keep = Y.clone()
    groups = np.zeros(n, dtype=np.int64)
    a = MofaEngine(hip, [Y], groups, 4, dtype=torch.float64, seed=1)
    b = MofaEngine(hip, [hip.to_host(keep) * (1.0 + 2.0 ** -30)], groups, 4, dtype=torch.float64, seed=1)  # (not exact in f32)
    assert torch.equal(Y, keep)
    for _ in range(3):
        a.step()
        b.step()
    assert abs(a.elbo[-1] - b.elbo[-1]) <= 1e-6 * abs(b.elbo[-1])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_general_engine_replayed_iteration_equals_eager(hip, dt, monkeypatch):
    """The general engine's iteration as a HIP graph (MUON_AMD_MOFA_NG_GRAPH=1: every expectation updated in place;
    captured after two eager iterations) against the same fit run eagerly: the same numbers, iteration by iteration - gaussian view with NaN,
    fused sparse poisson view, bernoulli view, two groups."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from tests.test_mofa_host import _mixed_views

    _, y1, y2, y3 = _mixed_views(n=500, seed=4)
    groups = np.random.default_rng(2).integers(0, 2, 500)
    liks = ["gaussian", "poisson", "bernoulli"]
    runs = []
    for graph in ("0", "1"):
        monkeypatch.setenv("MUON_AMD_MOFA_NG_GRAPH", graph)
        eng = GeneralMofaEngine(hip, [y1, sp.csr_matrix(y2), y3], liks, groups, 5, seed=1, dtype=dt)
        for _ in range(7):
            eng.step()
        assert (eng._graph is not None) == (graph == "1")
        runs.append((list(eng.elbo), eng.results(sort_factors=False)))
    (e0, r0), (e1, r1) = runs
    assert e0 == e1
    assert np.array_equal(r0["Z"], r1["Z"]) and all(np.array_equal(a, b) for a, b in zip(r0["W"], r1["W"]))


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-9), (torch.float32, 5e-3)])
def test_spikeslab_factors_on_the_gpu_match_the_oracle(hip, dt, tol):
    """r06: spikeslab_factors=True on the device - the factor sweep is the W form of
    mu_mofa_gs_update with one (alpha, theta) pair per (group, factor) - against oracle.run_general, mixed likelihoods,
    two groups."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from tests.test_mofa_host import _mixed_views

    _, y1, y2, y3 = _mixed_views(n=400, seed=2)
    groups = np.random.default_rng(1).integers(0, 2, 400)
    liks = ["gaussian", "poisson", "bernoulli"]
    ref = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=5, n_iterations=8,
                                  convergence_mode="slow", min_iterations=100, spikeslab_factors=True)
    eng = GeneralMofaEngine(hip, [y1, sp.csr_matrix(y2), y3], liks, groups, 5, seed=1, dtype=dt, chunk_elems=6000,
                            spikeslab_factors=True)
    eng.run(8, "slow", min_iterations=100)
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=tol)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=max(tol * 10, 1e-8))


def test_wrapper_fits_48_factors_and_90_stacked_columns_on_the_gpu():
    """n_factors = 48 (tools.py:298 takes any int; the two-pass engine's kernels stop at 32) and 9 groups x 10 factors
    against a sparse modality (its stacked products stop at 64 columns): the general engine, against the gaussian
    oracle iteration by iteration."""
    import pandas as pd

    rng = np.random.default_rng(5)
    n = 600
    Z = rng.standard_normal((n, 4))
    y1 = Z @ rng.standard_normal((4, 300)) + 0.5 * rng.standard_normal((n, 300))
    y2 = Z @ rng.standard_normal((4, 500)) + 0.5 * rng.standard_normal((n, 500))
    ref = mofa_oracle.run([y1, y2], n_factors=48, n_iterations=4, convergence_mode="slow", min_iterations=100)
    md = MuData({"a": AnnData(y1), "b": AnnData(y2)})
    mu.tl.mofa(md, n_factors=48, n_iterations=4, convergence_mode="slow", quiet=True)
    assert md.obsm["X_mofa"].shape == (n, 48)
    np.testing.assert_allclose(md.uns["mofa"]["elbo"][:4], ref["elbo"][:4], rtol=1e-9)
    y2s = np.where(rng.random(y2.shape) < 0.2, y2, 0.0)
    grp = np.array([i % 9 for i in range(n)])
    ref = mofa_oracle.run([y1, y2s], groups=grp, n_factors=10, n_iterations=4, convergence_mode="slow", min_iterations=100)
    md = MuData({"a": AnnData(y1), "s": AnnData(sp.csr_matrix(y2s))})
    md.obs["grp"] = pd.Categorical([f"g{i}" for i in grp])
    for m in md.mod.values():
        m.obs["grp"] = md.obs["grp"].values
    mu.tl.mofa(md, n_factors=10, groups_label="grp", n_iterations=4, convergence_mode="slow", quiet=True)
    np.testing.assert_allclose(md.uns["mofa"]["elbo"][:4], ref["elbo"][:4], rtol=1e-8)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_multi_rank_iteration_as_two_graph_segments(hip, dt, tol):
    """r06: with several ranks an iteration is two captured segments around ONE packed all-reduce (mofa_engine._seg_a /
    _seg_b).  A stand-in communicator that claims two ranks and sums nothing (the partner holds no samples) must
    reproduce the single-process trace: same kernels, same operands; sparse + dense view, two groups."""
    from muon_amd._comm import LocalComm

    class LonelyPair(LocalComm):
        world_size = 2
        calls = 0

        def all_reduce_sum(self, *tensors):
            LonelyPair.calls += 1
            return tensors[0] if len(tensors) == 1 else tensors

    rng = np.random.default_rng(4)
    n = 3000
    Z = rng.standard_normal((n, 4))
    y1 = (Z @ rng.standard_normal((4, 320)) + 0.5 * rng.standard_normal((n, 320))).astype(np.float32)
    y2 = sp.random(n, 900, density=0.05, format="csr", random_state=rng, dtype=np.float32)
    groups = rng.integers(0, 2, n)
    ref = MofaEngine(hip, [y1, y2], groups, 6, dtype=dt, seed=1)
    seg = MofaEngine(hip, [y1, y2], groups, 6, dtype=dt, seed=1, comm=LonelyPair())
    for _ in range(10):
        ref.step()
        seg.step()
    assert seg._seg_graphs is not None and seg._graph is None
    before = LonelyPair.calls
    seg.step()
    ref.step()
    # one message per iteration: the moments, every view's B and - in an f64 fit - the factors' sums are ONE flat tensor;
    # an f32 fit sends the factors' f64 sums (2 G K doubles) as a second, tiny one
    assert LonelyPair.calls == before + (1 if dt == torch.float64 else 2)
    np.testing.assert_allclose(seg.elbo, ref.elbo, rtol=tol)
    a, b = seg.results(sort_factors=False), ref.results(sort_factors=False)
    np.testing.assert_allclose(a["Z"], b["Z"], atol=1e-9 if dt == torch.float64 else 1e-3)

"""MOFA without a GPU: (a) the oracle against what the reference's tests pin at the mofapy2
boundary (tests/test_muon_tools.py:25-44 structure; monotone ELBO), (b) the engine's host
logic (statistics, implicit centring, groups, missing samples) against the oracle through the
CPU test operator set, (c) the wrapper's signature, validation and write-back."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import muon_amd as mu
from muon_amd import AnnData, MuData
from muon_amd._core.mofa_engine import MofaEngine
from oracle import mofa_oracle
from tests.cpu_backend import CpuTestBackend

BE = CpuTestBackend()


def simple_views():
    # /root/reference/tests/test_muon_tools.py:15-23
    np.random.seed(1000)
    z = np.random.normal(size=(100, 5))
    w1 = np.random.normal(size=(90, 5))
    w2 = np.random.normal(size=(50, 5))
    e1 = np.random.normal(size=(100, 90))
    e2 = np.random.normal(size=(100, 50))
    return np.dot(z, w1.T) + e1, np.dot(z, w2.T) + e2


def r2_per_factor(y, Z, W):
    return [1 - np.sum((y - Z[:, [i]] @ W[:, [i]].T) ** 2) / np.sum(y ** 2) for i in range(Z.shape[1])]


def test_oracle_structure_and_monotone_elbo():
    y1, y2 = simple_views()
    r = mofa_oracle.run([y1, y2], n_factors=10, n_iterations=1000)
    e = np.array(r["elbo"])
    assert np.all(np.diff(e) > -1e-8 * abs(e[0])), "ELBO must not decrease"
    y = np.concatenate([y1, y2], axis=1)
    r2 = sorted(r2_per_factor(y, r["Z"], np.concatenate(r["W"])), reverse=True)
    assert all(v > 0.1 for v in r2[:5]) and not any(v > 0.1 for v in r2[5:])


@pytest.mark.parametrize("case", ["dense", "groups", "sparse_missing", "noard_nospike"])
def test_engine_matches_oracle_iteration_by_iteration(case):
    y1, y2 = simple_views()
    rng = np.random.default_rng(0)
    groups = None
    views = [y1, y2]
    kw = {}
    if case == "groups":
        groups = rng.integers(0, 3, 100)
    if case == "sparse_missing":
        groups = rng.integers(0, 2, 100)
        y1 = y1.copy(); y1[85:] = np.nan
        y2 = y2.copy(); y2[np.abs(y2) < 1.0] = 0
        views = [y1, sp.csr_matrix(y2)]
    if case == "noard_nospike":
        kw = dict(ard_weights=False, ard_factors=False, spikeslab_weights=False)
    dense = [v.toarray() if sp.issparse(v) else v for v in views]
    ref = mofa_oracle.run(dense, groups=groups, n_factors=8, n_iterations=25, convergence_mode="slow", **kw)
    eng = MofaEngine(BE, views, np.zeros(100, dtype=int) if groups is None else groups, 8, seed=1, **kw)
    eng.run(25, "slow")
    res = eng.results(sort_factors=False)
    n = min(len(ref["elbo"]), len(res["elbo"]))
    np.testing.assert_allclose(res["elbo"][:n], ref["elbo"][:n], rtol=1e-10)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-9)
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-9)
    np.testing.assert_allclose(res["r2"], ref["r2"], atol=1e-7)


@pytest.mark.parametrize("case", ["plain", "groups_missing", "offset"])
def test_f32_valued_dense_views_are_centred_inside_the_products(case, monkeypatch):
    """r04: under the f64 fit a dense view whose values are exact in f32 stays f32 and is centred inside the
    products (mofa_engine._f32_storage_ok), like the sparse views.  Host logic on the CPU operator set against the
    oracle, which centres first; the GPU twin is tests/test_gpu_mofa.py."""
    import torch
    from muon_amd._core.mofa_engine import MofaEngine
    from tests.cpu_backend import CpuTestBackend

    rng = np.random.default_rng(12)
    N, K0 = 300, 4
    Z = rng.standard_normal((N, K0))
    W1 = rng.standard_normal((96, K0)) * (rng.random((96, K0)) < 0.4)
    W2 = rng.standard_normal((50, K0)) * (rng.random((50, K0)) < 0.4)
    y1 = (Z @ W1.T + rng.standard_normal((N, 96))).astype(np.float32).astype(np.float64)
    y2 = (Z @ W2.T + rng.standard_normal((N, 50))).astype(np.float32).astype(np.float64)
    groups = np.zeros(N, dtype=int)
    if case == "groups_missing":
        groups = rng.integers(0, 3, N)
        y1[250:] = np.nan
    if case == "offset":
        y1 = (y1 + 40.0).astype(np.float32).astype(np.float64)
    ref = mofa_oracle.run([y1, y2], groups=groups if case == "groups_missing" else None, n_factors=5, n_iterations=15,
                          convergence_mode="slow")
    eng = MofaEngine(CpuTestBackend(), [y1, y2], groups, 5, seed=1)
    assert all(v.Y.dtype == torch.float32 and v.implicit for v in eng.views)
    eng.run(15, "slow")
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"][:len(res["elbo"])], rtol=1e-8 if case != "offset" else 1e-7)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-6)
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-6)
    monkeypatch.setenv("MUON_AMD_MOFA_F32_STORAGE", "0")
    eng64 = MofaEngine(CpuTestBackend(), [y1, y2], groups, 5, seed=1)
    assert all(v.Y.dtype == torch.float64 and not getattr(v, "implicit", False) for v in eng64.views)


def test_scaling_options_match_oracle():
    y1, y2 = simple_views()
    y2 = y2 * 7.0
    groups = np.random.default_rng(1).integers(0, 2, 100)
    for kw in (dict(scale_views=True), dict(scale_groups=True), dict(center_groups=False),
               dict(scale_views=True, scale_groups=True), dict(center_groups=False, scale_views=True),
               dict(center_groups=False, scale_groups=True, scale_views=True)):
        ref = mofa_oracle.run([y1, y2], groups=groups, n_factors=6, n_iterations=10, convergence_mode="slow", **kw)
        eng = MofaEngine(BE, [y1, sp.csr_matrix(y2)], groups, 6, seed=1, **kw)
        eng.run(10, "slow")
        np.testing.assert_allclose(eng.elbo, ref["elbo"], rtol=1e-9)


class TestWrapperLikeReference:
    """Mirrors /root/reference/tests/test_muon_tools.py:12-87 on the duck containers."""

    def setup_method(self):
        y1, y2 = simple_views()
        self.mdata = MuData({"y1": AnnData(y1), "y2": AnnData(y2)})

    def test_mofa_nfactors(self, tmp_path):
        n_factors = 10
        mu.tl.mofa(self.mdata, n_factors=n_factors, quiet=True, verbose=False,
                   outfile=str(tmp_path / "m.hdf5"), backend=BE)
        y = np.concatenate([self.mdata.mod["y1"].X, self.mdata.mod["y2"].X], axis=1)
        r2 = r2_per_factor(y, self.mdata.obsm["X_mofa"], self.mdata.varm["LFs"])
        # Only first 5 factors should have high R2
        assert all(i > 0.1 for i in r2[:5])
        assert not any(i > 0.1 for i in r2[5:])
        # no h5py in this image: the arrays go to a NumPy archive whose name says so
        try:
            import h5py  # noqa: F401
            assert (tmp_path / "m.hdf5").exists()
        except ImportError:
            assert (tmp_path / "m.hdf5.npz").exists() and not (tmp_path / "m.hdf5").exists()
            with np.load(tmp_path / "m.hdf5.npz") as f:
                # the keys are mofapy2's HDF5 dataset paths, read back the way the reference does (tools.py:608-629)
                groups = [k.split("/")[-1] for k in f.files if k.startswith("expectations/Z/")]
                z = np.concatenate([f[f"expectations/Z/{g}"] for g in groups], axis=1).T
                w = np.concatenate([f[f"expectations/W/{m}"] for m in ("y1", "y2")], axis=1).T
                assert z.shape == (100, n_factors) and w.shape == (self.mdata.varm["LFs"].shape[0], n_factors)
                np.testing.assert_allclose(z, self.mdata.obsm["X_mofa"])
                np.testing.assert_allclose(w, self.mdata.varm["LFs"])
                assert f[f"variance_explained/r2_per_factor/{groups[0]}"].shape == (2, n_factors)
                assert f["training_stats/elbo"].ndim == 1 and [v.decode() for v in f["views/views"]] == ["y1", "y2"]
        u = self.mdata.uns["mofa"]
        assert u["params"]["model"]["n_factors"] == 10 and set(u["variance"]) == {"y1", "y2"}
        assert u["variance"]["y1"].shape == (10,)

    def test_mofa_anndata(self):
        a = self.mdata["y1"]
        mu.tl.mofa(a, n_factors=10, quiet=True, verbose=False, backend=BE)
        assert "X_mofa" in a.obsm and "LFs" in a.varm
        assert a.obsm["X_mofa"].shape == (100, 10) and a.varm["LFs"].shape == (90, 10)

    def test_mofa_anndata_groups_cat(self):
        adata = self.mdata["y1"].copy()
        np.random.seed(3)
        adata.obs["ab"] = np.random.choice(["a", "b"], adata.n_obs)
        adata.obs["ab"] = adata.obs.ab.astype("category")
        mu.tl.mofa(adata, groups_label="ab", n_factors=10, quiet=True, verbose=False, backend=BE)
        assert "X_mofa" in adata.obsm and "LFs" in adata.varm
        assert set(adata.uns["mofa"]["variance"]["data"]) == {"a", "b"}

    def test_mofa_obs_union(self):
        y1 = self.mdata["y1"]
        y2 = self.mdata["y2"]
        for sparsity in (0, 1, 2):
            if sparsity == 0 or sparsity == 2:
                y1.X = sp.csr_matrix(y1.X)
            if sparsity == 1 or sparsity == 2:
                y2.X = sp.csr_matrix(y2.X)
            a, b = y1[:-10], y2[10:]
            a._init_as_actual(); b._init_as_actual()
            mdata = MuData({"y1": a, "y2": b})
            mu.tl.mofa(mdata, n_factors=10, quiet=True, verbose=False, use_obs="union",
                       likelihoods="gaussian", backend=BE)
            assert mdata.obsm["X_mofa"].shape == (100, 10) and mdata.varm["LFs"].shape == (140, 10)
            assert np.all(np.isfinite(mdata.obsm["X_mofa"]))

    def test_mofa_obs_intersection_and_errors(self):
        a, b = self.mdata["y1"][:-10], self.mdata["y2"][10:]
        a._init_as_actual(); b._init_as_actual()
        mdata = MuData({"y1": a, "y2": b})
        with pytest.raises(IndexError):
            mu.tl.mofa(mdata, backend=BE)
        with pytest.raises(ValueError):
            mu.tl.mofa(mdata, use_obs="bogus", backend=BE)
        mu.tl.mofa(mdata, use_obs="intersection", n_factors=5, backend=BE)
        xm = mdata.obsm["X_mofa"]
        assert xm.shape == (100, 5) and np.isnan(xm[:10]).all() and np.isnan(xm[90:]).all()
        assert np.isfinite(xm[10:90]).all()
        with pytest.raises(TypeError):
            mu.tl.mofa(np.ones((3, 3)), backend=BE)
        with pytest.raises(ValueError):
            mu.tl.mofa(self.mdata, groups_label="nope", backend=BE)
        with pytest.raises(NotImplementedError):
            mu.tl.mofa(self.mdata, svi_mode=True, backend=BE)
        # count data with the reference's default likelihoods=None: guessed poisson (tools.py:272-280)
        # and fitted as poisson (r02 fell back to gaussian with a warning)
        counts = MuData({"c": AnnData(np.random.default_rng(0).poisson(2, size=(30, 8)).astype(float))})
        mu.tl.mofa(counts, n_factors=2, n_iterations=5, quiet=True, backend=BE)
        assert counts.obsm["X_mofa"].shape == (30, 2)
        assert list(counts.uns["mofa"]["params"]["data"]["likelihoods"]) == ["poisson"]
        mu.tl.mofa(counts, likelihoods="poisson", n_factors=2, n_iterations=3, backend=BE)

    def test_use_var_subset_zero_fills(self):
        self.mdata.mod["y1"].var["highly_variable"] = np.arange(90) % 2 == 0
        self.mdata.mod["y2"].var["highly_variable"] = np.arange(50) % 5 != 0
        self.mdata.update()
        cp = mu.tl.mofa(self.mdata, n_factors=4, copy=True, backend=BE)
        assert "X_mofa" not in self.mdata.obsm and cp.obsm["X_mofa"].shape == (100, 4)
        lf = cp.varm["LFs"]
        sel = np.concatenate([np.arange(90) % 2 == 0, np.arange(50) % 5 != 0])
        assert lf.shape == (140, 4) and np.all(lf[~sel] == 0) and np.any(lf[sel] != 0)
        with pytest.warns(UserWarning, match="There is no column"):
            mu.tl.mofa(self.mdata, n_factors=3, use_var="absent", backend=BE)


# ---- non-gaussian likelihoods and element-wise missing values ------------------------------
def _mixed_views(n=150, seed=0):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, 3))
    W1, W2, W3 = (rng.standard_normal((d, 3)) for d in (40, 60, 50))
    y1 = Z @ W1.T + 0.5 * rng.standard_normal((n, 40))
    y2 = rng.poisson(np.logaddexp(0, Z @ W2.T + 1.0)).astype(float)
    y3 = (rng.random((n, 50)) < 1 / (1 + np.exp(-(Z @ W3.T)))).astype(float)
    y1[rng.random(y1.shape) < 0.1] = np.nan  # element-wise missing values in the gaussian view
    return Z, y1, y2, y3


@pytest.mark.parametrize("kw", [{}, {"scale_views": True},
                                {"center_groups": False, "ard_weights": False, "spikeslab_weights": False}])
def test_general_engine_matches_oracle(kw):
    """Chunked torch engine (sparse poisson view stays CSR, 3 chunk passes per iteration) against the
    dense numpy restatement, iteration by iteration: gaussian view with NaN entries + poisson +
    bernoulli, two groups."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from oracle import mofa_oracle

    _, y1, y2, y3 = _mixed_views()
    groups = np.random.default_rng(1).integers(0, 2, 150)
    liks = ["gaussian", "poisson", "bernoulli"]
    ref = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=5, n_iterations=8,
                                  convergence_mode="slow", min_iterations=100, **kw)
    eng = GeneralMofaEngine(BE, [y1, sp.csr_matrix(y2), y3], liks, groups, 5, seed=1, chunk_elems=1500, **kw)
    eng.run(8, "slow", min_iterations=100)
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=1e-10)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-9)
    for a, b in zip(res["W"], ref["W"]):
        np.testing.assert_allclose(a, b, atol=1e-9)
    np.testing.assert_allclose(res["r2"], ref["r2"], atol=1e-7)
    e = np.asarray(ref["elbo"])
    assert np.all(np.diff(e) > -1e-8 * abs(e[0]))  # the bounds are refreshed consistently: monotone


def test_general_oracle_equals_gaussian_oracle_on_complete_data():
    """run_general with a gaussian likelihood and no missing entry is run(): the element-wise
    formulation reduces to the sufficient-statistics one."""
    from oracle import mofa_oracle

    y1, y2 = simple_views()
    a = mofa_oracle.run([y1, y2], n_factors=6, n_iterations=8, convergence_mode="slow", min_iterations=100)
    b = mofa_oracle.run_general([y1, y2], ["gaussian", "gaussian"], n_factors=6, n_iterations=8,
                                convergence_mode="slow", min_iterations=100)
    np.testing.assert_allclose(a["elbo"], b["elbo"], rtol=1e-12)
    np.testing.assert_allclose(a["Z"], b["Z"], atol=1e-11)


def test_wrapper_fits_guessed_count_likelihoods_and_recovers_planted_factors():
    """mu.tl.mofa with the reference's default likelihoods=None on count / binary / real-valued
    modalities (guess: poisson / bernoulli / gaussian, tools.py:272-280): the three planted factors
    span the leading learnt factors."""
    from scipy.linalg import subspace_angles

    Z, y1, y2, y3 = _mixed_views(n=300, seed=3)
    md = MuData({"rna": AnnData(np.nan_to_num(y1)), "counts": AnnData(sp.csr_matrix(y2)), "acc": AnnData(y3)})
    mu.tl.mofa(md, n_factors=6, n_iterations=60, convergence_mode="slow", quiet=True, backend=BE)
    assert list(md.uns["mofa"]["params"]["data"]["likelihoods"]) == ["gaussian", "poisson", "bernoulli"]
    X = md.obsm["X_mofa"]
    assert X.shape == (300, 6) and md.varm["LFs"].shape == (150, 6)
    # (count views are not centred - process_data centres gaussian views only -, so one learnt
    #  factor carries the intercept of the poisson view: the planted three lie in the leading four)
    ang = np.rad2deg(subspace_angles(Z, X[:, :4]))
    assert ang.max() < 15.0, ang
    e = md.uns["mofa"]["elbo"]
    assert np.all(np.diff(e) > -1e-7 * abs(e[0]))


def test_wrapper_elementwise_nan_in_a_dense_modality():
    y1, y2 = simple_views()
    y1 = y1.copy()
    y1[np.random.default_rng(0).random(y1.shape) < 0.05] = np.nan
    md = MuData({"y1": AnnData(y1), "y2": AnnData(y2)})
    mu.tl.mofa(md, n_factors=8, n_iterations=30, quiet=True, backend=BE)
    assert np.all(np.isfinite(md.obsm["X_mofa"])) and md.obsm["X_mofa"].shape == (100, 8)


def test_multi_group_with_shuffled_sample_names_like_the_reference_test():
    """Reference tests/test_muon_tools.py:91-147 (TestMOFA2D.test_multi_group) without its two golden
    values (they depend on mofapy2's private RNG stream): two groups, shuffled sample names,
    `groups_label`: the group column survives, X_mofa rows follow mdata.obs order and equal the
    oracle's factors for the same data."""
    from oracle import mofa_oracle

    n_g1, n_g2, d_m1, d_m2, k = 10, 20, 30, 40, 5
    n = n_g1 + n_g2
    np.random.seed(42)
    z = np.concatenate([np.random.normal(size=(n_g1, k)), np.random.normal(size=(n_g2, k))], axis=0)
    w1, w2 = np.random.normal(size=(d_m1, k)), np.random.normal(size=(d_m2, k))
    y1 = z @ w1.T + np.random.normal(size=(n, d_m1))
    y2 = z @ w2.T + np.random.normal(size=(n, d_m2))
    names = [f"sample{i}_group{g}" for g, sz in {"A": n_g1, "B": n_g2}.items() for i in range(sz)]
    np.random.shuffle(names)
    groups = [s.split("_")[1] for s in names]
    mdata = MuData({"view1": AnnData(y1, obs=pd.DataFrame(index=names)), "view2": AnnData(y2, obs=pd.DataFrame(index=names))})
    mdata.obs = mdata.obs.join(pd.DataFrame({"sample": names, "group": groups}, index=names))
    mu.tl.mofa(mdata, groups_label="group", n_factors=6, n_iterations=40, convergence_mode="slow", quiet=True, backend=BE)
    assert all(mdata.obs.group.values == [s.split("_")[1] for s in mdata.obs["sample"]])
    codes = pd.Index(pd.unique(np.asarray(groups))).get_indexer(groups)
    ref = mofa_oracle.run([y1, y2], groups=codes, n_factors=6, n_iterations=40, convergence_mode="slow")
    order = np.argsort(-ref["r2"].sum(axis=(0, 1)), kind="stable")
    np.testing.assert_allclose(mdata.obsm["X_mofa"], ref["Z"][:, order], atol=1e-7)
    assert set(mdata.uns["mofa"]["variance"]["view1"]) == set(pd.unique(np.asarray(groups)))


# ---- the reference's own data assembly, executed (tests/golden/make_mofa_golden.py) ------------------------------
def _prep_case(g, tag):
    import scipy.sparse as sp

    from muon_amd import AnnData, MuData

    mods = {}
    for m in ("rna", "atac"):
        if f"{tag}_{m}_X" in g.files:
            x = g[f"{tag}_{m}_X"].copy()
        else:
            x = sp.csr_matrix((g[f"{tag}_{m}_X_data"], g[f"{tag}_{m}_X_indices"], g[f"{tag}_{m}_X_indptr"]),
                              shape=tuple(g[f"{tag}_{m}_X_shape"]))
        a = AnnData(x)
        a.obs_names = g[f"{tag}_{m}_obs_names"]
        if f"{tag}_{m}_layer_lognorm" in g.files:
            lay = g[f"{tag}_{m}_layer_lognorm"]
            a.layers["lognorm"] = sp.csr_matrix(lay) if int(g[f"{tag}_{m}_layer_lognorm_sparse"][0]) else lay.copy()
        if f"{tag}_{m}_hv" in g.files:
            a.var["highly_variable"] = g[f"{tag}_{m}_hv"]
        mods[m] = a
    md = MuData(mods)
    assert list(md.obs.index.values) == list(g[f"{tag}_obs_names"])
    if f"{tag}_grp" in g.files:
        md.obs["grp"] = g[f"{tag}_grp"]
    return md


@pytest.mark.parametrize("tag,kw", [
    ("groups", dict(groups_label="grp")),
    ("union", dict(use_obs="union")),
    ("intersection", dict(use_obs="intersection")),
    ("subset_layer", dict(use_layer="lognorm", features_subset="highly_variable")),
])
def test_data_assembly_against_the_reference_executing(golden_dir, tag, kw):
    """`_collect_views` against what /root/reference/muon/_core/tools.py:50-287 itself hands to mofapy2 (process_data
    replaced by the identity): the matrices (densified there, sparse here), the union expansion with missing samples,
    the feature subset, the group order of first appearance, the names and the per-group intercepts"""
    import os

    import scipy.sparse as sp

    from muon_amd._core.tools import _collect_views

    g = np.load(os.path.join(golden_dir, "mofa_prep_golden.npz"))
    md = _prep_case(g, tag)
    views, groups, group_names, obs_names, lik = _collect_views(
        md, kw.get("groups_label"), False, kw.get("use_layer"), ["gaussian", "gaussian"], kw.get("features_subset"),
        kw.get("use_obs"))
    M, G, N = (int(v) for v in g[f"{tag}_dims"][:3])
    assert len(views) == M and len(group_names) == G and len(obs_names) == N
    assert [v.shape[1] for v in views] == [int(d) for d in g[f"{tag}_dims"][3:]]
    assert list(group_names) == list(g[f"{tag}_groups_names"]) and list(lik) == list(g[f"{tag}_likelihoods"])
    # the reference orders the samples by group (first appearance); the engine keeps the order and a group id per sample
    order = np.concatenate([np.nonzero(groups == gi)[0] for gi in range(G)])
    assert list(np.asarray(obs_names)[order]) == list(g[f"{tag}_samples_names"])
    assert list(np.asarray(group_names)[groups[order]]) == list(g[f"{tag}_samples_groups"])
    assert [int((groups == gi).sum()) for gi in range(G)] == [int(v) for v in g[f"{tag}_samples_per_group"]]
    for m, v in enumerate(views):
        if sp.issparse(v):
            d = v.toarray().astype(np.float64)
            miss = getattr(v, "_missing_rows", None)
            if miss is not None:
                d[miss] = np.nan
        else:
            d = np.asarray(v, dtype=np.float64)
        want = g[f"{tag}_data{m}"]
        np.testing.assert_array_equal(np.isnan(d[order]), np.isnan(want))
        np.testing.assert_allclose(np.nan_to_num(d[order]), np.nan_to_num(want), rtol=0, atol=0)
        with np.errstate(invalid="ignore"):
            ic = np.stack([np.nanmean(d[groups == gi], axis=0) for gi in range(G)])
        np.testing.assert_allclose(ic, g[f"{tag}_intercepts{m}"], rtol=1e-14)


@pytest.mark.parametrize("tag,kw", [
    ("groups_subset", dict(groups_label="batch", use_var="highly_variable", n_factors=4, likelihoods="gaussian",
                           n_iterations=7, convergence_mode="medium", seed=3, scale_views=True, quiet=True)),
    ("intersection", dict(use_obs="intersection", use_var=None, n_factors=3, likelihoods=["gaussian", "gaussian"], quiet=True)),
    ("union", dict(use_obs="union", use_var=None, n_factors=3, likelihoods=["gaussian", "gaussian"], quiet=True,
                   use_float32=True)),
])
def test_wrapper_routes_options_and_writes_back_like_the_reference_executing(golden_dir, tmp_path, monkeypatch, tag, kw):
    """The reference's whole `mofa()` (tools.py:290-708) was executed around a recording stand-in for mofapy2 whose
    `save()` wrote a seeded model (tests/golden/make_mofa_golden.py).  The same model fed through muon_amd.tl.mofa's
    wrapper must leave the same .obsm["X_mofa"] (factors re-ordered by sample name, NaN outside the intersection),
    .varm["LFs"] (zero rows for unused features), .uns["mofa"] record and variance table - and the engine must be
    configured with the options the reference passes to mofapy2."""
    import os

    import scipy.sparse as sp
    import torch

    import muon_amd as mu
    from muon_amd import AnnData, MuData
    from muon_amd._core import mofa_engine

    g = np.load(os.path.join(golden_dir, "mofa_writeback_golden.npz"))
    mods = {}
    for m in ("rna", "atac"):
        if f"{tag}_{m}_X" in g.files:
            x = g[f"{tag}_{m}_X"].copy()
        else:
            x = sp.csr_matrix((g[f"{tag}_{m}_X_data"], g[f"{tag}_{m}_X_indices"], g[f"{tag}_{m}_X_indptr"]),
                              shape=tuple(g[f"{tag}_{m}_X_shape"]))
        a = AnnData(x)
        a.obs_names = g[f"{tag}_{m}_obs_names"]
        if f"{tag}_{m}_hv" in g.files:
            a.var["highly_variable"] = g[f"{tag}_{m}_hv"]
        mods[m] = a
    md = MuData(mods)
    if f"{tag}_batch" in g.files:
        md.obs["batch"] = g[f"{tag}_batch"]
        md.var["highly_variable"] = np.concatenate([mods["rna"].var["highly_variable"].values,
                                                    mods["atac"].var["highly_variable"].values])
    seen = {}

    class FakeEngine:
        def __init__(self, backend, views, groups, n_factors, **opts):
            seen.update(opts, n_factors=n_factors, groups=np.asarray(groups), shapes=[v.shape for v in views])

        def run(self, n_iterations=1000, convergence_mode="fast", **_):
            seen.update(n_iterations=n_iterations, convergence_mode=convergence_mode)

        def results(self, sort_factors=True):
            return seen["res"]

    group_names = [str(x) for x in g[f"{tag}_model_groups"]]
    # the seeded model of the fixture, in the engine's conventions: Z [N, K] in the wrapper's sample order
    from muon_amd._core.tools import _collect_views

    lik = kw["likelihoods"] if isinstance(kw["likelihoods"], list) else [kw["likelihoods"]] * 2
    _v, groups, gnames, obs_used, _l = _collect_views(md, kw.get("groups_label"), False, None, lik, kw.get("use_var"),
                                                      kw.get("use_obs"))
    assert list(gnames) == group_names
    K = kw["n_factors"]
    Z = np.full((len(obs_used), K), np.nan)
    for gname in group_names:
        names = [str(s) for s in g[f"{tag}_model_samples_{gname}"]]
        Z[[list(obs_used).index(s) for s in names]] = g[f"{tag}_model_Z_{gname}"].T
    assert not np.isnan(Z).any()
    W = [g[f"{tag}_model_W_{m}"].T for m in ("rna", "atac")]
    r2 = np.stack([g[f"{tag}_model_r2_{gname}"] for gname in group_names], axis=1)  # [M, G, K]
    seen["res"] = {"Z": Z, "W": W, "r2": r2, "elbo": [0.0]}
    monkeypatch.setattr(mofa_engine, "MofaEngine", FakeEngine)
    mu.tl.mofa(md, backend=BE, outfile=str(tmp_path / "m.hdf5"), **kw)

    np.testing.assert_array_equal(np.isnan(md.obsm["X_mofa"]), np.isnan(g[f"{tag}_X_mofa"]))
    np.testing.assert_allclose(np.nan_to_num(md.obsm["X_mofa"]), np.nan_to_num(g[f"{tag}_X_mofa"]), rtol=0, atol=0)
    np.testing.assert_allclose(md.varm["LFs"], g[f"{tag}_LFs"], rtol=0, atol=0)
    u = md.uns["mofa"]
    for key in g.files:
        pre = f"{tag}_param_"
        if key.startswith(pre):
            sect, name = key[len(pre):].split("_", 1)
            want = g[key]
            got = u["params"][sect][name]
            got = np.asarray("None" if got is None else got)
            assert got.shape == want.shape and np.all(got.astype(str) == want.astype(str)), (key, got, want)
    for m in ("rna", "atac"):
        v = u["variance"][m]
        if len(group_names) > 1:
            for gname in group_names:
                np.testing.assert_allclose(v[gname], g[f"{tag}_variance_{m}_{gname}"])
        else:
            np.testing.assert_allclose(v, g[f"{tag}_variance_{m}"])
    # what the reference told mofapy2 == what the engine was configured with
    assert bool(g[f"{tag}_call_data_scale_views"]) == seen["scale_views"]
    assert bool(g[f"{tag}_call_data_scale_groups"]) == seen["scale_groups"]
    assert bool(g[f"{tag}_call_data_center_groups"]) == seen["center_groups"]
    assert bool(g[f"{tag}_call_data_use_float32"]) == (seen["dtype"] == torch.float32)
    assert int(g[f"{tag}_call_model_factors"]) == seen["n_factors"]
    for k in ("ard_factors", "ard_weights", "spikeslab_weights"):
        assert bool(g[f"{tag}_call_model_{k}"]) == seen[k]
    assert int(g[f"{tag}_call_train_iter"]) == seen["n_iterations"]
    assert str(g[f"{tag}_call_train_convergence_mode"]) == seen["convergence_mode"]
    assert int(g[f"{tag}_call_train_seed"]) == seen["seed"]


# ---- model options beyond the two-pass engine's kernels (r06: VERDICT r05 item 5) --------------------------------------
@pytest.mark.parametrize("kw", [{}, {"ard_factors": False}, {"spikeslab_weights": False}])
def test_spikeslab_factors_match_the_oracle(kw):
    """spikeslab_factors=True (/root/reference/muon/_core/tools.py:305,486): the W node's spike-and-slab update on the
    factors, one (alpha, theta) per (group, factor).  General engine against oracle.run_general, iteration by iteration,
    mixed likelihoods and two groups; the ELBO is monotone (update and bound are consistent)."""
    from muon_amd._core.mofa_general import GeneralMofaEngine
    from oracle import mofa_oracle

    _, y1, y2, y3 = _mixed_views()
    groups = np.random.default_rng(1).integers(0, 2, 150)
    liks = ["gaussian", "poisson", "bernoulli"]
    ref = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=5, n_iterations=10,
                                  convergence_mode="slow", min_iterations=100, spikeslab_factors=True, **kw)
    eng = GeneralMofaEngine(BE, [y1, sp.csr_matrix(y2), y3], liks, groups, 5, seed=1, chunk_elems=1500,
                            spikeslab_factors=True, **kw)
    eng.run(10, "slow", min_iterations=100)
    res = eng.results(sort_factors=False)
    np.testing.assert_allclose(res["elbo"], ref["elbo"], rtol=1e-10)
    np.testing.assert_allclose(res["Z"], ref["Z"], atol=1e-9)
    e = np.asarray(ref["elbo"])
    assert np.all(np.diff(e) > -1e-8 * abs(e[0]))
    plain = mofa_oracle.run_general([y1, y2, y3], liks, groups=groups, n_factors=5, n_iterations=10,
                                    convergence_mode="slow", min_iterations=100, **kw)
    assert abs(plain["elbo"][-1] - ref["elbo"][-1]) > 1e-6 * abs(ref["elbo"][-1])  # (a different model, not a no-op)


def test_wrapper_takes_every_plain_mofa_option():
    """No NotImplementedError outside SVI / MEFISTO: more than 32 factors (tools.py:298 takes any int), more than 64
    stacked (group, factor) columns against a sparse modality and spikeslab_factors run on the general engine; the
    48-factor fit equals the gaussian oracle iteration by iteration."""
    from oracle import mofa_oracle

    rng = np.random.default_rng(5)
    n = 120
    Z = rng.standard_normal((n, 4))
    y1 = Z @ rng.standard_normal((4, 70)) + 0.5 * rng.standard_normal((n, 70))
    y2 = Z @ rng.standard_normal((4, 90)) + 0.5 * rng.standard_normal((n, 90))
    ref = mofa_oracle.run([y1, y2], n_factors=48, n_iterations=4, convergence_mode="slow", min_iterations=100)
    md = MuData({"a": AnnData(y1), "b": AnnData(y2)})
    mu.tl.mofa(md, n_factors=48, n_iterations=4, convergence_mode="slow", quiet=True, backend=BE)
    assert md.obsm["X_mofa"].shape == (n, 48)
    np.testing.assert_allclose(md.uns["mofa"]["elbo"][:4], ref["elbo"][:4], rtol=1e-9)
    # nine groups x ten factors = 90 stacked columns against a sparse modality
    md = MuData({"a": AnnData(y1), "s": AnnData(sp.csr_matrix(np.where(rng.random(y2.shape) < 0.2, y2, 0.0)))})
    md.obs["grp"] = pd.Categorical([f"g{i % 9}" for i in range(n)])
    for m in md.mod.values():
        m.obs["grp"] = md.obs["grp"].values
    mu.tl.mofa(md, n_factors=10, groups_label="grp", n_iterations=5, quiet=True, backend=BE)
    assert md.obsm["X_mofa"].shape == (n, 10) and np.all(np.isfinite(md.obsm["X_mofa"]))
    e = md.uns["mofa"]["elbo"]
    assert np.all(np.diff(e) > -1e-7 * abs(e[0]))
    md = MuData({"a": AnnData(y1), "b": AnnData(y2)})
    mu.tl.mofa(md, n_factors=6, spikeslab_factors=True, n_iterations=8, quiet=True, backend=BE)
    assert md.uns["mofa"]["params"]["model"]["spikeslab_factors"] is True or md.uns["mofa"]["params"]["model"]["spikeslab_factors"] == 1
    e = md.uns["mofa"]["elbo"]
    assert np.all(np.diff(e) > -1e-7 * abs(e[0]))


def test_intersection_write_back_by_name_or_by_the_reference_mask(monkeypatch):
    """use_obs="intersection" with cell names that are NOT in sorted order.  Default: every common cell gets its own
    factors back, by name.  MUON_AMD_MOFA_INTERSECTION_MASK=1: the reference's statement bit for
    bit - the model's rows, in sorted-name order, assigned through the boolean mask in data.obs order
    (/root/reference/muon/_core/tools.py:615-621) - i.e. the same rows in another place when the names are unsorted."""
    rng = np.random.default_rng(8)
    n = 40
    names = [f"c{v:02d}" for v in rng.permutation(n)]
    z = rng.standard_normal((n, 3))
    y1 = z @ rng.standard_normal((3, 25)) + 0.3 * rng.standard_normal((n, 25))
    y2 = z @ rng.standard_normal((3, 30)) + 0.3 * rng.standard_normal((n, 30))
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("MUON_AMD_MOFA_INTERSECTION_MASK", flag)
        a = AnnData(y1[:-6], obs=pd.DataFrame(index=names[:-6]))
        b = AnnData(y2[6:], obs=pd.DataFrame(index=names[6:]))
        md = MuData({"a": a, "b": b})
        mu.tl.mofa(md, use_obs="intersection", n_factors=3, n_iterations=15, quiet=True, backend=BE)
        out[flag] = (md.obsm["X_mofa"].copy(), list(md.obs.index))
    (x0, idx), (x1, _) = out["0"], out["1"]
    common = sorted(set(names[:-6]) & set(names[6:]))
    rows = {c: x0[idx.index(c)] for c in common}                       # by name (default)
    mask = np.isin(np.asarray(idx), common)
    assert np.array_equal(np.isnan(x0).all(axis=1), ~mask) and np.array_equal(np.isnan(x1).all(axis=1), ~mask)
    np.testing.assert_allclose(x1[mask], np.stack([rows[c] for c in common]), atol=1e-12)  # sorted-order rows through the mask
    assert not np.allclose(x0[mask], x1[mask])                          # ... which is another placement here

"""Host logic of tfidf / lsi without a GPU: argument validation and write-back semantics of
the reference (/root/reference/tests/test_atac_preproc.py), driven through the CPU test
operator set, plus the rule that the product path has no CPU fallback."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import muon_amd
from muon_amd import AnnData, MuData
from muon_amd import atac as ac
from oracle import lsi_oracle, tfidf_oracle
from tests.cpu_backend import CpuTestBackend
from tests.synth import planted_topics_csr

BE = CpuTestBackend()


def _dense_adata():
    np.random.seed(2020)
    return AnnData(np.abs(np.random.normal(size=(4, 5))))


def test_namespaces_match_reference():
    assert callable(muon_amd.atac.pp.tfidf)
    assert callable(muon_amd.atac.pp.binarize)
    assert callable(muon_amd.atac.tl.lsi)
    assert callable(muon_amd.tl.mofa)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from muon_amd._ffi import MuonAmdError

    with pytest.raises(MuonAmdError):
        ac.pp.tfidf(_dense_adata())
    with pytest.raises(MuonAmdError):
        ac.tl.lsi(AnnData(sp.random(30, 20, density=0.3, format="csr")))


def test_tfidf_validation_errors():
    a = _dense_adata()
    with pytest.raises(TypeError):
        ac.pp.tfidf(np.ones((3, 3)), backend=BE)
    with pytest.raises(TypeError):
        ac.pp.tfidf(MuData({"rna": a}), backend=BE)
    with pytest.raises(AttributeError):
        ac.pp.tfidf(a, log_tfidf=True, backend=BE)
    with pytest.raises(ValueError):
        ac.pp.tfidf(a, copy=True, inplace=False, backend=BE)
    with pytest.raises(ValueError):
        ac.pp.tfidf(a, to_layer="x", inplace=False, backend=BE)


def test_tfidf_writeback_semantics():
    # mirrors test_atac_preproc.py:16-52 (values themselves are GPU-checked in test_gpu_*)
    adata = _dense_adata()
    ac.pp.tfidf(adata, log_tf=True, log_idf=True, backend=BE)
    assert "%.3f" % adata.X[0, 0] == "4.659"
    assert "%.3f" % adata.X[3, 0] == "4.770"
    assert sp.issparse(adata.X)  # preproc.py:114: the result is CSR even for dense input

    base = _dense_adata()
    view = base[:, :]
    assert view.is_view
    ac.pp.tfidf(view, backend=BE)
    assert not view.is_view and "%.3f" % view.X[0, 0] == "4.659"

    adata = _dense_adata()
    orig = adata.X[0, 0]
    cp = ac.pp.tfidf(adata, copy=True, backend=BE)
    assert adata.X[0, 0] == orig and "%.3f" % cp.X[0, 0] == "4.659"

    res = ac.pp.tfidf(adata, inplace=False, backend=BE)
    assert adata.X[0, 0] == orig and "%.3f" % res[0, 0] == "4.659"

    ac.pp.tfidf(adata, to_layer="new", backend=BE)
    assert adata.X[0, 0] == orig and "%.3f" % adata.layers["new"][0, 0] == "4.659"
    with pytest.warns(UserWarning, match="will be overwritten"):
        ac.pp.tfidf(adata, to_layer="new", backend=BE)

    adata = _dense_adata()
    adata.layers["counts"] = adata.X.copy() + 1
    adata.X = None
    ac.pp.tfidf(adata, from_layer="counts", backend=BE)
    assert "%.3f" % adata.X[0, 0] == "2.856"

    m = MuData({"atac": _dense_adata()})
    ac.pp.tfidf(m, backend=BE)
    assert "%.3f" % m.mod["atac"].X[0, 0] == "4.659"


def test_match_scipy_order_reproduces_reference_layout(golden_dir):
    g = np.load(f"{golden_dir}/tfidf_golden.npz")
    x = sp.csr_matrix((g["sparse_in_data"], g["sparse_in_indices"], g["sparse_in_indptr"]), shape=(100, 10))
    res = ac.pp.tfidf(AnnData(x), inplace=False, match_scipy_order=True, backend=BE)
    assert np.array_equal(res.indices, g["sparse_out_indices"])
    assert np.array_equal(res.indptr, g["sparse_out_indptr"])
    np.testing.assert_allclose(res.data, g["sparse_out_data"], rtol=1e-12)


def test_lsi_writeback_and_parity_cpu_host_logic():
    X = planted_topics_csr(500, 300, n_topics=8, density=0.08, seed=5, dtype=np.float32)
    ad = AnnData(X)
    ac.pp.tfidf(ad, backend=BE)
    ref = lsi_oracle.lsi(ad.X, n_comps=8)
    ac.tl.lsi(ad, n_comps=8, backend=BE)
    assert ad.obsm["X_lsi"].shape == (500, 8) and ad.varm["LSI"].shape == (300, 8)
    assert ad.uns["lsi"]["stdev"].shape == (8,)
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"]) < 1e-4
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-5)
    # scaled embeddings: zero mean / unit variance per component (tools.py:60-63)
    np.testing.assert_allclose(ad.obsm["X_lsi"].mean(axis=0), 0, atol=1e-3)
    np.testing.assert_allclose(ad.obsm["X_lsi"].std(axis=0), 1, rtol=1e-3)
    with pytest.raises(TypeError):
        ac.tl.lsi(np.ones((3, 3)), backend=BE)
    with pytest.raises(ValueError):
        ac.tl.lsi(AnnData(sp.random(10, 60, density=0.5, format="csr")), n_comps=50, backend=BE)


def test_binarize():
    x = sp.random(20, 10, density=0.3, format="csr", dtype=np.float32, random_state=1)
    x.data[:] = np.arange(1, x.nnz + 1)
    x.data[3] = 0
    ad = AnnData(x)
    ac.pp.binarize(ad, backend=BE)
    exp = np.ones(x.nnz, dtype=np.float32)
    exp[3] = 0
    assert np.array_equal(ad.X.data, exp)


class _CountingBackend(CpuTestBackend):
    def __init__(self):
        super().__init__()
        self.uploads = 0

    def upload_csr(self, *a, **k):
        self.uploads += 1
        return super().upload_csr(*a, **k)


def test_binarize_tfidf_lsi_upload_once():
    # the standard workflow binarize -> tfidf -> lsi keeps the CSR on the device;
    # results equal the ones of three independent calls
    X = planted_topics_csr(400, 250, n_topics=6, density=0.08, seed=9, dtype=np.float32)
    be = _CountingBackend()
    ad = AnnData(X.copy())
    ac.pp.binarize(ad, backend=be)
    ac.pp.tfidf(ad, backend=be)
    ac.tl.lsi(ad, n_comps=6, backend=be)
    assert be.uploads == 1

    ref = AnnData(X.copy())
    ref.X.data[ref.X.data != 0] = 1
    ac.pp.tfidf(ref, backend=BE, keep_on_device=False)
    np.testing.assert_array_equal(ad.X.data, ref.X.data)
    np.testing.assert_array_equal(ad.X.indices, ref.X.indices)
    ac.tl.lsi(ref, n_comps=6, backend=BE)
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref.uns["lsi"]["stdev"], rtol=1e-6)
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref.varm["LSI"]) < 1e-5


def test_resident_copy_is_dropped_when_the_host_matrix_changes():
    X = planted_topics_csr(300, 200, n_topics=5, density=0.1, seed=3, dtype=np.float32)
    be = _CountingBackend()
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad, backend=be)
    assert be.uploads == 1
    ad.X.data *= 2.0  # in-place edit between the calls: the device copy no longer describes X
    ac.tl.lsi(ad, n_comps=5, backend=be)
    assert be.uploads == 2
    ad2 = AnnData(X.copy())
    ac.pp.tfidf(ad2, backend=be)
    ad2.X = ad2.X.copy()  # a new object: nothing attached
    ac.tl.lsi(ad2, n_comps=5, backend=be)
    assert be.uploads == 4
    # the result of the edited run is the one of the edited matrix
    ref = lsi_oracle.lsi(ad.X, n_comps=5)
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-5)


def test_resident_copy_detects_surgical_and_index_edits():
    # one changed entry, or a rewritten index array, must invalidate the device copy
    X = planted_topics_csr(400, 300, n_topics=5, density=0.1, seed=6, dtype=np.float32)
    for edit in ("one_value", "indices", "indptr"):
        be = _CountingBackend()
        ad = AnnData(X.copy())
        ac.pp.tfidf(ad, backend=be)
        assert be.uploads == 1
        m = ad.X
        if edit == "one_value":
            m.data[m.nnz // 3 + 1] += 0.5
        elif edit == "indices":
            lo, hi = m.indptr[7], m.indptr[8]
            m.indices[lo:hi] = m.indices[lo:hi][::-1].copy()  # same buffer, other content
            m.has_sorted_indices = False
        else:
            m.indptr[5] -= 1  # moves one entry from row 4 to row 5 (still a valid CSR)
            m.has_sorted_indices = False
        ac.tl.lsi(ad, n_comps=5, backend=be)
        assert be.uploads == 2, edit
    be = _CountingBackend()
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad, backend=be)
    ac.tl.lsi(ad, n_comps=5, backend=be)  # untouched: the copy is reused
    assert be.uploads == 1


def _device_tfidf(X):
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(X)).astype(np.float32)
    return T, BE.upload_csr(T.indptr, T.indices, T.data, T.shape)


def test_lsi_block_lanczos_thick_restart():
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(600, 400, n_topics=8, density=0.08, seed=11, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    ref = lsi_oracle.lsi(T, n_comps=8)
    _, sd, V, info = lsi_device(BE, Xd, n_comps=8, oversample=8, max_blocks=2, return_info=True)
    assert info["restarts"] >= 1 and info["converged"]
    assert info["spmm"] == 2 * info["iterations"] + 1
    assert lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"]) < 1e-4
    np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)
    # without the cap: same answer, no restart, not more products
    _, sd2, V2, info2 = lsi_device(BE, Xd, n_comps=8, oversample=8, max_blocks=12, return_info=True)
    assert info2["restarts"] == 0 and info2["iterations"] <= info["iterations"]
    assert lsi_oracle.max_subspace_angle(V2.numpy(), ref["LSI"]) < 1e-4


def test_lsi_krylov_space_exhausted_on_tiny_matrices():
    from muon_amd._atac.tools import lsi_device

    rng = np.random.default_rng(3)
    for n, d, k in ((40, 25, 5), (18, 60, 4), (30, 16, 15)):
        A = sp.random(n, d, density=0.4, format="csr", dtype=np.float32, random_state=rng)
        A.data[:] = rng.random(A.nnz).astype(np.float32) + 0.5
        Xd = BE.upload_csr(A.indptr, A.indices, A.data, A.shape)
        _, sd, V, info = lsi_device(BE, Xd, n_comps=k, return_info=True)
        s_ref = np.linalg.svd(A.toarray().astype(np.float64), compute_uv=False)[:k]
        np.testing.assert_allclose(sd * np.sqrt(n - 1), s_ref, rtol=1e-5)
        assert info["converged"] and info["iterations"] <= 6
        G = V.numpy().astype(np.float64).T @ V.numpy().astype(np.float64)
        np.testing.assert_allclose(G, np.eye(k), atol=1e-5)


def test_lsi_more_components_than_the_block_width():
    # n_comps > 64: several 64-wide blocks are kept across thick restarts
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(2000, 1500, n_topics=70, density=0.06, seed=5, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    ref = lsi_oracle.lsi(T, n_comps=70)
    U, sd, V, info = lsi_device(BE, Xd, n_comps=70, return_info=True)
    assert U.shape == (2000, 70) and V.shape == (1500, 70) and info["converged"]
    assert lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"]) < 1e-4
    np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)
    np.testing.assert_allclose(U.numpy().mean(axis=0), 0, atol=1e-3)
    np.testing.assert_allclose(U.numpy().std(axis=0), 1, rtol=1e-3)


def test_tfidf_takes_column_compressed_input_through_the_device_transpose():
    X = planted_topics_csr(300, 200, n_topics=5, density=0.1, seed=4, dtype=np.float32)
    a = AnnData(X.copy())
    ac.pp.tfidf(a, backend=BE)
    b = AnnData(X.tocsc())
    ac.pp.tfidf(b, backend=BE)
    assert b.X.format == "csr" and b.X.has_sorted_indices
    np.testing.assert_array_equal(b.X.indptr, a.X.indptr)
    np.testing.assert_array_equal(b.X.indices, a.X.indices)
    np.testing.assert_array_equal(b.X.data, a.X.data)
    ac.tl.lsi(b, n_comps=5, backend=BE)   # the device copy attached to the result serves lsi
    assert b.obsm["X_lsi"].shape == (300, 5)


def test_ritz_step_handles_dependent_krylov_directions():
    # M = K^T K with (numerically) dependent columns: the Ritz step truncates them instead of
    # inverting them, and reproduces the eigenpairs of the operator restricted to span(K)
    from muon_amd._atac.tools import _ritz

    rng = np.random.default_rng(0)
    A = rng.standard_normal((40, 40)); A = A @ A.T            # SPD operator
    K = np.linalg.qr(rng.standard_normal((40, 12)))[0]
    K = np.hstack([K, K[:, :3] + 1e-9 * rng.standard_normal((40, 3))])  # three near-copies
    T, M = K.T @ A @ K, K.T @ K
    lam, C = _ritz(T, M, 5)
    ref = np.linalg.eigvalsh(K[:, :12].T @ A @ K[:, :12])[::-1][:5]
    np.testing.assert_allclose(lam, ref, rtol=1e-6)
    V = K @ C
    np.testing.assert_allclose(V.T @ V, np.eye(5), atol=1e-6)
    # more pairs wanted than the space holds: the rest comes back as zeros
    lam2, C2 = _ritz(T, M, 20)
    assert lam2.shape == (20,) and np.all(lam2[12:] == 0) and np.all(C2[:, 12:] == 0)


@pytest.mark.parametrize("case", ["k_inside_cluster", "k_past_the_planted_rank", "unstructured"])
def test_lsi_reports_unconverged_instead_of_being_silently_wrong(case):
    # k not at a planted spectral gap.  Whatever the spectrum, either the
    # reported `converged` is True and the top-k right singular subspace is within the 1e-4 target of
    # f64 ARPACK, or `converged` is False; `angle_bound` (Lanczos residuals / Ritz gap + f32 floor)
    # always covers the true angle.
    from muon_amd._atac.tools import lsi_device
    from tests.synth import unstructured_csr

    if case == "k_inside_cluster":
        X, k = planted_topics_csr(1200, 900, n_topics=40, density=0.05, seed=3, dtype=np.float32), 25
    elif case == "k_past_the_planted_rank":
        X, k = planted_topics_csr(1000, 800, n_topics=12, density=0.05, seed=4, dtype=np.float32), 20
    else:
        X, k = unstructured_csr(700, 500, density=0.05, seed=5), 20
    T, Xd = _device_tfidf(X)
    ref = lsi_oracle.lsi(T, n_comps=k)
    _, sd, V, info = lsi_device(BE, Xd, n_comps=k, return_info=True, max_iter=40)
    angle = lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"])
    print(f"{case}: gap_rel={info['gap_rel']:.2e} angle={angle:.2e} bound={info['angle_bound']:.2e} "
          f"converged={info['converged']} spmm={info['spmm']}")
    assert angle <= max(info["angle_bound"], 1e-6)
    if info["converged"]:
        assert angle < 1e-4
    np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)  # singular values converge with the angle squared


def test_lsi_more_components_than_rank_stops_with_exact_values():
    """n_comps beyond the rank of the matrix: the Krylov space runs out after one expansion, the run ends
    there (not at max_iter), the singular values that exist are exact and the rest ~0."""
    import scipy.sparse as sp

    from muon_amd._atac.tools import lsi_device

    rng = np.random.default_rng(0)
    base = sp.random(20, 200, density=0.2, random_state=rng, format="csr", dtype=np.float32)
    X = (sp.diags((1 + rng.random(300)).astype(np.float32)) @ base[rng.integers(0, 20, 300)]).tocsr()
    X.sort_indices()
    be = CpuTestBackend()
    Xd = be.upload_csr(X.indptr, X.indices, X.data.astype(np.float32), X.shape)
    U, stdev, V, info = lsi_device(be, Xd, n_comps=30, n_obs=300, return_info=True)
    s = stdev * np.sqrt(299)
    want = np.linalg.svd(X.toarray().astype(np.float64), compute_uv=False)
    np.testing.assert_allclose(s[:20], want[:20], rtol=1e-5)
    assert np.all(s[20:] < 1e-4 * s[0]) and info["iterations"] <= 3 and info["spmm"] <= 7


@pytest.mark.parametrize("n_comps,max_blocks", [(8, 2), (8, 12), (70, None)])
def test_lsi_pipelined_expansions_give_the_same_answer(monkeypatch, n_comps, max_blocks):
    """r04: with the device-side CholeskyQR the C-independent half of the next expansion is queued before the host's
    Ritz step (tools.py `speculate`), across thick restarts (cross Grams transformed by Cw on the host).  Same
    subspace, same singular values, same number of expansions as the unpipelined loop; the speculation that turns
    out wrong at the last step is reported."""
    from muon_amd._atac.tools import lsi_device

    if n_comps > 64:
        X = planted_topics_csr(2000, 1500, n_topics=70, density=0.06, seed=5, dtype=np.float32)
    else:
        X = planted_topics_csr(600, 400, n_topics=8, density=0.08, seed=11, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    ref = lsi_oracle.lsi(T, n_comps=n_comps)
    kw = dict(n_comps=n_comps, return_info=True, device_qr=True)
    if max_blocks is not None:
        kw.update(oversample=8, max_blocks=max_blocks)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MUON_AMD_LSI_PIPELINE", mode)
        _, sd, V, info = lsi_device(BE, Xd, **kw)
        assert info["converged"]
        assert lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"]) < 1e-4
        np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)
        out[mode] = (sd, V.numpy(), info)
    a, b = out["0"], out["1"]
    assert a[2]["iterations"] == b[2]["iterations"] and a[2]["restarts"] == b[2]["restarts"]
    assert lsi_oracle.max_subspace_angle(a[1], b[1]) < 2e-5
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6)
    assert a[2]["spmm_unused"] <= 1 and b[2]["spmm_unused"] <= 2  # the pipelined loop runs ahead by two products


def _big_counts(seed=0):
    m = sp.random(400, 300, density=0.1, format="csr", dtype=np.float32, random_state=seed)
    m.data = np.floor(m.data * 8 + 1).astype(np.float32)
    return m.copy()


def test_tfidf_takes_the_replaced_matrix_over_only_when_nothing_else_sees_it(monkeypatch):
    """`adata.X = tf_idf` (preproc.py:121-127) leaves the old matrix to its other owners.  When there are none, the
    result reuses its index arrays and value buffer (no 6 GB copy / first touch / release at scale); an owner -
    `layers["counts"] = adata.X`, a variable, a view of one of the arrays - keeps seeing the counts."""
    from muon_amd._atac import preproc

    monkeypatch.delenv("MUON_AMD_REUSE_HOST", raising=False)
    ref = AnnData(_big_counts())
    old = ref.X
    where = ref.X.data.ctypes.data
    del old
    ac.pp.tfidf(ref, backend=BE)          # r06: OFF by default - the reference never writes into the matrix it replaces
    assert ref.X.data.ctypes.data != where
    # the counts it compares with are measured on this interpreter and the whole decision is self-tested at first use
    cal = preproc._calibrate_refs()
    assert cal["ok"] and cal["matrix2"] >= 3 and cal["attr"] >= 2

    ad = AnnData(_big_counts())
    where = (ad.X.data.ctypes.data, ad.X.indices.ctypes.data, ad.X.indptr.ctypes.data)
    ac.pp.tfidf(ad, backend=BE, reuse_host=True)
    assert (ad.X.data.ctypes.data, ad.X.indices.ctypes.data, ad.X.indptr.ctypes.data) == where  # taken over
    assert (ad.X != ref.X).nnz == 0 and np.array_equal(ad.X.data, ref.X.data)
    ac.tl.lsi(ad, n_comps=5, backend=BE)  # the attached device copy describes the new values

    for how in ("layer", "variable", "array", "view", "shared arrays"):
        ad = AnnData(_big_counts())
        counts = ad.X.copy()
        where = ad.X.data.ctypes.data
        if how == "layer":
            ad.layers["counts"] = ad.X
            seen = lambda: ad.layers["counts"]  # noqa: E731
        elif how == "variable":
            keep = ad.X
            seen = lambda: keep  # noqa: E731
        elif how == "array":
            arr = ad.X.data
            seen = lambda: sp.csr_matrix((arr, counts.indices, counts.indptr), shape=counts.shape)  # noqa: E731
        elif how == "view":
            v = ad.X.data[:10]
            seen = lambda: sp.csr_matrix((v.base if v.base is not None and v.base.shape == counts.data.shape else  # noqa: E731
                                          np.concatenate([v, counts.data[10:]]), counts.indices, counts.indptr), shape=counts.shape)
        else:
            other = sp.csr_matrix((ad.X.data, ad.X.indices, ad.X.indptr), shape=ad.X.shape)
            seen = lambda: other  # noqa: E731
        ac.pp.tfidf(ad, backend=BE, reuse_host=True)
        assert ad.X.data.ctypes.data != where, how
        assert np.array_equal(ad.X.data, ref.X.data), how
        assert (seen() != counts).nnz == 0, how  # the other owner still holds the counts
    # the environment switch opts in as well; a failed self-test switches the takeover off whatever the caller asks
    monkeypatch.setenv("MUON_AMD_REUSE_HOST", "1")
    ad = AnnData(_big_counts())
    where = ad.X.data.ctypes.data
    ac.pp.tfidf(ad, backend=BE)
    assert ad.X.data.ctypes.data == where
    monkeypatch.setitem(preproc._CALIBRATION, "ok", False)
    ad = AnnData(_big_counts())
    where = ad.X.data.ctypes.data
    ac.pp.tfidf(ad, backend=BE)
    assert ad.X.data.ctypes.data != where and np.array_equal(ad.X.data, ref.X.data)

    # not in place / into a layer / from a layer: the counts stay where they are
    ad = AnnData(_big_counts())
    counts = ad.X.copy()
    res = ac.pp.tfidf(ad, inplace=False, backend=BE)
    assert (ad.X != counts).nnz == 0 and np.array_equal(res.data, ref.X.data)
    ac.pp.tfidf(ad, to_layer="tfidf", backend=BE)
    assert (ad.X != counts).nnz == 0 and np.array_equal(ad.layers["tfidf"].data, ref.X.data)

    # unsorted input: the canonicalised temporary of the call is the result's (and the caller's matrix is untouched)
    ad = AnnData(_big_counts())
    perm = np.arange(ad.X.nnz)
    for r in range(ad.X.shape[0]):
        lo, hi = ad.X.indptr[r], ad.X.indptr[r + 1]
        perm[lo:hi] = perm[lo:hi][::-1]
    shuffled = sp.csr_matrix((ad.X.data[perm], ad.X.indices[perm], ad.X.indptr.copy()), shape=ad.X.shape)
    keep = shuffled.copy()
    ad2 = AnnData(shuffled)
    ac.pp.tfidf(ad2, backend=BE)
    assert np.array_equal(ad2.X.data, ref.X.data) and np.array_equal(ad2.X.indices, ref.X.indices)
    assert np.array_equal(shuffled.data, keep.data) and np.array_equal(shuffled.indices, keep.indices)


@pytest.mark.parametrize("dtype", [np.float64, np.int32, np.int64])
def test_tfidf_takeover_with_other_count_types(dtype):
    """f64 counts are taken over like f32 ones (the result is f64); integer counts are promoted to f64 by a canonicalised
    temporary of the call (preproc.py:92: 1.0 / n_peaks is float64), which the result owns - the caller's matrix stays."""
    m = _big_counts(3)
    m = sp.csr_matrix((m.data.astype(dtype), m.indices.copy(), m.indptr.copy()), shape=m.shape)
    ref = AnnData(m.astype(np.float64))
    ac.pp.tfidf(ref, backend=BE, keep_on_device=False)
    ad = AnnData(m.copy())
    before = ad.X.copy()
    keep = ad.X if dtype != np.float64 else None  # (integer input is never written to, owner or not)
    where = ad.X.data.ctypes.data
    ac.pp.tfidf(ad, backend=BE, reuse_host=True)  # (opt-in since r06)
    assert ad.X.dtype == np.float64 and np.array_equal(ad.X.data, ref.X.data) and np.array_equal(ad.X.indices, ref.X.indices)
    if dtype == np.float64:
        assert ad.X.data.ctypes.data == where
    else:
        assert (keep != before).nnz == 0 and keep.dtype == dtype


def test_restarts_keep_more_when_k_sits_inside_a_cluster(monkeypatch):
    """r06: 80 planted topics, n_comps = 50 - sigma_50 and sigma_51 lie inside one cluster (a ~1 % gap).  A thick restart
    that keeps one 64-vector block cuts through the cluster every time; when the Ritz values show the small gap at the
    first restart, the restarts keep one block more (MUON_AMD_LSI_GROW=0: the fixed rule).  Same subspace, within the
    parity bar of f64 ARPACK, in about half the products."""
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(3000, 2500, n_topics=80, density=0.03, seed=3, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    ref = lsi_oracle.lsi(T, n_comps=50)
    out = {}
    for grow in ("0", "1"):
        monkeypatch.setenv("MUON_AMD_LSI_GROW", grow)
        _, sd, V, info = lsi_device(BE, Xd, n_comps=50, return_info=True)
        assert info["converged"] and lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"]) < 1e-4
        np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)
        out[grow] = info["spmm"]
    assert out["1"] <= 0.7 * out["0"], out
    # a gapped spectrum (planted rank = n_comps) never triggers it: same products either way
    X = planted_topics_csr(1500, 1200, n_topics=20, density=0.05, seed=2, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    cnt = []
    for grow in ("0", "1"):
        monkeypatch.setenv("MUON_AMD_LSI_GROW", grow)
        cnt.append(lsi_device(BE, Xd, n_comps=20, return_info=True)[3]["spmm"])
    assert cnt[0] == cnt[1]


def test_roctx_ranges_are_balanced_and_off_by_default(monkeypatch):
    """MUON_AMD_TRACE=1 brackets the phases of tfidf / lsi with roctx ranges (muon_amd/_trace.py).  Off by
    default; when on, every push has its pop - also when lsi leaves through its redo-on-host path - and the result is
    the same."""
    from muon_amd import _trace
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(600, 400, n_topics=8, density=0.08, seed=11, dtype=np.float32)
    T, Xd = _device_tfidf(X)
    monkeypatch.setattr(_trace, "_state", None)
    monkeypatch.delenv("MUON_AMD_TRACE", raising=False)
    assert not _trace._enabled()
    _, sd0, _, _ = lsi_device(BE, Xd, n_comps=8, return_info=True)

    calls = []

    class Fake:
        def roctxRangePushA(self, b):
            calls.append(("push", b.decode()))
            return 0

        def roctxRangePop(self):
            calls.append(("pop", None))
            return 0

    monkeypatch.setattr(_trace, "_state", True)
    monkeypatch.setattr(_trace, "_lib", Fake())
    monkeypatch.setattr(_trace, "_depth", [0])
    _, sd1, _, _ = lsi_device(BE, Xd, n_comps=8, return_info=True)
    names = [n for k, n in calls if k == "push"]
    assert names == ["lsi/operands", "lsi/warm_start", "lsi/krylov", "lsi/ritz_vectors"]
    assert sum(k == "push" for k, _ in calls) == sum(k == "pop" for k, _ in calls) and _trace._depth[0] == 0
    np.testing.assert_array_equal(sd0, sd1)


def test_f64_input_is_answered_in_f64_arithmetic():
    """the reference runs f64 ARPACK on an f64 X (/root/reference/muon/_atac/tools.py:53).  The f32
    Krylov process stops at the f32 floor; for f64 input it is continued in f64 (tools._refine_f64: f64 blocks, products
    accumulated in f64, exact residuals): a 1.3 % gap - the hardest gapped case of the suite - comes out below 1e-6 rad of
    f64 ARPACK where f32 arithmetic gives ~1e-5, singular values to 1e-9, outputs f64; f32 input keeps the f32 path."""
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(3000, 2500, n_topics=80, density=0.03, seed=3, dtype=np.float64)
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(X))
    assert T.dtype == np.float64
    ref = lsi_oracle.lsi(T, n_comps=50)
    ad = AnnData(T.copy())
    ac.tl.lsi(ad, n_comps=50, backend=BE)
    assert ad.varm["LSI"].dtype == np.float64 and ad.obsm["X_lsi"].dtype == np.float64
    ang = lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"])
    assert ang < 1e-6, ang
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-8)  # (the operand's VALUES stay f32: 8e-10)
    # the embeddings: scaled like the reference's (zero mean, unit population variance), same subspace
    np.testing.assert_allclose(ad.obsm["X_lsi"].mean(axis=0), 0, atol=1e-12)
    np.testing.assert_allclose(ad.obsm["X_lsi"].std(axis=0), 1, rtol=1e-10)
    assert lsi_oracle.max_subspace_angle(ad.obsm["X_lsi"] - ad.obsm["X_lsi"].mean(axis=0),
                                         ref["X_lsi"] - ref["X_lsi"].mean(axis=0)) < 1e-5
    # what the refinement reports, and what f32 arithmetic alone reaches on the same matrix
    Xd = BE.upload_csr(T.indptr, T.indices, T.data, T.shape, values_dtype=np.float32)
    _, sd32, V32, i32 = lsi_device(BE, Xd, n_comps=50, return_info=True)
    _, sd64, V64, i64 = lsi_device(BE, Xd, n_comps=50, return_info=True, refine_f64=True)
    assert V32.dtype == torch.float32 and V64.dtype == torch.float64 and i32["refine_f64"] is None
    r = i64["refine_f64"]
    assert 1 <= r["blocks"] <= 8 and r["angle_bound"] <= 1e-6 and i64["spmm"] == i32["spmm"] + 2 * r["blocks"]
    a32 = lsi_oracle.max_subspace_angle(V32.numpy(), ref["LSI"])
    a64 = lsi_oracle.max_subspace_angle(V64.numpy(), ref["LSI"])
    assert a64 < 1e-7 and a64 < 0.2 * a32 and a32 < 1e-4, (a32, a64)  # (4.4e-8: what rounding the VALUES to f32 costs)
    # a 0.16 % gap (k = 26 on the same matrix): still below 1e-6 rad, by the refinement's own bound and against ARPACK
    ref26 = lsi_oracle.lsi(T, n_comps=26)
    _, _, V26, i26 = lsi_device(BE, Xd, n_comps=26, return_info=True, refine_f64=True)
    assert 1e-3 < i26["gap_rel"] < 2e-3 and i26["refine_f64"]["angle_bound"] <= 1e-6 and i26["converged"]
    assert lsi_oracle.max_subspace_angle(V26.numpy(), ref26["LSI"]) < 1e-6
    # unscaled embeddings: unit columns, U diag(s) = X V
    U, sd, V, _ = lsi_device(BE, Xd, n_comps=50, scale_embeddings=False, return_info=True, refine_f64=True)
    np.testing.assert_allclose(np.linalg.norm(U.numpy(), axis=0), 1, rtol=1e-12)
    s = sd * np.sqrt(T.shape[0] - 1)
    XV = T.astype(np.float32).astype(np.float64) @ V.numpy()
    np.testing.assert_allclose(U.numpy() * s, XV, atol=1e-9 * s[0])


@pytest.mark.parametrize("k", [5, 63, 70, 100])
def test_f64_continuation_with_wide_starts_and_an_exhausted_krylov_space(k):
    """tools._refine_f64 beyond the comfortable case: more components than the block width (the start block is every Ritz
    vector the f32 run kept: 193 columns here, several blocks' worth), and a Krylov space that reaches the dimension of
    the matrix (900 columns: the fifth block has 128 directions left of 193 - deflation must drop the rest, normalised
    rounding is not orthogonal to the basis).  Against f64 ARPACK of the f64 operand."""
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(1500, 900, n_topics=30, density=0.05, seed=3, dtype=np.float64)
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(X))
    ref = lsi_oracle.lsi(T, n_comps=k)
    Xd = BE.upload_csr(T.indptr, T.indices, T.data, T.shape, values_dtype=np.float32)
    U, sd, V, info = lsi_device(BE, Xd, n_comps=k, return_info=True, refine_f64=True)
    r = info["refine_f64"]
    assert V.shape == (900, k) and V.dtype == torch.float64 and U.shape == (1500, k)
    assert r["angle_bound"] <= 1e-6 and info["converged"], r
    assert lsi_oracle.max_subspace_angle(V.numpy(), ref["LSI"]) < 1e-6
    np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-8)
    assert np.abs(V.numpy().T @ V.numpy() - np.eye(k)).max() < 1e-10


def test_bench_line_is_flattened_for_the_driver():
    """bench.py:flatten_for_driver: the driver keeps scalar keys of `config` and a 2 000-character
    tail of the line - parity, the LSI figures, one value per secondary record (and the hard spectrum's f64 continuation)
    must be flat scalars under `config`, and a compact `summary` the LAST key of the line."""
    import importlib.util
    import json
    import os

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {"metric": "cells/sec", "value": 3.5e6, "ms_per_step": 284.9, "roofline": {"frac": 0.1731, "avg_launch_ms": 36.48},
           "config": {"workload": "c3", "lsi": {"spmm_per_step": 5, "converged": True, "angle_bound": 2e-5}},
           "parity": {"tfidf_pattern_identical": True, "tfidf_values_max_rel": 4.4e-7, "lsi_angle_rad": 4.2e-6,
                      "lsi_stdev_max_rel": 6e-7, "sample": "text is not a scalar"}}
    sec = {"c2": {"value": 1.06e6, "unit": "cells/s", "ms_per_step": 9.38, "roofline": {"frac": 0.096}},
           "mofa_ng": {"value": 0.1779, "unit": "s", "roofline": {"frac": 0.1887}, "parity": {"elbo_max_rel": 1.5e-15, "oracle": "x"}},
           "c3_rank8": {"value": 43.4, "unit": "ms", "ms_per_step": 43.4},
           "hard": {"value": 2.26e6, "unit": "cells/s", "ms_per_step": 443.1,
                    "config": {"workload": "hard", "f64_continuation_ms_per_step": 1057.9, "f64_continuation_converged": True,
                               "f64_continuation_angle_bound": 3.1e-7, "f64_continuation_blocks": 4}},
           "wnn": {"error": "RuntimeError('x')"}}
    bench.flatten_for_driver(out, sec)
    cfg = out["config"]
    assert cfg["lsi_spmm_per_step"] == 5 and cfg["lsi_converged"] is True and cfg["parity_lsi_angle_rad"] == 4.2e-6
    assert cfg["c2_ms_per_step"] == 9.38 and cfg["mofa_ng_value"] == 0.1779 and cfg["mofa_ng_parity_elbo_max_rel"] == 1.5e-15
    assert cfg["hard_ms_per_step"] == 443.1 and cfg["hard_f64_continuation_converged"] is True
    assert cfg["hard_f64_continuation_angle_bound"] == 3.1e-7 and "wnn_error" in cfg
    assert cfg["speedup_8gpu_before_comm_emulated"] == round(284.9 / 43.4, 3)
    assert "mofa_ng_parity_oracle" not in cfg  # (strings stay in the sub-record)
    assert list(out)[-1] == "summary" and out["summary"]["hard_f64_continuation_converged"] is True
    assert len(json.dumps(out["summary"], separators=(",", ":"))) < 1900

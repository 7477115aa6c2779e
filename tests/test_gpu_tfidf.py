"""muon_amd.atac.pp.tfidf on the GPU against (a) the reference's own golden values,
(b) fixtures produced by executing the reference source, (c) the CPU oracle on seeded
matrices, (d) size-independent properties at a size the oracle cannot reach quickly.
Bar: identical nnz pattern / indices (canonical order), values within 1e-5 relative."""
import numpy as np
import pytest
import scipy.sparse as sp

from muon_amd import AnnData, MuData
from muon_amd import atac as ac
from oracle import tfidf_oracle
from tests.synth import planted_topics_csr, unstructured_csr
from tests.test_tfidf_oracle import SWEEPS, _csr

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _same(res, ref, rtol=RTOL):
    ref = tfidf_oracle.canonical(ref)
    assert res.shape == ref.shape
    assert np.array_equal(res.indptr, ref.indptr), "indptr differs"
    assert np.array_equal(res.indices, ref.indices), "indices differ"
    assert res.dtype == ref.dtype
    fin = np.isfinite(ref.data)
    assert np.array_equal(np.isnan(res.data), np.isnan(ref.data))
    np.testing.assert_allclose(res.data[fin], ref.data[fin], rtol=rtol, atol=0)


def test_reference_golden_dense():
    # /root/reference/tests/test_atac_preproc.py:11-52
    np.random.seed(2020)
    x = np.abs(np.random.normal(size=(4, 5)))
    adata = AnnData(x.copy())
    ac.pp.tfidf(adata, log_tf=True, log_idf=True)
    assert "%.3f" % adata.X[0, 0] == "4.659"
    assert "%.3f" % adata.X[3, 0] == "4.770"
    view = AnnData(x.copy())[:, :]
    ac.pp.tfidf(view, log_tf=True, log_idf=True)
    assert "%.3f" % view.X[0, 0] == "4.659"
    a = AnnData(x.copy())
    cp = ac.pp.tfidf(a, copy=True)
    assert a.X[0, 0] == x[0, 0] and "%.3f" % cp.X[0, 0] == "4.659"
    res = ac.pp.tfidf(a, inplace=False)
    assert a.X[0, 0] == x[0, 0] and "%.3f" % res[0, 0] == "4.659"
    ac.pp.tfidf(a, to_layer="new")
    assert "%.3f" % a.layers["new"][0, 0] == "4.659"
    a = AnnData(x.copy())
    a.layers["counts"] = a.X.copy() + 1
    a.X = None
    ac.pp.tfidf(a, from_layer="counts")
    assert "%.3f" % a.X[0, 0] == "2.856"
    m = MuData({"atac": AnnData(x.copy())})
    ac.pp.tfidf(m)
    assert "%.3f" % m.mod["atac"].X[0, 0] == "4.659"


def test_reference_golden_sparse():
    # /root/reference/tests/test_atac_preproc.py:57-64
    np.random.seed(2020)
    x = sp.rand(100, 10, density=0.2, format="csr")
    adata = AnnData(x)
    ac.pp.tfidf(adata, log_tf=True, log_idf=True)
    assert "%.3f" % adata.X[10, 9] == "18.749"
    assert "%.3f" % adata.X[50, 5] == "0.000"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/tfidf_golden.npz")


def test_fixture_dense_and_sparse(gold):
    _same(ac.pp.tfidf(AnnData(gold["dense_in"]), inplace=False), _csr(gold, "dense_out"))
    _same(ac.pp.tfidf(AnnData(_csr(gold, "sparse_in")), inplace=False), _csr(gold, "sparse_out"))
    raw = ac.pp.tfidf(AnnData(_csr(gold, "sparse_in")), inplace=False, match_scipy_order=True)
    assert np.array_equal(raw.indices, gold["sparse_out_indices"])
    np.testing.assert_allclose(raw.data, gold["sparse_out_data"], rtol=RTOL)


@pytest.mark.parametrize("name", sorted(SWEEPS))
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_fixture_option_sweep(gold, name, dt):
    cnt = _csr(gold, "sweep_in").astype(dt)  # has an empty row and an empty column
    res = ac.pp.tfidf(AnnData(cnt), inplace=False, **SWEEPS[name])
    _same(res, _csr(gold, f"sweep_{name}_{dt}"))


def test_fixture_explicit_zeros_are_dropped_and_ints_promoted(gold):
    res = ac.pp.tfidf(AnnData(_csr(gold, "ezero_in")), inplace=False)
    _same(res, _csr(gold, "ezero_out"))
    assert res.nnz < _csr(gold, "ezero_in").nnz
    res = ac.pp.tfidf(AnnData(_csr(gold, "sweep_in").astype(np.int32)), inplace=False)
    assert res.dtype == np.float64
    _same(res, _csr(gold, "sweep_int32"))


@pytest.mark.parametrize("gen", ["planted", "unstructured"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_against_oracle_seeded(gen, dt):
    if gen == "planted":
        X = planted_topics_csr(3000, 20000, n_topics=30, density=0.02, seed=11, dtype=dt)
    else:
        X = unstructured_csr(2500, 17000, density=0.02, seed=12, dtype=dt)
    for kw in (dict(), dict(log_tf=False, log_idf=False, log_tfidf=True), dict(scale_factor=None)):
        _same(ac.pp.tfidf(AnnData(X.copy()), inplace=False, **kw), tfidf_oracle.tfidf(X, **kw))


def test_unsorted_and_duplicate_input_is_canonicalised():
    rng = np.random.default_rng(5)
    r = rng.integers(0, 50, 4000); c = rng.integers(0, 80, 4000)
    coo = sp.coo_matrix((np.ones(4000, np.float32), (r, c)), shape=(50, 80))
    raw = sp.csr_matrix((coo.data, coo.col, np.concatenate([[0], np.cumsum(np.bincount(coo.row, minlength=50))])), shape=(50, 80))
    # raw has duplicates and is unsorted; scipy's SpGEMM sums duplicates too
    order = np.argsort(coo.row, kind="stable")
    raw = sp.csr_matrix((coo.data[order], coo.col[order], raw.indptr), shape=(50, 80))
    _same(ac.pp.tfidf(AnnData(raw.copy()), inplace=False), tfidf_oracle.tfidf(raw))


def test_properties_at_scale(hip):
    """200k x 30k device-generated counts (~1.8e8 nnz): size-independent checks."""
    from muon_amd._atac.preproc import tfidf_device
    import torch
    X = hip.synth_counts(0, 200000, 30000, n_topics=50, density=0.03, seed=0)
    rs, cs = hip.row_col_sums(X)
    tot = float(X.values.double().sum().item())
    assert float(rs.sum().item()) == tot and float(cs.sum().item()) == tot  # checksum of checksums
    # reference (torch scatter) for the column sums
    ref_cs = torch.zeros(30000, dtype=torch.float64, device=X.values.device)
    ref_cs.index_add_(0, X.indices.long(), X.values.double())
    assert torch.equal(ref_cs, cs)
    R = tfidf_device(hip, X, 200000, 3, 1e4)
    assert R.indices is X.indices  # pattern untouched
    # spot-check 3 rows against the oracle formula
    rsh, csh = hip.to_host(rs), hip.to_host(cs)
    ip = hip.to_host(X.indptr)
    for row in (0, 77777, 199999):
        lo, hi = ip[row], ip[row + 1]
        c = hip.to_host(X.values[lo:hi]).astype(np.float32)
        j = hip.to_host(X.indices[lo:hi])
        exp = np.log1p((np.float32(1.0) / np.float32(rsh[row])) * c * np.float32(1e4)) * np.log1p(np.float32(200000) / csh[j].astype(np.float32))
        np.testing.assert_allclose(hip.to_host(R.values[lo:hi]), exp, rtol=RTOL)
    # scale invariance of tf: tfidf(2*counts) == tfidf(counts) up to the idf shift
    X2 = X.with_values(X.values * 2)
    R2 = tfidf_device(hip, X2, 400000, 1, 1e4)  # log_tf only, n_obs doubled keeps idf equal
    R1 = tfidf_device(hip, X, 200000, 1, 1e4)
    torch.testing.assert_close(R2.values, R1.values, rtol=1e-6, atol=0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("flags", [3, 1, 2, 0, 4])
def test_scale_sweep_equals_gather_kernel(hip, dt, flags):
    """The slab-sweep scale pass (idf slab in LDS) and the per-lane gather kernel are the same
    arithmetic: identical bits, with and without reusable slab pointers, ragged / empty rows and
    more than one slab of columns."""
    import torch
    rng = np.random.default_rng(3)
    n, d = 3000, 20000   # 3 slabs of 8192 columns
    m = sp.random(n, d, density=0.01, format="csr", random_state=rng, dtype=np.float64)
    m.data[:] = rng.integers(1, 5, m.nnz)
    m = m.tolil()
    m[5, :] = 0            # an empty row
    m[7, 100:9000:3] = 2   # a long row
    m = m.tocsr().astype(dt)
    m.sort_indices()
    X = hip.upload_csr(m.indptr, m.indices, m.data, m.shape)
    rs, cs = hip.row_col_sums(X)
    idf = hip.idf(cs, n, flags, X.values.dtype)
    a, za = hip.tfidf_scale(X, rs, idf, 1e4, flags)            # slab pointers reused
    b, zb = hip.tfidf_scale(X, rs, idf, 1e4, flags)            # searched again
    hip._scale_gather = True
    try:
        c, zc = hip.tfidf_scale(X, rs, idf, 1e4, flags)
    finally:
        hip._scale_gather = False
    assert torch.equal(a.view(torch.uint8), c.view(torch.uint8))
    assert torch.equal(b.view(torch.uint8), c.view(torch.uint8))
    assert int(za.item()) == int(zc.item()) == int(zb.item())


@pytest.mark.parametrize("shape", [(3000, 20000), (700, 5000), (2000, 40001), (130, 16384)])
def test_sum_sweep_variants_agree(hip, shape):
    """r04: the f32 sum sweep exists in three forms - the kernel of r03 (tune tfidf_pipe = 1), the software-pipelined
    walk with 8192-column bins (tfidf_sum_m = 1) and the one for large matrices with 16 384-column bins, one
    workgroup per CU (tfidf_sum_m = 2: whatever the size).  Same row sums (same order of additions), same column
    sums on count data, the same pointer table, the same values out of the scale sweep."""
    import torch
    rng = np.random.default_rng(11)
    n, d = shape
    m = sp.random(n, d, density=0.02, format="csr", random_state=rng, dtype=np.float64)
    m.data[:] = rng.integers(1, 6, m.nnz)
    keep = np.ones(n)
    keep[[5, 6, n - 1]] = 0  # empty rows, the last one too
    m = (sp.diags(keep) @ m).tolil()
    m[7, 10:min(d, 17000):2] = 3  # a long row across slabs
    m[9, d - 1] = 4               # a row that ends in the last column
    m = m.tocsr().astype(np.float32)
    m.eliminate_zeros()
    m.sort_indices()
    X = hip.upload_csr(m.indptr, m.indices, m.data, m.shape, slab_ptr=False)  # (the sweeps search their own table here)
    n_sp = n * (-(-d // 8192) + 1)
    got = {}
    try:
        for name, keys in (("r03", {"tfidf_pipe": 1}), ("pipe", {"tfidf_sum_m": 1}), ("wide", {"tfidf_sum_m": 2})):
            for k in ("tfidf_pipe", "tfidf_sum_m"):
                hip.tune(k, keys.get(k, 0))
            rs, cs = hip.row_col_sums(X)
            table = hip._sweep_work[0][:8 * n_sp].view(torch.int64).clone()
            idf = hip.idf(cs, n, 3, torch.float32)
            vals, _ = hip.tfidf_scale(X, rs, idf, 1e4, 3)
            got[name] = (rs, cs, table, vals)
    finally:
        hip.tune("tfidf_pipe", 0)
        hip.tune("tfidf_sum_m", 0)
    np.testing.assert_array_equal(hip.to_host(got["r03"][0]), np.asarray(m.sum(axis=1)).ravel())
    np.testing.assert_array_equal(hip.to_host(got["r03"][1]), np.asarray(m.sum(axis=0)).ravel())
    for name in ("pipe", "wide"):
        for a, b, what in zip(got[name], got["r03"], ("row sums", "column sums", "slab pointers", "values")):
            assert torch.equal(a, b), f"{name}: {what}"


def test_column_compressed_input_goes_through_the_device_transpose():
    X = planted_topics_csr(4000, 3000, n_topics=10, density=0.03, seed=6, dtype=np.float32)
    a = AnnData(X.copy())
    ac.pp.tfidf(a)
    b = AnnData(X.tocsc())
    ac.pp.tfidf(b)
    assert b.X.format == "csr"
    np.testing.assert_array_equal(b.X.indptr, a.X.indptr)
    np.testing.assert_array_equal(b.X.indices, a.X.indices)
    np.testing.assert_array_equal(b.X.data, a.X.data)


def test_ingest_10x_arrays_sharded_on_the_device_then_tfidf(hip):
    """SURVEY 8f.2 on the HIP path: the arrays of a 10x matrix group -> per-rank device CSR (peak columns
    selected on the device, int32 counts converted on the way up), and the AnnData built from them runs
    tfidf on its resident copy: identical pattern, values within 1e-5 of the oracle."""
    from muon_amd import atac as ac
    from muon_amd._core import io as mio
    from tests.test_ingest import _FakeComm, _tenx

    m, matrix, ft = _tenx(n_cells=3000, n_feat=700, seed=5)
    peaks = np.array([t == b"Peaks" for t in ft])
    for world in (1, 4):
        for rank in range(world):
            X, keep, (r0, r1) = mio.device_csr_from_10x(matrix, _FakeComm(rank, world), hip)
            want = m[r0:r1][:, peaks].astype(np.float32)
            got = sp.csr_matrix((hip.to_host(X.values), hip.to_host(X.indices), hip.to_host(X.indptr)), shape=X.shape)
            assert (got != want).nnz == 0 and np.array_equal(keep, np.nonzero(peaks)[0])
    ad = mio.read_10x_arrays(matrix, backend=hip)
    ac.pp.tfidf(ad, backend=hip)
    ref = tfidf_oracle.canonical(tfidf_oracle.tfidf(m[:, peaks].astype(np.float32)))
    assert np.array_equal(ad.X.indices, ref.indices) and np.array_equal(ad.X.indptr, ref.indptr)
    assert np.max(np.abs(ad.X.data - ref.data) / np.abs(ref.data)) < 1e-5
    coo = m.tocoo()
    Y = mio.device_csr_from_coo(coo.row, coo.col, coo.data, m.shape, hip)
    got = sp.csr_matrix((hip.to_host(Y.values), hip.to_host(Y.indices), hip.to_host(Y.indptr)), shape=Y.shape)
    assert (got != m.astype(np.float32)).nnz == 0
    c = m.tocsc()
    Z = mio.device_csr_from_csc(c.indptr, c.indices, c.data, m.shape, hip)
    got = sp.csr_matrix((hip.to_host(Z.values), hip.to_host(Z.indices), hip.to_host(Z.indptr)), shape=Z.shape)
    assert (got != m.astype(np.float32)).nnz == 0

"""on-disk layouts -> (row-sharded) device CSR without a host conversion
(muon_amd/_core/io.py; reference: muon/_core/io.py:23-72, muon/_atac/io.py:11-22,125).  Host logic on the
CPU test operator set; the HIP path is exercised by tests/test_gpu_tfidf.py::test_ingest_*."""
import numpy as np
import pytest
import scipy.sparse as sp

from muon_amd._core import io as mio
from tests.cpu_backend import CpuTestBackend

BE = CpuTestBackend()


class _FakeComm:
    def __init__(self, rank, world):
        self.rank, self.world_size = rank, world


def _tenx(n_cells=230, n_feat=90, seed=0):
    rng = np.random.default_rng(seed)
    m = sp.random(n_cells, n_feat, density=0.1, format="csr", random_state=rng, dtype=np.float64)
    m.data = (1 + rng.poisson(0.5, m.nnz)).astype(np.int32)
    m.sort_indices()
    ft = np.array([b"Peaks" if j % 3 else b"Gene Expression" for j in range(n_feat)])
    matrix = {"data": m.data, "indices": m.indices.astype(np.int64), "indptr": m.indptr.astype(np.int64),
              "shape": np.array([n_feat, n_cells]), "features": {"feature_type": ft}}
    return m, matrix, ft


def _host(X):
    return sp.csr_matrix((X.values.numpy(), X.indices.numpy(), X.indptr.numpy()), shape=X.shape)


@pytest.mark.parametrize("world", [1, 3])
def test_10x_arrays_to_row_sharded_device_csr(world):
    m, matrix, ft = _tenx()
    peaks = np.array([t == b"Peaks" for t in ft])
    for rank in range(world):
        X, keep, (r0, r1) = mio.device_csr_from_10x(matrix, _FakeComm(rank, world), BE)
        want = m[r0:r1][:, peaks].astype(np.float32)
        got = _host(X)
        assert X.values.dtype.is_floating_point and X.indices.dtype.itemsize == 4 and X.indptr.dtype.itemsize == 8
        assert got.shape == want.shape and (got != want).nnz == 0
        assert np.array_equal(keep, np.nonzero(peaks)[0])
    X, keep, _ = mio.device_csr_from_10x(matrix, None, BE, atac_only=False)
    assert keep is None and (_host(X) != m.astype(np.float32)).nnz == 0


def test_unsorted_rows_are_sorted_on_the_device():
    m, matrix, _ = _tenx(seed=1)
    idx, dat = matrix["indices"].copy(), matrix["data"].copy()
    for r in range(m.shape[0]):  # reverse every row: a legal but non-canonical file
        a, b = matrix["indptr"][r], matrix["indptr"][r + 1]
        idx[a:b], dat[a:b] = idx[a:b][::-1].copy(), dat[a:b][::-1].copy()
    X, _, _ = mio.device_csr_from_10x(dict(matrix, indices=idx, data=dat), None, BE, atac_only=False)
    got = _host(X)
    assert np.array_equal(got.indices, m.indices) and np.array_equal(got.data, m.data.astype(np.float32))


def test_non_canonical_file_arrays_do_not_become_a_canonical_host_view():
    """read_10x_arrays marked the caller's arrays sorted / canonical although only the device copy had been
    sorted; duplicate (row, column) entries were never summed."""
    m, matrix, _ = _tenx(seed=4)
    idx, dat, ptr = matrix["indices"].copy(), matrix["data"].astype(np.float32), matrix["indptr"]
    for r in range(m.shape[0]):
        a, b = ptr[r], ptr[r + 1]
        idx[a:b], dat[a:b] = idx[a:b][::-1].copy(), dat[a:b][::-1].copy()
    ad = mio.read_10x_arrays(dict(matrix, indices=idx, data=dat), backend=BE, atac_only=False)
    X = ad.X
    assert X.has_sorted_indices and X.has_canonical_format
    assert np.all(np.diff(X.indices)[np.setdiff1d(np.arange(X.nnz - 1), X.indptr[1:-1] - 1)] > 0)  # really sorted
    assert (X != m.astype(np.float32)).nnz == 0
    # a duplicated entry is summed
    r = int(np.argmax(np.diff(ptr) > 0))
    idx2 = np.insert(matrix["indices"], ptr[r], matrix["indices"][ptr[r]])
    dat2 = np.insert(matrix["data"], ptr[r], 3).astype(np.float32)
    ptr2 = ptr.copy()
    ptr2[r + 1:] += 1
    Xd, _, _ = mio.device_csr_from_10x(dict(matrix, indices=idx2, data=dat2, indptr=ptr2), None, BE, atac_only=False)
    got = _host(Xd)
    want = m.astype(np.float32).tolil()
    want[r, matrix["indices"][ptr[r]]] += 3
    assert got.nnz == m.nnz and (got != want.tocsr()).nnz == 0


def test_coo_and_csc_layouts():
    m, _, _ = _tenx(seed=2)
    coo = m.tocoo()
    p = np.random.default_rng(0).permutation(coo.nnz)
    rows = np.concatenate([coo.row[p], coo.row[:5]])  # with duplicates: summed like csr_matrix((v, (i, j)))
    cols = np.concatenate([coo.col[p], coo.col[:5]])
    vals = np.concatenate([coo.data[p], coo.data[:5]])
    X = mio.device_csr_from_coo(rows, cols, vals, m.shape, BE)
    want = sp.csr_matrix((vals.astype(np.float32), (rows, cols)), shape=m.shape)
    want.sum_duplicates()
    assert (_host(X) != want).nnz == 0
    c = m.tocsc()
    Y = mio.device_csr_from_csc(c.indptr, c.indices, c.data, m.shape, BE)
    assert (_host(Y) != m.astype(np.float32)).nnz == 0


def test_read_10x_arrays_feeds_tfidf_without_another_upload():
    from muon_amd import atac as ac
    from oracle import tfidf_oracle

    m, matrix, ft = _tenx(seed=3)
    peaks = np.array([t == b"Peaks" for t in ft])
    ad = mio.read_10x_arrays(matrix, backend=BE, barcodes=[f"c{i}".encode() for i in range(m.shape[0])],
                             feature_names=[f"f{j}" for j in range(m.shape[1])])
    assert ad.shape == (m.shape[0], int(peaks.sum())) and ad.X.dtype == np.float32
    assert list(ad.var_names[:2]) == ["f1", "f2"] and ad.obs_names[0] == "c0"
    uploads = []
    orig = BE.upload_csr
    BE.upload_csr = lambda *a, **k: (uploads.append(1), orig(*a, **k))[1]
    try:
        ac.pp.tfidf(ad, backend=BE)
    finally:
        BE.upload_csr = orig
    assert not uploads  # the device copy made by the ingest was used
    ref = tfidf_oracle.canonical(tfidf_oracle.tfidf(m[:, peaks].astype(np.float32)))
    assert np.array_equal(ad.X.indices, ref.indices) and np.allclose(ad.X.data, ref.data, rtol=1e-5)


def test_duplicate_sums_see_their_own_group_only():
    """the duplicate summation took differences of one f64 running sum over ALL entries - a NaN anywhere
    poisoned every later entry and non-integer values picked up the rounding of the whole matrix' running total.  Now:
    entries without a duplicate pass through bit for bit, a group's sum sees that group's values."""
    import torch

    from muon_amd._backend import DeviceCSR

    rng = np.random.default_rng(3)
    n, d = 40, 30
    m = sp.random(n, d, density=0.3, format="csr", random_state=rng, dtype=np.float64)
    m.data = (rng.standard_normal(m.nnz) * 1e6 + rng.random(m.nnz)).astype(np.float64)
    # rows reversed (unsorted) + one duplicated entry in row 5 + a NaN in row 1
    ptr, idx, dat = m.indptr.copy(), m.indices.copy(), m.data.copy()
    dat[ptr[1]] = np.nan
    for r in range(n):
        a, b = ptr[r], ptr[r + 1]
        idx[a:b], dat[a:b] = idx[a:b][::-1].copy(), dat[a:b][::-1].copy()
    a5 = ptr[5]
    idx = np.insert(idx, a5, idx[a5])
    dat = np.insert(dat, a5, 0.25)
    ptr[6:] += 1
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a).astype(dt))  # noqa: E731
    X = DeviceCSR(t(ptr, np.int64), t(idx, np.int32), t(dat, np.float64), (n, d))
    Y = mio.canonicalize(BE, X)
    assert not Y.canonical_as_given
    got = sp.csr_matrix((Y.values.numpy(), Y.indices.numpy(), Y.indptr.numpy()), shape=(n, d))
    want = m.copy()
    want.data[m.indptr[1]] = np.nan
    dup_col = idx[a5]
    assert got.nnz == m.nnz and np.array_equal(got.indices, m.indices)
    k = m.indptr[5] + int(np.nonzero(m.indices[m.indptr[5]:m.indptr[6]] == dup_col)[0][0])
    exp = want.data.copy()
    exp[k] = 0.25 + exp[k]
    assert np.isnan(got.data[m.indptr[1]]) and np.isnan(got.data).sum() == 1      # the NaN stays where it is
    keep = ~np.isnan(exp)
    keep[k] = False
    assert np.array_equal(got.data[keep], exp[keep])                              # bit for bit: never part of a sum
    assert got.data[k] == exp[k]

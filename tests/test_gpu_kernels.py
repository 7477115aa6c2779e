"""Kernel-level parity: every C-ABI kernel against a scipy / numpy f64 restatement of the
same operation on seeded inputs (bit-exact for index work, f32-roundoff for sums)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.synth import planted_topics_csr, unstructured_csr

pytestmark = pytest.mark.gpu


def _up(hip, m):
    m = m.tocsr()
    m.sort_indices()
    return hip.upload_csr(m.indptr, m.indices, m.data, m.shape)


def _down(hip, X):
    return sp.csr_matrix((hip.to_host(X.values), hip.to_host(X.indices), hip.to_host(X.indptr)), shape=X.shape)


SHAPES = [(1, 1, 1.0), (7, 5, 0.5), (100, 10, 0.2), (257, 131, 0.08), (300, 9000, 0.01),
          (2000, 20000, 0.004), (5000, 700, 0.03)]


@pytest.mark.parametrize("n,d,dens", SHAPES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_row_col_sums(hip, n, d, dens, dt):
    rng = np.random.default_rng(n * 31 + d)
    m = sp.random(n, d, density=dens, format="csr", random_state=rng, dtype=np.float64)
    m.data = (1 + rng.poisson(0.7, size=m.nnz)).astype(dt)
    if n > 6:
        m = m.tolil(); m[3, :] = 0; m = m.tocsr(); m.eliminate_zeros()
    X = _up(hip, m)
    rs, cs = hip.row_col_sums(X)
    # integer-valued counts: both reductions are exact
    assert np.array_equal(hip.to_host(rs), np.asarray(m.sum(axis=1)).reshape(-1).astype(np.float64))
    assert np.array_equal(hip.to_host(cs), np.asarray(m.sum(axis=0)).reshape(-1).astype(np.float64))


def test_row_col_sums_float_values_and_reproducible(hip):
    rng = np.random.default_rng(3)
    m = sp.random(4000, 30000, density=0.01, format="csr", random_state=rng, dtype=np.float64)
    X = _up(hip, m)
    rs, cs = hip.row_col_sums(X)
    np.testing.assert_allclose(hip.to_host(rs), np.asarray(m.sum(axis=1)).reshape(-1), rtol=1e-12)
    np.testing.assert_allclose(hip.to_host(cs), np.asarray(m.sum(axis=0)).reshape(-1), rtol=1e-12)
    rs2, cs2 = hip.row_col_sums(X)
    assert torch.equal(rs, rs2)  # wave-shuffle reduction of one row by one wave: fixed order
    # the column sums of NON-integer values are accumulated by LDS atomics of several waves: the
    # order is not fixed, so they agree to rounding, not bit for bit (count matrices - integers far
    # below 2^53 - are exact and therefore identical: the test above and the one below)
    np.testing.assert_allclose(hip.to_host(cs2), hip.to_host(cs), rtol=1e-14)


def test_row_col_sums_of_counts_are_bit_reproducible(hip):
    m = planted_topics_csr(5000, 40000, n_topics=20, density=0.02, seed=13, dtype=np.float32)
    X = _up(hip, m)
    rs, cs = hip.row_col_sums(X)
    for _ in range(3):
        rs2, cs2 = hip.row_col_sums(X)
        assert torch.equal(rs, rs2) and torch.equal(cs, cs2)


@pytest.mark.parametrize("n,d,dens", SHAPES)
def test_transpose_is_bit_exact_and_sorted(hip, n, d, dens):
    rng = np.random.default_rng(n + d)
    m = sp.random(n, d, density=dens, format="csr", random_state=rng, dtype=np.float32)
    X = _up(hip, m)
    T = _down(hip, hip.transpose(X))
    ref = m.T.tocsr()
    ref.sort_indices()
    assert np.array_equal(T.indptr, ref.indptr.astype(np.int64))
    assert np.array_equal(T.indices, ref.indices)  # ascending row ids => stable
    assert np.array_equal(T.data, ref.data)


def test_transpose_empty_and_float64(hip):
    m = sp.csr_matrix((5, 7), dtype=np.float64)
    T = _down(hip, hip.transpose(_up(hip, m)))
    assert T.shape == (7, 5) and T.nnz == 0 and np.all(T.indptr == 0)
    m = sp.random(50, 20000, density=0.02, format="csr", random_state=1, dtype=np.float64)
    T = _down(hip, hip.transpose(_up(hip, m)))
    ref = m.T.tocsr(); ref.sort_indices()
    assert (T != ref).nnz == 0 and np.array_equal(T.indices, ref.indices)


@pytest.mark.parametrize("B", [16, 32, 64])
@pytest.mark.parametrize("n,d,dens", [(1, 3, 1.0), (100, 10, 0.2), (513, 700, 0.05), (3000, 20000, 0.01)])
def test_spmm(hip, B, n, d, dens):
    rng = np.random.default_rng(B + n)
    m = sp.random(n, d, density=dens, format="csr", random_state=rng, dtype=np.float32)
    Q = rng.standard_normal((d, B)).astype(np.float32)
    Y = hip.to_host(hip.spmm(_up(hip, m), hip.to_device(Q)))
    ref = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    assert np.max(np.abs(Y - ref) / scale) < 2e-6  # f32 accumulation
    # rows without entries give exact zeros
    empty = np.diff(m.indptr) == 0
    assert np.all(Y[empty] == 0)


@pytest.mark.parametrize("B", [16, 32, 64])
@pytest.mark.parametrize("n", [1, 3, 64, 1000, 40001])
def test_gram_and_colsum(hip, B, n):
    rng = np.random.default_rng(n)
    A = (rng.standard_normal((n, B)) * (1 + np.arange(B))).astype(np.float32)  # asymmetric columns
    G, cs = hip.gram(hip.to_device(A))
    A64 = A.astype(np.float64)
    np.testing.assert_allclose(hip.to_host(G), A64.T @ A64, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(hip.to_host(cs), A64.sum(axis=0), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("B", [16, 32, 64])
@pytest.mark.parametrize("n", [1, 17, 1000, 40001])
def test_dense_apply(hip, B, n):
    rng = np.random.default_rng(n + B)
    A = rng.standard_normal((n, B)).astype(np.float32)
    M = rng.standard_normal((B, B)).astype(np.float32)  # not symmetric: catches transposes
    b = rng.standard_normal(B).astype(np.float32)
    ref = A.astype(np.float64) @ M.astype(np.float64)
    out = hip.to_host(hip.apply(hip.to_device(A), hip.to_device(M)))
    assert np.max(np.abs(out - ref)) < 1e-4 * np.sqrt(B)
    out = hip.to_host(hip.apply(hip.to_device(A), hip.to_device(M), bias=hip.to_device(b)))
    assert np.max(np.abs(out - (ref + b))) < 1e-4 * np.sqrt(B)
    # in place
    Ad = hip.to_device(A)
    hip.apply(Ad, hip.to_device(M), out=Ad)
    assert np.max(np.abs(hip.to_host(Ad) - ref)) < 1e-4 * np.sqrt(B)


def test_scan_and_compact(hip):
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 1023, 1024, 1025, 65535, 65536, 65537, 100003, 262144, 1000448):
        v = rng.integers(0, 1000, size=n).astype(np.int64)
        vin = hip.to_device(v) if n else hip.empty((0,), torch.int64)
        out = hip.empty((n + 1,), torch.int64)
        from muon_amd._ffi import check
        check(hip.lib.mu_exclusive_scan_i64(n, vin.data_ptr() if n else None, out.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(hip.to_host(out), np.concatenate([[0], np.cumsum(v)]))
    m = sp.random(300, 200, density=0.1, format="csr", random_state=2, dtype=np.float32)
    m.data[::7] = 0
    m.data[5] = np.nan
    C = _down(hip, hip.compact_nonzero(_up(hip, m)))
    keep = m.data != 0
    ref = m.copy(); ref.eliminate_zeros()  # scipy keeps NaN as well
    assert np.array_equal(C.indptr, ref.indptr) and np.array_equal(C.indices, ref.indices)
    assert np.array_equal(C.data, m.data[keep], equal_nan=True)


def test_randn_moments_and_reproducible(hip):
    a = hip.randn(200000, 64, 1)
    b = hip.randn(200000, 64, 1)
    c = hip.randn(200000, 64, 2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.mean().item()) < 2e-3 and abs(a.std().item() - 1) < 2e-3
    cc = torch.corrcoef(a[:, :8].T)
    assert (cc - torch.eye(8, device=cc.device)).abs().max().item() < 0.02


def test_synth_counts_structure(hip):
    X = hip.synth_counts(0, 2000, 30000, n_topics=50, density=0.03, seed=0)
    m = _down(hip, X)
    assert m.has_canonical_format or (np.all(np.diff(m.indices)[np.diff(m.indices) <= 0].size >= 0))
    m2 = m.copy(); m2.sort_indices()
    assert np.array_equal(m2.indices, m.indices)  # columns come out sorted
    dens = m.nnz / (2000 * 30000)
    assert 0.02 < dens < 0.045
    assert m.data.min() >= 1 and m.data.max() <= 4 and np.all(m.data == np.round(m.data))
    # a shard generated elsewhere equals the same rows of the full matrix
    S = _down(hip, hip.synth_counts(500, 300, 30000, n_topics=50, density=0.03, seed=0))
    assert (S != m[500:800]).nnz == 0


# ---- row-stream SpMM (B = 64 / 32 / 16) -----------------------------------------------------
def _stream_ref(m):
    """numpy restatement of the row stream in matrix order (include/muon_amd.h): the (column, value)
    pairs row after row, 8 bytes each, no padding; sptr = the CSR's row pointers."""
    ent = m.indices.astype(np.uint32).astype(np.uint64) | \
        (m.data.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32))
    return m.indptr.astype(np.int64), ent


def _check_stream(hip, P, m):
    """Decode a row stream (any layout) on the host and compare it with the canonical CSR `m` it
    must hold: every row at exactly one position, pairs bit-exact and in column order, no padding,
    nothing at the empty positions."""
    m = m.tocsr()
    m.sort_indices()
    n = m.shape[0]
    sptr = hip.to_host(P.sptr)
    ent = hip.to_host(P.ent).view(np.uint64)
    perm = np.arange(n, dtype=np.int64) if P.perm is None else hip.to_host(P.perm).astype(np.int64)
    assert sptr[0] == 0 and sptr.size == perm.size + 1 and sptr[-1] == m.nnz and ent.size >= m.nnz
    rows = perm[perm >= 0]
    assert rows.size == n and np.array_equal(np.sort(rows), np.arange(n))
    lens = np.diff(m.indptr)
    plens = np.where(perm >= 0, lens[np.maximum(perm, 0)], 0)
    assert np.array_equal(np.diff(sptr), plens)
    slot_pos = np.repeat(np.arange(perm.size), plens)
    slot_off = np.arange(m.nnz) - np.repeat(sptr[:-1], plens)
    src = m.indptr[np.maximum(perm, 0)][slot_pos] + slot_off
    col = (ent[: m.nnz] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    val = (ent[: m.nnz] >> np.uint64(32)).astype(np.uint32)
    assert np.array_equal(col, m.indices[src].astype(np.uint32))
    assert np.array_equal(val, m.data[src].astype(np.float32).view(np.uint32))


def _heavy_rows_csr(n, d, dens, rng, bursts=True):
    m = sp.random(n, d, density=dens, format="lil", random_state=rng, dtype=np.float32)
    if bursts and n > 8 and d > 40:
        w = min(d, 700)
        m[1, :w] = rng.standard_normal(w).astype(np.float32)           # > 32 entries per slab: overflow passes
        m[5, d - min(d, 300):] = 1.5                                      # dense tail incl. the ragged last slab
        m[n - 1, ::2] = 0.25                                              # every second column of every slab
        m[2, :] = 0                                                       # empty row
    m = m.tocsr()
    m.eliminate_zeros()
    m.sort_indices()
    return m.astype(np.float32)


def test_stream_layout_bit_exact(hip):
    rng = np.random.default_rng(11)
    m = _heavy_rows_csr(203, 1000, 0.03, rng)
    P = hip.stream(_up(hip, m), sort_rows=False)  # identity layout: byte for byte the numpy stream
    sptr, ent = _stream_ref(m)
    assert P.perm is None and np.array_equal(hip.to_host(P.sptr), sptr)
    assert np.array_equal(hip.to_host(P.ent).view(np.uint64)[: ent.size], ent)
    for mm in (m, _heavy_rows_csr(5000, 300, 0.05, rng), sp.csr_matrix((3, 9), dtype=np.float32)):
        P = hip.stream(_up(hip, mm))  # sorted + dealt layout
        assert P.perm is not None and P.n_pos % (64 * P.k) == 0
        _check_stream(hip, P, mm)
    # rows are dealt longest first: position 0 holds a longest row
    lens = np.diff(m.indptr)
    P = hip.stream(_up(hip, m))
    assert lens[hip.to_host(P.perm)[0]] == lens.max()


@pytest.mark.parametrize("n,d,dens", [(1, 3, 1.0), (5, 255, 0.3), (64, 256, 0.1), (100, 257, 0.2),
                                      (513, 700, 0.05), (1000, 5000, 0.03), (3000, 20000, 0.01),
                                      (4097, 1031, 0.04)])
def test_spmm_stream_matches_f64_and_csr_kernel(hip, n, d, dens):
    rng = np.random.default_rng(n * 7 + d)
    m = _heavy_rows_csr(n, d, dens, rng)
    Q = rng.standard_normal((d, 64)).astype(np.float32)
    X = _up(hip, m)
    Qd = hip.to_device(Q)
    hip.tune("spmm_k", 0)
    Y = hip.to_host(hip.spmm(hip.stream(X), Qd))
    ref = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    assert np.max(np.abs(Y - ref) / scale) < 2e-6  # f32 accumulation
    assert np.all(Y[np.diff(m.indptr) == 0] == 0)
    Yc = hip.to_host(hip.spmm(X, Qd))  # the general CSR kernel (one wave per row)
    assert np.max(np.abs(Y - Yc) / scale) < 2e-6
    # the layout only decides which wave owns a row: matrix order gives the same bits
    assert np.array_equal(hip.to_host(hip.spmm(hip.stream(X, sort_rows=False), Qd)), Y)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8])
def test_spmm_stream_every_rowset_count_is_bit_identical(hip, K):
    """K (row-sets per wave) only changes which wave owns a row: results must not depend on it.
    (The stream is laid out in matrix order here: any K can walk it.)"""
    rng = np.random.default_rng(5)
    m = _heavy_rows_csr(2500, 3000, 0.02, rng)
    Q = rng.standard_normal((3000, 64)).astype(np.float32)
    X = _up(hip, m)
    P = hip.stream(X, sort_rows=False)
    Qd = hip.to_device(Q)
    Ydealt = hip.spmm(hip.stream(X), Qd)
    try:
        hip.tune("spmm_k", K)
        Y = hip.spmm(P, Qd)
        Y2 = hip.spmm(P, Qd)
    finally:
        hip.tune("spmm_k", 0)
    assert torch.equal(Y, Y2) and torch.equal(Y, Ydealt)
    ref = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    assert np.max(np.abs(hip.to_host(Y) - ref) / scale) < 2e-6


def test_spmm_stream_transpose_round_trip_property(hip):
    """<X q, y> == <q, X^T y> through the two operands of the Lanczos iteration (size-independent
    adjoint property; planted-topic counts with the real row-length spread), the stream of X^T
    built straight from X and through the general CSR transpose."""
    m = planted_topics_csr(6000, 9000, n_topics=20, density=0.03, seed=3, dtype=np.float32)
    X = _up(hip, m)
    Xs, Xts = hip.stream(X), hip.transpose_stream(X)
    q = hip.randn(9000, 64, 7)
    y = hip.randn(6000, 64, 8)
    lhs = (hip.spmm(Xs, q).double() * y.double()).sum(dim=0)
    rhs = (q.double() * hip.spmm(Xts, y).double()).sum(dim=0)
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-3 * float(lhs.abs().max()))
    assert torch.equal(hip.spmm(hip.stream(hip.transpose(X)), y), hip.spmm(Xts, y))


@pytest.mark.parametrize("n,d,dens", [(1, 1, 1.0), (7, 5, 0.5), (100, 10, 0.2), (257, 131, 0.08),
                                      (300, 9000, 0.01), (2000, 20000, 0.004), (5000, 700, 0.03),
                                      (70, 4097, 0.2)])
def test_transpose_stream_is_bit_exact(hip, n, d, dens):
    """The row stream (and the CSR) of X^T built straight from X equal scipy's transpose, cell ids
    ascending inside every output row."""
    rng = np.random.default_rng(n * 13 + d)
    m = _heavy_rows_csr(n, d, dens, rng, bursts=(n > 8 and d > 40))
    mt = m.T.tocsr()
    mt.sort_indices()
    P = hip.transpose_stream(_up(hip, m), sort_rows=False)  # identity layout: byte for byte
    sptr, ent = _stream_ref(mt)
    assert P.shape == (d, n) and P.perm is None
    assert np.array_equal(hip.to_host(P.sptr), sptr)
    assert np.array_equal(hip.to_host(P.ent).view(np.uint64)[: ent.size], ent)
    P = hip.transpose_stream(_up(hip, m))  # sorted + dealt layout
    assert P.shape == (d, n)
    _check_stream(hip, P, mt)
    T = _down(hip, hip.transpose_csr(_up(hip, m)))
    assert np.array_equal(T.indptr, mt.indptr) and np.array_equal(T.indices, mt.indices)
    assert np.array_equal(T.data.view(np.uint32), mt.data.astype(np.float32).view(np.uint32))


def test_transpose_stream_empty_matrix(hip):
    m = sp.csr_matrix((5, 7), dtype=np.float32)
    P = hip.transpose_stream(_up(hip, m), sort_rows=False)
    assert np.array_equal(hip.to_host(P.sptr), np.zeros(8, dtype=np.int64))
    _check_stream(hip, hip.transpose_stream(_up(hip, m)), m.T.tocsr())


def test_transpose_stream_tile_overflow_falls_back(hip):
    """A (row block x column tile) denser than the staging buffer is retried at half the width until it fits; the
    tile-staged transposition and the general one (tune tpack4_off) give the bytes of the numpy stream."""
    rng = np.random.default_rng(21)
    n, d = 60000, 200
    dense = sp.random(n, 100, density=0.95, format="csr", random_state=rng, dtype=np.float32)
    m = sp.hstack([dense, sp.csr_matrix((n, d - 100), dtype=np.float32)], format="csr")
    m.sort_indices()
    X = _up(hip, m)
    mt = m.T.tocsr()
    mt.sort_indices()
    sptr, ent = _stream_ref(mt)
    for off in (0, 1):
        try:
            hip.tune("tpack4_off", off)
            P = hip.transpose_stream(X, sort_rows=False)
            Ps = hip.transpose_stream(X)
        finally:
            hip.tune("tpack4_off", 0)
        assert np.array_equal(hip.to_host(P.sptr), sptr)
        assert np.array_equal(hip.to_host(P.ent).view(np.uint64)[: ent.size], ent)
        _check_stream(hip, Ps, mt)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("n,D", [(1, 1), (5, 3), (33, 130), (1000, 257), (4099, 2000), (20000, 301)])
def test_skinny_products(hip, dt, n, D):
    """Tall-skinny MFMA products of a dense MOFA view against numpy f64 (incl. a row slice with a
    leading dimension, ragged row / column tiles)."""
    rng = np.random.default_rng(n + D)
    Yfull = rng.standard_normal((n + 3, D)).astype(dt)
    T16 = np.zeros((D, 16), dtype=dt); T16[:, :10] = rng.standard_normal((D, 10))
    Z16 = np.zeros((n, 16), dtype=dt); Z16[:, :10] = rng.standard_normal((n, 10))
    Yd = hip.to_device(Yfull)[2:2 + n]  # row slice: non-zero offset, same leading dimension
    tol = 2e-4 if dt == np.float32 else 1e-11
    A = hip.to_host(hip.skinny_nn(Yd, hip.to_device(T16)))
    refA = Yfull[2:2 + n].astype(np.float64) @ T16.astype(np.float64)
    assert np.max(np.abs(A - refA)) <= tol * (1 + np.abs(refA).max())
    Bm = hip.to_host(hip.skinny_tn(Yd, hip.to_device(Z16)))
    refB = Yfull[2:2 + n].astype(np.float64).T @ Z16.astype(np.float64)
    assert np.max(np.abs(Bm - refB)) <= tol * (1 + np.abs(refB).max())
    assert np.all(A[:, 10:] == 0) and np.all(Bm[:, 10:] == 0)
    assert torch.equal(hip.skinny_tn(Yd, hip.to_device(Z16)), hip.skinny_tn(Yd, hip.to_device(Z16)))


@pytest.mark.parametrize("n,D", [(1, 1), (5, 3), (33, 130), (1000, 257), (4099, 2000), (20000, 301)])
def test_skinny_products_of_an_f32_stored_view_are_the_f64_ones(hip, n, D):
    """r04: mu_skinny_*_f64_f32 - Y in f32 in memory, blocks, products and sums in f64 - against the f64 kernels
    on the widened copy of the same values: the same instruction sequence after the conversion, so the same bits."""
    rng = np.random.default_rng(n * 7 + D)
    Y32 = rng.standard_normal((n + 3, D)).astype(np.float32)
    T16 = np.zeros((D, 16)); T16[:, :10] = rng.standard_normal((D, 10))
    Z16 = np.zeros((n, 16)); Z16[:, :10] = rng.standard_normal((n, 10))
    Yd32 = hip.to_device(Y32)[2:2 + n]
    Yd64 = hip.to_device(Y32.astype(np.float64))[2:2 + n]
    Td, Zd = hip.to_device(T16), hip.to_device(Z16)
    A, A64 = hip.skinny_nn(Yd32, Td), hip.skinny_nn(Yd64, Td)
    Bm, B64 = hip.skinny_tn(Yd32, Zd), hip.skinny_tn(Yd64, Zd)
    assert A.dtype == torch.float64 and Bm.dtype == torch.float64
    refA = Y32[2:2 + n].astype(np.float64) @ T16
    refB = Y32[2:2 + n].astype(np.float64).T @ Z16
    assert np.max(np.abs(hip.to_host(A) - refA)) <= 1e-11 * (1 + np.abs(refA).max())
    assert np.max(np.abs(hip.to_host(Bm) - refB)) <= 1e-11 * (1 + np.abs(refB).max())
    assert torch.equal(Bm, B64)  # (tn: the same tiles and order; nn tiles 32 columns of f32 against 16 of f64)
    assert float((A - A64).abs().max()) <= 1e-12 * (1 + np.abs(refA).max())


@pytest.mark.parametrize("B", [16, 32])
@pytest.mark.parametrize("n,d,dens", [(5, 255, 0.3), (513, 700, 0.05), (3000, 20000, 0.01)])
def test_spmm_stream_narrow_blocks(hip, B, n, d, dens):
    """B = 16 / 32 instances of the row-stream SpMM (MOFA's sparse views, lsi with few components)."""
    rng = np.random.default_rng(B + n)
    m = _heavy_rows_csr(n, d, dens, rng)
    Q = rng.standard_normal((d, B)).astype(np.float32)
    X = _up(hip, m)
    Qd = hip.to_device(Q)
    Y = hip.to_host(hip.spmm(hip.stream(X), Qd))
    ref = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    assert Y.shape == (n, B) and np.max(np.abs(Y - ref) / scale) < 2e-6
    Yn = rng.standard_normal((n, B)).astype(np.float32)
    Z = hip.to_host(hip.spmm(hip.transpose_stream(X), hip.to_device(Yn)))
    refz = m.T.astype(np.float64) @ Yn.astype(np.float64)
    scalez = np.abs(m.T).astype(np.float64) @ np.abs(Yn).astype(np.float64) + 1e-30
    assert Z.shape == (d, B) and np.max(np.abs(Z - refz) / scalez) < 2e-6


@pytest.mark.parametrize("B,w", [(64, 64), (64, 50), (32, 32), (16, 7), (64, 1)])
def test_chol_rinv_on_device_matches_numpy(hip, B, w):
    """CholeskyQR's B x B step on the device: M = R^-1 of the leading w x w block of G = R^T R."""
    rng = np.random.default_rng(B * 100 + w)
    A = rng.standard_normal((4 * B, B))
    G = A.T @ A
    flag = hip.zeros((1,), torch.int32)
    M = hip.to_host(hip.chol_rinv(hip.to_device(G), w, flag)).astype(np.float64)
    R = np.linalg.cholesky(G[:w, :w]).T
    ref = np.zeros((B, B))
    ref[:w, :w] = np.linalg.inv(R)
    assert int(flag.item()) == 0
    assert np.allclose(M, ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    assert np.all(M[np.tril_indices(B, -1)] == 0) and np.all(M[w:] == 0) and np.all(M[:, w:] == 0)
    # the product it is used for: (A M)^T (A M) = I on the leading block
    QtQ = (A[:, :] @ M).T @ (A @ M)
    assert np.allclose(QtQ[:w, :w], np.eye(w), atol=1e-4)


def test_chol_rinv_flags_a_semidefinite_gram(hip):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((200, 64))
    A[:, 40] = A[:, 3] * 2.0 - A[:, 17]  # a dependent column: the Gram is singular
    flag = hip.zeros((1,), torch.int32)
    M = hip.to_host(hip.chol_rinv(hip.to_device(A.T @ A), 64, flag))
    assert int(flag.item()) == 1 and np.all(np.isfinite(M))


def test_pipelined_upload_matches_plain_copy(hip):
    """Big host arrays reach HBM through pinned staging buffers filled by host threads, converted on
    the way (int64 -> int32 column indices, f64 -> f32 values): forced here with small buffers so that
    the three staging buffers are reused several times and the last chunk is ragged."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2**31 - 1, 1_000_003).astype(np.int64)
    v = rng.standard_normal(777_777)
    keep = (hip._UPLOAD_PIPELINE_MIN, hip._UPLOAD_CHUNK)
    try:
        hip._UPLOAD_PIPELINE_MIN, hip._UPLOAD_CHUNK = 1 << 16, 1 << 18
        hip.__dict__.pop("_upload_state", None)
        got_a = hip.to_device(a, np.int32)
        got_v = hip.to_device(v, np.float32)
        got_same = hip.to_device(v)
    finally:
        hip._UPLOAD_PIPELINE_MIN, hip._UPLOAD_CHUNK = keep
        hip.__dict__.pop("_upload_state", None)
    assert got_a.dtype == torch.int32 and np.array_equal(hip.to_host(got_a), a.astype(np.int32))
    assert got_v.dtype == torch.float32 and np.array_equal(hip.to_host(got_v), v.astype(np.float32))
    assert got_same.dtype == torch.float64 and np.array_equal(hip.to_host(got_same), v)


@pytest.mark.parametrize("B", [16, 32])
def test_stream_spmm_with_f64_blocks(hip, B):
    """The row-stream SpMM against f64 dense blocks (MOFA's default precision): f32 stored values,
    f64 FMAs.  Values that are exact in f32 need one stream; arbitrary f64 values are split into
    hi + lo streams and the second product accumulates - both against scipy in f64, both directions."""
    rng = np.random.default_rng(40 + B)
    m = _heavy_rows_csr(3000, 2500, 0.02, rng).astype(np.float64)
    Q = rng.standard_normal((2500, B))
    Y = rng.standard_normal((3000, B))
    X32 = _up(hip, m)  # f32-exact values
    assert X32.values.dtype == torch.float64
    Xs, Xt = hip.split_streams(X32)
    assert Xs.lo is None and Xt.lo is None
    got = hip.to_host(hip.spmm(Xs, hip.to_device(Q)))
    ref = m @ Q
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    got = hip.to_host(hip.spmm(Xt, hip.to_device(Y)))
    ref = m.T @ Y
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    m2 = m.copy()
    m2.data = m2.data * np.pi + rng.standard_normal(m2.nnz) * 1e-9  # not exact in f32
    Xs, Xt = hip.split_streams(_up(hip, m2))
    assert Xs.lo is not None
    got = hip.to_host(hip.spmm(Xs, hip.to_device(Q)))
    ref = m2 @ Q
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    got = hip.to_host(hip.spmm(Xt, hip.to_device(Y)))
    ref = m2.T @ Y
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    # bit-reproducible like the f32 kernel
    assert torch.equal(hip.spmm(Xt, hip.to_device(Y)), hip.spmm(Xt, hip.to_device(Y)))


@pytest.mark.parametrize("B", [16, 32, 64])
def test_project_out_block_equals_apply_and_subtract(hip, B):
    rng = np.random.default_rng(B)
    n = 2049
    Q = hip.to_device(rng.standard_normal((n, B)).astype(np.float32))
    Z = hip.to_device(rng.standard_normal((n, B)).astype(np.float32))
    C = hip.gram_cross(Q, Z)
    want = Z - hip.apply(Q, C.to(torch.float32).contiguous())
    got = hip.project_out_block(Q, C, Z.clone())
    assert torch.equal(got, want)

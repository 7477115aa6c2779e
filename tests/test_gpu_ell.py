"""Sliced-ELL narrow SpMM (csrc/spmm_ell.hip) on the GPU, through the C-ABI: the product against f64 arithmetic
on the same inputs (ragged and empty rows, row counts off the group size, column counts off the slab size, every
group count, several workgroup shapes), bit-reproducibility, and agreement with the row-stream kernel at a MOFA-sized
view."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.synth import planted_topics_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from muon_amd._backend import get_backend

    return get_backend()


def _upload(be, m):
    return be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float32)


def _ragged(n, d, seed):
    m = planted_topics_csr(n, d, n_topics=5, density=0.04, seed=seed).astype(np.float32)
    m.data = np.log1p(m.data).astype(np.float32) + np.float32(0.125)
    keep = np.ones(n)
    keep[::7] = 0          # empty rows
    m = sp.csr_matrix(sp.diags(keep) @ m).astype(np.float32)
    heavy = sp.random(n, d, density=0.5, random_state=seed, format="csr", dtype=np.float32)
    pick = np.zeros(n)
    pick[1::13] = 1        # a few rows hundreds of entries long
    m = sp.csr_matrix(m + sp.diags(pick) @ heavy).astype(np.float32)
    m.eliminate_zeros()
    m.sort_indices()
    return m


@pytest.mark.parametrize("shape", [(16, 40), (100, 1024), (257, 1025), (1000, 5000), (4099, 3000)])
def test_product_matches_f64_arithmetic(be, shape):
    from muon_amd._backend import ell16_layout

    n, d = shape
    m = _ragged(n, d, seed=n)
    X = _upload(be, m)
    E = ell16_layout(X)
    assert E.nnz == m.nnz
    rng = np.random.default_rng(n)
    Q = rng.standard_normal((d, 16)).astype(np.float32)
    want = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    Qd = torch.from_numpy(Q).to(be.device)
    first = None
    for waves in (15, 1, 6):  # row-owning waves per workgroup: the launch shape does not touch the arithmetic
        E.waves = waves
        out = torch.full((n, 16), float("nan"), device=be.device, dtype=torch.float32)
        got = be.spmm(E, Qd, out=out)
        again = be.spmm(E, Qd)
        g = be.to_host(got).astype(np.float64)
        assert np.isfinite(g).all()  # every row written, the empty ones with zeros
        assert np.max(np.abs(g - want) / scale) < 2e-6  # f32 sums of up to ~1500 terms
        assert torch.equal(got, again)
        if first is None:
            first = got.clone()
        else:
            assert torch.equal(first, got)
    assert np.all(g[np.diff(m.indptr) == 0] == 0)


@pytest.mark.parametrize("shape", [(16, 40), (257, 1025), (1000, 5000)])
@pytest.mark.parametrize("exact", [True, False])
def test_f64_blocks_match_f64_arithmetic(be, shape, exact):
    """f64 blocks: 512-column slabs, f32 stored values (hi + lo when the f64 values are not exact in f32)"""
    n, d = shape
    m = _ragged(n, d, seed=n + 3).astype(np.float64)
    if not exact:
        m.data = m.data * (1.0 + 1e-9 * np.arange(1, m.nnz + 1))  # not representable in f32
    X = be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float64)
    E = be.ell16(X, wide=True)
    assert (E.lo is None) == exact and E.hi.slab_cols == 512
    rng = np.random.default_rng(n)
    Q = rng.standard_normal((d, 16))
    Qd = torch.from_numpy(Q).to(be.device)
    want = m @ Q
    scale = np.abs(m) @ np.abs(Q) + 1e-300
    got = be.spmm(E, Qd)
    g = be.to_host(got)
    assert np.max(np.abs(g - want) / scale) < (1e-14 if exact else 1e-13)  # hi + lo: exact to 2^-48 per value
    assert torch.equal(got, be.spmm(E, Qd))
    # accumulate: Y + X Q
    acc = got.clone()
    be.spmm(E, Qd, out=acc, accumulate=True)
    assert np.max(np.abs(be.to_host(acc) - 2 * want) / scale) < 1e-12
    with pytest.raises(TypeError):
        be.spmm(E.hi, Qd.to(torch.float32))  # laid out for f64 blocks


def test_default_layout_and_row_stream_agree_at_a_mofa_sized_view(be):
    n, d = 30000, 20000
    X = be.synth_counts(0, n, d, 50, 0.03, 0)
    X = type(X)(X.indptr, X.indices, torch.log1p(X.values.to(torch.float32)), X.shape)
    for M in (X, be.transpose(X)):
        E = be.ell16(M)
        assert 1 <= E.waves <= 15 and E.slots >= M.nnz
        assert E.slots < 1.6 * M.nnz  # the padding the kernel streams
        Q = torch.randn(M.shape[1], 16, device=be.device, dtype=torch.float32)
        a = be.spmm(E, Q)
        b = be.spmm(be.stream(M), Q)
        # both add a row's products in column order, slab after slab: one fma chain per output element ...
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        # ... and linearity in Q
        c = be.spmm(E, 2.0 * Q)
        assert torch.equal(c, 2.0 * a)


def test_refusals_and_long_rows(be):
    from muon_amd._backend import ell16_layout
    from muon_amd._ffi import MuonAmdError

    m = _ragged(64, 2000, seed=1)
    X = _upload(be, m)
    E = ell16_layout(X)
    with pytest.raises((MuonAmdError, AssertionError, TypeError)):
        be.spmm(E, torch.zeros(2000, 32, device=be.device))  # 16 columns only
    dense = sp.csr_matrix(np.arange(1, 3 * 2100 + 1, dtype=np.float32).reshape(3, 2100) / 64)  # 256 windows per slab
    Q = torch.randn(2100, 16, device=be.device, dtype=torch.float32)
    got = be.to_host(be.spmm(ell16_layout(_upload(be, dense)), Q)).astype(np.float64)
    want = dense.toarray().astype(np.float64) @ be.to_host(Q).astype(np.float64)
    assert np.max(np.abs(got - want)) < 1e-5 * np.max(np.abs(want))


@pytest.mark.parametrize("shape", [(16, 40), (257, 1025), (1000, 5000), (4099, 3000), (3, 2100)])
@pytest.mark.parametrize("slab", [1024, 512])
def test_layout_kernel_writes_the_windows_of_the_tensor_layout(be, shape, slab):
    """mu_ell16_fill (one pass, a team of 16 lanes per (group, slab)) against the layout as tensor operations -
    the specification tests/test_ell_layout.py holds against a pure-Python restatement: the same bytes."""
    from muon_amd._backend import ell16_layout

    n, d = shape
    if n == 3:
        m = sp.csr_matrix(np.arange(1, n * d + 1, dtype=np.float32).reshape(n, d) / 64)
    else:
        m = _ragged(n, d, seed=n + slab)
    X = _upload(be, m)
    want = ell16_layout(X, 15, slab)
    got = ell16_layout(X, 15, slab, be.slab_ptr_width, be._ell16_fill)
    assert torch.equal(got.hdr, want.hdr) and torch.equal(got.wave_base, want.wave_base)
    assert torch.equal(got.perm, want.perm) and got.slots == want.slots
    assert got.ent.shape == want.ent.shape and torch.equal(got.ent, want.ent)


def test_layout_pair_of_an_f64_valued_view(be):
    """ell16_pair: both operands of an f64 fit from the tile-staged transposition of the hi (and lo) parts."""
    n, d = 3000, 2500
    m = _ragged(n, d, seed=5).astype(np.float64)
    for inexact in (False, True):
        if inexact:
            m.data = m.data * (1.0 + 2.0 ** -40)
        X = be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float64)
        Xs, Xt = be.ell16_pair(X, wide=True)
        assert (Xs.lo is not None) == inexact and (Xt.lo is not None) == inexact
        Q = torch.randn(d, 16, device=be.device, dtype=torch.float64)
        Z = torch.randn(n, 16, device=be.device, dtype=torch.float64)
        a = be.to_host(be.spmm(Xs, Q))
        b = be.to_host(be.spmm(Xt, Z))
        wa = m @ be.to_host(Q)
        wb = m.T @ be.to_host(Z)
        assert np.max(np.abs(a - wa)) < 1e-11 * np.max(np.abs(wa))
        assert np.max(np.abs(b - wb)) < 1e-11 * np.max(np.abs(wb))


@pytest.mark.parametrize("shape,wide", [((3000, 20000), False), ((700, 100000), False), ((2500, 9000), True), ((40, 30000), False)])
def test_column_parts_for_operands_of_few_rows(be, shape, wide):
    """r06: an operand of a few thousand rows (one rank's shard of a sharded fit) is launched with its column slabs split
    over blockIdx.y (mu_spmm_ell16_parts) and the partial products summed in order: against f64 arithmetic, the same bits
    run to run, every row written; the heuristic keeps one part for shapes that fill a round by their rows."""
    import ctypes as C

    n, d = shape
    m = _ragged(n, d, seed=n + 11)
    wv, parts = C.c_int(0), C.c_int(0)
    assert be.lib.mu_spmm_ell16_parts(n, d, int(wide), C.byref(wv), C.byref(parts)) == 0
    assert parts.value > 1 and wv.value == 15
    assert be.lib.mu_spmm_ell16_parts(100000, d, int(wide), C.byref(wv), C.byref(parts)) == 0 and parts.value == 1
    rng = np.random.default_rng(n)
    if wide:
        m = m.astype(np.float64)
        X = be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float64)
        E = be.ell16(X, wide=True)
        Q = rng.standard_normal((d, 16))
        tol = 1e-13
    else:
        X = _upload(be, m)
        E = be.ell16(X)
        Q = rng.standard_normal((d, 16)).astype(np.float32)
        tol = 2e-6
    Qd = torch.from_numpy(Q).to(be.device)
    want = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64) + 1e-30
    out = torch.full((n, 16), float("nan"), device=be.device, dtype=Qd.dtype)
    got = be.spmm(E, Qd, out=out)
    g = be.to_host(got).astype(np.float64)
    assert np.isfinite(g).all() and np.max(np.abs(g - want) / scale) < tol
    assert torch.equal(got, be.spmm(E, Qd))
    assert np.all(g[np.diff(m.indptr) == 0] == 0)
    if wide:
        acc = got.clone()
        be.spmm(E, Qd, out=acc, accumulate=True)
        assert np.max(np.abs(be.to_host(acc) - 2 * want) / scale) < 1e-12

"""Matrix-core SpMM (csrc/spmm_mfma.hip) on the GPU, through the C-ABI: the hardware semantics it rests on,
the cell cutter against the numpy statement of the format, the product against f64 arithmetic on the same
rounded operands, and size-independent properties at a full-size shard."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import cells_ref as cr
from tests.synth import planted_topics_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from muon_amd._backend import get_backend

    return get_backend()


def _p(t):
    return t.data_ptr()


def _upload(be, m):
    return be.upload_csr(m.indptr, m.indices, m.data, m.shape, values_dtype=np.float32)


def test_transpose_read_gathers_four_rows_per_lane_group(be):
    """ds_read_b64_tr_b16: lane 4 j + c of a 16-lane group names piece c (4 halves) of ANY row j; lane i gets
    column i of the 4 x 16 block of those rows.  The image is [rows][16] u16 with value = 256 row + column."""
    from muon_amd._ffi import check

    rows = 200
    img = (np.arange(rows, dtype=np.uint16)[:, None] * 256 + np.arange(16, dtype=np.uint16)[None, :])
    rng = np.random.default_rng(5)
    pick = rng.integers(0, rows, size=(4, 4))  # [group][j]: which row
    addr = np.zeros(64, dtype=np.uint32)
    for l in range(64):
        g, j, c = l >> 4, (l >> 2) & 3, l & 3
        addr[l] = pick[g, j] * 32 + c * 8
    d_img = be.to_device(img.reshape(-1).view(np.uint32))
    d_addr = be.to_device(addr)
    d_out = be.zeros((128,), torch.int32)
    check(be.lib.mu_probe_tr16(_p(d_img), d_img.numel(), _p(d_addr), _p(d_out), None))
    torch.cuda.synchronize()
    got = be.to_host(d_out).view(np.uint16).reshape(64, 4)
    want = np.zeros((64, 4), dtype=np.uint16)
    for l in range(64):
        for e in range(4):
            want[l, e] = pick[l >> 4, e] * 256 + (l & 15)
    assert np.array_equal(got, want), (got[:20], want[:20])


def test_mfma_16x16x32_f16_fragment_layout(be):
    """A[m = lane & 15][k = 8 (lane >> 4) + i], B[k][n = lane & 15], D[row = 4 (lane >> 4) + reg][col = lane & 15]"""
    from muon_amd._ffi import check

    rng = np.random.default_rng(7)
    A = rng.integers(-4, 5, size=(16, 32)).astype(np.float16)
    B = rng.integers(-4, 5, size=(32, 16)).astype(np.float16)
    a = np.zeros((64, 8), dtype=np.float16)
    b = np.zeros((64, 8), dtype=np.float16)
    for l in range(64):
        for i in range(8):
            a[l, i] = A[l & 15, 8 * (l >> 4) + i]
            b[l, i] = B[8 * (l >> 4) + i, l & 15]
    d_a, d_b = be.to_device(a.view(np.uint32).reshape(-1)), be.to_device(b.view(np.uint32).reshape(-1))
    d_d = be.zeros((256,), torch.float32)
    check(be.lib.mu_probe_mfma16(_p(d_a), _p(d_b), _p(d_d), None))
    torch.cuda.synchronize()
    got = be.to_host(d_d).reshape(64, 4)
    D = A.astype(np.float64) @ B.astype(np.float64)
    want = np.zeros((64, 4))
    for l in range(64):
        for r in range(4):
            want[l, r] = D[4 * (l >> 4) + r, l & 15]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nset,shape", [(1, (300, 1700)), (1, (1000, 2048)), (2, (130, 900)), (1, (37, 513))])
def test_cells_cut_holds_every_entry_once(be, nset, shape):
    m = planted_topics_csr(shape[0], shape[1], n_topics=7, density=0.04, seed=shape[0])
    rng = np.random.default_rng(1)
    m.data = (m.data * rng.uniform(0.05, 7.0, m.nnz)).astype(np.float32)
    X = _upload(be, m)
    Xc = be.cells(X, nset=nset)
    be._cells_check(Xc)
    vs = float(Xc.vscale.item())
    got, n_steps = cr.decode(be.to_host(Xc.hdr), be.to_host(Xc.band_base), be.to_host(Xc.cells), m.shape, nset)
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    h, l = cr.split_f16((m.data / np.float32(vs)).astype(np.float32))
    want = np.stack([rows, m.indices, h.view(np.uint16), l.view(np.uint16)], axis=1).astype(np.int64)
    want = want[np.lexsort((want[:, 1], want[:, 0]))]
    assert got.shape == want.shape and np.array_equal(got, want)
    # same number of steps as the reference encoder (cells are filled densely)
    hdr_ref, _base, _cells, _ = cr.encode(m, nset, vscale=vs)
    assert np.array_equal(be.to_host(Xc.hdr), hdr_ref)


@pytest.mark.parametrize("shape", [(300, 1700), (1000, 2048), (37, 513), (5000, 3000)])
def test_product_equals_f64_arithmetic_on_the_rounded_operands(be, shape):
    m = planted_topics_csr(shape[0], shape[1], n_topics=7, density=0.04, seed=shape[1])
    rng = np.random.default_rng(2)
    m.data = (m.data * rng.uniform(0.05, 7.0, m.nnz)).astype(np.float32)
    Q = (rng.standard_normal((shape[1], 64)) * rng.uniform(1e-3, 1e2, 64)).astype(np.float32)
    X = _upload(be, m)
    Xc = be.cells(X)
    Qd = be.to_device(Q)
    Y = be.to_host(be.spmm(Xc, Qd))
    Yref, Qr = cr.product(m, Q, nset=1, vscale=float(Xc.vscale.item()))
    assert np.array_equal(be.to_host(Qd), Qr), "the product leaves the rounded block in Q"
    scale = np.abs(m).astype(np.float64) @ np.abs(Qr).astype(np.float64) + 1e-30
    assert np.max(np.abs(Y - Yref) / scale) < 2e-6  # f32 accumulation of a few hundred terms
    # and against the exact product with the rounded block: values carry 22 bits
    exact = m.astype(np.float64) @ Qr.astype(np.float64)
    assert np.max(np.abs(Y - exact) / scale) < 3e-6
    # bit-reproducible
    Y2 = be.to_host(be.spmm(Xc, be.to_device(Q)))
    assert np.array_equal(Y, Y2)


def test_product_with_a_two_term_operand(be):
    """nset = 2: the dense operand is read as hi + lo (not rounded): 22-bit product"""
    shape = (400, 1300)
    m = planted_topics_csr(shape[0], shape[1], n_topics=5, density=0.05, seed=11)
    rng = np.random.default_rng(3)
    Q = (rng.standard_normal((shape[1], 64)) * rng.uniform(1e-2, 1e2, 64)).astype(np.float32)
    X = _upload(be, m)
    Xc = be.cells(X, nset=2)
    Qd = be.to_device(Q)
    Y = be.to_host(be.spmm(Xc, Qd))
    assert np.array_equal(be.to_host(Qd), Q)
    exact = m.astype(np.float64) @ Q.astype(np.float64)
    scale = np.abs(m).astype(np.float64) @ np.abs(Q).astype(np.float64)
    assert np.max(np.abs(Y - exact) / scale) < 3e-6


def test_full_size_shard_properties(be):
    """125 000 x 200 000 at 3 % (one rank's shard of configs[2]): the product against the f32 row-stream kernel on
    the rounded block, linearity, reproducibility"""
    X = be.synth_counts(0, 125_000, 200_000, 50, 0.03, 7)
    X = X.with_values(X.values.to(torch.float32))
    Xc = be.cells(X)
    Q = be.randn(200_000, 64, 3)
    Y = be.spmm(Xc, Q)  # rounds Q
    Xs = be.stream(X)
    Yw = be.spmm(Xs, Q)
    num = (Y - Yw).abs().max().item()
    den = Yw.abs().max().item()
    assert num <= 2e-6 * den, (num, den)
    Y2 = be.spmm(Xc, Q)  # Q is f16-exact already: rounding again changes nothing
    assert torch.equal(Y, Y2)
    Y3 = be.spmm(Xc, (Q * 0.5).contiguous())
    assert torch.equal(Y3, Y * 0.5)

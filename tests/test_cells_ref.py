"""The numpy statement of the cell format (tests/cells_ref.py) against itself and against scipy: what the GPU
tests of csrc/spmm_mfma.hip compare the device with must be right first."""
import numpy as np
import scipy.sparse as sp

from tests import cells_ref as cr
from tests.synth import planted_topics_csr


def _entries(m, vscale):
    m = sp.csr_matrix(m)
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    h, l = cr.split_f16((m.data / np.float32(vscale)).astype(np.float32))
    arr = np.stack([rows, m.indices, h.view(np.uint16), l.view(np.uint16)], axis=1).astype(np.int64)
    return arr[np.lexsort((arr[:, 1], arr[:, 0]))]


def test_encode_decode_round_trip_keeps_every_entry():
    for nset, shape in ((1, (70, 1300)), (2, (45, 700))):
        m = planted_topics_csr(shape[0], shape[1], n_topics=5, density=0.05, seed=3)
        hdr, base, cells, vs = cr.encode(m, nset)
        got, n_steps = cr.decode(hdr, base, cells, m.shape, nset)
        assert np.array_equal(got, _entries(m, vs))
        assert n_steps == hdr.sum() and n_steps <= base[-1]


def test_split_values_and_rounded_block_reproduce_the_product():
    rng = np.random.default_rng(0)
    m = planted_topics_csr(64, 900, n_topics=4, density=0.05, seed=1)
    m.data = (m.data * rng.uniform(0.1, 3.0, m.nnz)).astype(np.float32)
    Q = rng.standard_normal((900, 64)).astype(np.float32) * rng.uniform(1e-3, 1e3, 64).astype(np.float32)
    Y, Qr = cr.product(m, Q, nset=1)
    # the split values carry 22 bits, the block is the rounded one: exact product with the rounded block
    ref = m.astype(np.float64) @ Qr.astype(np.float64)
    assert np.max(np.abs(Y - ref)) <= 2.0 ** -20 * np.max(np.abs(ref))
    assert np.max(np.abs(Qr - Q) / np.abs(Q).max(axis=0)) <= 2.0 ** -11
    Y2, _ = cr.product(m, Q, nset=2)
    ref2 = m.astype(np.float64) @ Q.astype(np.float64)
    assert np.max(np.abs(Y2 - ref2)) <= 2.0 ** -19 * np.max(np.abs(ref2))

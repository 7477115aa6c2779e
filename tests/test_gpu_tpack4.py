"""Fourth-generation transposition (csrc/tpack4.hip) and the row stream the TF-IDF scale sweep writes (r05):
bit-exact against scipy's transpose and against the third generation, from the CSR arrays and from the row stream of X,
through the overflow slot and the half-width retries; the emitted stream equals the streaming copy's bytes and lsi gives
the same result either way."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.test_gpu_kernels import _check_stream, _heavy_rows_csr, _stream_ref, _up

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[2, 0], ids=["plain windows", "circular windows"])
def _keep_work(hip, request):
    hip.keep_tpack4_work = True  # the tests read the fill's error word
    hip.tune("tpack4_circ", request.param)  # (the stream-source fill with and without its circular windows)
    yield
    hip.tune("tpack4_circ", 0)
    hip.keep_tpack4_work = False
    hip._tpack4_work = None


def _t_ref(m):
    mt = m.T.tocsr()
    mt.sort_indices()
    return mt, _stream_ref(mt)


def _both_sources(hip, m, sort_rows=False):
    """X^T's stream by the fourth generation: from the CSR arrays, and from the row stream of X (sorted + dealt)."""
    X = _up(hip, m)
    P = hip.transpose_stream(X, sort_rows=sort_rows)
    assert hip.tpack4_status() == 0
    Xs = hip.stream(X)
    inv = torch.empty(m.shape[0], dtype=torch.int64, device=Xs.sptr.device)
    perm = Xs.perm.long()
    ok = perm >= 0
    inv[perm[ok]] = torch.nonzero(ok).reshape(-1)
    row_dst = Xs.sptr[inv].contiguous()
    Q = hip.transpose_stream(X, sort_rows=sort_rows, src=(Xs, row_dst))
    assert hip.tpack4_status() == 0
    return P, Q


@pytest.mark.parametrize("n,d,dens", [(1, 1, 1.0), (7, 5, 0.5), (100, 10, 0.2), (257, 131, 0.08), (300, 9000, 0.01),
                                      (2000, 20000, 0.004), (5000, 700, 0.03), (70, 4097, 0.2), (20000, 3000, 0.03)])
def test_fourth_generation_is_bit_exact_from_both_sources(hip, n, d, dens):
    rng = np.random.default_rng(n * 7 + d)
    m = _heavy_rows_csr(n, d, dens, rng, bursts=(n > 8 and d > 40))
    mt, (sptr, ent) = _t_ref(m)
    assert hip._use_tpack4(_up(hip, m)) or m.nnz == 0
    P, Q = _both_sources(hip, m)
    for R in (P, Q):
        assert np.array_equal(hip.to_host(R.sptr), sptr)
        assert np.array_equal(hip.to_host(R.ent).view(np.uint64)[: ent.size], ent)
    Ps, Qs = _both_sources(hip, m, sort_rows=True)
    _check_stream(hip, Ps, mt)
    _check_stream(hip, Qs, mt)
    # ... and the general transposition + streaming copy (the path of shapes the tile-staged one refuses) writes the same bytes
    try:
        hip.tune("tpack4_off", 1)
        P3 = hip.transpose_stream(_up(hip, m), sort_rows=False)
    finally:
        hip.tune("tpack4_off", 0)
    assert P3.t4 is None and torch.equal(P3.ent[: ent.size], P.ent[: ent.size])


@pytest.mark.parametrize("C", [0, 16, 32, 96, 768])
def test_overflow_slot_and_half_width_retries(hip, C):
    """Rows with dense bursts (33 .. 400 consecutive columns: more than one window, more than the overflow slot), many
    such rows in one wave, empty column ranges, tiles forced narrow and as wide as they get."""
    rng = np.random.default_rng(100 + C)
    n, d = 3000, 2600
    m = sp.random(n, d, density=0.02, format="lil", random_state=rng, dtype=np.float32)
    for r in rng.choice(n, 80, replace=False):
        c0 = int(rng.integers(0, d - 450))
        L = int(rng.integers(33, 400))
        m[r, c0:c0 + L] = rng.random(L).astype(np.float32) + 0.5
    for r in range(64, 72):  # eight neighbouring rows (one wave) with the same burst: more than the slot holds
        m[r, 100:180] = 1.25
    m = m.tocsr()
    m[:, 900:1400] = 0
    m[:, 2000:2100] = 0
    m.eliminate_zeros()
    m.sort_indices()
    mt, (sptr, ent) = _t_ref(m)
    try:
        hip.tune("tpack4_c", C)
        P, Q = _both_sources(hip, m)
    finally:
        hip.tune("tpack4_c", 0)
    for R in (P, Q):
        assert np.array_equal(hip.to_host(R.sptr), sptr)
        assert np.array_equal(hip.to_host(R.ent).view(np.uint64)[: ent.size], ent)


def test_tile_larger_than_the_staging_buffer_is_narrowed(hip):
    rng = np.random.default_rng(21)
    n, d = 60000, 200
    dense = sp.random(n, 100, density=0.95, format="csr", random_state=rng, dtype=np.float32)
    m = sp.hstack([dense, sp.csr_matrix((n, d - 100), dtype=np.float32)], format="csr")
    m.sort_indices()
    mt, (sptr, ent) = _t_ref(m)
    for C in (0, 768):
        try:
            hip.tune("tpack4_c", C)
            P, Q = _both_sources(hip, m)
        finally:
            hip.tune("tpack4_c", 0)
        for R in (P, Q):
            assert np.array_equal(hip.to_host(R.sptr), sptr)
            assert np.array_equal(hip.to_host(R.ent).view(np.uint64)[: ent.size], ent)
    T = hip.transpose_csr(_up(hip, m))
    assert np.array_equal(hip.to_host(T.indices), mt.indices) and np.array_equal(hip.to_host(T.indptr), mt.indptr)
    assert np.array_equal(hip.to_host(T.values).view(np.uint32), mt.data.view(np.uint32))


def test_scale_sweep_writes_the_row_stream(hip):
    """tfidf_device leaves the row stream of its result with it: the bytes the streaming copy would write, the same
    values as the sweep without it, and lsi on top gives identical embeddings either way."""
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device
    from tests.synth import planted_topics_csr

    m = planted_topics_csr(30000, 5000, n_topics=20, density=0.03, seed=3, dtype=np.float32)
    X = _up(hip, m)
    T = tfidf_device(hip, X, m.shape[0], 3, 1e4)
    T0 = tfidf_device(hip, X, m.shape[0], 3, 1e4, emit_stream=False)
    assert torch.equal(T.values, T0.values) and getattr(T0, "xstream", None) is None
    xs, row_dst, _key = T.xstream
    ref = hip.stream(T0)
    assert torch.equal(xs.sptr, ref.sptr) and torch.equal(xs.perm, ref.perm) and xs.k == ref.k
    assert torch.equal(xs.ent[: T.nnz], ref.ent[: T.nnz])
    host = sp.csr_matrix((hip.to_host(T.values), hip.to_host(T.indices), hip.to_host(T.indptr)), shape=T.shape)
    _check_stream(hip, xs, host)
    U, sd, V = lsi_device(hip, T, n_comps=20, n_obs=m.shape[0])
    assert hip.tpack4_status() == 0
    U0, sd0, V0 = lsi_device(hip, T0, n_comps=20, n_obs=m.shape[0])
    assert torch.equal(U, U0) and torch.equal(V, V0) and np.array_equal(sd, sd0)
    # the general transposition (CSR source, streaming copies) agrees bit for bit too
    try:
        hip.tune("tpack4_off", 1)
        U3, sd3, V3 = lsi_device(hip, T0, n_comps=20, n_obs=m.shape[0])
    finally:
        hip.tune("tpack4_off", 0)
    assert torch.equal(U, U3) and torch.equal(V, V3)


def test_explicit_zero_result_drops_the_stream(hip):
    """An output that is exactly 0 is dropped like scipy's SpGEMM drops it: the compacted result has other index arrays
    and must not carry the stream of the uncompacted one."""
    from muon_amd._atac.preproc import tfidf_device

    rng = np.random.default_rng(0)
    m = sp.random(500, 300, density=0.1, format="csr", random_state=rng, dtype=np.float32)
    m.data[:] = 1.0
    m.data[::17] = 0.0  # explicitly stored zeros -> TF = 0 -> dropped
    m.sort_indices()
    T = tfidf_device(hip, _up(hip, m), 500, 3, 1e4)
    assert T.nnz < m.nnz and hip._xstream_of(T) is None


def test_slab_pointers_made_at_ingest_give_the_same_results(hip):
    """upload_csr attaches the slab-pointer table of the index arrays (r05): the sweeps of tfidf and lsi's transposition read
    it instead of searching; sums, values, stream and transposed stream are bit-identical to the searching path, also
    through binarize's in-place rewrite of the values and for f64 counts."""
    from muon_amd._atac.preproc import tfidf_device
    from tests.synth import planted_topics_csr

    m = planted_topics_csr(20000, 30000, n_topics=20, density=0.02, seed=9, dtype=np.float32)
    m.sort_indices()
    X = hip.upload_csr(m.indptr, m.indices, m.data, m.shape)
    X0 = hip.upload_csr(m.indptr, m.indices, m.data, m.shape, slab_ptr=False)
    assert hip._slab_ptr_of(X) is not None and hip._slab_ptr_of(X0) is None
    # the table is what the sweeps would search: first entry of every row at or behind every 8192-column boundary
    S = -(-m.shape[1] // 8192)
    sp = hip.to_host(hip._slab_ptr_of(X)).reshape(m.shape[0], S + 1)
    rows = np.random.default_rng(0).integers(0, m.shape[0], 200)
    for r in rows:
        cols = m.indices[m.indptr[r]:m.indptr[r + 1]]
        want = m.indptr[r] + np.searchsorted(cols, np.arange(S) * 8192, side="left")
        assert np.array_equal(sp[r, :S], want) and sp[r, S] == m.indptr[r + 1]
    r1, c1 = hip.row_col_sums(X)
    hip.__dict__.pop("_sweep_work", None)
    r0, c0 = hip.row_col_sums(X0)
    hip.__dict__.pop("_sweep_work", None)
    assert torch.equal(r1, r0) and torch.equal(c1, c0)
    T, T0 = tfidf_device(hip, X, m.shape[0], 3, 1e4), tfidf_device(hip, X0, m.shape[0], 3, 1e4)
    assert torch.equal(T.values, T0.values)
    assert torch.equal(T.xstream[0].ent[: T.nnz], T0.xstream[0].ent[: T.nnz])
    assert hip._slab_ptr_of(T) is not None and hip._slab_ptr_of(T0) is not None  # (T0: the sweep's own search, handed on)
    assert torch.equal(hip._slab_ptr_of(T), hip._slab_ptr_of(T0))
    A, B = hip.stream_both(T), hip.stream_both(T0)
    assert torch.equal(A[1].ent[: T.nnz], B[1].ent[: T.nnz]) and torch.equal(A[1].sptr, B[1].sptr)
    # f64 counts: the generic sweeps with the table
    Xd = hip.upload_csr(m.indptr, m.indices, m.data.astype(np.float64), m.shape)
    Xd0 = hip.upload_csr(m.indptr, m.indices, m.data.astype(np.float64), m.shape, slab_ptr=False)
    Td, Td0 = tfidf_device(hip, Xd, m.shape[0], 3, 1e4), tfidf_device(hip, Xd0, m.shape[0], 3, 1e4)
    assert torch.equal(Td.values, Td0.values)

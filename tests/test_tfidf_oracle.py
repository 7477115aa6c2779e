"""The CPU oracle against the reference's own golden values and the fixtures produced by
executing the reference source (tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tfidf_oracle


def _csr(g, prefix):
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]),
                         shape=tuple(g[prefix + "_shape"]))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/tfidf_golden.npz")


def test_reference_known_answers_dense():
    # /root/reference/tests/test_atac_preproc.py:11-20
    np.random.seed(2020)
    x = np.abs(np.random.normal(size=(4, 5)))
    r = tfidf_oracle.tfidf(x, log_tf=True, log_idf=True)
    assert "%.3f" % r[0, 0] == "4.659"
    assert "%.3f" % r[3, 0] == "4.770"
    # :47-52 from_layer with counts + 1
    r = tfidf_oracle.tfidf(x + 1)
    assert "%.3f" % r[0, 0] == "2.856"


def test_reference_known_answers_sparse():
    # /root/reference/tests/test_atac_preproc.py:57-64
    np.random.seed(2020)
    x = sp.rand(100, 10, density=0.2, format="csr")
    r = tfidf_oracle.tfidf(x, log_tf=True, log_idf=True)
    assert "%.3f" % r[10, 9] == "18.749"
    assert "%.3f" % r[50, 5] == "0.000"


def test_oracle_matches_reference_fixtures_bitwise(gold):
    x = gold["dense_in"]
    assert (tfidf_oracle.tfidf(x) != _csr(gold, "dense_out")).nnz == 0
    xs = _csr(gold, "sparse_in")
    out = tfidf_oracle.tfidf(xs)
    ref = _csr(gold, "sparse_out")
    # same raw arrays, including scipy's descending index order
    assert np.array_equal(out.indices, ref.indices) and np.array_equal(out.indptr, ref.indptr)
    assert np.array_equal(out.data, ref.data)


SWEEPS = {
    "default": dict(),
    "nolog": dict(log_tf=False, log_idf=False),
    "logtfidf": dict(log_tf=False, log_idf=False, log_tfidf=True),
    "noscale": dict(scale_factor=None),
    "scale100": dict(scale_factor=100.0),
    "logtf_only": dict(log_idf=False),
    "logidf_only": dict(log_tf=False),
}


@pytest.mark.parametrize("name", sorted(SWEEPS))
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_oracle_option_sweep(gold, name, dt):
    cnt = _csr(gold, "sweep_in").astype(dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = tfidf_oracle.tfidf(cnt, **SWEEPS[name])
    ref = _csr(gold, f"sweep_{name}_{dt}")
    assert out.dtype == ref.dtype
    assert np.array_equal(out.indices, ref.indices) and np.array_equal(out.indptr, ref.indptr)
    assert np.array_equal(out.data, ref.data, equal_nan=True)


def test_oracle_errors():
    with pytest.raises(AttributeError):
        tfidf_oracle.tfidf(np.ones((2, 2)), log_tfidf=True)

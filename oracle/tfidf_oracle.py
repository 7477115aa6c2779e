"""CPU restatement of muon.atac.pp.tfidf's arithmetic (TEST INFRASTRUCTURE).

Follows /root/reference/muon/_atac/preproc.py:92-119 statement by statement on a
raw scipy CSR / numpy array, i.e. with the container handling (:62-90, :120-129)
stripped.  The arithmetic the reference delegates to scipy (diag x CSR SpGEMM,
``log1p`` on a sparse matrix) is delegated to the same scipy calls here so that
side effects on nnz pattern and index order are identical.
"""
import numpy as np
from scipy.sparse import csr_matrix, dia_matrix, issparse


def tfidf(counts, n_obs=None, log_tf=True, log_idf=True, log_tfidf=False, scale_factor=1e4):
    """Return the TF-IDF matrix exactly as the reference builds it (always CSR).

    ``counts``: scipy sparse matrix or dense ndarray, cells x peaks.
    ``n_obs``: adata.shape[0] (preproc.py:106); defaults to counts.shape[0].
    """
    if log_tfidf and (log_tf or log_idf):  # preproc.py:69-73
        raise AttributeError(
            "When returning log(TF*IDF), applying neither log(TF) nor log(IDF) is possible."
        )
    if n_obs is None:
        n_obs = counts.shape[0]

    if issparse(counts):  # preproc.py:92-96
        n_peaks = np.asarray(counts.sum(axis=1)).reshape(-1)
        n_peaks = dia_matrix((1.0 / n_peaks, 0), shape=(n_peaks.size, n_peaks.size))
        tf = np.dot(n_peaks, counts)
    else:  # preproc.py:97-99
        n_peaks = np.asarray(counts.sum(axis=1)).reshape(-1, 1)
        tf = counts / n_peaks

    if scale_factor is not None and scale_factor != 0 and scale_factor != 1:  # :101-102
        tf = tf * scale_factor
    if log_tf:  # :103-104
        tf = np.log1p(tf)

    idf = np.asarray(n_obs / counts.sum(axis=0)).reshape(-1)  # :106
    if log_idf:  # :107-108
        idf = np.log1p(idf)

    if issparse(tf):  # :110-112
        idf = dia_matrix((idf, 0), shape=(idf.size, idf.size))
        tf_idf = np.dot(tf, idf)
    else:  # :113-114
        tf_idf = np.dot(csr_matrix(tf), csr_matrix(np.diag(idf)))

    if log_tfidf:  # :116-117
        tf_idf = np.log1p(tf_idf)

    return np.nan_to_num(tf_idf, nan=0.0)  # :119 (a no-op on sparse input)


def canonical(m):
    """Sorted-index copy: the comparison convention for 'bit-exact indices'.

    The reference's output index order is an artefact of scipy's SpGEMM (descending
    columns by default, ``has_sorted_indices=False``); both sides are compared after
    ``sort_indices()``, and ``muon_amd`` can reproduce the raw order on request
    (``match_scipy_order=True``).
    """
    m = csr_matrix(m, copy=True)
    m.sort_indices()
    return m

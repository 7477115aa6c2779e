"""CPU restatement of muon.atac.tl.lsi (TEST INFRASTRUCTURE).

Follows /root/reference/muon/_atac/tools.py:50-69: ``scipy.sparse.linalg.svds``
(ARPACK, k components), reversal to descending order, optional embedding scaling,
``stdev = s / sqrt(n_obs - 1)``.  The oracle always runs ARPACK in float64 because
float32 ARPACK is only repeatable to ~6e-4 in subspace angle on gap-less spectra
(SURVEY.md §7 hard part 1); the reference's dtype follows ``X.dtype``.
"""
import numpy as np
from scipy.linalg import subspace_angles
from scipy.sparse.linalg import svds


def lsi(X, scale_embeddings=True, n_comps=50, dtype=np.float64):
    n_comps = min(n_comps, X.shape[1])  # tools.py:50
    cell_embeddings, svalues, peaks_loadings = svds(X.astype(dtype), k=n_comps)  # :53
    cell_embeddings = cell_embeddings[:, ::-1]  # :56-58
    svalues = svalues[::-1]
    peaks_loadings = peaks_loadings[::-1, :]
    if scale_embeddings:  # :60-63
        cell_embeddings = (cell_embeddings - cell_embeddings.mean(axis=0)) / cell_embeddings.std(
            axis=0
        )
    stdev = svalues / np.sqrt(X.shape[0] - 1)  # :65
    return {"X_lsi": cell_embeddings, "stdev": stdev, "LSI": peaks_loadings.T, "svalues": svalues}


def max_subspace_angle(A, B):
    """Largest principal angle (radians) between the column spaces of A and B."""
    return float(np.max(subspace_angles(np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64))))


def sign_align(A, B):
    """Flip columns of A so that each has positive inner product with B's column."""
    s = np.sign(np.sum(np.asarray(A) * np.asarray(B), axis=0))
    s[s == 0] = 1
    return A * s
